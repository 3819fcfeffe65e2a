"""The oracle's strptime (oracle/otime.c, a restatement of src/flb_strptime.c) against the REAL source file
compiled from the reference (oracle/_ref/libstrptime_ref.so) -- live where /root/reference exists, and
through the committed answers (tests/golden/strptime_kat.json) everywhere."""
import ctypes
import json
import os

import pytest

import oracle_binding as ob
from strptime_cases import corpus

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "oracle", "_ref", "libstrptime_ref.so")


class TM(ctypes.Structure):          # struct tm (glibc, x86-64)
    _fields_ = [(n, ctypes.c_int) for n in ("sec", "min", "hour", "mday", "mon", "year", "wday", "yday", "isdst")] + \
               [("gmtoff", ctypes.c_long), ("zone", ctypes.c_char_p)]


class OTM(ctypes.Structure):         # struct otm (oracle/otime.h)
    _fields_ = [("tm", TM), ("gmtoff", ctypes.c_long)]


def _fields(tm, gmtoff):
    return [tm.sec, tm.min, tm.hour, tm.mday, tm.mon, tm.year, tm.wday, tm.yday, gmtoff]


_ref = None


def run_reference(fmt, text):
    global _ref
    if _ref is None:
        _ref = ctypes.CDLL(REF)
        _ref.flb_strptime.restype = ctypes.c_void_p
        _ref.flb_strptime.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(TM)]
    buf = ctypes.create_string_buffer(text.encode("latin-1"))
    tm = TM()
    r = _ref.flb_strptime(buf, fmt.encode(), ctypes.byref(tm))
    return (None, None) if not r else (r - ctypes.addressof(buf), _fields(tm, tm.gmtoff))


def run_oracle(fmt, text):
    L = ob.lib()
    L.o_strptime.restype = ctypes.c_void_p
    L.o_strptime.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(OTM)]
    buf = ctypes.create_string_buffer(text.encode("latin-1"))
    o = OTM()
    r = L.o_strptime(buf, fmt.encode(), ctypes.byref(o))
    return (None, None) if not r else (r - ctypes.addressof(buf), _fields(o.tm, o.gmtoff))


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libstrptime_ref.so not built (needs /root/reference)")
def test_oracle_strptime_matches_the_real_reference():
    for fmt, text in corpus(seed=77, extra=6000):
        assert run_oracle(fmt, text) == run_reference(fmt, text), (fmt, text)


def test_oracle_strptime_golden_vectors():
    kat = json.load(open(os.path.join(HERE, "golden", "strptime_kat.json")))
    assert len(kat["cases"]) > 5000
    for c in kat["cases"]:
        assert run_oracle(c["fmt"], c["text"]) == (c["consumed"], c["tm"]), (c["fmt"], c["text"])
