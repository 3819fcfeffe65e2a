"""The oracle's flb_regex layer (oracle/oflb.c oflb_regex_create / oflb_regex_match: the /pat/imx option
syntax of src/flb_regex.c:60-152 in front of the regex engine) against the REAL src/flb_regex.c compiled
over the real Onigmo (oracle/_ref/libflbregex_ref.so): same patterns accepted, same match decisions."""
import base64
import ctypes
import json
import os
import random

import pytest

import oracle_binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "oracle", "_ref", "libflbregex_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libflbregex_ref.so not built (needs /root/reference)")


def test_option_syntax_and_match_decisions():
    ref = ctypes.CDLL(REF)
    ref.flb_regex_create.restype = ctypes.c_void_p
    ref.flb_regex_create.argtypes = [ctypes.c_char_p]
    ref.flb_regex_match.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    ref.flb_regex_destroy.argtypes = [ctypes.c_void_p]
    L = ob.lib()
    L.oflb_regex_create.restype = ctypes.c_void_p
    L.oflb_regex_create.argtypes = [ctypes.c_char_p]
    L.oflb_regex_match.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    L.oflb_regex_destroy.argtypes = [ctypes.c_void_p]
    kat = json.load(open(os.path.join(HERE, "golden", "regex_kat.json")))
    rng = random.Random(12)
    base = [(base64.b64decode(k["pattern"]), [base64.b64decode(c[0]) for c in k["cases"][:12]]) for k in kat if k["compiles"]]
    extra = [(b"hello", [b"HELLO world", b"hello", b"hell"]), (b"^a.c$", [b"a\nc", b"abc", b"ABC"]), (b"a b # comment", [b"ab", b"a b", b"a b # comment"]),
             (b"/", [b"/", b""]), (b"//", [b"", b"/", b"//"]), (b"/a", [b"/a", b"a"]), (b"a/", [b"a/", b"a"]), (b"/a/b/i", [b"a/b", b"A/B"]),
             (b"/a/", [b"a", b"/a/"]), (b"/a/q", [b"/a/q", b"a"]), (b"/a/ii", [b"A"]), (b"/a/I", [b"A", b"/a/I"]), (b"/(?<n>x+)/mi", [b"XX\n", b"y"])]
    wraps = [lambda p: p, lambda p: b"/" + p + b"/", lambda p: b"/" + p + b"/i", lambda p: b"/" + p + b"/m", lambda p: b"/" + p + b"/x",
             lambda p: b"/" + p + b"/imx", lambda p: b"/" + p + b"/xi", lambda p: b"/" + p + b"/z", lambda p: b"/" + p + b"/i ", lambda p: b"/" + p]
    checked = unsupported = 0
    for pat, subjects in base + extra:
        if b"\x00" in pat:
            continue
        for w in (wraps if (pat, subjects) in extra else rng.sample(wraps, 3)):
            full = w(pat)
            r = ref.flb_regex_create(full)
            o = L.oflb_regex_create(full)
            # the oracle's engine (like the GPU compiler) may refuse a construct the reference accepts -- then
            # creation fails loudly; it must never accept what the reference refuses
            assert r or not o, full
            if not r or not o:
                unsupported += bool(r) and not o
                if r:
                    ref.flb_regex_destroy(r)
                continue
            for s in subjects + [s.swapcase() for s in subjects[:4]]:
                a = ref.flb_regex_match(r, s, len(s))
                b = L.oflb_regex_match(o, s, len(s))
                assert (a > 0) == (b > 0), (full, s, a, b)
                checked += 1
            ref.flb_regex_destroy(r)
            L.oflb_regex_destroy(o)
    assert checked > 4000 and unsupported < 60, (checked, unsupported)
