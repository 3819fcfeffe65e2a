"""The oracle's msgpack unpacker and canonical re-pack (oracle/omp.c, a restatement of
lib/msgpack-c/src/unpack.c + objectc.c) against the REAL msgpack-c compiled from the reference
(oracle/_ref/libmsgpack_ref.so): same return code and offset for every msgpack_unpack_next call, same
re-packed bytes -- on valid streams, truncated ones, reserved bytes and random garbage."""
import ctypes
import json
import os
import random

import pytest

import oracle_binding as ob
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "oracle", "_ref", "libmsgpack_ref.so")
MAXC = 4096


def _run(fn, data):
    out = ctypes.c_void_p(); size = ctypes.c_size_t()
    codes = (ctypes.c_int * MAXC)(); ends = (ctypes.c_size_t * MAXC)()
    n = fn(data, len(data), ctypes.byref(out), ctypes.byref(size), codes, ends, MAXC)
    b = ctypes.string_at(out, size.value) if size.value else b""
    ctypes.CDLL(None).free(out)
    return [int(codes[i]) for i in range(n)], [int(ends[i]) for i in range(n)], b


def _bind(fn):
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                   ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int]
    return fn


def run_oracle(data):
    return _run(_bind(ob.lib().omp_roundtrip), data)


def _rand_obj(rng, depth=0):
    k = rng.randrange(13 if depth < 4 else 9)
    if k == 0: return None
    if k == 1: return rng.random() < 0.5
    if k == 2: return rng.choice([0, 1, 127, 128, 255, 256, 65535, 65536, 2**32 - 1, 2**32, 2**63 - 1, 2**64 - 1, rng.getrandbits(rng.randrange(1, 65))])
    if k == 3: return -rng.choice([1, 32, 33, 128, 129, 32768, 32769, 2**31, 2**31 + 1, 2**63, rng.getrandbits(rng.randrange(1, 64)) + 1])
    if k == 4: return rng.choice([0.0, -0.0, 1.5, 1e300, float("inf"), rng.random() * 10 ** rng.randrange(-5, 6)])
    if k == 5: return bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 1, 31, 32, 255, 256, rng.randrange(0, 400)])))
    if k == 6: return "s" * rng.choice([0, 5, 31, 32, 255, 256, 70000 if rng.random() < 0.02 else 3])
    if k == 7: return synth.Raw(b"\xc4" + bytes([3]) + b"bin")                                  # bin 8
    if k == 8: return synth.Raw(rng.choice([b"\xd4\x01\x00", b"\xd5\x02ab", b"\xd6\x03abcd", b"\xd7\x00" + bytes(8), b"\xd8\x05" + bytes(16), b"\xc7\x03\x07abc",
                                            b"\xc8\x00\x01\x09z", b"\xca\x3f\xc0\x00\x00", b"\xd0\x05", b"\xd1\x00\x05", b"\xd2\x00\x00\x00\x05", b"\xd3" + bytes(7) + b"\x05",
                                            b"\xcc\x05", b"\xcd\x00\x05", b"\xce\x00\x00\x00\x05", b"\xcf" + bytes(7) + b"\x05", b"\xd9\x01a", b"\xda\x00\x01a", b"\xdb\x00\x00\x00\x01a",
                                            b"\xdc\x00\x01\x01", b"\xdd\x00\x00\x00\x01\x01", b"\xde\x00\x01\x01\x02", b"\xdf\x00\x00\x00\x01\x01\x02", b"\xc5\x00\x01x", b"\xc6\x00\x00\x00\x01x"]))
    if k in (9, 10): return [_rand_obj(rng, depth + 1) for _ in range(rng.choice([0, 1, 2, 15, 16, 3]))]
    return synth.KV([(_rand_obj(rng, depth + 1), _rand_obj(rng, depth + 1)) for _ in range(rng.choice([0, 1, 2, 15, 16, 3]))])


def corpus(seed, n):
    rng = random.Random(seed)
    out = [b"", b"\xc1", b"\x92", b"\x92\x01", b"\xc0\xc1\xc0", b"\xd9", b"\xd9\x05ab", b"\xdd\xff\xff\xff\xff", b"\xdf\x00\x00\x00\x02\x01", b"\x81\xa1k"]
    out += [b"\x91" * k + b"\x01" for k in (30, 31, 32, 33, 34, 40)] + [b"\x81\x01" * k + b"\x02" for k in (31, 32, 33)]   # the 32-deep container stack
    out += [b"\x91" * 33, b"\x92\x92\xd7\x00" + bytes(8) + b"\x80\x81\xa1k" + b"\x91" * 31 + b"\x01"]
    for _ in range(n):
        s = b"".join(synth.mp(_rand_obj(rng)) for _ in range(rng.randrange(1, 5)))
        r = rng.random()
        if r < 0.25 and s: s = s[: rng.randrange(len(s))]                                        # truncated
        elif r < 0.35 and s:
            b = bytearray(s); b[rng.randrange(len(b))] = rng.getrandbits(8); s = bytes(b)          # one byte flipped
        elif r < 0.40: s = bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 40)))         # garbage
        out.append(s)
    return out


def _alloc_failure(s, off):
    """NOMEM because malloc refused the element table of an array32 / map32 with an absurd count (the
    executor stops on the last byte of the count, lib/msgpack-c/src/unpack.c:190-235): depends on the
    machine's memory, not on the bytes -- not part of the pin"""
    return off >= 4 and s[off - 4] in (0xdd, 0xdf) and int.from_bytes(s[off - 3:off + 1], "big") >= 1 << 24


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libmsgpack_ref.so not built (needs /root/reference)")
def test_oracle_msgpack_matches_the_real_library():
    ref = _bind(ctypes.CDLL(REF).ref_msgpack_roundtrip)
    for s in corpus(5, 6000):
        want = _run(ref, s)
        if want[0][-1] == -2 and _alloc_failure(s, want[1][-1]):
            continue
        assert run_oracle(s) == want, s[:80]


def test_oracle_msgpack_golden_vectors():
    kat = json.load(open(os.path.join(HERE, "golden", "msgpack_kat.json")))
    assert len(kat["cases"]) > 2000
    for c in kat["cases"]:
        s = bytes.fromhex(c["in"])
        assert run_oracle(s) == (c["codes"], c["ends"], bytes.fromhex(c["out"])), s[:80]
