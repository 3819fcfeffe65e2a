#!/usr/bin/env python3
"""Generates tests/golden/unescape_kat.json: known answers of flb_unescape_string_utf8 (the decoder of
logfmt's quoted values, src/flb_unescape.c:186-277) produced by the REAL src/flb_unescape.c compiled
from /root/reference (oracle/_ref/libunescape_ref.so; `make -C oracle ref`).  Run in the build
container (the GPU box has no /root/reference); the JSON file is committed."""
import ctypes, json, os, random, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from test_kv_oracle import ESC_CASES, REF_UNESC

ref = ctypes.CDLL(REF_UNESC)
ref.flb_unescape_string_utf8.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
rng = random.Random(20260921)
alphabet = [b"\\", b"u", b"U", b"x", b"D", b"8", b"3", b"d", b"c", b"0", b"1", b"7", b"9", b"f", b"n", b'"', b"'", b"/", b"a", b"v", b"z", b"\xc3", b"\xa9",
            b"\xff", b"\x00", b" ", b"e", b"E", b"F", b"b", b"t", b"r"]
cases = list(ESC_CASES)
for _ in range(6000):
    cases.append(b"".join(rng.choice(alphabet) for _ in range(rng.randrange(0, 28))))
out = []
for s in cases:
    buf = ctypes.create_string_buffer(len(s) + 8)
    n = ref.flb_unescape_string_utf8(s, len(s), buf)
    out.append({"in": s.hex(), "out": buf.raw[:n].hex()})
json.dump({"generator": "tests/golden/gen_unescape_kat.py", "source": "oracle/_ref/libunescape_ref.so (src/flb_unescape.c)", "cases": out},
          open(os.path.join(HERE, "unescape_kat.json"), "w"))
print(len(out), "cases")
