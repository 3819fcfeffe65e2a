#!/usr/bin/env python3
"""Generates tests/golden/msgpack_kat.json: known answers of the reference's msgpack-c (msgpack_unpack_next return
codes / offsets and the msgpack_pack_object re-pack of every object) produced by the REAL library compiled
from /root/reference (oracle/_ref/libmsgpack_ref.so; `make -C oracle ref`).  Run in the build container; the
JSON file is committed."""
import ctypes, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import test_msgpack_pin as t

ref = t._bind(ctypes.CDLL(t.REF).ref_msgpack_roundtrip)
cases = []
for s in t.corpus(20260921, 3000):
    if len(s) > 3000:
        continue
    codes, ends, out = t._run(ref, s)
    if codes[-1] == -2 and t._alloc_failure(s, ends[-1]):
        continue
    cases.append({"in": s.hex(), "codes": codes, "ends": ends, "out": out.hex()})
json.dump({"generator": "tests/golden/gen_msgpack_kat.py", "source": "oracle/_ref/libmsgpack_ref.so (lib/msgpack-c)", "cases": cases},
          open(os.path.join(HERE, "msgpack_kat.json"), "w"))
print(len(cases), "cases")
