#!/usr/bin/env python3
"""Writes tests/golden/sp_cases.json: what the REFERENCE's stream processor (oracle/_ref/ref_sp: src/stream_processor/*.c compiled
in place) answers for seeded chunks -- per case the query, the chunks, the (ret, packaged bytes) of every flb_sp_do and the bytes
the window timer packages.  Run here (needs /root/reference for `make -C oracle ref`); the fixture is what travels."""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_sp
import sp_synth

rng = random.Random(0x5B)
cases = []
for qi, q in enumerate(sp_synth.QUERIES):
    for rep in range(4):
        clean = rep < 2
        conv = rep != 3
        chunks = [sp_synth.chunk(rng, rng.choice([1, 7, 40]), clean) for _ in range(rng.choice([1, 2]))]
        r = ref_sp.RefSp(q, str_conv=conv)
        assert r.ok, q
        do = []
        for c in chunks:
            ret, out = r.do(c)
            do.append([ret, out.hex()])
        timer = r.timer().hex()
        r.close()
        cases.append({"sql": q, "str_conv": conv, "clean": clean, "chunks": [c.hex() for c in chunks], "do": do, "timer": timer})
with open(os.path.join(HERE, "sp_cases.json"), "w") as f:
    json.dump(cases, f, indent=0)
print(len(cases), "cases")
