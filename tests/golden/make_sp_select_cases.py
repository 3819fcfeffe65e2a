#!/usr/bin/env python3
"""Writes tests/golden/sp_select_cases.json: what the REFERENCE's sp_process_data (flb_sp.c:1607-1850, through oracle/_ref/ref_sp --
src/stream_processor/*.c compiled in place) answers for SELECTs without aggregation functions over seeded chunks: per case the
query, the chunks (modern [[ts, meta], map] records, legacy [ts, map] ones, a chunk cut inside its last record) and the
(return value, bytes) of every flb_sp_do.  Run here (needs /root/reference for `make -C oracle ref`); the fixture is what travels."""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_sp
import sp_synth

rng = random.Random(0x5E1)
cases = []
for q in sp_synth.SELECT_QUERIES:
    for rep in range(4):
        chunks = [sp_synth.select_chunk(rng, rng.choice([1, 9, 40]), legacy=rep == 2) for _ in range(rng.choice([1, 2]))]
        if rep == 3:
            chunks[-1] = chunks[-1][:len(chunks[-1]) - rng.randrange(1, 12)]
        r = ref_sp.RefSp(q)
        assert r.ok and r.select_only, q
        do = []
        for c in chunks:
            ret, out = r.do(c)
            do.append([ret, out.hex()])
        assert r.timer() == b""
        r.close()
        cases.append({"sql": q, "chunks": [c.hex() for c in chunks], "do": do})
with open(os.path.join(HERE, "sp_select_cases.json"), "w") as f:
    json.dump(cases, f, indent=0)
print(len(cases), "cases")
