#!/usr/bin/env python3
"""Generates tests/golden/json_kat.json: known answers of the reference's JSON -> msgpack path
(flb_pack_json, src/flb_pack.c:389-508) produced by the REAL yyjson reader compiled from
/root/reference (oracle/_ref/libyyjson_ref.so; `make -C oracle ref`).  Also records the reference's
own json/.mp sample pairs (tests/internal/data/pack/*.json + *.mp, used by tests/internal/pack.c).
Run in the build container (the GPU box has no /root/reference); the JSON file is committed."""
import glob, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import jsonfuzz as jf

ref = jf.reference()
assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
cases = []
for c in jf.corpus(20260921, 2500):
    if len(c) > 4096:
        continue
    r = ref(c)
    cases.append({"in": c.hex(), "ret": r[0], "out": r[1].hex() if r[1] is not None else None, "root_type": r[2], "records": r[3],
                  "consumed": r[4]})
pairs = []
for mp in sorted(glob.glob("/root/reference/tests/internal/data/pack/*.mp")):
    pairs.append({"name": os.path.basename(mp)[:-3], "json": open(mp[:-3] + ".json", "rb").read().hex(), "mp": open(mp, "rb").read().hex()})
json.dump({"generator": "tests/golden/gen_json_kat.py", "source": "oracle/_ref/libyyjson_ref.so (lib/yyjson-0.12.0 + src/flb_pack.c:389-508 glue)",
           "cases": cases, "reference_pairs": pairs}, open(os.path.join(HERE, "json_kat.json"), "w"))
print(len(cases), "cases,", len(pairs), "reference pairs")
