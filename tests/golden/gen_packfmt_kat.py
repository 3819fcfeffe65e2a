#!/usr/bin/env python3
"""Generates tests/golden/packfmt_kat.json: known answers of the reference's flb_pack_msgpack_to_json_format produced by
the REAL function compiled from /root/reference (oracle/_ref/ref_packfmt; `make -C oracle ref`).  Run in the build
container; the JSON file is committed."""
import base64, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import packfmt_cases, test_packfmt_oracle as t

cases = packfmt_cases.corpus(20260922, 1500)
outs = t.run_reference(cases)
kat = []
for (cfg, data), out in zip(cases, outs):
    c = dict(cfg)
    c["date_key"] = None if cfg["date_key"] is None else base64.b64encode(cfg["date_key"]).decode()
    kat.append({"cfg": c, "in": base64.b64encode(data).decode(), "out": None if out is None else base64.b64encode(out).decode()})
json.dump({"generator": "tests/golden/gen_packfmt_kat.py", "source": "oracle/_ref/ref_packfmt (src/flb_pack.c, src/flb_utils.c, ...)",
           "cases": kat}, open(os.path.join(HERE, "packfmt_kat.json"), "w"))
print(len(kat), "cases", sum(1 for k in kat if k["out"] is None), "NULL")
