#!/usr/bin/env python3
"""Generates tests/golden/decoders_kat.json: what the reference's OWN filter_parser (plugins/filter_parser/filter_parser.c over
src/flb_parser*.c and src/flb_parser_decoder.c, compiled in place: oracle/_ref/ref_filters, `make -C oracle ref`) answers for
every (Decode_Field set, parser format, Reserve_Data / Preserve_Key) configuration of tests/test_decoders_oracle.py on its
seeded chunk: return code + SHA-256 of the output bytes.  Run in the build container (the GPU box has no /root/reference);
the JSON file is committed so that the oracle's restatement stays pinned where the reference is absent."""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_filters as rf
import test_decoders_oracle as T

data = T._chunks()
cases, names = [], []
for di, decs in enumerate(T.DECODER_SETS):
    for pi, pa in enumerate(T.PARSERS):
        for rp in (0, 1):
            cases.append(rf.parser_case("msg", [dict(pa, decoders=decs)], data, bool(rp), bool(rp)))
            names.append([di, pi, rp])
res = rf.run(cases)
out = [{"case": n, "ret": ret, "sha256": hashlib.sha256(o).hexdigest(), "bytes": len(o)} for n, (ret, o) in zip(names, res)]
json.dump({"generator": "tests/golden/gen_decoders_kat.py", "source": "oracle/_ref/ref_filters (filter_parser.c + flb_parser_decoder.c, yyjson backend)",
           "chunk_sha256": hashlib.sha256(data).hexdigest(), "cases": out}, open(os.path.join(HERE, "decoders_kat.json"), "w"), indent=0)
print(len(out), "cases")
