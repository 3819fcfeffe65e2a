#!/usr/bin/env python3
"""Generates tests/golden/strptime_kat.json: known answers of flb_strptime (src/flb_strptime.c:248-816) produced
by the REAL source file compiled from /root/reference (oracle/_ref/libstrptime_ref.so; `make -C oracle ref`):
for every (format, text) pair the number of characters consumed (null = no match) and the struct tm fields
+ gmtoff it left behind.  Run in the build container; the JSON file is committed."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from strptime_cases import corpus
from test_strptime_pin import run_reference

cases = []
for f, t in corpus():
    consumed, fields = run_reference(f, t)
    cases.append({"fmt": f, "text": t, "consumed": consumed, "tm": fields})
json.dump({"generator": "tests/golden/gen_strptime_kat.py", "source": "oracle/_ref/libstrptime_ref.so (src/flb_strptime.c)",
           "fields": ["sec", "min", "hour", "mday", "mon", "year", "wday", "yday", "gmtoff"], "cases": cases},
          open(os.path.join(HERE, "strptime_kat.json"), "w"))
print(len(cases), "cases")
