#!/usr/bin/env python3
"""Cuts the first 400 records of the reference fixture tests/internal/data/mp/apache_10k.mp
(10 000 legacy-format [ext-ts, {"log": line}] records; pinned to 10 000 by
tests/internal/mp.c:18-38) into tests/golden/apache_400.mp and records, for every line, the
capture spans the REAL Onigmo produces for conf/parsers.conf 'apache2' and 'apache'
(tests/golden/apache_400_spans.json).  Needs /root/reference + oracle/_ref."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from rxdiff import load_ref, RefRegex
import synth

SRC = "/root/reference/tests/internal/data/mp/apache_10k.mp"
APACHE2 = rb'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$'
APACHE = rb'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^\"]*?)(?: +\S*)?)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>[^\"]*)")?$'

def main():
    b = open(SRC, "rb").read()
    recs = []; p = 0
    while p < len(b) and len(recs) < 400:
        o, q = synth._un(b, p)
        recs.append((p, q, o)); p = q
    cut = b[:recs[-1][1]]
    open(os.path.join(HERE, "apache_400.mp"), "wb").write(cut)
    R = load_ref()
    out = {}
    for name, pat in (("apache2", APACHE2), ("apache", APACHE)):
        r = RefRegex(R, pat)
        out[name] = [r.search(o[1][1][0][1]) for _, _, o in recs]
    json.dump(out, open(os.path.join(HERE, "apache_400_spans.json"), "w"), separators=(",", ":"))
    print(len(recs), len(cut))

if __name__ == "__main__":
    main()
