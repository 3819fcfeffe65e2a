#!/usr/bin/env python3
"""Generates tests/golden/regex_kat.json by running the REAL reference regex engine
(oracle/_ref/libonig_ref.so = Onigmo 6.2.0 compiled from /root/reference/lib/onigmo by
oracle/Makefile) over a pattern corpus x seeded random inputs.  Run in the build container
(needs /root/reference); the JSON travels with the repo so the oracle and the HIP path can be
pinned on boxes where the reference is absent.

    python3 tests/golden/gen_regex_kat.py
"""
import json, os, random, sys, base64
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from rxdiff import load_ref, RefRegex, PATTERNS, rand_input, rand_input_illformed

def main():
    R = load_ref()
    assert R is not None, "build oracle/_ref first (make -C oracle ref)"
    rng = random.Random(0xF1B17)
    rng2 = random.Random(0x111F0)          # ill-formed inputs: their own stream, the first 60 cases of a pattern stay what they were
    out = []
    for pat in PATTERNS:
        r = RefRegex(R, pat)
        ent = {"pattern": base64.b64encode(pat).decode(), "compiles": r.ok, "cases": []}
        if r.ok:
            ent["names"] = r.names()
            u8 = any(c >= 0x80 for c in pat)
            seen = set()
            for i in range(60):
                s = rand_input(rng, pat, maxlen=28, utf8=u8)
                if s in seen:
                    continue
                seen.add(s)
                m = r.search(s)
                ent["cases"].append([base64.b64encode(s).decode(), m])
            for i in range(30):
                s = rand_input_illformed(rng2, pat, maxlen=20)
                if s in seen:
                    continue
                seen.add(s)
                ent["cases"].append([base64.b64encode(s).decode(), r.search(s)])
        out.append(ent)
    with open(os.path.join(HERE, "regex_kat.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("patterns", len(out), "cases", sum(len(e["cases"]) for e in out))

if __name__ == "__main__":
    main()
