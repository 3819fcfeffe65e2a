#!/usr/bin/env python3
"""first random multiline case of a seed where the device and the oracle differ, printed in full"""
import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import msgpack
import flbamd_loader, ml_synth
from test_multiline_oracle import oracle_run
from test_multiline_gpu import device_run
g = flbamd_loader.load(); g.init(0)
seed = int(sys.argv[1])
sub = len(sys.argv) > 2 and sys.argv[2] == "sub"
rng = random.Random(seed)
for it in range(60):
    if sub:
        cfg, frames, kw = ml_synth.random_sub_case(rng, bad_times=False)
    else:
        cfg, frames, kw = ml_synth.random_case(rng)
        if rng.random() < 0.3:
            cfg["buffer_limit_bytes"] = rng.choice([0, 1, 8, 40, 200, 1000])
    want, n, trunc = oracle_run(cfg, frames, clock_of_the_call=True, **kw)
    got, gn, st = device_run(g, cfg, frames, **kw)
    if got != want:
        print("case", it, cfg, kw)
        for f in frames: print("frame", f)
        def recs(b):
            u = msgpack.Unpacker(raw=True); u.feed(b)
            out = []
            try:
                for x in u: out.append(x)
            except Exception as e:
                out.append(("unpack error", str(e)))
            return out
        w, gt = recs(want), recs(got)
        k = next((i for i in range(min(len(w), len(gt))) if w[i] != gt[i]), min(len(w), len(gt)))
        print("records want", len(w), "got", len(gt), "first difference at record", k)
        for i in range(max(0, k - 3), min(max(len(w), len(gt)), k + 4)):
            print(i, "want", w[i] if i < len(w) else None)
            print(i, "got ", gt[i] if i < len(gt) else None)
        break
