#!/usr/bin/env python3
"""debug: row-by-row keep decisions of a filter_grep instance on NDJSON-made events, one-pass kernel against the three launches (run twice:
FLBGPU_GREP_LANE=0 and default), saved under gpurun_out/"""
import os, sys, json, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flbamd_loader, ndjson_synth as ns
g = flbamd_loader.load(); g.init(0); L = g.lib()
n = 20000
lines = ns.lines(n, seed=7)
data = b"".join(lines); off = g.split_lines(data)
d_data = L.flbgpu_dev_alloc(len(data) + 16); d_off = L.flbgpu_dev_alloc(off.nbytes)
L.flbgpu_memcpy_h2d(d_data, data, len(data)); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
pk = g.JsonPacker()
ev = pk.run_dev(g.DevChunk(d_data, d_off, n, len(data)), events=True, ts=(1, 0))
which = sys.argv[1]
rules = ns.GREP32_REGEX if which.startswith("r") else ns.GREP32_EXCLUDE
if len(sys.argv) > 2: rules = [rules[int(sys.argv[2])]]
fg = g.FilterGrep(rules, "OR")
r, k = fg.filter_dev(ev)
o = np.zeros(n + 1, dtype=np.uint64)
if r == g.MODIFIED:
    L.flbgpu_memcpy_d2h(o.ctypes.data, k.row_off, o.nbytes)
    lens = np.diff(o.astype(np.int64))
else:
    lens = np.full(n, -1)
tag = "lane" if os.environ.get("FLBGPU_GREP_LANE", "1") != "0" else "old"
np.save(os.path.join(ROOT, "gpurun_out", "dbg_%s_%s.npy" % (which + (sys.argv[2] if len(sys.argv) > 2 else ""), tag)), lens)
print(tag, which, "ret", r, "kept", int((lens > 0).sum()))
