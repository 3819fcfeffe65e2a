#!/usr/bin/env python3
"""text -> in_tail's line packing -> [filter_parser, filter_grep] on the device, per-kernel times"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
from bench import APACHE2, TIME_FMT, GREP_RULE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
g = flbamd_loader.load(); g.init(0); L = g.lib()
data, off, ep = synth.apache_records(n)
ev = np.asarray(data).reshape(n, 277)
txt = np.empty((n, 257), dtype=np.uint8); txt[:, :256] = ev[:, 21:]; txt[:, 256] = 10
d_txt = L.flbgpu_dev_alloc(txt.nbytes); L.flbgpu_memcpy_h2d(d_txt, txt.ctypes.data, txt.nbytes)
tl = g.TailLines()
lines, chunk, proc = tl.process_dev(d_txt, txt.nbytes, sec=1700000000)
p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE])
ch = g.FilterChain([fp, fg])
for rep in range(2):
    ch.filter_dev(chunk)
    fp.profile(True)
    for _ in range(5): ch.filter_dev(chunk)
    L.flbgpu_sync()
    prof = dict(fp.profile_read()); fp.profile(False)
    print("  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in prof.items()), ch.last_stats())
