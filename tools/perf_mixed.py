#!/usr/bin/env python3
"""secondary.mixed_shapes by kernel: the headline pair on line lengths 80-600 B, 10 % multi-key bodies, 1 % legacy events, and on
sub-mixes (only the lengths / only the layouts) to see which property costs what"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth, bench
from bench import APACHE2, TIME_FMT, GREP_RULE
g = flbamd_loader.load(); g.init(0); L = g.lib()
recs = bench.mixed_shape_records()
def is_plain(r): return r[:4] == b"\x92\x92\xd7\x00" and r[12:14] == b"\x80\x81"
variants = {"all": recs,
            "plain layout only (all lengths)": [r for r in recs if is_plain(r)],
            "<= 277 B plain": [r for r in recs if is_plain(r) and len(r) <= 300],
            "400 + 600 B plain": [r for r in recs if is_plain(r) and len(r) > 380],
            "multi-key + legacy only": [r for r in recs if not is_plain(r)]}
for name, rs in variants.items():
    if not rs: continue
    tiles = max(1, 3_000_000 // len(rs))
    pool = b"".join(rs); mdata = pool * tiles
    sizes = np.array([len(x) for x in rs], dtype=np.uint64)
    moff = np.zeros(len(rs) * tiles + 1, dtype=np.uint64); np.cumsum(np.tile(sizes, tiles), out=moff[1:])
    mn = len(rs) * tiles
    d_md = L.flbgpu_dev_alloc(len(mdata) + 16); d_mo = L.flbgpu_dev_alloc(moff.nbytes)
    L.flbgpu_memcpy_h2d(d_md, mdata, len(mdata)); L.flbgpu_memcpy_h2d(d_mo, moff.ctypes.data, moff.nbytes)
    mch = g.DevChunk(d_md, d_mo, mn, len(mdata))
    p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
    fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE]); ch = g.FilterChain([fp, fg])
    for _ in range(2): ch.filter_dev(mch)
    fp.profile(True)
    import time
    L.flbgpu_sync(); t0 = time.perf_counter()
    for _ in range(5): ch.filter_dev(mch)
    L.flbgpu_sync(); dt = (time.perf_counter() - t0) / 5
    prof = dict(fp.profile_read()); fp.profile(False)
    print("%-34s %8d recs %6.1f MB  %.3f ms  %6.1f GB/s  %5.2f Grec/s | " % (name, mn, len(mdata) / 1e6, dt * 1e3, len(mdata) / dt / 1e9, mn / dt / 1e9) +
          "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in prof.items()), flush=True)
    fp.close(); fg.close(); p.close(); L.flbgpu_dev_free(d_md); L.flbgpu_dev_free(d_mo)
