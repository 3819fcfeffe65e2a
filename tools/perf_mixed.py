#!/usr/bin/env python3
"""bench.py's mixed_shapes chunk (line lengths 80-600 B, 10 % multi-key bodies, 1 % legacy events) through the pair, per kernel.
usage: perf_mixed.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flbamd_loader
import bench as b

def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    g = flbamd_loader.load(); g.init(0); L = g.lib()
    recs = b.mixed_shape_records()
    tiles = 3_000_000 // len(recs)
    mdata = b"".join(recs) * tiles
    sizes = np.array([len(x) for x in recs], dtype=np.uint64)
    moff = np.zeros(len(recs) * tiles + 1, dtype=np.uint64)
    np.cumsum(np.tile(sizes, tiles), out=moff[1:])
    mn = len(recs) * tiles
    d_md = L.flbgpu_dev_alloc(len(mdata) + 16); d_mo = L.flbgpu_dev_alloc(moff.nbytes)
    L.flbgpu_memcpy_h2d(d_md, mdata, len(mdata)); L.flbgpu_memcpy_h2d(d_mo, moff.ctypes.data, moff.nbytes)
    mch = g.DevChunk(d_md, d_mo, mn, len(mdata))
    p = g.Parser(b.APACHE2, time_fmt=b.TIME_FMT, time_key="time")
    fp = g.FilterParser("log", [p]); fg = g.FilterGrep([b.GREP_RULE])
    ch = g.FilterChain([fp, fg])
    for _ in range(3):
        ch.filter_dev(mch)
    L.flbgpu_sync()
    for rnd in range(2):
        fp.profile(True); fg.profile(True)
        ts = []
        for _ in range(steps):
            L.flbgpu_sync(); t0 = time.perf_counter()
            ch.filter_dev(mch)
            L.flbgpu_sync(); ts.append((time.perf_counter() - t0) * 1e3)
        pr = dict(fp.profile_read()); pr.update(fg.profile_read())
        fp.profile(False); fg.profile(False)
        print("calls ms:", " ".join("%.2f" % t for t in ts))
        print("median %.3f ms  mean %.3f ms   %.0f GB/s at the median" % (sorted(ts)[len(ts) // 2], sum(ts) / len(ts), len(mdata) / sorted(ts)[len(ts) // 2] / 1e6))
        print("   " + "  ".join("%s %.3f x%d" % (k, v[0] / max(v[1], 1), v[1]) for k, v in pr.items()))
        print("   paths", fp.paths())

if __name__ == "__main__":
    main()
