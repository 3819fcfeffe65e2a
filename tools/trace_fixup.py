#!/usr/bin/env python3
"""The fix-up launch of k_parser_reg on the mixed-shapes chunk (bench.mixed_shape_records: 11 % of the rows have another layout): the
s_memtime stamps of its phases per wave-iteration (FLBGPU_TRACE + FLBGPU_TRACE_FIXUP=1).   python3 tools/trace_fixup.py"""
import os, sys, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, bench
from bench import APACHE2, TIME_FMT, GREP_RULE
g = flbamd_loader.load(); g.init(0); L = g.lib()
rs = bench.mixed_shape_records()
tiles = 3_000_000 // len(rs)
mdata = b"".join(rs) * tiles
sizes = np.array([len(x) for x in rs], dtype=np.uint64)
moff = np.zeros(len(rs) * tiles + 1, dtype=np.uint64); np.cumsum(np.tile(sizes, tiles), out=moff[1:])
mn = len(rs) * tiles
d_md = L.flbgpu_dev_alloc(len(mdata) + 16); d_mo = L.flbgpu_dev_alloc(moff.nbytes)
L.flbgpu_memcpy_h2d(d_md, mdata, len(mdata)); L.flbgpu_memcpy_h2d(d_mo, moff.ctypes.data, moff.nbytes)
mch = g.DevChunk(d_md, d_mo, mn, len(mdata))
p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE]); ch = g.FilterChain([fp, fg])
for _ in range(2): ch.filter_dev(mch)
path = "/tmp/flbgpu_trace_fixup.bin"
os.environ["FLBGPU_TRACE"] = "8"; os.environ["FLBGPU_TRACE_FILE"] = path; os.environ["FLBGPU_TRACE_FIXUP"] = "1"
fp.profile(True)
ch.filter_dev(mch); L.flbgpu_sync()
prof = dict(fp.profile_read()); fp.profile(False)
print("traced run:", "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in prof.items()))
raw = open(path, "rb").read()
grid, waves, iters, k = struct.unpack("<4I", raw[:16])
t = np.frombuffer(raw[16:], dtype=np.uint64).reshape(grid, waves, iters, k).astype(np.int64)
done = (t[..., 7] != 0) & (t[..., 0] != 0)
print("fix-up iterations per wave: mean %.2f  max %d" % (done.sum(-1).mean(), done.sum(-1).max()))
NAMES = ["row offsets", "ingest / general locate", "decode + shift (guarded burst)", "walk", "fields + rules", "time", "stores"]
d = np.diff(t, axis=-1)[done]
tot = (t[..., 7] - t[..., 0])[done]
print("per wave-iteration, ticks (median / mean / p90):   total %d / %d / %d" % (np.median(tot), tot.mean(), np.percentile(tot, 90)))
for i, nm in enumerate(NAMES):
    c = d[:, i]
    print("    %-36s %8d %8d %8d  %5.1f %%" % (nm, np.median(c), c.mean(), np.percentile(c, 90), 100.0 * c.mean() / tot.mean()))
first = t[..., 0][done].min(); last = t[..., 7][done].max()
print("span of the traced iterations: %d ticks" % (last - first))
