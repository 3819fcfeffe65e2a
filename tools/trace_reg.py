#!/usr/bin/env python3
"""k_parser_reg's timeline: s_memtime stamps of the phases of every wave's iterations (FLBGPU_TRACE), per-phase medians and how
the 16 waves of one CU sit relative to each other.   python3 tools/trace_reg.py [records] [nbuf,...]"""
import os, sys, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
from bench import APACHE2, TIME_FMT, GREP_RULE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cfgs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "3"]
g = flbamd_loader.load(); g.init(0); L = g.lib()
data, off, ep = synth.apache_records(n)
d_data = L.flbgpu_dev_alloc(data.nbytes); d_off = L.flbgpu_dev_alloc(off.nbytes)
L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
chunk = g.DevChunk(d_data, d_off, n, data.nbytes)
NAMES = ["row offsets", "ingest (header + value -> registers)", "decode + shift", "walk", "fields + rules", "time", "stores"]
if "fine" in os.environ.get("FLBGPU_LIB", ""):       # tools/variant.sh fine "-DREG_TRACE_FINE": the stamps spent behind the walk
    NAMES = ["offsets + ingest + decode + walk", "walk's result, time text", "span reads", "field sizes", "rules", "time", "stores"]
for nbuf in cfgs:
    os.environ["FLBGPU_STAGE_NBUF"] = nbuf
    p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
    fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE])
    ch = g.FilterChain([fp, fg])
    ch.filter_dev(chunk)
    fp.profile(True)
    for _ in range(3): ch.filter_dev(chunk)
    L.flbgpu_sync()
    prof = dict(fp.profile_read()); fp.profile(False)
    print("nbuf", nbuf, "untraced:", "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in prof.items()))
    path = "/tmp/flbgpu_trace_%s.bin" % nbuf
    os.environ["FLBGPU_TRACE"] = "48"; os.environ["FLBGPU_TRACE_FILE"] = path
    fp.profile(True)
    ch.filter_dev(chunk)
    L.flbgpu_sync()
    prof = dict(fp.profile_read()); fp.profile(False)
    del os.environ["FLBGPU_TRACE"]
    raw = open(path, "rb").read()
    grid, waves, iters, k = struct.unpack("<4I", raw[:16])
    t = np.frombuffer(raw[16:], dtype=np.uint64).reshape(grid, waves, iters, k).astype(np.int64)
    done = (t[..., 7] != 0) & (t[..., 0] != 0)
    print("nbuf", nbuf, "traced run:", "  ".join("%s %.3f" % (kk, v[0] / max(v[1], 1)) for kk, v in prof.items()), " iterations/wave", done.sum(-1).mean())
    d = np.diff(t, axis=-1)[done]                   # [m, 7]
    tot = (t[..., 7] - t[..., 0])[done]
    print("  per wave-iteration, ticks (median / mean / p90):   total %d / %d / %d" % (np.median(tot), tot.mean(), np.percentile(tot, 90)))
    for i, nm in enumerate(NAMES):
        print("    %-40s %8d %8d %8d   %5.1f %%" % (nm, np.median(d[:, i]), d[:, i].mean(), np.percentile(d[:, i], 90), 100 * d[:, i].mean() / tot.mean()))
    # kernel span and the ticks' unit
    span = t[..., 7][done].max() - t[..., 0][done].min()
    print("  kernel span %d ticks = %.3f ms traced -> %.1f ticks/us" % (span, prof["k_parser_reg"][0], span / (prof["k_parser_reg"][0] * 1e3)))
    # phase alignment inside one workgroup: start of iteration 5 of each of its waves, relative to the earliest
    for b in (0, grid // 2):
        s5 = t[b, :, 5, 0] - t[b, :, 5, 0].min()
        w5 = t[b, :, 5, 3] - t[b, :, 5, 0].min()
        print("  wg %d iteration 5: start offsets of its waves %s" % (b, " ".join("%d" % x for x in s5)))
        print("  wg %d iteration 5: walk start offsets          %s" % (b, " ".join("%d" % x for x in w5)))
    # how many of a workgroup's waves are inside the walk at a time (sampled over the kernel)
    b = 0
    t0, t1 = t[b, :, :, 0][done[b]].min(), t[b, :, :, 7][done[b]].max()
    xs = np.linspace(t0, t1, 2000)
    inwalk = np.zeros_like(xs); inmem = np.zeros_like(xs)
    for w in range(waves):
        for it in range(iters):
            if not done[b, w, it]: continue
            inwalk += (xs >= t[b, w, it, 3]) & (xs < t[b, w, it, 4])
            inmem += ((xs >= t[b, w, it, 0]) & (xs < t[b, w, it, 2]))
    print("  wg 0: waves inside the walk: mean %.1f  (hist %s)   inside rowoff+ingest: mean %.1f" % (inwalk.mean(), np.bincount(inwalk.astype(int), minlength=17).tolist(), inmem.mean()))
    fp.close(); fg.close(); p.close()
