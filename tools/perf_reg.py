#!/usr/bin/env python3
"""k_parser_reg: occupancy sensitivity (FLBGPU_TILE_WAVES) x skip builds (FLBGPU_DEBUG_SKIP) at 10 M records"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
from bench import APACHE2, TIME_FMT, GREP_RULE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = flbamd_loader.load(); g.init(0); L = g.lib()
data, off, ep = synth.apache_records(n)
d_data = L.flbgpu_dev_alloc(data.nbytes); d_off = L.flbgpu_dev_alloc(off.nbytes)
L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
chunk = g.DevChunk(d_data, d_off, n, data.nbytes)
p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE])
ch = g.FilterChain([fp, fg])
for waves in sys.argv[2].split(",") if len(sys.argv) > 2 else ("16", "12", "8", "4"):
    for skip in sys.argv[3].split(",") if len(sys.argv) > 3 else ("0", "1", "3"):
        os.environ["FLBGPU_DEBUG_SKIP"] = skip; os.environ["FLBGPU_TILE_WAVES"] = waves
        ch.filter_dev(chunk)
        fp.profile(True)
        for _ in range(5): ch.filter_dev(chunk)
        L.flbgpu_sync()
        prof = dict(fp.profile_read()); fp.profile(False)
        print("waves", waves, "skip", skip, "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in prof.items()), flush=True)
