#!/usr/bin/env python3
"""k_parser_reg: staged ingest (ring of LDS buffers) against per-lane loads, with the FLBGPU_DEBUG_SKIP builds
(1 walk, 2 value loads, 4 result, 16 rules, 32/64 time) -- where the time goes in each"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
from bench import APACHE2, TIME_FMT, GREP_RULE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cfgs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0:16", "1:16", "2:16", "3:16", "4:12", "3:14"]      # staging buffers : waves per CU
g = flbamd_loader.load(); g.init(0); L = g.lib()
data, off, ep = synth.apache_records(n)
d_data = L.flbgpu_dev_alloc(data.nbytes); d_off = L.flbgpu_dev_alloc(off.nbytes)
L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
chunk = g.DevChunk(d_data, d_off, n, data.nbytes)
for cfg in cfgs:
    nbuf, waves = cfg.split(":")
    for skip in (("0", "16", "48", "112", "2", "1", "3", "115") if waves == "16" and nbuf in ("0", "3") else ("0",)):
        # (a fresh filter per build: a skip build that sends every record down the slow path makes a filter decline the single pass)
        p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
        fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE])
        ch = g.FilterChain([fp, fg])
        os.environ["FLBGPU_DEBUG_SKIP"] = skip; os.environ["FLBGPU_STAGE_NBUF"] = nbuf; os.environ["FLBGPU_TILE_WAVES"] = waves
        ch.filter_dev(chunk)
        fp.profile(True)
        for _ in range(5): ch.filter_dev(chunk)
        L.flbgpu_sync()
        prof = dict(fp.profile_read()); fp.profile(False)
        st = ch.last_stats()
        fp.close(); fg.close(); p.close()
        print("nbuf", nbuf, "waves", waves, "skip", skip, "kept", int(st[1]["out_records"]), "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in prof.items()), flush=True)
