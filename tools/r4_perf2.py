#!/usr/bin/env python3
"""k_parser_reg timing experiments whose results are WRONG (debug flags): a fresh filter per configuration, its first call only
(the filter would decline the register kernel afterwards)"""
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
for skip in sys.argv[2].split(","):
    for rep in range(2):
        os.environ["FLBGPU_DEBUG_SKIP"] = skip
        p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
        fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE])
        ch = g.FilterChain([fp, fg])
        fp.profile(True)
        ch.filter_dev(chunk)
        L.flbgpu_sync()
        prof = dict(fp.profile_read()); fp.profile(False)
        print("skip", skip, "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in prof.items()), flush=True)
