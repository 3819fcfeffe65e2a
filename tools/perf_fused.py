#!/usr/bin/env python3
"""A/B timing of flb_filter_do over [filter_parser(apache2), filter_grep] on a device-resident chunk: the fused
pair (fused_kernels.inc) against the unfused kernels (FLBGPU_NO_FUSE=1), same process, same chunk.
    python3 tools/perf_fused.py [records]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
from bench import APACHE2, TIME_FMT, GREP_RULE

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    g = flbamd_loader.load(); g.init(0); L = g.lib()
    data, off, ep = synth.apache_records(n)
    d_data = L.flbgpu_dev_alloc(data.nbytes); d_off = L.flbgpu_dev_alloc(off.nbytes)
    L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
    chunk = g.DevChunk(d_data, d_off, n, data.nbytes)
    p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
    fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE])
    ch = g.FilterChain([fp, fg])
    res = {}
    for mode in ("fused", "unfused", "fused"):
        if mode == "unfused": os.environ["FLBGPU_NO_FUSE"] = "1"
        else: os.environ.pop("FLBGPU_NO_FUSE", None)
        ch.filter_dev(chunk)
        fp.profile(True); fg.profile(True)
        L.flbgpu_sync(); t0 = time.perf_counter()
        for _ in range(5):
            r, o = ch.filter_dev(chunk)
        L.flbgpu_sync(); dt = (time.perf_counter() - t0) / 5
        prof = dict(fp.profile_read()); prof.update({"grep:" + k: v for k, v in fg.profile_read().items()})
        fp.profile(False); fg.profile(False)
        print("%-8s %.3f ms/step  %.1f M records/s  out %d bytes  kept %s" % (mode, dt * 1e3, n / dt / 1e6, int(o.bytes), ch.last_stats()[1]["out_records"]))
        print("         " + "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in prof.items()))

if __name__ == "__main__":
    main()
