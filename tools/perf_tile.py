#!/usr/bin/env python3
"""A/B timing of filter_parser's pass 1: the single-pass tile kernel (tile_kernels.inc) against the phase kernels
(FLBGPU_NO_TILE=1), same process, same device-resident chunk, for the pair [filter_parser, filter_grep] and for
filter_parser alone; the outputs of the two must be byte-identical.
    python3 tools/perf_tile.py [records]"""
import os, sys, time, zlib, ctypes
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
    modes = [("phase", {"FLBGPU_NO_TILE": "1"}), ("reg", {}), ("reg12", {"FLBGPU_TILE_WAVES": "12"}), ("reg8", {"FLBGPU_TILE_WAVES": "8"}),
             ("tile", {"FLBGPU_TILE_MODE": "tile"}), ("reg", {})]
    sums = {}
    for what in ("pair", "parser"):
        for mode, env in modes:
            for k in ("FLBGPU_NO_TILE", "FLBGPU_TILE_WAVES", "FLBGPU_TILE_MODE"):
                os.environ.pop(k, None)
            os.environ.update(env)
            run = (lambda: ch.filter_dev(chunk)) if what == "pair" else (lambda: fp.filter_dev(chunk))
            run()
            fp.profile(True); fg.profile(True)
            L.flbgpu_sync(); t0 = time.perf_counter()
            for _ in range(5):
                r, o = run()
            L.flbgpu_sync(); dt = (time.perf_counter() - t0) / 5
            prof = dict(fp.profile_read())
            fp.profile(False); fg.profile(False)
            nb = int(o.bytes)
            host = (ctypes.c_uint8 * nb)()
            L.flbgpu_memcpy_d2h(host, ctypes.c_void_p(o.data), nb)
            crc = zlib.crc32(bytes(host))
            sums.setdefault(what, set()).add((nb, crc))
            print("%-6s %-6s %.3f ms/step  %.1f M records/s  out %d bytes crc %08x" % (what, mode, dt * 1e3, n / dt / 1e6, nb, crc))
            print("         " + "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in prof.items()))
    for what, s in sums.items():
        print(what, "outputs identical across modes:", len(s) == 1)

if __name__ == "__main__":
    main()
