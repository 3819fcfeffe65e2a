#!/usr/bin/env python3
"""Timing of the msgpack -> JSON formatter on a device-resident chunk of parsed apache records (filter_parser's
output), per kernel.   python3 tools/perf_fmt.py [records] [json_format] [date_format]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
from bench import APACHE2, TIME_FMT

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    fmt = sys.argv[2] if len(sys.argv) > 2 else "lines"
    df = sys.argv[3] if len(sys.argv) > 3 else "double"
    g = flbamd_loader.load(); g.init(0); L = g.lib()
    data, off, ep = synth.apache_records(n)
    d_data = L.flbgpu_dev_alloc(data.nbytes); d_off = L.flbgpu_dev_alloc(off.nbytes)
    L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
    chunk = g.DevChunk(d_data, d_off, n, data.nbytes)
    fp = g.FilterParser("log", [g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")])
    r, parsed = fp.filter_dev(chunk)
    assert r == g.MODIFIED
    for esc in (1, 0):
        jf = g.JsonFormatter(fmt, df, b"date", escape_unicode=esc)
        rc, o = jf.format_dev(parsed)
        assert rc == 0, g.last_error()
        jf.profile(True)
        L.flbgpu_sync(); t0 = time.perf_counter()
        for _ in range(5):
            rc, o = jf.format_dev(parsed)
        L.flbgpu_sync(); dt = (time.perf_counter() - t0) / 5
        prof = jf.profile_read()
        ib, ob = int(parsed.bytes), int(o.bytes)
        print("escape_unicode=%d %s/%s: %.3f ms/step  %.1f M records/s  in %d B out %d B  (in+out)/t = %.0f GB/s" % (esc, fmt, df, dt * 1e3, n / dt / 1e6, ib, ob, (ib + ob) / dt / 1e9))
        print("         " + "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in prof.items()))
        jf.close()

if __name__ == "__main__":
    main()
