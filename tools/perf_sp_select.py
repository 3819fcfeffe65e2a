#!/usr/bin/env python3
"""flb_sp's plain SELECT on a device-resident chunk of BASELINE configs[4]'s record shape: the two passes' event-timed milliseconds."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, sp_synth

def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    g = flbamd_loader.load(); g.init(0); L = g.lib()
    data, off = sp_synth.config4_chunk(m)
    d_d = L.flbgpu_dev_alloc(data.nbytes + 16); d_o = L.flbgpu_dev_alloc(off.nbytes)
    L.flbgpu_memcpy_h2d(d_d, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_o, off.ctypes.data, off.nbytes)
    ch = g.DevChunk(d_d, d_o, m, data.nbytes)
    for sql in ("SELECT status, host AS h, latency FROM STREAM:x WHERE status >= 400;", "SELECT * FROM STREAM:x WHERE status >= 400;", "SELECT host FROM STREAM:x;"):
        t = g.StreamTask(sql)
        ret, out = t.do_dev(ch)
        t.profile(True)
        t0 = time.perf_counter()
        for _ in range(3):
            t.do_dev(ch)
        dt = (time.perf_counter() - t0) / 3
        pr = t.profile(False)
        print("%-75s records %8d bytes %10d  call %.2f ms  kernels %s  sha %s" % (sql, ret, len(out), dt * 1e3, {k: round(v[0] / max(v[1], 1), 3) for k, v in pr.items()}, hashlib.sha256(out).hexdigest()[:12]), flush=True)
        t.close()

if __name__ == "__main__":
    main()
