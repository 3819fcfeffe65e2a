#!/usr/bin/env python3
"""BASELINE configs[4] shape: flb_sp GROUP BY status / AVG(latency) over a tumbling window, chunk resident in HBM; per-kernel
times, and the reference's own flb_sp (oracle/_ref/ref_sp) on a bounded sample of the same records beside it"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, sp_synth, ref_sp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
sql = sys.argv[2] if len(sys.argv) > 2 else sp_synth.CONFIG4_SQL
g = flbamd_loader.load(); g.init(0); L = g.lib()
data, off = sp_synth.config4_chunk(n)
d_data = L.flbgpu_dev_alloc(data.nbytes + 16); d_off = L.flbgpu_dev_alloc(off.nbytes)
L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
chunk = g.DevChunk(d_data, d_off, n, data.nbytes)
t = g.StreamTask(sql)
t.do_dev(chunk); t.timer()
t.profile(True)
steps = 5
t0 = time.perf_counter()
for _ in range(steps):
    rec, _ = t.do_dev(chunk)
L.flbgpu_sync()
dt = (time.perf_counter() - t0) / steps
prof = t.profile(False)
out = t.timer()
res = {"records": n, "chunk_bytes": int(data.nbytes), "ms_per_chunk": round(dt * 1e3, 3), "records_per_s": round(n / dt, 1),
       "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in prof.items()}, "window_records": rec, "groups_bytes": len(out)}
ke = res["kernel_ms"]["k_sp_extract"]
res["extract_GBps"] = round(data.nbytes / (ke / 1e3) / 1e9, 1) if ke else None
if ref_sp.available():
    m = min(n, 1_000_000)
    sample = data[: int(off[m])].tobytes()
    r = ref_sp.RefSp(sql)
    t0 = time.perf_counter()
    r.do(sample)
    want = r.timer()
    dt_ref = time.perf_counter() - t0
    r.close()
    t2 = g.StreamTask(sql)
    t2.do(sample); got = t2.timer(); t2.close()
    res["reference"] = {"records": m, "seconds": round(dt_ref, 3), "records_per_s": round(m / dt_ref, 1), "identical": got == want}
print(json.dumps(res))
