#!/usr/bin/env python3
"""multiline on the device: text in HBM -> records in HBM through a built-in parser; wall time per call (whole buffer, one read)
and, under rocprofv3 --kernel-trace --stats, the per-kernel split.  usage: perf_ml.py [lines] [parser] [reps]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, ml_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
name = sys.argv[2] if len(sys.argv) > 2 else "java"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
g = flbamd_loader.load(); g.init(0); L = g.lib()
rng = random.Random(1)
block = ml_synth.cri_text(rng, 20000, damage=0.005, bad_times=False, ascii_only=not os.environ.get("ML_UTF8"), long_lines=0.0 if os.environ.get("ML_NOLONG") else 0.02) if name == "cri" else ml_synth.java_service_log(rng, 20000) if name == "java" and not os.environ.get("ML_SHORT") else ml_synth.random_text(rng, 20000, ml_synth.SEED_LINES[name], crlf=0.0, empty=0.02, long_line=0.0, nul_lead=0.0)
text = block * (n // 20000)
nl = text.count(b"\n")
d = L.flbgpu_dev_alloc(len(text)); L.flbgpu_memcpy_h2d(d, text, len(text))
p = g.MultilineParser(builtin=name)
print("product (states, classes, live):", p.product() if name != "cri" else None, flush=True)
s = p.stream()
for rep in range(reps):
    L.flbgpu_sync(); t0 = time.perf_counter()
    out, recs, proc = s.append_dev(d, len(text), 1700000000, 5, flush=True)
    L.flbgpu_sync(); dt = time.perf_counter() - t0
    print("lines %d bytes %d -> records %d out %d  %.3f ms  %.2f GB/s in  %.1f M lines/s" % (nl, len(text), recs, out.bytes, dt * 1e3, len(text) / dt / 1e9, nl / dt / 1e6), flush=True)
