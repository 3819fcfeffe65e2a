"""Device record indexer: time per call on a device-resident chunk of apache records."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
g = flbamd_loader.load(); g.init(0)
L = g.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
data, off, ep = synth.apache_records(n)
blob = bytes(data)
d = L.flbgpu_dev_alloc(len(blob) + 16)
L.flbgpu_memcpy_h2d(d, blob, len(blob))
ix = g.Indexer()
ch, cons = ix.index_dev(d, len(blob))
assert ch.n == n and cons == len(blob), (ch.n, cons)
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    ch, cons = ix.index_dev(d, len(blob))
dt = (time.perf_counter() - t0) / reps
print("device indexer: %d records (%.2f GB) %.3f ms = %.2f G records/s, %.2f TB/s of chunk bytes; %s" %
      (n, len(blob) / 1e9, dt * 1e3, n / dt / 1e9, len(blob) / dt / 1e12, ix.stats()))
