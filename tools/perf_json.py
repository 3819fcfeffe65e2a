import sys, os, time, json, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flbamd_loader
g = flbamd_loader.load(); g.init(0); L = g.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = random.Random(1)
base = []
for i in range(4096):
    d = {"time": "2026-09-21T10:%02d:%02d.%03dZ" % (rng.randrange(60), rng.randrange(60), rng.randrange(1000)), "level": rng.choice(["info", "warn", "error", "debug"]),
         "msg": "request %d finished %s" % (rng.randrange(10 ** 6), rng.choice(["ok", "timeout", "refused"])), "code": rng.randrange(200, 600),
         "latency": round(rng.random() * 100, 3), "svc": {"name": rng.choice(["api", "db", "cache"]), "pod": "pod-%d" % rng.randrange(1000)},
         "path": "/v1/items/%d?x=%d" % (rng.randrange(10 ** 5), rng.randrange(100)), "bytes": rng.randrange(10 ** 6)}
    base.append(json.dumps(d).encode() + b"\n")
reps = (n + len(base) - 1) // len(base)
data = b"".join(base) * reps
off = g.split_lines(data)
n = len(off) - 1
print("lines", n, "bytes", len(data), "avg", len(data) / n)
d_data = L.flbgpu_dev_alloc(len(data) + 16); d_off = L.flbgpu_dev_alloc(off.nbytes)
L.flbgpu_memcpy_h2d(d_data, data, len(data)); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
chunk = g.DevChunk(d_data, d_off, n, len(data))
p = g.JsonPacker()
ev = p.run_dev(chunk, events=True, ts=(1, 0))
t0 = time.perf_counter()
for _ in range(3):
    ev = p.run_dev(chunk, events=True, ts=(1, 0))
dt = (time.perf_counter() - t0) / 3
print("json->events %.2f ms  %.1f M lines/s  in %.1f GB/s  out bytes %d" % (dt * 1e3, n / dt / 1e6, len(data) / dt / 1e9, ev.bytes), p.stats())
rules = [("regex", "level ^(error|warn)$")] + [("exclude", "msg pattern%d" % i) for i in range(31)]
fg = g.FilterGrep(rules)
fg.filter_dev(ev)
t0 = time.perf_counter()
for _ in range(3):
    r, kept = fg.filter_dev(ev)
dt2 = (time.perf_counter() - t0) / 3
print("grep 32 rules %.2f ms %.1f M rec/s kept" % (dt2 * 1e3, n / dt2 / 1e6), fg.counts())

# filter_parser with a Format json parser on {"log": "<json line>"} records
import synth
recs = [synth.v2_record(1700000000 + i, 0, {"log": base[i % len(base)].rstrip(b"\n")}) for i in range(len(base))]
blob = b"".join(recs) * reps
roff = np.zeros(len(recs) * reps + 1, dtype=np.uint64)
roff[1:] = np.cumsum(np.tile(np.array([len(r) for r in recs], dtype=np.uint64), reps))
nr = len(roff) - 1
d_b = L.flbgpu_dev_alloc(len(blob) + 16); d_o = L.flbgpu_dev_alloc(roff.nbytes)
L.flbgpu_memcpy_h2d(d_b, blob, len(blob)); L.flbgpu_memcpy_h2d(d_o, roff.ctypes.data, roff.nbytes)
pj = g.Parser(format="json", time_fmt="%Y-%m-%dT%H:%M:%S.%LZ", time_key="time")
fpj = g.FilterParser("log", [pj])
ck = g.DevChunk(d_b, d_o, nr, len(blob))
fpj.filter_dev(ck)
fpj.profile(True)
t0 = time.perf_counter()
for _ in range(3):
    r, o = fpj.filter_dev(ck)
dt3 = (time.perf_counter() - t0) / 3
print("filter_parser(json) %.2f ms  %.1f M rec/s  in %d B out %d B" % (dt3 * 1e3, nr / dt3 / 1e6, len(blob), o.bytes),
      {k: round(v[0] / v[1], 3) for k, v in fpj.profile_read().items()})
