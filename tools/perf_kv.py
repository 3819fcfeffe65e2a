"""filter_parser with a Format logfmt parser: records/s on device-resident chunks (secondary measurement)."""
import sys, os, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flbamd_loader, synth
g = flbamd_loader.load(); g.init(0)
L = g.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = random.Random(3)
base = []
for i in range(2048):
    t = 'time=2026-09-21T10:%02d:%02d.%03d level=%s msg="request %d finished %s" code=%d latency=%.3f svc=%s path=/v1/items/%d?x=%d bytes=%d ok' % (
        rng.randrange(60), rng.randrange(60), rng.randrange(1000), rng.choice(["info", "warn", "error"]), rng.randrange(10 ** 6),
        rng.choice(["ok", "timeout", "refused \\\"by peer\\\""]), rng.randrange(200, 600), rng.random() * 100, rng.choice(["api", "db", "cache"]),
        rng.randrange(10 ** 5), rng.randrange(100), rng.randrange(10 ** 6))
    base.append(synth.mp([[synth.ext_ts(1700000000 + i, i), {}], {"log": t}]))
reps = (n + len(base) - 1) // len(base)
blob = b"".join(base) * reps
n = len(base) * reps
d = L.flbgpu_dev_alloc(len(blob) + 16)
L.flbgpu_memcpy_h2d(d, blob, len(blob))
for fmt, kw in (("logfmt", dict(time_fmt="%Y-%m-%dT%H:%M:%S.%L", time_key="time")),):
    f = g.FilterParser("log", [g.Parser(format=fmt, **kw)])
    raw = g.DevChunk(d, None, 0, len(blob))
    r, o = f.filter_dev(raw)
    assert r == g.MODIFIED and int(o.n) == n
    ix = g.Indexer(); ch, _ = ix.index_dev(d, len(blob))
    t0 = time.perf_counter()
    for _ in range(3):
        r, o = f.filter_dev(ch)
    L.flbgpu_sync()
    dt = (time.perf_counter() - t0) / 3
    print("filter_parser(%s): %d records (%.2f GB in, %.2f GB out) %.2f ms = %.1f M records/s" % (fmt, n, len(blob) / 1e9, int(o.bytes) / 1e9, dt * 1e3, n / dt / 1e6))
