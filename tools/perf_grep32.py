#!/usr/bin/env python3
"""configs[2] alone: NDJSON -> events, then the 16 Regex (OR) + 16 Exclude (OR) filter_grep chain, with per-kernel times.
FLBGPU_GREP_NO_LDS_RULES=1 keeps the rule tables in global memory (A/B of the LDS staging in k_grep_match)."""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader
import bench as b
import torch
g = flbamd_loader.load(); g.init(0)

nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
steps = 10
rng = random.Random(7)
base = []
for i in range(4096):
    d = {"time": "2026-09-21T10:%02d:%02d.%03dZ" % (rng.randrange(60), rng.randrange(60), rng.randrange(1000)),
         "level": rng.choice(["info", "warn", "error", "debug"]),
         "msg": "request %d finished %s" % (rng.randrange(10 ** 6), rng.choice(["ok", "timeout", "refused"])),
         "code": rng.randrange(200, 600), "latency": round(rng.random() * 100, 3),
         "svc": {"name": rng.choice(["api", "db", "cache"]), "pod": "pod-%d" % rng.randrange(1000)},
         "path": "/v1/items/%d?x=%d" % (rng.randrange(10 ** 5), rng.randrange(100)), "bytes": rng.randrange(10 ** 6)}
    base.append(json.dumps(d).encode() + b"\n")
data = b"".join(base) * ((nl + len(base) - 1) // len(base))
off = g.split_lines(data)
nl = len(off) - 1
L = g.lib()
d_data = L.flbgpu_dev_alloc(len(data) + 16); d_off = L.flbgpu_dev_alloc(off.nbytes)
L.flbgpu_memcpy_h2d(d_data, data, len(data)); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
chunk = g.DevChunk(d_data, d_off, nl, len(data))
pk = g.JsonPacker()
fg1 = g.FilterGrep(b.GREP32_REGEX, "OR"); fg2 = g.FilterGrep(b.GREP32_EXCLUDE, "OR")
ch = g.FilterChain([fg1, fg2])
ev = pk.run_dev(chunk, events=True, ts=(1, 0))
ch.filter_dev(ev)
torch.cuda.synchronize()
fg1.profile(True); fg2.profile(True)
t0 = time.perf_counter()
for _ in range(steps):
    r, o = ch.filter_dev(ev)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
p1, p2 = fg1.profile_read(), fg2.profile_read()
st = ch.last_stats()
print(json.dumps({"lines": nl, "event_bytes": int(ev.bytes), "ms_per_step": round(dt * 1e3, 3),
                  "lds_rules": not os.environ.get("FLBGPU_GREP_NO_LDS_RULES"),
                  "kept": [int(s["out_records"]) for s in st],
                  "regex_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in p1.items()},
                  "exclude_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in p2.items()}}))
