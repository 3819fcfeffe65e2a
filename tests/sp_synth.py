"""seeded chunks and queries for the stream-processor tests (hostile value mix: numeric strings, floats, nested maps, duplicates)"""
import random
import struct

import msgpack

QUERIES = [
    "SELECT COUNT(*), AVG(latency), SUM(latency), MIN(latency), MAX(latency) FROM STREAM:x;",
    "SELECT host, COUNT(*), AVG(latency), SUM(bytes), MIN(latency), MAX(bytes) FROM STREAM:x WINDOW TUMBLING (1 MINUTE) GROUP BY host;",
    "SELECT status, COUNT(*) AS n, AVG(bytes) AS avg_bytes FROM TAG:'app.*' WINDOW TUMBLING (5 SECOND) WHERE status >= 200 GROUP BY status;",
    "SELECT svc['name'], COUNT(*), MAX(svc['n']) AS m FROM STREAM:x WHERE @record.contains(host) AND NOT (latency < 50 OR flag) GROUP BY svc['name'];",
    "SELECT COUNT(*), SUM(status) FROM STREAM:x WHERE status = 200 OR host > 'a' AND @record.time() > 1700000005.5;",
    "SELECT host, svc['name'], COUNT(latency), AVG(status) FROM STREAM:x WHERE latency IS NOT NULL GROUP BY host, svc['name'];",
    "SELECT MIN(bytes), MAX(bytes), SUM(bytes), AVG(bytes) FROM STREAM:x WHERE host = 'cc' OR host <= 'a' OR bytes <> 7;",
    "CREATE STREAM agg WITH (tag='agg.out') AS SELECT host, COUNT(*) FROM STREAM:x WHERE flag IS NULL AND host != '' GROUP BY host;",
    "SELECT code, COUNT(*), SUM(code) FROM STREAM:x WINDOW TUMBLING (2 HOUR) WHERE code < 500.5 AND code GROUP BY code;",
]


def chunk(rng, n, clean=False):
    """n V2 records [[ts, {}], body]; clean = one value class per GROUP BY column and finite sums"""
    out = bytearray()
    for i in range(n):
        d = {}
        if rng.random() < 0.95:
            d["status"] = rng.choice([200, 200, 404, 500, 301] if clean else [200, 200, 404, 500, "200", "404", 301, True, " 7", "12x"])
        if rng.random() < 0.9:
            pool = [rng.random() * 100, rng.randrange(100), 0.0, 0, -3, 2.5]
            if not clean:
                pool += ["%.3f" % (rng.random() * 10), "12", "1e3", "1.5e3", ".5", "5.", "x.y", "0x1.8", None, [1], {"a": 1}, "", "-0.0",
                         "123456789012345678901", "1.7976931348623157e309", "9223372036854775808", 2 ** 63 + 5, -2 ** 63, 1e300]
            d["latency"] = rng.choice(pool)
        if rng.random() < 0.9:
            d["bytes"] = rng.choice([rng.randrange(10 ** 6), rng.randrange(100), 7])
        if rng.random() < 0.8:
            d["svc"] = {"name": rng.choice(["api", "db", "cache"]), "n": rng.randrange(3)}
        if rng.random() < 0.7:
            d["host"] = rng.choice(["a", "b", "cc", ""])
        if rng.random() < 0.1:
            d["flag"] = rng.choice([True, False, None])
        if rng.random() < 0.9:
            d["code"] = rng.choice([200, 404, 503]) if clean else rng.choice([200, 404, 503, "200", "503", 0, -1])
        if not clean and rng.random() < 0.05:
            d[5] = 1
        items = list(d.items())
        rng.shuffle(items)
        body = msgpack.packb(dict(items), use_bin_type=True)
        if not clean and items and rng.random() < 0.05 and len(items) < 15:
            body = bytes([0x80 | (len(items) + 1)]) + body[1:] + msgpack.packb(items[0][0]) + msgpack.packb(7)
        ts = b"\xd7\x00" + struct.pack(">II", 1700000000 + i, rng.randrange(10 ** 9))
        out += b"\x92\x92" + ts + b"\x80" + body
    return bytes(out)
