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
    # time / record functions next to aggregates (package_results, flb_sp.c:1199-1206)
    "SELECT host, NOW() AS n, COUNT(*), RECORD_TAG(), UNIX_TIMESTAMP(), RECORD_TIME() AS rt FROM STREAM:x WINDOW TUMBLING (5 SECOND) GROUP BY host;",
]


# HOPPING windows (sql.y:275-278): the hop timer closes a slot every ADVANCE BY, the window timer packages and drops the oldest.
# Number GROUP BY keys only: a slot's nodes share the string of a string key with the window's node (memcpy of groupby_nums,
# flb_sp.c:1985) and both free it -- the reference binary dies with "double free" on such queries.
HOPPING_QUERIES = [
    "SELECT COUNT(*), AVG(latency), SUM(latency), MIN(latency), MAX(latency) FROM STREAM:x WINDOW HOPPING (5 SECOND, ADVANCE BY 1 SECOND);",
    "SELECT code, COUNT(*), AVG(latency), SUM(bytes), MIN(latency), MAX(bytes) FROM STREAM:x WINDOW HOPPING (1 MINUTE, ADVANCE BY 20 SECOND) GROUP BY code;",
    "SELECT status, COUNT(*) AS n, SUM(bytes) AS b, AVG(bytes) FROM TAG:'app.*' WINDOW HOPPING (10 SECOND, ADVANCE BY 5 SECOND) WHERE status >= 200 GROUP BY status;",
    "SELECT code, svc['n'], COUNT(latency), SUM(latency), AVG(status) FROM STREAM:x WINDOW HOPPING (1 HOUR, ADVANCE BY 1 MINUTE) WHERE latency IS NOT NULL GROUP BY code, svc['n'];",
]


def hopping_schedule(rng, steps):
    """a timer schedule as the engine produces it (flb_sp_fd_event): 'c' = a chunk arrives, 'h' = the hop timer, 't' = the
    window timer; a hop timer precedes every window timer but the order of the two at one instant is the event loop's"""
    ev = []
    for _ in range(steps):
        ev += ["c"] * rng.choice([0, 1, 1, 2])
        r = rng.random()
        ev += ["h"] if r < 0.45 else ["h", "t"] if r < 0.8 else ["t", "h"] if r < 0.9 else ["t"]
    return ev


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
        if not clean and rng.random() < 0.05:
            d[rng.choice([b"host", b"bytes", b"latency", b"svc"])] = rng.choice([3, "bin key", {"name": "x", "n": 9}])   # a bin key: flb_sp_key.c:189
        items = list(d.items())
        rng.shuffle(items)
        body = msgpack.packb(dict(items), use_bin_type=True)
        if not clean and items and rng.random() < 0.05 and len(items) < 15:
            body = bytes([0x80 | (len(items) + 1)]) + body[1:] + msgpack.packb(items[0][0]) + msgpack.packb(7)
        ts = b"\xd7\x00" + struct.pack(">II", 1700000000 + i, rng.randrange(10 ** 9))
        out += b"\x92\x92" + ts + b"\x80" + body
    return bytes(out)


def config4_chunk(n, seed=0x5ca1e, base_sec=1700000000):
    """BASELINE configs[4]'s shape, vectorised: n fixed-layout V2 records
    [[ts, {}], {"status": u16, "latency": f64, "bytes": u32, "host": "hNN"}] (status 70 % 200, the rest 301 / 404 / 500)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    head = b"\x92\x92\xd7\x00" + b"\0" * 8 + b"\x80\x84"
    parts = [head, b"\xa6status\xcd", b"\0\0", b"\xa7latency\xcb", b"\0" * 8, b"\xa5bytes\xce", b"\0" * 4, b"\xa4host\xa3h00"]
    tmpl = b"".join(parts)
    L = len(tmpl)
    a = np.tile(np.frombuffer(tmpl, dtype=np.uint8), (n, 1))
    o_ts = 4
    o_status = len(parts[0]) + len(parts[1])
    o_lat = o_status + 2 + len(parts[3])
    o_bytes = o_lat + 8 + len(parts[5])
    o_host = o_bytes + 4 + len(parts[7]) - 2
    sec = (base_sec + np.arange(n) // 1000).astype(">u4")
    a[:, o_ts:o_ts + 4] = sec.view(np.uint8).reshape(n, 4)
    st = rng.choice(np.array([200, 200, 200, 200, 200, 200, 200, 301, 404, 500], dtype=">u2"), n)
    a[:, o_status:o_status + 2] = st.view(np.uint8).reshape(n, 2)
    lat = (rng.random(n) * 250.0).astype(">f8")
    a[:, o_lat:o_lat + 8] = lat.view(np.uint8).reshape(n, 8)
    by = rng.integers(0, 1 << 20, n).astype(">u4")
    a[:, o_bytes:o_bytes + 4] = by.view(np.uint8).reshape(n, 4)
    h = rng.integers(0, 64, n)
    a[:, o_host] = 48 + h // 10
    a[:, o_host + 1] = 48 + h % 10
    off = (np.arange(n + 1, dtype=np.uint64) * L)
    return a.reshape(-1), off


CONFIG4_SQL = "SELECT status, COUNT(*), AVG(latency) FROM STREAM:x WINDOW TUMBLING (60 SECOND) GROUP BY status;"


# SELECTs without aggregation functions: flb_sp_do's other branch, sp_process_data (flb_sp.c:1607-1850)
SELECT_QUERIES = [
    "SELECT * FROM STREAM:x;",
    "SELECT host, status FROM STREAM:x;",
    "SELECT host AS h, svc['name'], svc['n'] AS n, svc FROM TAG:'app.*' WHERE status >= 200;",
    "SELECT *, host, bytes AS b FROM STREAM:x WHERE @record.contains(host) AND NOT (latency < 50 OR flag);",
    "SELECT latency, blob, tags, big, nope FROM STREAM:x WHERE latency IS NOT NULL;",
    "SELECT svc['name']['deep'], svc['missing'], host['x'] FROM STREAM:x;",
    "CREATE STREAM sel WITH (tag='sel.out') AS SELECT code, host FROM STREAM:x WHERE @record.time() > 1700000005.5 OR host = 'cc';",
    "SELECT status, host FROM STREAM:x WINDOW TUMBLING (5 SECOND) WHERE code < 500.5 AND code GROUP BY status;",
    "SELECT *, k00, k01, k02 FROM STREAM:x;",
    "SELECT NOW(), UNIX_TIMESTAMP() AS u, RECORD_TAG(), RECORD_TIME() AS rt, host FROM STREAM:x WHERE status >= 200;",
    "SELECT RECORD_TIME() FROM STREAM:x;",
]


def select_chunk(rng, n, legacy=False):
    """n records for the projections: every value class msgpack_pack_object re-packs (ints in every width -- also wider than
    needed --, float32 / float64, str8 of a short string, bin, ext, arrays, nested maps), duplicate keys, maps of more than 15
    entries (a map16 header to patch; a fixmap of 14 that `*` plus named keys push past 15), non-string keys"""
    out = bytearray()
    for i in range(n):
        pairs = []
        def add(k, v):
            pairs.append(msgpack.packb(k, use_bin_type=True) + v)
        P = lambda v: msgpack.packb(v, use_bin_type=True)
        if rng.random() < 0.9:
            v = rng.choice([200, 404, 500, 301, "200", True])
            add("status", rng.choice([P(v), b"\xcd" + struct.pack(">H", v) if isinstance(v, int) and not isinstance(v, bool) else P(v)]))
        if rng.random() < 0.85:
            v = rng.choice([rng.random() * 100, rng.randrange(100), -3, None, "12", 2 ** 63 + 5, -2 ** 63])
            add("latency", rng.choice([P(v), b"\xca" + struct.pack(">f", 1.5), b"\xd3" + struct.pack(">q", -7), b"\xcf" + struct.pack(">Q", 9)]))
        if rng.random() < 0.9:
            add("bytes", P(rng.choice([rng.randrange(10 ** 6), 7, 70000, 2 ** 33])))
        if rng.random() < 0.8:
            add("svc", P(rng.choice([{"name": rng.choice(["api", "db"]), "n": rng.randrange(3)}, {"name": {"deep": rng.choice([1, "x", [2]])}}, {}, "flat", {"n": None}])))
        if rng.random() < 0.7:
            h = rng.choice(["a", "b", "cc", "", "h" * 40])
            add("host", rng.choice([P(h), b"\xd9" + bytes([len(h)]) + h.encode(), b"\xda" + struct.pack(">H", len(h)) + h.encode()]))
        if rng.random() < 0.15:
            add("flag", P(rng.choice([True, False, None])))
        if rng.random() < 0.8:
            add("code", P(rng.choice([200, 404, 503, "200", 0, -1])))
        if rng.random() < 0.3:
            add("blob", P(rng.choice([b"\x00\x01\x02", b"", b"x" * 300])))
        if rng.random() < 0.3:
            add("tags", P(rng.choice([[1, "a", [2.5]], [], [{"k": 1}]])))
        if rng.random() < 0.2:
            add("big", rng.choice([P(msgpack.ExtType(5, b"12345678")), b"\xc7\x03\x7d123", b"\xc7\x04\x07abcd", P(msgpack.ExtType(1, b"z" * 20))]))
        if rng.random() < 0.06:
            add(99, P(1))
        if rng.random() < 0.06:
            add(b"host", P("a bin key"))
        if rng.random() < 0.1 and pairs:
            pairs.append(rng.choice(pairs))                             # a duplicate key
        r = rng.random()
        if r < 0.12:
            for j in range(rng.choice([14, 16, 20]) - len(pairs)):
                add("k%02d" % j, P(j))
        rng.shuffle(pairs)
        cnt = len(pairs)
        body = b"".join(pairs)
        if cnt < 16 and rng.random() < 0.9:
            hdr = bytes([0x80 | cnt])
        elif cnt < 65536 and rng.random() < 0.9:
            hdr = b"\xde" + struct.pack(">H", cnt)
        else:
            hdr = b"\xdf" + struct.pack(">I", cnt)
        if legacy:
            t = rng.choice([b"\xd7\x00" + struct.pack(">II", 1700000000 + i, 5), P(1700000000 + i), b"\xcb" + struct.pack(">d", 1700000000.25 + i)])
            out += b"\x92" + t + hdr + body
        else:
            ts = b"\xd7\x00" + struct.pack(">II", 1700000000 + i, rng.randrange(10 ** 9))
            meta = rng.choice([b"\x80", b"\x81\xa1m\x01", b"\xde\x00\x00"])
            out += b"\x92\x92" + ts + meta + hdr + body
    return bytes(out)
