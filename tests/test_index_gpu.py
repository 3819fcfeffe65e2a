"""Device record indexer (flbgpu_index_dev) against the sequential host walk (flbgpu_index_host =
what msgpack_unpack_next yields, src/flb_log_event_decoder.c:296-333): same boundaries, same
`consumed`, on clean chunks and on every way the speculation can be misled."""
import random
import struct

import msgpack
import numpy as np
import pytest

import flbamd_loader
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


def event(body, sec=1700000000, nsec=5, meta=None):
    ts = msgpack.ExtType(0, struct.pack(">II", sec, nsec))
    return msgpack.packb([[ts, meta or {}], body], use_bin_type=True)


def check(g, blob, ix=None):
    ix = ix or g.Indexer()
    n0, off0, c0 = g.index_host(blob)
    n1, off1, c1 = ix.index(blob)
    assert (n1, c1) == (n0, c0), (n0, c0, n1, c1, ix.stats())
    assert np.array_equal(np.asarray(off0[: n0 + 1], dtype=np.uint64), off1)
    return ix.stats()


def test_apache_chunk(g):
    data, off, ep = synth.apache_records(20000)
    st = check(g, bytes(data))
    assert st["rounds"] == 1 and st["off_chain_rows"] == 0


def test_empty_and_tiny(g):
    ix = g.Indexer()
    check(g, b"", ix)
    check(g, event({"a": 1}), ix)
    check(g, b"\x92", ix)
    check(g, b"\xc1", ix)
    check(g, b"\x01", ix)
    check(g, event({"a": 1}) + b"\x92", ix)


def test_spurious_candidates(g):
    rnd = random.Random(7)
    recs = []
    for i in range(30000):
        body = {"i": rnd.getrandbits(64), "b": bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(0, 40))),
                "s": "‒ܒ" * rnd.randrange(0, 5), "x": b"\x92" * rnd.randrange(0, 70),
                "n": [rnd.randrange(0x90, 0x94) for _ in range(rnd.randrange(0, 6))]}
        recs.append(event(body, sec=rnd.getrandbits(32), nsec=rnd.getrandbits(32)))
    st = check(g, b"".join(recs))
    assert st["candidates"] > 2 * 30000


def test_embedded_records_in_strings(g):
    """payloads that are themselves valid chains of events: the false chain runs inside the string and
    (when the string is the last value) rejoins the true chain at the next record"""
    rnd = random.Random(11)
    inner = [event({"k": "v" * rnd.randrange(0, 30)}) for _ in range(50)]
    recs = []
    for i in range(8000):
        fake = b"".join(rnd.sample(inner, rnd.randrange(1, 6)))
        if i % 3 == 0:
            body = {"log": fake}                               # ends exactly where the record ends
        elif i % 3 == 1:
            body = {"log": fake, "tail": i}
        else:
            body = {"a": fake[:-1], "b": fake[1:], "c": fake + b"\x92\x92"}
        recs.append(event(body))
    check(g, b"".join(recs))


def test_records_that_are_no_candidates(g):
    rnd = random.Random(3)
    parts = []
    for i in range(6000):
        r = rnd.random()
        if r < 0.02:
            parts.append(msgpack.packb({"plain": "map", "i": i}))            # not an event at all
        elif r < 0.04:
            parts.append(b"\xdc\x00\x02" + event({"a": i})[1:])              # array16 head
        elif r < 0.05:
            parts.append(msgpack.packb(i))
        elif r < 0.06:
            parts.append(msgpack.packb("str" * rnd.randrange(1, 20)))
        else:
            parts.append(event({"log": "x" * rnd.randrange(0, 200), "i": i}))
    ix = g.Indexer()
    st = check(g, b"".join(parts), ix)
    assert st["off_chain_rows"] > 100 and st["rounds"] > 10
    # starts off the chain
    check(g, msgpack.packb(5) + msgpack.packb({"a": 1}) + b"".join(parts[:200]), ix)
    # nothing is a candidate
    check(g, b"".join(msgpack.packb(i) for i in range(300)), ix)


def test_long_records(g):
    big = {("k%d" % i): i for i in range(3000)}                             # > the lane's object budget
    parts = [event({"a": 1}), event(big), event({"b": 2}), event({"n": [[1, 2, 3]] * 3000}), event({"c": 3})]
    st = check(g, b"".join(parts))
    assert st["off_chain_rows"] == 2
    check(g, event(big))
    check(g, event(big)[:-5])


@pytest.mark.parametrize("cut", [1, 2, 5, 12, 13, 14, 40])
def test_truncated_tail(g, cut):
    recs = [event({"log": "y" * 50, "i": i}) for i in range(5000)]
    blob = b"".join(recs)
    check(g, blob[:-cut])


def test_garbage(g):
    recs = [event({"log": "y" * (i % 90), "i": i}) for i in range(9000)]
    ix = g.Indexer()
    check(g, b"".join(recs) + b"\xc1\xc1\xc1", ix)
    check(g, b"".join(recs[:4000]) + b"\xc1" + b"".join(recs[4000:]), ix)
    check(g, b"".join(recs[:4000]) + b"\x92\xc1" + b"".join(recs[4000:]), ix)
    check(g, b"".join(recs[:4000]) + b"\xdd\xff\xff\xff\xff" + b"".join(recs[4000:]), ix)
    check(g, b"\xc1" + b"".join(recs[:100]), ix)


def test_random_fuzz(g):
    rnd = random.Random(99)
    ix = g.Indexer()
    for it in range(12):
        parts = []
        for i in range(rnd.randrange(1, 3000)):
            r = rnd.random()
            if r < 0.9:
                parts.append(event({"log": bytes(rnd.choice(b"\x92\x92\xd7\x00ab\x80\x81\xa1") for _ in range(rnd.randrange(0, 60)))},
                                   nsec=rnd.choice([0x92929292, 0x92, 5])))
            elif r < 0.95:
                parts.append(bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 20))))
            else:
                parts.append(msgpack.packb([rnd.getrandbits(16), {"a": None}]))
        check(g, b"".join(parts), ix)


def test_host_level_call_uses_device_indexer(g):
    """chunks above the staging threshold are indexed on the device inside flbgpu_filter_run"""
    import oracle_binding as ob
    data, off, ep = synth.apache_records(60000)                            # ~16.6 MB
    blob = bytes(data)
    fg = g.FilterGrep([("regex", r"log \" 5\d\d ")])
    og = ob.Grep([("regex", r"log \" 5\d\d ")])
    assert fg.filter(blob) == og.filter(blob)
    # ... and with a record that is no candidate plus a truncated tail
    cutpos = int(off[20000])
    blob2 = blob[:cutpos] + msgpack.packb({"x": 1}) + blob[cutpos:-7]
    assert fg.filter(blob2) == og.filter(blob2)


def test_device_filters_take_raw_chunks(g):
    """flbgpu_filter_run_dev / chain_run_dev with row_off == NULL index the chunk themselves"""
    import oracle_binding as ob
    L = g.lib()
    recs = [event({"log": "GET /x %d" % i, "code": str(200 + (i * 7) % 400)}) for i in range(30000)]
    for blob in (b"".join(recs), b"".join(recs)[:-3], b"".join(recs[:100]) + msgpack.packb(7) + b"".join(recs[100:]), b"\xc1"):
        d = L.flbgpu_dev_alloc(len(blob) + 16)
        L.flbgpu_memcpy_h2d(d, blob, len(blob))
        fg = g.FilterGrep([("regex", r"code ^5\d\d$")])
        fm = g.FilterLogToMetrics("counter", [("label_field", "code")])
        raw = g.DevChunk(d, None, 0, len(blob))
        r, o = fg.filter_dev(raw)
        want_r, want = ob.Grep([("regex", r"code ^5\d\d$")]).filter(blob)
        assert r == want_r
        if r == g.MODIFIED:
            got = np.empty(int(o.bytes), dtype=np.uint8)
            L.flbgpu_memcpy_d2h(got.ctypes.data, o.data, int(o.bytes))
            assert got.tobytes() == want
        om = ob.L2M("counter", [("label_field", "code")])
        assert fm.filter_dev(raw)[0] == om.filter(blob)
        assert [(x["labels"], x["value"]) for x in fm.snapshot()] == [(x["labels"], x["value"]) for x in om.snapshot()[2]]
        ch = g.FilterChain([g.FilterGrep([("exclude", "code ^2")]), fg])
        r3, o3 = ch.filter_dev(raw)
        q1, w1 = ob.Grep([("exclude", "code ^2")]).filter(blob)
        q3, w3 = ob.Grep([("regex", r"code ^5\d\d$")]).filter(w1 if q1 == g.MODIFIED else blob)
        assert r3 == (g.MODIFIED if g.MODIFIED in (q1, q3) else g.NOTOUCH)
        L.flbgpu_dev_free(d)
