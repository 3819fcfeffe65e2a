"""the stream processor's aggregate queries on the device (flbgpu_sp_* through the C ABI) against the reference:
committed answers of the reference's own flb_sp (tests/golden/sp_cases.json), the oracle restatement (oracle/osp.py) on seeded
hostile chunks, and -- at the bench size -- the reference binary itself (oracle/_ref/ref_sp) plus size-independent properties.

Bar: byte-identical records (group order, key names, value types, integers) -- except float32 fields fed by float SUM / AVG,
where the reference adds in arrival order and the device rounds the exact sum once: those may differ by one float32 ULP
(the tolerance north_star states for the sums)."""
import ctypes
import json
import os
import random
import struct
import sys

import msgpack
import pytest

import time as _time
LOCAL_IS_UTC = _time.localtime(86400).tm_gmtoff == 0          # NOW() formats localtime: the committed answers were written under UTC

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import flbamd_loader
import osp
import ref_sp
import sp_synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


def _rows(buf):
    u = msgpack.Unpacker(raw=True, strict_map_key=False)
    u.feed(buf)
    return list(u)


def _f32_ulps(a, b):
    ia, ib = struct.unpack("<i", struct.pack("<f", a))[0], struct.unpack("<i", struct.pack("<f", b))[0]
    ia = ia if ia >= 0 else -(ia & 0x7FFFFFFF)
    ib = ib if ib >= 0 else -(ib & 0x7FFFFFFF)
    return abs(ia - ib)


def assert_same_records(got, want, ctx):
    """identical bytes, or identical structure with float32 values within one ULP"""
    if got == want:
        return 0
    a, b = _rows(got), _rows(want)
    assert len(a) == len(b), ctx
    soft = 0
    for ra, rb in zip(a, b):
        assert ra[0] == rb[0], ctx
        assert list(ra[1].keys()) == list(rb[1].keys()), ctx
        for k in ra[1]:
            va, vb = ra[1][k], rb[1][k]
            assert type(va) is type(vb), (ctx, k, va, vb)
            if isinstance(va, float) and va != vb:
                if va != va and vb != vb:
                    continue
                assert _f32_ulps(va, vb) <= 1, (ctx, k, va, vb)
                soft += 1
            else:
                assert va == vb, (ctx, k, va, vb)
    return soft


def test_reference_answers(g):
    with open(os.path.join(HERE, "golden", "sp_cases.json")) as f:
        cases = json.load(f)
    exact = refused = 0
    for c in cases:
        if "NOW()" in c["sql"] and not LOCAL_IS_UTC:
            continue
        try:
            osp_ok = True
            o = osp.Task(c["sql"], str_conv=c["str_conv"])
            for ch in c["chunks"]:
                o.do(bytes.fromhex(ch))
        except osp.Unsupported:
            osp_ok = False
        t = g.StreamTask(c["sql"], str_conv=c["str_conv"], tag="t")
        try:
            for ch, (ret, out) in zip(c["chunks"], c["do"]):
                rec, got = t.do(bytes.fromhex(ch))
                assert rec == ret, c["sql"]
                assert_same_records(got, bytes.fromhex(out), c["sql"])
            assert_same_records(t.timer(), bytes.fromhex(c["timer"]), c["sql"])
            assert osp_ok, "the device answered a case the oracle refuses: " + c["sql"]
            exact += 1
        except RuntimeError:
            assert not osp_ok and not c["clean"], (c["sql"], g.last_error())
            refused += 1
        finally:
            t.close()
    assert exact >= 24, (exact, refused)


def test_hostile_chunks_against_the_oracle(g):
    rng = random.Random(0xBEE)
    compared = soft = 0
    for q in sp_synth.QUERIES:
        for rep in range(4):
            clean = rep < 2
            conv = rep != 3
            chunks = [sp_synth.chunk(rng, rng.choice([1, 65, 3000]), clean) for _ in range(rng.choice([1, 3]))]
            o = osp.Task(q, str_conv=conv)
            t = g.StreamTask(q, str_conv=conv, tag="t")
            try:
                want = []
                try:
                    for c in chunks:
                        want.append(o.do(c))
                    want_timer = o.timer()
                except osp.Unsupported:
                    assert not clean
                    with pytest.raises(RuntimeError):
                        for c in chunks:
                            t.do(c)
                        t.timer()
                    continue
                for c, (ret, out) in zip(chunks, want):
                    rec, got = t.do(c)
                    assert rec == ret, q
                    soft += assert_same_records(got, out, q)
                soft += assert_same_records(t.timer(), want_timer, q)
                compared += 1
            finally:
                t.close()
    assert compared >= 20
    print("float32 fields off by one ULP:", soft)


def test_window_life_cycle(g):
    """records accumulate over chunks until the timer; an idle timer packages nothing; the next window starts empty"""
    rng = random.Random(5)
    q = "SELECT host, COUNT(*), SUM(bytes) FROM STREAM:x WINDOW TUMBLING (5 SECOND) GROUP BY host;"
    t = g.StreamTask(q)
    o = osp.Task(q)
    assert t.window == "tumbling" and t.window_size == 5 and t.key_names == ["host", "COUNT(*)", "SUM(bytes)"]
    assert t.timer() == b"" == o.timer()
    for rnd in range(3):
        for _ in range(rnd + 1):
            c = sp_synth.chunk(rng, 500, clean=True)
            assert t.do(c) == o.do(c)
        assert t.timer(now=(77, 5)) == o.timer(now=(77, 5))
        assert t.timer() == b""
    t.close()
    s = g.StreamTask("CREATE STREAM agg WITH (tag='agg.out') AS SELECT COUNT(*) FROM TAG:'app.*';")
    assert s.stream_name == "agg" and s.stream_prop("tag") == "agg.out" and s.source_type == "tag" and s.source == "app.*" and s.window == "default"
    s.close()


def test_refusals(g):
    for bad in ["SELECT id, MIN(id) FROM STREAM:FLB;", "SELECT id, * FROM STREAM:FLB;", "SELECT NOW(), * FROM STREAM:FLB;", "SELECT COUNT(*) FROM STREAM:s WINDOW HOPPING (5 SECOND, ADVANCE BY 5 SECOND);"]:
        with pytest.raises(ValueError):
            g.StreamTask(bad)
    # a GROUP BY column that mixes strings and numbers inside one window: the reference's tree comparator is not an order
    t = g.StreamTask("SELECT k, COUNT(*) FROM STREAM:s GROUP BY k;")
    rec = lambda v: b"\x92\xd7\x00" + struct.pack(">II", 1, 0) + msgpack.packb({"k": v})
    with pytest.raises(RuntimeError):
        t.do(rec(1) + rec("abc"))
    t.close()
    t = g.StreamTask("SELECT k, COUNT(*) FROM STREAM:s GROUP BY k;", str_conv=True)
    ret, out = t.do(rec(7) + rec("7") + rec(" 7") + rec(True))
    assert [r[1] for r in _rows(out)] == [{b"k": 7, b"COUNT(*)": 3}, {b"k": 1, b"COUNT(*)": 1}]
    t.close()


def test_truncated_chunk_stops_at_the_first_bad_record(g):
    q = "SELECT COUNT(*), SUM(bytes) FROM STREAM:x;"
    rng = random.Random(9)
    c = sp_synth.chunk(rng, 50, clean=True)
    cut = c[:len(c) - 7]
    t, o = g.StreamTask(q), osp.Task(q)
    assert t.do(cut) == o.do(cut)
    t.close()


@pytest.mark.skipif(not ref_sp.available(), reason="oracle/_ref/ref_sp not built")
def test_bench_size_against_the_reference_binary(g):
    """BASELINE configs[4]: GROUP BY status, AVG(latency) over a tumbling window -- 1 M records in 4 chunks, the reference's own
    flb_sp on the same bytes; plus properties that do not need it (counts add up, merging is chunk-order independent)"""
    rng = random.Random(0xC0FFEE)
    statuses = [200] * 7 + [301, 404, 500]
    chunks = []
    for ci in range(4):
        out = bytearray()
        for i in range(250_000):
            body = {"status": rng.choice(statuses), "latency": rng.random() * 250.0, "bytes": rng.randrange(1 << 20), "host": "h%d" % rng.randrange(64)}
            out += b"\x92\x92\xd7\x00" + struct.pack(">II", 1700000000 + i, 0) + b"\x80" + msgpack.packb(body)
        chunks.append(bytes(out))
    q = "SELECT status, host, COUNT(*), AVG(latency), SUM(bytes), MIN(latency), MAX(bytes) FROM STREAM:x WINDOW TUMBLING (60 SECOND) WHERE status < 500 GROUP BY status, host;"
    r = ref_sp.RefSp(q)
    t = g.StreamTask(q)
    for c in chunks:
        ret_r, _ = r.do(c)
        ret_t, _ = t.do(c)
        assert ret_r == ret_t
    want, got = r.timer(), t.timer()
    r.close()
    soft = assert_same_records(got, want, q)
    rows = _rows(got)
    assert len(rows) == 3 * 64 and sum(x[1][b"COUNT(*)"] for x in rows) == ret_t
    # chunk order does not matter to the aggregate state: same groups, same numbers (first-seen order aside)
    t2 = g.StreamTask(q)
    for c in reversed(chunks):
        t2.do(c)
    rows2 = _rows(t2.timer())
    key = lambda x: (x[1][b"status"], x[1][b"host"])
    assert sorted(rows, key=key) == sorted(rows2, key=key)
    t.close(); t2.close()
    print("float32 fields off by one ULP vs the reference:", soft, "of", 2 * len(rows))


def test_shards_merge_like_one_task(g):
    """one window sharded over ranks: every shard runs on its own records with a disjoint index base; the merge of the exported
    states packages what a single task (and the reference) packages for all the records -- whatever the shard count"""
    rng = random.Random(0x5AD)
    q = "SELECT host, status, COUNT(*), AVG(latency), SUM(bytes), MIN(bytes), MAX(latency) FROM STREAM:x WINDOW TUMBLING (10 SECOND) WHERE status <> 404 GROUP BY host, status;"
    chunks = [sp_synth.chunk(rng, 2000, clean=True) for _ in range(6)]
    o = osp.Task(q)
    for c in chunks:
        o.do(c)
    want = o.timer()
    for nshards in (1, 2, 3):
        tasks = [g.StreamTask(q) for _ in range(nshards)]
        for ci, c in enumerate(chunks):
            t = tasks[ci % nshards]
            t.set_index_base(ci * 2000)
            t.do(c)
        snaps = [t.export() for t in tasks]
        got = tasks[0].package_merged(snaps)
        assert_same_records(got, want, (q, nshards))
        # the order the states arrive in does not matter
        assert tasks[-1].package_merged(snaps[::-1]) == got
        for t in tasks:
            t.close()


def test_rccl_timer_single_rank(g, rccl_ok):
    """flbgpu_sp_timer_all_reduce over a real RCCL communicator (one rank: the exchange is the identity)"""
    rng = random.Random(0xCC1)
    q = "SELECT host, COUNT(*), AVG(latency) FROM STREAM:x WINDOW TUMBLING (5 SECOND) GROUP BY host;"
    c = sp_synth.chunk(rng, 3000, clean=True)
    t, o = g.StreamTask(q), osp.Task(q)
    t.do(c); o.do(c)
    comm = g.RcclComm(1, 0)
    try:
        assert_same_records(t.timer_all_reduce(comm), o.timer(), q)
        assert t.timer() == b""                                     # pruned
    finally:
        comm.close()
        t.close()


def assert_same_hopping_records(got, want, ctx):
    """as assert_same_records; a float32 field fed by a float SUM / AVG may also differ by 1e-6 absolute: both sides subtract
    f64 slot sums from f64 window sums and keep the rounding residue of their own order of additions"""
    if got == want:
        return
    a, b = _rows(got), _rows(want)
    assert len(a) == len(b), ctx
    for ra, rb in zip(a, b):
        assert ra[0] == rb[0] and list(ra[1].keys()) == list(rb[1].keys()), ctx
        for k in ra[1]:
            va, vb = ra[1][k], rb[1][k]
            assert type(va) is type(vb), (ctx, k, va, vb)
            if isinstance(va, float) and va != vb:
                assert _f32_ulps(va, vb) <= 1 or abs(va - vb) <= 1e-6, (ctx, k, va, vb)
            else:
                assert va == vb, (ctx, k, va, vb)


def test_hopping_windows(g):
    """WINDOW HOPPING (n, ADVANCE BY m): chunks, hop timers (flbgpu_sp_hop) and window timers interleaved as the engine's
    event loop would; every packaged record and every window.records equal to the oracle's, which
    tests/test_sp_oracle.py::test_hopping_windows_against_the_reference pins on the reference binary"""
    rng = random.Random(0x40B)
    packaged = 0
    for q in sp_synth.HOPPING_QUERIES:
        for rep in range(4):
            t = g.StreamTask(q)
            o = osp.Task(q)
            assert t.window == "hopping" and (t.window_size, t.window_advance) == (o.q.window_size, o.q.advance_by)
            log = []
            for ev in sp_synth.hopping_schedule(rng, 12):
                log.append(ev)
                if ev == "c":
                    c = sp_synth.chunk(rng, rng.choice([1, 8, 60, 300]), clean=True)
                    assert t.do(c)[0] == o.do(c)[0], (q, "".join(log))
                elif ev == "h":
                    assert t.hop() == 0 == o.hop()
                else:
                    a, b = t.timer(now=(9, rep)), o.timer(now=(9, rep))
                    assert_same_hopping_records(a, b, (q, "".join(log)))
                    packaged += len(a) > 0
            t.close()
    assert packaged > 40
    # a string GROUP BY value: the reference frees the key twice and dies; refused
    t = g.StreamTask("SELECT host, COUNT(*) FROM STREAM:x WINDOW HOPPING (5 SECOND, ADVANCE BY 1 SECOND) GROUP BY host;")
    with pytest.raises(RuntimeError):
        t.do(sp_synth.chunk(rng, 50, clean=True))
    t.close()
    # shards of a hopping window are not exchanged
    t = g.StreamTask(sp_synth.HOPPING_QUERIES[0])
    with pytest.raises(RuntimeError):
        t.export()
    t.close()


# ---- SELECTs without aggregation functions: flb_sp_do's other branch, sp_process_data (flb_sp.c:1607-1850).  Bit-exact.
def test_select_reference_answers(g):
    with open(os.path.join(HERE, "golden", "sp_select_cases.json")) as f:
        cases = json.load(f)
    n = 0
    for c in cases:
        if "NOW()" in c["sql"] and not LOCAL_IS_UTC:
            continue
        t = g.StreamTask(c["sql"], tag="t")
        assert t.select_only and t.window == "default"
        try:
            for ch, (ret, out) in zip(c["chunks"], c["do"]):
                assert t.do(bytes.fromhex(ch)) == (ret, bytes.fromhex(out)), c["sql"]
                n += 1
            assert t.timer() == b""
        finally:
            t.close()
    assert n >= 40


def test_select_hostile_chunks_against_the_oracle(g):
    rng = random.Random(0x5E1)
    compared = out_bytes = 0
    for q in sp_synth.SELECT_QUERIES:
        t = g.StreamTask(q, tag="t")
        o = osp.Task(q)
        try:
            for rep in range(6):
                c = sp_synth.select_chunk(rng, rng.choice([1, 65, 3000]), legacy=rep == 4)
                if rep == 5:
                    c = c[:len(c) - rng.randrange(1, 30)]             # msgpack_unpack_next stops inside the last record
                want = o.do(c)
                assert t.do(c) == want, q
                compared += 1
                out_bytes += len(want[1])
            assert t.do(b"") == (0, b"")
        finally:
            t.close()
    assert compared >= 50 and out_bytes > 1_000_000


def test_select_quirks(g):
    body = {"k%02d" % i: i for i in range(14)}
    rec = b"\x92\x92\xd7\x00" + struct.pack(">II", 7, 0) + b"\x80" + msgpack.packb(body)
    for q in ["SELECT *, k00, k01 FROM STREAM:s;", "SELECT nope FROM STREAM:s;", "SELECT * FROM STREAM:s WHERE k00 = 5;",
              "SELECT k03 AS x, k04, nope FROM STREAM:s WINDOW TUMBLING (5 SECOND) GROUP BY k03;"]:
        t = g.StreamTask(q)
        assert t.do(rec * 3) == osp.Task(q).do(rec * 3), q
        t.close()
    t = g.StreamTask("SELECT *, k00, k01 FROM STREAM:s;")
    assert t.do(rec)[1][13] == 0x90                                   # 0x80 | 16 entries: the reference's fixmap patch, kept
    # a record that is not [time, map]: the reference reads ptr[1] as a map whatever it is -- refused loudly here
    with pytest.raises(RuntimeError):
        t.do(b"\x92\x01\x02")
    t.close()


def test_select_at_size_device_resident(g):
    """BASELINE configs[4]'s record shape, 1 M records resident in HBM: SELECT with WHERE.  The head of the output against the oracle,
    the whole of it through properties (every record of the same length here: count and size follow from the statuses)"""
    import numpy as np
    data, off = sp_synth.config4_chunk(1_000_000)
    q = "SELECT status, host AS h, latency FROM STREAM:x WHERE status >= 400;"
    t = g.StreamTask(q)
    L = g.lib()
    L.flbgpu_dev_alloc.restype = ctypes.c_void_p
    L.flbgpu_dev_alloc.argtypes = [ctypes.c_size_t]
    L.flbgpu_dev_free.argtypes = [ctypes.c_void_p]
    L.flbgpu_memcpy_h2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    d_d = L.flbgpu_dev_alloc(data.nbytes + 16)
    d_o = L.flbgpu_dev_alloc(off.nbytes)
    L.flbgpu_memcpy_h2d(d_d, data.ctypes.data, data.nbytes)
    L.flbgpu_memcpy_h2d(d_o, off.ctypes.data, off.nbytes)
    ret, out = t.do_dev(g.DevChunk(d_d, d_o, 1_000_000, data.nbytes))
    assert t.do(data.tobytes()) == (ret, out)                         # the host entry point indexes the chunk itself: same answer
    # and the result as a device chunk (flbgpu_sp_select_dev): the same bytes, one row per incoming row
    ret_d, dch = t.select_dev(g.DevChunk(d_d, d_o, 1_000_000, data.nbytes))
    assert ret_d == ret and dch.n == 1_000_000 and dch.bytes == len(out)
    hb = np.empty(len(out), dtype=np.uint8)
    L.flbgpu_memcpy_d2h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.flbgpu_memcpy_d2h(hb.ctypes.data, dch.data, hb.nbytes)
    assert hb.tobytes() == out
    ho = np.empty(1_000_001, dtype=np.uint64)
    L.flbgpu_memcpy_d2h(ho.ctypes.data, dch.row_off, ho.nbytes)
    assert int(ho[-1]) == len(out) and bool(np.all(np.diff(ho.astype(np.int64)) >= 0)) and int((np.diff(ho.astype(np.int64)) > 0).sum()) == ret
    L.flbgpu_dev_free(d_d); L.flbgpu_dev_free(d_o)
    raw = data.reshape(1_000_000, -1)
    RL = raw.shape[1]
    st = raw[:, 22].astype(np.uint32) * 256 + raw[:, 23]
    assert ret == int((st >= 400).sum())
    one = len(osp.Task(q).do(data[:RL * 64].tobytes())[1]) // int((st[:64] >= 400).sum())
    assert len(out) == ret * one
    head = 20_000
    want = osp.Task(q).do(data[:RL * head].tobytes())
    assert out[:len(want[1])] == want[1] and want[0] == int((st[:head] >= 400).sum())
    # the last record that leaves is the last record that passes
    last = int(np.nonzero(st >= 400)[0][-1])
    assert out[-one:] == osp.Task(q).do(raw[last].tobytes())[1]
    t.close()
