"""Two filter_grep instances that follow each other in a chain run as ONE pass on a device-resident chunk (flbgpu.cpp run_grep_pair).
flb_filter_do (src/flb_filter.c:121-325) hands the second instance what the first keeps; what leaves the pair, and what each
instance reports on its own (MODIFIED / NOTOUCH, records in and out, bytes out: plugins/filter_grep/grep.c:356-385), must be what
the two calls one after the other give -- compared here against the oracle's filters AND against the library's own one-by-one calls."""
import random
import numpy as np
import pytest
import oracle_binding as ob
import flbamd_loader
from test_gpu_parity import _rec, first_diff, oracle_chain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    return flbamd_loader.load()


@pytest.fixture(autouse=True)
def _no_ahead_launch(monkeypatch):
    # a chunk of up to 8 MB is launched ahead of its sizes (flbgpu.cpp SpecCall), its instances one by one; the pair is what larger
    # chunks take -- the small cases below go the large chunks' way (the library reads the variable at every call)
    monkeypatch.setenv("FLBGPU_NO_SPEC", "1")


def _records(n, seed, odd=True):
    rng = random.Random(seed)
    recs = []
    for i in range(n):
        d = {b"level": rng.choice([b"info", b"warn", b"error", b"debug", b""]),
             b"msg": b"request %d finished %s" % (rng.randrange(10 ** 6), rng.choice([b"ok", b"timeout", b"refused"])),
             b"code": rng.randrange(200, 600),
             b"svc": {b"name": rng.choice([b"api", b"db", b"cache"]), b"pod": b"pod-%d" % rng.randrange(100)},
             b"path": b"/v%d/items/%d" % (rng.choice([1, 1, 2]), rng.randrange(10 ** 4))}
        if odd:
            x = rng.random()
            if x < 0.02: d.pop(b"msg")
            elif x < 0.04: d[b"level"] = 7                              # not a string: the rule sees no value
            elif x < 0.06: d[b"level"] = b"info"; d[b"msg"] = b"x" * 700       # str16
            elif x < 0.08: d = {b"level": b"warn", **d, b"level ": b"zz"}
            elif x < 0.09: d[b"msg"] = {b"nested": b"timeout"}
        recs.append(_rec(d, 5, i))
    return recs


def _upload(g, recs, tail=b""):
    blob = b"".join(recs) + tail
    off = np.zeros(len(recs) + 1, dtype=np.uint64)
    np.cumsum(np.fromiter((len(r) for r in recs), dtype=np.uint64, count=len(recs)), out=off[1:])
    L = g.lib()
    d_data = L.flbgpu_dev_alloc(len(blob) + 16); d_off = L.flbgpu_dev_alloc(off.nbytes)
    assert d_data and d_off
    L.flbgpu_memcpy_h2d(d_data, blob, len(blob)); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
    return g.DevChunk(d_data, d_off, len(recs), len(b"".join(recs))), (d_data, d_off), blob


def _download(g, ch):
    L = g.lib()
    out = np.empty(max(int(ch.bytes), 1), dtype=np.uint8)
    if ch.bytes:
        L.flbgpu_memcpy_d2h(out.ctypes.data, ch.data, int(ch.bytes))
    off = np.zeros(int(ch.n) + 1, dtype=np.uint64)
    L.flbgpu_memcpy_d2h(off.ctypes.data, ch.row_off, off.nbytes)
    return bytes(out[: int(ch.bytes)]), off


CASES = [
    # (rules of the first instance, op, rules of the second, op)
    ([("regex", "level ^(info|warn|error)$")], None, [("exclude", "msg timeout")], None),                 # both drop
    ([("regex", "level ."), ("regex", "code ^[2-5]")], "OR", [("exclude", "msg refused"), ("exclude", "path ^/v2")], "OR"),
    ([("regex", "code .")], None, [("exclude", "msg timeout")], None),                                    # code is an integer: the rule sees no value, the first instance drops everything
    ([("regex", "path ^/v")], None, [("exclude", "level ^debug$")], None),                                # the first keeps everything: NOTOUCH
    ([("exclude", "level ^debug$")], None, [("regex", "path ^/v")], None),                                # the second keeps everything it gets
    ([("regex", "path ^/v")], None, [("regex", "path items")], None),                                     # neither changes anything
    ([("regex", "level ^nothing$")], None, [("regex", "path .")], None),                                  # the first drops everything
    ([("regex", "level ^info$")], None, [("regex", "level ^warn$")], None),                               # the second drops everything
    ([("regex", "level ^info$"), ("regex", "msg ok$")], "AND", [("regex", "level ^info$"), ("exclude", "msg 1")], None),   # shared names, legacy list
    ([("regex", "$svc['name'] ^(api|db)$")], None, [("exclude", "$svc['pod'] 7$"), ("exclude", "level warn")], "OR"),       # sub-keys
    ([("regex", "a x"), ("regex", "b x"), ("regex", "c x"), ("regex", "d x"), ("regex", "level .")], "OR",
     [("exclude", "e x"), ("exclude", "f x"), ("exclude", "g x"), ("exclude", "h x"), ("exclude", "i x")], "OR"),            # ten names: more than the walk's slots
]


def _run(g, recs, r1, op1, r2, op2, tail=b""):
    chunk, bufs, blob = _upload(g, recs, tail)
    f1, f2 = g.FilterGrep(r1, op1), g.FilterGrep(r2, op2)
    ch = g.FilterChain([f1, f2])
    ret, out = ch.filter_dev(chunk)
    stats = ch.last_stats()
    got = (ret, _download(g, out)) if ret == g.MODIFIED else (ret, None)
    counts = (f1.counts(), f2.counts())
    # the same instances' rules, one call after the other (flbgpu_filter_run_dev: no pair there)
    s1, s2 = g.FilterGrep(r1, op1), g.FilterGrep(r2, op2)
    ra, oa = s1.filter_dev(chunk)
    cur = oa if ra == g.MODIFIED else chunk
    seq_stats = [dict(ret=ra, in_records=s1.counts()[0], out_records=s1.counts()[1] if ra == g.MODIFIED else s1.counts()[0],
                      out_bytes=int(cur.bytes))]
    if ra == g.MODIFIED and cur.bytes == 0:
        seq_stats.append(dict(ret=0, in_records=0, out_records=0, out_bytes=0))
        seq = (g.MODIFIED, _download(g, cur))
    else:
        rb, obb = s2.filter_dev(cur)
        cur2 = obb if rb == g.MODIFIED else cur
        seq_stats.append(dict(ret=rb, in_records=s2.counts()[0], out_records=s2.counts()[1] if rb == g.MODIFIED else s2.counts()[0],
                              out_bytes=int(cur2.bytes)))
        seq = (g.MODIFIED, _download(g, cur2)) if (ra == g.MODIFIED or rb == g.MODIFIED) else (g.NOTOUCH, None)
    want = oracle_chain([ob.Grep(r1, op1), ob.Grep(r2, op2)], blob)
    for f in (f1, f2, s1, s2):
        f.close()
    L = g.lib()
    L.flbgpu_dev_free(bufs[0]); L.flbgpu_dev_free(bufs[1])
    return got, stats, seq, seq_stats, want, counts


@pytest.mark.parametrize("n", [1, 63, 3000, 70000])
def test_pair_is_what_the_two_calls_give(g, n):
    recs = _records(n, seed=n)
    for r1, op1, r2, op2 in CASES:
        got, stats, seq, seq_stats, want, counts = _run(g, recs, r1, op1, r2, op2)
        assert got[0] == seq[0] == want[0], (r1, r2, stats, seq_stats)
        assert stats == seq_stats, (r1, r2)
        if got[0] == g.MODIFIED:
            assert got[1][0] == seq[1][0] == want[1], (r1, r2, first_diff(want[1], got[1][0]))
            assert (got[1][1] == seq[1][1]).all(), (r1, r2)
        assert counts[0] == (stats[0]["in_records"], stats[0]["out_records"])
        assert counts[1] == (stats[1]["in_records"], stats[1]["out_records"])


def test_pair_with_a_decoder_error_and_with_emptied_rows(g):
    recs = _records(5000, seed=3)
    r1, r2 = [("regex", "level ^(info|warn)$")], [("exclude", "msg timeout")]
    bad = list(recs)
    bad[2500] = b"\x93\x01\x02\x03"                                 # not a log event: both instances answer NOTOUCH (grep.c:356-385)
    got, stats, seq, seq_stats, want, _ = _run(g, bad, r1, None, r2, None)
    assert got[0] == seq[0] == want[0] == g.NOTOUCH
    assert stats == seq_stats
    # rows emptied by an instance in front, a group marker pair, an empty map
    odd = list(recs[:2000])
    odd[10] = _rec({}, 5, 10)
    for i in range(100, 400):
        odd[i] = b""
    got, stats, seq, seq_stats, want, _ = _run(g, odd, r1, None, r2, None)
    assert got[0] == seq[0] == g.MODIFIED and stats == seq_stats
    assert got[1][0] == seq[1][0] == want[1]


def test_pair_inside_a_longer_chain(g):
    recs = _records(20000, seed=11)
    chunk, bufs, blob = _upload(g, recs)
    rules = [[("regex", "level ^(info|warn|error)$")], [("exclude", "msg timeout")], [("regex", "path ^/v1")], [("exclude", "$svc['name'] ^db$")]]
    fs = [g.FilterGrep(r) for r in rules]
    ch = g.FilterChain(fs)
    fs[0].profile(True); fs[2].profile(True)
    ret, out = ch.filter_dev(chunk)
    # (the four instances ran as two passes)
    assert "k_grep_lane(two instances)" in fs[0].profile_read() and "k_grep_lane(two instances)" in fs[2].profile_read()
    want = oracle_chain([ob.Grep(r) for r in rules], blob)
    assert ret == want[0] == g.MODIFIED
    got, _ = _download(g, out)
    assert got == want[1], first_diff(want[1], got)
    st = ch.last_stats()
    work = blob
    for i, r in enumerate(rules):
        rr, work2 = ob.Grep(r).filter(work)
        assert st[i]["ret"] == rr
        if rr == ob.MODIFIED:
            assert st[i]["out_records"] == ob.count_records(work2) and st[i]["out_bytes"] == len(work2)
            work = work2
    for f in fs:
        f.close()
    L = g.lib()
    L.flbgpu_dev_free(bufs[0]); L.flbgpu_dev_free(bufs[1])
