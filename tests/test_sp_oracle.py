"""the stream-processor restatement (oracle/osp.py) against the reference: committed answers of the reference's own
src/stream_processor (tests/golden/sp_cases.json, written by tests/golden/make_sp_cases.py through oracle/_ref/ref_sp), the
reference's unit-test expectations, and -- when the binary is here -- live fuzzing against it"""
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
import osp
import ref_sp
import sp_synth


def _cases():
    with open(os.path.join(HERE, "golden", "sp_cases.json")) as f:
        return json.load(f)


def _rows(buf):
    u = msgpack.Unpacker(raw=True, strict_map_key=False)
    u.feed(buf)
    return [r[1] for r in u]


def test_oracle_matches_reference_answers():
    n_checked = n_refused = 0
    for c in _cases():
        if "NOW()" in c["sql"] and not LOCAL_IS_UTC:
            continue
        t = osp.Task(c["sql"], str_conv=c["str_conv"])
        try:
            for ch, (ret, out) in zip(c["chunks"], c["do"]):
                got = t.do(bytes.fromhex(ch))
                assert got == (ret, bytes.fromhex(out)), c["sql"]
            assert t.timer() == bytes.fromhex(c["timer"]), (c["sql"], _rows(bytes.fromhex(c["timer"])))
            n_checked += 1
        except osp.Unsupported:
            assert not c["clean"], c["sql"]          # only the hostile chunks may hit the refused corner (mixed group-key classes)
            n_refused += 1
    assert n_checked >= 24, (n_checked, n_refused)


def test_reference_unit_test_expectations():
    """tests/internal/include/sp_cb_functions.h:484-530 (cb_select_aggr) on the shape of data/stream_processor/samples.mp:
    ids 0..10, bytes ten times 10 and one 10.5 -> MIN 0, MAX 10, COUNT 11, SUM(bytes) 110.5, AVG(bytes) 10.04545"""
    chunk = b""
    for i in range(11):
        body = {"id": i, "bytes": 10.5 if i == 10 else (10.0 if i == 1 else 10), "bool": i < 8}
        chunk += b"\x92\xd7\x00" + struct.pack(">II", 1590000000 + i, 0) + msgpack.packb(body)
    t = osp.Task("SELECT MIN(id), MAX(id), COUNT(*), SUM(bytes), AVG(bytes) FROM STREAM:FLB;")
    ret, out = t.do(chunk)
    row = _rows(out)
    assert ret == 11 and len(row) == 1
    assert row[0][b"MIN(id)"] == 0 and row[0][b"MAX(id)"] == 10 and row[0][b"COUNT(*)"] == 11
    assert row[0][b"SUM(bytes)"] == 110.5 and abs(row[0][b"AVG(bytes)"] - 10.04545) < 1e-5
    # sp_select_keys.h:84 "SELECT bool, MIN(id), ... GROUP BY bool": two rows, first-seen order
    t = osp.Task("SELECT bool, MIN(id), MAX(id), COUNT(*), SUM(bytes), AVG(bytes) FROM STREAM:FLB GROUP BY bool;")
    rows = _rows(t.do(chunk)[1])
    assert [r[b"bool"] for r in rows] == [1, 0] and [r[b"COUNT(*)"] for r in rows] == [8, 3]


def test_parser_follows_the_grammar():
    q = osp.parse("select a.b['x']['y'] as k, count(*) , Avg(v) from tag:'t.*' window tumbling (2 minute) "
                  "where not a = 1 and b <> 'it''s' or c group by a.b['x']['y'];")
    assert q.keys[0].out_name == "k" and q.keys[0].gb == 0 and q.keys[1].out_name == "COUNT(*)" and q.keys[2].out_name == "AVG(v)"
    assert q.window == "tumbling" and q.window_size == 120 and q.source_type == "tag" and q.source == "t.*"
    # NOT takes everything to its right; AND / OR associate to the right
    assert q.cond.a[1] == "NOT" and q.cond.a[2].a[1] == "AND" and q.cond.a[2].a[3].a[1] == "OR"
    assert osp.parse("SELECT SUM(x['a']) FROM STREAM:s;").keys[0].out_name == "SUM(x['a'])"
    with pytest.raises(osp.ParseError):
        osp.parse("SELECT a, COUNT(*) FROM STREAM:s;")                 # a plain key that is not a GROUP BY key
    with pytest.raises(osp.ParseError):
        osp.parse("SELECT COUNT(*) FROM STREAM:s WHERE time > 3;")      # TIME is a keyword of the lexer
    q = osp.parse("SELECT COUNT(*) FROM STREAM:s WINDOW HOPPING (5 MINUTE, ADVANCE BY 10 SECOND);")
    assert (q.window, q.window_size, q.advance_by) == ("hopping", 300, 10)
    with pytest.raises(osp.Unsupported):
        osp.parse("SELECT COUNT(*) FROM STREAM:s WINDOW HOPPING (5 SECOND, ADVANCE BY 5 SECOND);")


@pytest.mark.skipif(not ref_sp.available(), reason="oracle/_ref/ref_sp not built")
def test_live_fuzz_against_the_reference():
    rng = random.Random(0xA11CE)
    compared = 0
    for q in sp_synth.QUERIES:
        for rep in range(5):
            clean = rep < 3
            conv = rep != 4
            chunks = [sp_synth.chunk(rng, rng.choice([1, 30, 400]), clean) for _ in range(rng.choice([1, 2, 3]))]
            r = ref_sp.RefSp(q, str_conv=conv)
            assert r.ok
            t = osp.Task(q, str_conv=conv)
            try:
                for c in chunks:
                    assert r.do(c) == t.do(c), q
                assert r.timer() == t.timer(), q
                compared += 1
            except osp.Unsupported:
                assert not clean
            finally:
                r.close()
    assert compared >= 30


@pytest.mark.skipif(not ref_sp.available(), reason="oracle/_ref/ref_sp not built")
def test_hopping_windows_against_the_reference():
    """sp_process_hopping_slot / flb_sp_window_prune's HOPPING branch: chunks, hop timers and window timers interleaved;
    every packaged byte and every record count equal to the reference binary's"""
    rng = random.Random(0x40B)
    compared = packaged = 0
    for q in sp_synth.HOPPING_QUERIES:
        for rep in range(8):
            clean = rep < 6
            r = ref_sp.RefSp(q, str_conv=rep != 7)
            assert r.ok, q
            t = osp.Task(q, str_conv=rep != 7)
            try:
                for ev in sp_synth.hopping_schedule(rng, 14):
                    # (the oracle first: it raises Unsupported where the reference binary would die)
                    if ev == "c":
                        c = sp_synth.chunk(rng, rng.choice([1, 8, 60]), clean)
                        b = t.do(c)
                        assert r.do(c) == b, q
                    elif ev == "h":
                        assert t.hop() == 0 and r.hop() == 0
                    else:
                        b = t.timer()
                        a = r.timer()
                        assert a == b, q
                        packaged += len(a) > 0
                compared += 1
            except osp.Unsupported:
                assert not clean
            finally:
                r.close()
    assert compared >= 24 and packaged > 100


def _select_cases():
    with open(os.path.join(HERE, "golden", "sp_select_cases.json")) as f:
        return json.load(f)


def test_select_oracle_matches_reference_answers():
    """SELECTs without aggregation functions (sp_process_data, flb_sp.c:1607-1850): the committed answers of the reference binary"""
    n = out_bytes = 0
    for c in _select_cases():
        if "NOW()" in c["sql"] and not LOCAL_IS_UTC:
            continue
        t = osp.Task(c["sql"])
        assert t.q.select_only and t.q.window == "default"
        for ch, (ret, out) in zip(c["chunks"], c["do"]):
            assert t.do(bytes.fromhex(ch)) == (ret, bytes.fromhex(out)), c["sql"]
            n += 1
            out_bytes += len(out) // 2
        assert t.timer() == b""
    assert n >= 40 and out_bytes > 30000


def test_select_map_header_quirk():
    """flb_sp.c:1801-1815: the header keeps the width msgpack_pack_map chose for the incoming size; a fixmap byte is patched
    with whatever was counted -- `*` plus two named keys over 14 entries writes 0x80 | 16 = 0x90, an (empty) fixarray head"""
    body = {"k%02d" % i: i for i in range(14)}
    rec = b"\x92\x92\xd7\x00" + struct.pack(">II", 7, 0) + b"\x80" + msgpack.packb(body)
    ret, out = osp.Task("SELECT *, k00, k01 FROM STREAM:s;").do(rec)
    assert ret == 1 and out[:13] == rec[:13] and out[13] == 0x90
    assert out[14:] == msgpack.packb(body)[1:] + msgpack.packb("k00") + b"\x00" + msgpack.packb("k01") + b"\x01"
    # a map16 header stays a map16 header however few entries leave; a key with a value the processor has no class for goes alone
    big = {"k%02d" % i: i for i in range(16)}
    big["arr"] = [1]
    rec = b"\x92\xd7\x00" + struct.pack(">II", 7, 0) + msgpack.packb(big)
    ret, out = osp.Task("SELECT k03 AS x, arr, nope FROM STREAM:s WINDOW TUMBLING (5 SECOND) GROUP BY k03;").do(rec)
    assert (ret, out) == (1, rec[:11] + b"\xde\x00\x02\xa1x\x03\xa3arr")
    # no key of the record selected: nothing leaves, the record still counts; no record passing WHERE: (0, nothing)
    assert osp.Task("SELECT nope FROM STREAM:s;").do(rec) == (1, b"")
    assert osp.Task("SELECT * FROM STREAM:s WHERE k00 = 5;").do(rec) == (0, b"")


def test_raw_repack_is_msgpack_pack_object():
    """the oracle's byte-level re-pack against the C restatement of msgpack_pack_object (omp.c, itself pinned on the reference's
    msgpack-c by test_msgpack_pin.py)"""
    import oracle_binding as ob
    rng = random.Random(77)
    for _ in range(40):
        c = sp_synth.select_chunk(rng, 30, legacy=rng.random() < 0.3)
        off, mine = 0, b""
        while off < len(c):
            piece, off = osp._raw_repack(c, off)
            mine += piece
        assert mine == ob.repack(c)


@pytest.mark.skipif(not ref_sp.available(), reason="oracle/_ref/ref_sp not built")
def test_select_live_fuzz_against_the_reference():
    rng = random.Random(0x5E1EC7)
    compared = 0
    for q in sp_synth.SELECT_QUERIES:
        for rep in range(12):
            r = ref_sp.RefSp(q)
            assert r.ok and r.select_only, q
            t = osp.Task(q)
            try:
                for _ in range(2):
                    c = sp_synth.select_chunk(rng, rng.choice([1, 20, 200]), legacy=rep % 5 == 4)
                    if rep % 6 == 5:
                        c = c[:len(c) - rng.randrange(1, 30)]
                    assert r.do(c) == t.do(c), q
                    compared += 1
                assert r.timer() == b""
            finally:
                r.close()
    assert compared >= 200


@pytest.mark.skipif(not ref_sp.available(), reason="oracle/_ref/ref_sp not built")
def test_shim_grammar_rejects_what_the_reference_rejects():
    # tests/internal/include/sp_invalid_queries.h: shapes the grammar refuses
    for bad in ["SELECT id, MIN(id) FROM STREAM:FLB;", "SELECT * FROM STREAM:FLB WHERE;", "SELECT COUNT(*) FROM STREAM:FLB GROUP BY;",
                "SELECT COUNT(*) FROM STREAM:FLB", "SELECT COUNT() FROM STREAM:FLB;", "SELECT a, * FROM STREAM:FLB;", "SELECT *, * FROM STREAM:FLB;"]:
        r = ref_sp.RefSp(bad)
        assert not r.ok, bad
        r.close()
        with pytest.raises((osp.ParseError, osp.Unsupported)):
            osp.parse(bad)
