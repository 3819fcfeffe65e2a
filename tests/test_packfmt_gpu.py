"""msgpack -> JSON output formatter on the device (flbgpu_pack_msgpack_to_json_format and the JsonFormatter object)
against the reference's known answers (tests/golden/packfmt_kat.json, produced by the real
flb_pack_msgpack_to_json_format) and against the oracle on larger chunks."""
import os, random, struct
import pytest
import flbamd_loader
import oracle_binding as ob
import packfmt_cases, synth
from test_packfmt_oracle import load_kat, oracle_out

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


def dev_out(g, cfg, data):
    return g.msgpack_to_json_format(data, cfg["json_format"], cfg["date_format"], cfg["date_key"], cfg["escape_unicode"], cfg["nan_to_null"])


def test_reference_known_answers(g):
    kat = load_kat()
    for cfg, data, want in kat:
        got = dev_out(g, cfg, data)
        assert got == want, (cfg, data.hex(), got, want)


def test_random_corpus_against_oracle(g):
    cases = packfmt_cases.corpus(4242, 1500)
    for cfg, data in cases:
        assert dev_out(g, cfg, data) == oracle_out(cfg, data), (cfg, data.hex())


def test_large_chunks_and_object_reuse(g):
    r = random.Random(9)
    for fmt in ("json", "stream", "lines"):
        for df in ("double", "iso8601", "epoch", "java_sql_timestamp", "epoch_ms"):
            esc = r.randrange(2)
            f = g.JsonFormatter(fmt, df, b"date", escape_unicode=esc)
            for _ in range(2):
                chunk = b"".join(packfmt_cases.rand_chunk(r, 40) for _ in range(50))
                cfg = dict(json_format=g.JSON_FORMAT[fmt], date_format=g.JSON_DATE[df], date_key=b"date")
                want = ob.msgpack_to_json_format(chunk, cfg["json_format"], cfg["date_format"], b"date", esc, 0)
                assert f.format(chunk) == want
            f.close()


def test_apache_records_lines(g):
    """the bench workload's records (parsed apache lines), 200 k of them, all five date formats"""
    data, off, ep = synth.apache_records(200000)
    raw = bytes(data)
    from test_gpu_parity import APACHE2, TF
    r, parsed = g.FilterParser("log", [g.Parser(APACHE2, time_fmt=TF, time_key="time")]).filter(raw)
    assert r == g.MODIFIED
    for df in range(5):
        want = ob.msgpack_to_json_format(parsed, 3, df, b"date", 1, 0)
        got = g.msgpack_to_json_format(parsed, 3, df, b"date", 1, 0)
        assert got == want


def test_skip_limit_and_groups_large(g):
    """1000 consecutive skipped rows end the decoder's walk; group attributes reach rows far behind the opener"""
    from synth import mp, ext_ts, KV
    opener = mp([[ext_ts(0xffffffff, 0), KV([(b"g", 1)])], KV([(b"resource", b"r1")])])
    closer = mp([[ext_ts(0xfffffffe, 0), KV([])], KV([])])
    rec = lambda i: mp([[ext_ts(1700000000 + i, 5), KV([])], KV([(b"i", i)])])
    chunk = opener + b"".join(rec(i) for i in range(3000)) + closer + b"".join(rec(i) for i in range(10))
    for fmt in (1, 2, 3):
        assert g.msgpack_to_json_format(chunk, fmt, 0, b"date", 1, 0) == ob.msgpack_to_json_format(chunk, fmt, 0, b"date", 1, 0)
    for k in (998, 999, 1000, 1001, 2500):
        chunk = rec(1) + closer * k + rec(2) + rec(3)
        want = ob.msgpack_to_json_format(chunk, 3, 0, b"date", 1, 0)
        assert g.msgpack_to_json_format(chunk, 3, 0, b"date", 1, 0) == want, k
        assert want.count(b"\n") == (3 if k < 1000 else 1)
