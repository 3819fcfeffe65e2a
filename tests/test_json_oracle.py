"""JSON -> msgpack (flb_pack_json): the oracle restatement (oracle/ojson.c) against the golden
vectors recorded from the real reference reader, against the reference's own json/.mp sample
pairs, and -- where oracle/_ref is present -- live against the real reader on a fresh fuzz corpus."""
import json, os
import pytest
import jsonfuzz as jf

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "json_kat.json")))


def test_oracle_matches_golden_vectors():
    o = jf.oracle()
    assert len(KAT["cases"]) > 2000
    for c in KAT["cases"]:
        r = o(bytes.fromhex(c["in"]))
        want = (c["ret"], bytes.fromhex(c["out"]) if c["out"] is not None else None, c["root_type"], c["records"], c["consumed"])
        assert r == want, (bytes.fromhex(c["in"])[:80], r[0], r[2:], want[0], want[2:])


def test_oracle_on_reference_sample_pairs():
    # tests/internal/data/pack/*.json / *.mp (tests/internal/pack.c:490-560)
    o = jf.oracle()
    assert len(KAT["reference_pairs"]) >= 7
    for p in KAT["reference_pairs"]:
        r = o(bytes.fromhex(p["json"]))
        assert r[0] == 0 and r[1] == bytes.fromhex(p["mp"]), p["name"]


def test_oracle_live_against_real_reader():
    ref = jf.reference()
    if ref is None:
        pytest.skip("oracle/_ref/libyyjson_ref.so not built (needs /root/reference)")
    o = jf.oracle()
    for c in jf.corpus(7, 6000):
        assert o(c) == ref(c), c[:100]


def test_semantics_spelled_out():
    o = jf.oracle()
    mp = lambda js: o(js)[1]
    assert mp(b'-0') == b'\x00' and mp(b'-0.0') == b'\xcb\x80' + b'\x00' * 7          # "-0" is an integer
    assert mp(b'18446744073709551615') == b'\xcf' + b'\xff' * 8
    assert mp(b'18446744073709551616')[0] == 0xcb and mp(b'-9223372036854775809')[0] == 0xcb
    assert o(b'1e400')[0] == -1 and mp(b'1e-400') == b'\xcb' + b'\x00' * 8
    assert mp(b'"\\ud83d"') == b'\xa3\xef\xbf\xbd' and mp(b'"\\u12G4"') == b'\xa5\\u124'   # the offending char is swallowed
    assert mp(b'"a\nb\xff"') == b'\xa4a\nb\xff'                                            # raw control chars / bad UTF-8 pass
    assert o(b'{"a":1}{"b":2} x') == (0, b'\x81\xa1a\x01\x81\xa1b\x02', 1, 2, 15)           # stops at the first bad value
    assert o(b'x') == (-1, None, 0, 0, 0) and o(b'  ') == (0, b'', 0, 0, 2)
    assert o(b'[1,]')[0] == -1 and o(b'{"a":1,}')[0] == -1 and o(b'01')[0] == -1
