"""Decode_Field / Decode_Field_As on the device (csrc/dec_dev.inc, k_parser_dec) against the oracle, which is pinned on the
reference's own flb_parser_decoder.c inside cb_filter (tests/test_decoders_oracle.py): the 11 decoder sets x 5 parser
configurations x Reserve_Data / Preserve_Key of that test, the docker parser of conf/parsers.conf:44-58 with its decoders
enabled, and a list that mixes a parser with decoders and one without."""
import json, random
import pytest

import flbamd_loader
import oracle_binding as ob
import synth
from test_decoders_oracle import DECODER_SETS, PARSERS, VALUES, _chunks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


def both(g, key, pargs_list, data, reserve=False, preserve=False):
    want = ob.FilterParser(key, [ob.Parser(**p) for p in pargs_list], reserve, preserve).filter(data)
    gps = [g.Parser(**p) for p in pargs_list]
    f = g.FilterParser(key, gps, reserve, preserve)
    got = f.filter(data)
    f.close()
    for p in gps: p.close()
    return want, got


def first_diff(a, b):
    a, b = a or b"", b or b""
    n = min(len(a), len(b))
    i = next((k for k in range(n) if a[k] != b[k]), n)
    return i, len(a), len(b), a[max(0, i - 40):i + 40], b[max(0, i - 40):i + 40]


def test_decoder_sets_on_every_parser_format(g):
    data = _chunks()
    for di, decs in enumerate(DECODER_SETS):
        for pi, pa in enumerate(PARSERS):
            for reserve, preserve in ((False, False), (True, True), (True, False)):
                want, got = both(g, "msg", [dict(pa, decoders=decs)], data, reserve, preserve)
                assert got[0] == want[0], (decs, pa, reserve)
                assert got[1] == want[1], (decs, pa, reserve, preserve, first_diff(want[1], got[1]))


def test_docker_parser_with_its_decoders(g):
    """conf/parsers.conf:44-58: Format json, Time_Key time, Time_Format %Y-%m-%dT%H:%M:%S.%L, with the commented decoder lines on"""
    rng = random.Random(3)
    recs = []
    for i in range(600):
        inner = rng.choice([{"level": "info", "msg": "m%d" % i, "n": i}, {"time": "2019-01-01T00:00:00.5", "k": [1, {"x": None}]}, "plain", 5])
        log = (json.dumps(inner) if not isinstance(inner, str) else inner) + "\n"
        if rng.random() < 0.2:
            log = log.replace('"', '\\"')
        line = json.dumps({"log": log, "stream": rng.choice(["stdout", "stderr"]), "time": "2020-03-04T05:06:%02d.%03dZ" % (i % 60, i % 1000)})
        if rng.random() < 0.03:
            line = line[:-2]
        recs.append(synth.v2_record(50 + i, 0, {"log": line.encode()}))
    data = b"".join(recs)
    base = dict(format="json", time_key="time", time_fmt="%Y-%m-%dT%H:%M:%S.%L", time_keep=True)
    for decs in ([(True, "json", "log")], [(True, "escaped_utf8", "log", "do_next"), (True, "json", "log")], [(False, "json", "log")],
                 [(True, "escaped", "log", "do_next"), (False, "json", "log")]):
        for keep in (True, False):
            want, got = both(g, "log", [dict(base, time_keep=keep, decoders=decs)], data)
            assert got == want, (decs, keep, first_diff(want[1], got[1]))


def test_parser_list_with_and_without_decoders(g):
    data = _chunks()
    plist = [dict(PARSERS[0], decoders=DECODER_SETS[5]), dict(format="logfmt"), dict(format="json", decoders=[(False, "json", "log")])]
    for reserve in (False, True):
        want, got = both(g, "msg", plist, data, reserve, reserve)
        assert got == want, first_diff(want[1], got[1])


def test_flb_parser_do_with_decoders(g):
    """flb_parser_do's map (the parser's output before filter_parser touches it)"""
    for decs in DECODER_SETS[:8]:
        po = ob.Parser(regex=r"^<(?<log>.*)> <(?<other>.*)>$", decoders=decs)
        pg = g.Parser(regex=r"^<(?<log>.*)> <(?<other>.*)>$", decoders=decs)
        for v in VALUES:
            for w in (b"x", VALUES[0]):
                line = b"<" + v + b"> <" + w + b">"
                assert pg.do(line) == po.do(line), (decs, line)
        pg.close()
