"""Calls on small chunks launch their kernels ahead of the counters and wait once (flbgpu.cpp SpecCall).  Whatever the launched
kernels do not cover runs the stage again the usual way -- so the two ways must answer alike, byte for byte, on chunks built to hit
every such case: rows for the generic kernel, a bad record in the middle, records for the exact writer, an output larger than the
room the writer was given, chunks where a filter answers NOTOUCH.  FLBGPU_NO_SPEC=1 (read per call) is the usual way."""
import os, random
import pytest
import oracle_binding as ob
import synth
import flbamd_loader

pytestmark = pytest.mark.gpu
APACHE2 = r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$'
TF = "%d/%b/%Y:%H:%M:%S %z"


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


def both_ways(f, blob):
    os.environ.pop("FLBGPU_NO_SPEC", None)
    ahead = f.filter(blob)
    ahead2 = f.filter(blob)                   # (and again: the buffers of the first call are reused)
    os.environ["FLBGPU_NO_SPEC"] = "1"
    try:
        usual = f.filter(blob)
    finally:
        os.environ.pop("FLBGPU_NO_SPEC", None)
    assert ahead[0] == usual[0] and ahead[1] == usual[1]
    assert ahead2[0] == usual[0] and ahead2[1] == usual[1]
    return usual


def mixed_chunk(n, seed, long_value=False, bad_at=None, legacy=False):
    rng = random.Random(seed)
    data, off, ep = synth.apache_records(n)
    data = bytes(data)
    recs = [data[int(off[i]):int(off[i + 1])] for i in range(n)]
    out = []
    for i, r in enumerate(recs):
        k = rng.random()
        if k < 0.05:
            out.append(synth.v2_record(1700000000 + i, 5, {"log": "not an access log line %d" % i}))
        elif k < 0.08:
            out.append(synth.v2_record(1700000000 + i, 5, {"stream": "stdout", "log": '10.0.0.%d - u [10/Oct/2000:13:55:36 -0700] "GET /x HTTP/1.0" 5%02d 12' % (i % 250, i % 100), "n": i}))
        elif k < 0.10 and legacy:
            out.append(synth.legacy_record(1700000000 + i, {"log": '10.0.0.1 - u [10/Oct/2000:13:55:36 -0700] "GET /y HTTP/1.0" 503 1'}))
        elif k < 0.11 and long_value:
            out.append(synth.v2_record(1700000000 + i, 5, {"log": '10.0.0.1 - u [10/Oct/2000:13:55:36 -0700] "GET /' + "z" * 5000 + ' HTTP/1.0" 500 1'}))
        else:
            out.append(r)
        if bad_at is not None and i == bad_at:
            out.append(b"\xc1\xc1\xc1")
    return b"".join(out)


def filters(g, kind):
    p = g.Parser(APACHE2, time_fmt=TF, time_key="time")
    fp = g.FilterParser("log", [p])
    if kind == "parser":
        return fp, [fp, p]
    fg = g.FilterGrep([("regex", r"code ^5\d\d$")])
    if kind == "grep":
        return g.FilterGrep([("regex", r"log 50[0-9] ")]), [fp, p, fg]
    if kind == "pair":
        return g.FilterChain([fp, fg]), [fp, p, fg]
    fg2 = g.FilterGrep([("exclude", r"host ^10\.0\.0\.7$")])
    return g.FilterChain([fp, fg, fg2]), [fp, p, fg, fg2]


@pytest.mark.parametrize("kind", ["parser", "grep", "pair", "three"])
@pytest.mark.parametrize("shape", ["plain", "mixed", "long", "bad", "legacy"])
def test_ahead_equals_usual(g, kind, shape):
    n = 3000
    if shape == "plain":
        data, off, ep = synth.apache_records(n)
        blob = bytes(data)
    else:
        blob = mixed_chunk(n, 5, long_value=shape == "long", bad_at=1700 if shape == "bad" else None, legacy=shape == "legacy")
    f, owned = filters(g, kind)
    both_ways(f, blob)
    both_ways(f, blob[: len(blob) // 3])
    both_ways(f, blob[:-7])                     # cut inside the last record


def test_ahead_against_the_oracle(g):
    blob = mixed_chunk(4000, 9, long_value=True, legacy=True)
    f, owned = filters(g, "pair")
    r, out = both_ways(f, blob)
    op = ob.Parser(regex=APACHE2, time_fmt=TF, time_key="time")
    ro, oo = ob.FilterParser("log", [op]).filter(blob)
    ro2, oo2 = ob.Grep([("regex", r"code ^5\d\d$")]).filter(oo)
    assert r == ob.MODIFIED and out == oo2


def test_output_larger_than_the_room(g):
    """twenty one-byte fields under forty-byte names: the output is many times the input, the writer launched ahead has no room for it"""
    names = ["a_field_with_quite_a_long_name_number_%02d" % i for i in range(20)]
    rx = "^" + "".join("(?<%s>.)" % nm for nm in names)
    recs = b"".join(synth.v2_record(1700000000 + i, 0, {"log": "abcdefghijklmnopqrstuvwxyz"}) for i in range(4000))
    p = g.Parser(rx)
    f = g.FilterParser("log", [p])
    r, out = both_ways(f, recs)
    assert r == ob.MODIFIED and len(out) > 10 * len(recs)
    ro, oo = ob.FilterParser("log", [ob.Parser(regex=rx)]).filter(recs)
    assert out == oo


def test_nothing_kept_and_everything_kept(g):
    data, off, ep = synth.apache_records(2000)
    blob = bytes(data)
    for rule in [("regex", r"log ^nothing matches this$"), ("regex", r"log .")]:
        f = g.FilterGrep([rule])
        r, out = both_ways(f, blob)
        ro, oo = ob.Grep([rule]).filter(blob)
        assert r == ro and (out or b"") == (oo or b"")
