"""filter_parser with Format logfmt / ltsv parsers on the GPU (pkv_dev.inc) against the oracle
(src/flb_parser_logfmt.c, src/flb_parser_ltsv.c, src/flb_unescape.c restated in oracle/oflb.c):
byte-identical filter output on the grammar corners and on random texts."""
import random

import pytest

import flbamd_loader
import oracle_binding as ob
import synth
from test_kv_oracle import ESC_CASES

pytestmark = pytest.mark.gpu
TFMT = "%Y-%m-%dT%H:%M:%S.%L"


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


def rec(body, sec=1700000000, nsec=7):
    return synth.mp([[synth.ext_ts(sec, nsec), {}], body])


def both(g, data, pargs, reserve=False, preserve=False, key="log"):
    po, pg = ob.Parser(**pargs), g.Parser(**pargs)
    want = ob.FilterParser(key, [po], reserve, preserve).filter(data)
    f = g.FilterParser(key, [pg], reserve, preserve)
    got = f.filter(data)
    f.close(); pg.close()
    return want, got


def diff(a, b):
    if a is None or b is None:
        return "one side None"
    for i in range(min(len(a), len(b))):
        if a[i] != b[i]:
            return "byte %d: oracle %r gpu %r" % (i, a[max(0, i - 24):i + 24], b[max(0, i - 24):i + 24])
    return "length %d vs %d" % (len(a), len(b))


LOGFMT_TEXTS = [b'str="text" int=100 double=1.23 bool=true', b'str="text" int=100 time=2022-10-31T12:00:01.123 z=1', b"", b'  ="= ', b"bare",
                b"k= j=", b'k="" j="x', b'k="a\\"b" z=1', b'k="a\\', b"a=1\nb=2", b"a=1 \nb=2", b"a=1\r\nb=2", b"a=1\rb=2", b"k=v=w x",
                b"k\x80\xff=\x01v", b"a=1 bare b=2", b"a=1 b= c=3", b"time=2020-01-02T03:04:05.5 a=1 time=2021-03-04T00:00:00.25",
                b"time=garbage a=1", b"time=2020-01-02T03:04:05.0", b"time= a=1", b"time=20x a=1", b'time="2022-10-31T12:00:01.123" q="w"',
                b'level=info msg="request done" path=/v1/x?y=1 dur=12ms ok', b"x" * 300 + b"=" + b"y" * 70000, b'a="' + b"\\n" * 200 + b'"']
LOGFMT_TEXTS += [b'k="' + e + b'" tail=1' for e in ESC_CASES]
# one `struct flb_tm` per walk (src/flb_parser_logfmt.c:70, src/flb_parser_ltsv.c:88): under Time_Strict Off a later value that
# parses only partially keeps the fields the earlier value set
LOGFMT_TEXTS += [b"time=2020-01-02T03:04:05.5 a=1 time=2021-03x", b"time=2019-12-31T23:59:58.25 time=2021 b=2 time=x", b"time=2020-05-06T07:08 time=2021-01-02T03:04:05.5 time=1999-"]
LTSV_TEXTS = [b"str:text\tint:100\tdouble:1.23\tbool:true", b"str:text\ttime:2022-10-31T12:00:01.123\tz:1", b"", b"nolabel", b":v\ta:1",
              b"a:\tb:x y:z", b"a:1\t\tb:2", b"a:1\nb:2", b"a:1\r\nb:2", b"a b:1", b"a:1\tb", b"a:x\x00y\tb:2", b"time:garbage\ta:1",
              b"time:2020-01-02T03:04:05.5", b"host:10.0.0.1\tident:-\tuser:bob\treq:GET /x HTTP/1.1\tstatus:200\tsize:512",
              b'json_str:{"str":"text", "int":100}', b"l-a.b_c:" + b"v" * 40000,
              b"time:2020-01-02T03:04:05.5\ta:1\ttime:2021-03x", b"time:2019-12-31T23:59:58.25\ttime:2021\tb:2\ttime:x", b"time:2020-05-06T07:08\ttime:2021-01-02T03:04:05.5\ttime:1999-"]


@pytest.mark.parametrize("fmt,texts", [("logfmt", LOGFMT_TEXTS), ("ltsv", LTSV_TEXTS)])
def test_kv_corner_cases(g, fmt, texts):
    data = b"".join(rec({"log": t, "other": 1}) for t in texts)
    for pargs in (dict(format=fmt), dict(format=fmt, time_fmt=TFMT, time_key="time"), dict(format=fmt, time_fmt=TFMT, time_key="time", time_keep=True),
                  dict(format=fmt, time_fmt=TFMT, time_key="time", time_strict=False), dict(format=fmt, no_bare_keys=True)):
        for reserve, preserve in ((False, False), (True, False), (True, True)):
            want, got = both(g, data, pargs, reserve, preserve)
            assert want[0] == got[0] and want[1] == got[1], (fmt, pargs, reserve, preserve, diff(want[1], got[1]))


@pytest.mark.parametrize("fmt", ["logfmt", "ltsv"])
def test_kv_random_texts(g, fmt):
    rng = random.Random(5 if fmt == "logfmt" else 6)
    if fmt == "logfmt":
        atoms = [b"key", b"k2", b"=", b'"', b"\\", b" ", b"\t", b"\n", b"\r", b"time", b"2022-10-31T12:00:01.5", b"u00e9", b"uD83D", b"\\uDE00", b"x41",
                 b"\x00", b"\xc3\xa9", b"v", b"1", b"n", b'\\"', b"\\\\"]
    else:
        atoms = [b"key", b"k-2", b":", b"\t", b" ", b"\n", b"\r", b"time", b"2022-10-31T12:00:01.5", b"\x00", b"\xc3\xa9", b"v", b"1", b"_", b".", b'"']
    recs = []
    for i in range(6000):
        t = b"".join(rng.choice(atoms) for _ in range(rng.randrange(0, 14)))
        body = {"log": t} if i % 7 else synth.KV([("log", t), ("n", i), ("log", b"a=1" if fmt == "logfmt" else b"a:1")])
        recs.append(rec(body, sec=i))
    data = b"".join(recs)
    for pargs in (dict(format=fmt, time_fmt=TFMT, time_key="time"), dict(format=fmt, time_fmt=TFMT, time_key="time", time_keep=True, time_strict=False),
                  dict(format=fmt, no_bare_keys=True)):
        for reserve, preserve in ((False, False), (True, True)):
            want, got = both(g, data, pargs, reserve, preserve)
            assert want[0] == got[0] and want[1] == got[1], (fmt, pargs, reserve, preserve, diff(want[1], got[1]))


@pytest.mark.parametrize("fmt", ["logfmt", "ltsv"])
def test_kv_parsers_with_types(g, fmt):
    """Types on logfmt / ltsv parsers: flb_parser_typecast per pair on the raw value text
    (src/flb_parser_logfmt.c:176-182, src/flb_parser_ltsv.c:149-155, src/flb_parser.c:2067-2164)"""
    rng = random.Random(9)
    sep, kvs = (" ", "=") if fmt == "logfmt" else ("\t", ":")
    vals = ["100", "-7", "0x1F", "ff", "1.5", "0.1", "1e400", "9007199254740993", "2.4703282292062328e-324", "8.41e21", "true", "FALSE", "maybe", "",
            "12abc", " 5", "3.141592653589793238462643383279", "text"]
    recs = []
    for i in range(4000):
        pairs = [(k, rng.choice(vals)) for k in rng.sample(["i", "h", "f", "b", "s", "other", "time"], rng.randrange(1, 6))]
        t = sep.join(k + kvs + (v if k != "time" else "2022-10-31T12:00:01.5") for k, v in pairs)
        if fmt == "logfmt" and rng.random() < 0.2:
            t += ' q="a\\nb" bare'
        recs.append(rec({"log": t.encode()}, sec=i))
    data = b"".join(recs)
    for pargs in (dict(format=fmt, types="i:integer h:hex f:float b:bool s:string"),
                  dict(format=fmt, types="f:float i:integer", time_fmt=TFMT, time_key="time"),
                  dict(format=fmt, types="other:float f:float f:integer", time_fmt=TFMT, time_key="time", time_keep=True)):
        for reserve, preserve in ((False, False), (True, True)):
            want, got = both(g, data, pargs, reserve, preserve)
            assert want[0] == got[0] and want[1] == got[1], (fmt, pargs, reserve, preserve, diff(want[1], got[1]))
    # in a list behind a regex parser (the generic kernel tries the whole list)
    want, got = both_list(g, data, [dict(regex=r"^(?<never>ZZZ)$"), dict(format=fmt, types="i:integer f:float")])
    assert want == got, diff(want[1], got[1])


def both_list(g, data, pargs_list, reserve=False, preserve=False, key="log"):
    po = [ob.Parser(**a) for a in pargs_list]
    pg = [g.Parser(**a) for a in pargs_list]
    want = ob.FilterParser(key, po, reserve, preserve).filter(data)
    f = g.FilterParser(key, pg, reserve, preserve)
    got = f.filter(data)
    f.close()
    for p in pg:
        p.close()
    return want, got


def test_lists_that_mix_parser_formats(g):
    """filter_parser tries its parsers in order on every candidate value (filter_parser.c:259-300):
    json / logfmt / ltsv / regex parsers in one list, in every position"""
    rng = random.Random(31)
    APACHE = r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)$'
    texts = [b'{"a":1,"time":"2022-10-31T12:00:01.123","n":{"x":[1,2.5,null]}}', b'{"broken": ', b'[1,2]', b'{"f":0.1,"g":1e400,"h":123456789012345678901234567890}',
             b'10.0.0.1 - bob [10/Oct/2000:13:55:36 -0700] "GET /a HTTP/1.0" 200 2326', b'level=info msg="hi there" time=2022-10-31T12:00:01.5 ok',
             b"host:h1\tstatus:200\ttime:2022-10-31T12:00:01.25", b"", b"plain words only", b"k=v", b"a:1", b'{"k":"v"} trailing', b"=", b"\xff\xfe",
             b'{"deep":' + b"[" * 70 + b"]" * 70 + b"}", b'msg="caf\\u00e9 \\uD83D\\uDE00" n=1']
    # around msgpack-c's 32-deep container stack (flb_parser_json_do unpacks what it packed)
    texts += [b'{"d":' + b"[" * k + b"]" * k + b"}" for k in (30, 31, 32, 33)] + [b'{"d":' + b'{"o":' * k + b"1" + b"}" * k + b"}" for k in (30, 31, 32)]
    recs = []
    for i in range(3000):
        t = rng.choice(texts)
        body = {"log": t, "i": i} if i % 5 else synth.KV([("log", rng.choice(texts)), ("i", i), ("log", t)])
        recs.append(rec(body, sec=1000 + i))
    data = b"".join(recs)
    js = dict(format="json", time_fmt=TFMT, time_key="time")
    lf = dict(format="logfmt", time_fmt=TFMT, time_key="time")
    lt = dict(format="ltsv", time_fmt=TFMT, time_key="time")
    rx = dict(regex=APACHE, time_fmt="%d/%b/%Y:%H:%M:%S %z", time_key="time")
    for plist in ([js, rx], [rx, js], [js, lt, rx, lf], [lt, js], [rx, lf], [lf, js], [dict(js, time_keep=True), dict(rx, time_keep=True)]):
        for reserve, preserve in ((False, False), (True, True)):
            want, got = both_list(g, data, plist, reserve, preserve)
            assert want[0] == got[0] and want[1] == got[1], ([p.get("format", "regex") for p in plist], reserve, preserve, diff(want[1], got[1]))
