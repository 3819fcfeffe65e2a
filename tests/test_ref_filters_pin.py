"""The oracle's filter_grep / filter_parser (oracle/oflb.c: the restated cb_filter bodies, event codec, parser glue,
record accessor) against the reference's OWN plugins: plugins/filter_grep/grep.c and plugins/filter_parser/filter_parser.c
compiled from /root/reference with everything under them (oracle/_ref/ref_filters, oracle/ref_filters_shim.c) and driven
through cb_init / cb_filter on the same bytes.  Return codes and output bytes must be identical.  Runs where the
reference tree (or a prebuilt oracle/_ref) is present."""
import os, random, struct, sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_binding as ob
import ref_filters as rf
import synth

pytestmark = pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built (no reference tree)")

APACHE2 = r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$'
APACHE = r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^\"]*?)(?: +\S*)?)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>[^\"]*)")?$'
TF = "%d/%b/%Y:%H:%M:%S %z"


def _rec(body, sec=1, nsec=0, meta=None):
    return synth.v2_record(sec, nsec, body, meta)


def _lines_chunk(seed, n=1500):
    rng = random.Random(seed)
    data, off, _ = synth.apache_records(n)
    blob = bytes(data)
    out = []
    for i in range(n):
        m = bytearray(blob[int(off[i]) + 21:int(off[i + 1])])
        r = rng.random()
        if r < 0.06: m[rng.randrange(len(m))] = rng.choice(b"\xff\xe9\x80\xc3")
        elif r < 0.12: m = m[: rng.randrange(len(m))]
        elif r < 0.16: k = rng.randrange(len(m)); m[k:k] = b"x" * rng.randrange(1, 400)
        elif r < 0.19: m = bytearray()
        elif r < 0.22: m += b"\n" + m
        elif r < 0.25: m = bytearray(b"\n") + m
        r2 = rng.random()
        if r2 < 0.05: rec = _rec(synth.KV([("stream", "stdout"), ("log", bytes(m)), ("n", i)]))
        elif r2 < 0.08: rec = _rec(synth.KV([("log", b"first"), ("log", bytes(m))]))
        elif r2 < 0.11: rec = synth.legacy_record(1700000000 + i, {"log": bytes(m)})
        elif r2 < 0.13: rec = _rec({"log": bytes(m)}, meta={"k": "v"})
        elif r2 < 0.14: rec = _rec({"log": i})
        elif r2 < 0.15: rec = synth.mp([[synth.ext_ts(0xffffffff, 0), {}], {"g": 1}])
        else: rec = _rec({"log": bytes(m)}, sec=1700000000 + i, nsec=i % 1000)
        out.append(rec)
    return b"".join(out)


def _check(cases, wants, what):
    got = rf.run(cases)
    for (ret, out), (wret, wout), w in zip(got, wants, what):
        assert ret == wret, (w, ret, wret)
        if wret == ob.MODIFIED:
            assert out == (wout or b""), (w, len(out), len(wout or b""))


def test_filter_parser_glue():
    chunk = _lines_chunk(11)
    cases, wants, what = [], [], []
    variants = [
        ([dict(regex=APACHE2, time_fmt=TF, time_key="time")], False, False),
        ([dict(regex=APACHE, time_fmt=TF, time_key="time", time_keep=True)], False, False),
        ([dict(regex=APACHE2, time_fmt=TF, time_key="time")], True, False),
        ([dict(regex=APACHE2, time_fmt=TF, time_key="time")], True, True),
        ([dict(regex=APACHE2, time_fmt=TF, time_key="time")], False, True),
        ([dict(regex=APACHE2, time_fmt=TF, time_key="time", types="code:integer size:integer")], False, False),
        ([dict(regex=APACHE2, time_fmt="%d/%b/%Y:%H:%M:%S", time_key="time")], False, False),
        ([dict(regex=r"^(?<a>\d+)$"), dict(regex=APACHE2, time_fmt=TF, time_key="time"), dict(regex=r"^(?<first>\S+)")], False, False),
        ([dict(regex=APACHE2, skip_empty=False)], True, False),
        ([dict(format="json", time_key="time", time_fmt="%Y-%m-%dT%H:%M:%S")], False, False),
        ([dict(format="logfmt")], True, False),
        ([dict(format="ltsv")], False, False),
    ]
    extra = b"".join([_rec({"log": '{"a":1,"time":"2024-01-02T03:04:05","b":{"c":[1,2.5,"x"]}}'}), _rec({"log": 'k=v msg="a b" n=1 bare'}),
                      _rec({"log": "a:1\tb:two\ttime:x"}), _rec({"log": "{bad json"}), _rec({"log": ""})])
    for data in (chunk, extra, chunk[:300] + b"\x92\x92\xd7\x00", b""):
        for plist, reserve, preserve in variants:
            cases.append(rf.parser_case("log", plist, data, reserve, preserve))
            wants.append(ob.FilterParser("log", [ob.Parser(**p) for p in plist], reserve, preserve).filter(data))
            what.append((plist[0].get("regex", plist[0].get("format"))[:20], reserve, preserve, len(data)))
    # record accessor key
    data = b"".join(_rec({"k": {"sub": ["x", "GET /a HTTP/1.1"]}, "log": "POST /b"}) for _ in range(5))
    pa = dict(regex=r"^(?<method>[A-Z]+) (?<path>[^ ]*)(?: (?<proto>.*))?$")
    for key in ("$k['sub'][1]", "$log", "log", "$nokey"):
        for reserve, preserve in ((False, False), (True, False), (True, True)):
            cases.append(rf.parser_case(key, [pa], data, reserve, preserve))
            wants.append(ob.FilterParser(key, [ob.Parser(**pa)], reserve, preserve).filter(data))
            what.append((key, reserve, preserve))
    _check(cases, wants, what)


def test_filter_grep_glue():
    chunk = _lines_chunk(12)
    parsed = ob.FilterParser("log", [ob.Parser(regex=APACHE2, time_fmt=TF, time_key="time")]).filter(chunk)[1]
    recs = [_rec({"log": "aaa"}), _rec({"log": "bbb"}), _rec({"log": "abc", "x": "1"}), _rec({"other": "aaa"}), _rec({"log": 5}),
            _rec(synth.KV([("log", "zzz"), ("log", "aaa")])), synth.legacy_record(5, {"log": "aaa"}), synth.legacy_record(1.5, {"log": "xaax"}),
            _rec({"k": {"sub": ["x", "HELLO"]}}), _rec({"log": "h\xc3\xa9llo w\xc3\xb6rld"}), _rec({"log": ""})]
    grp = synth.mp([[synth.ext_ts(0xffffffff, 0), {}], {"g": 1}])
    small = b"".join(recs)
    rule_sets = [([("regex", r"code ^5\d\d$")], None), ([("exclude", "method GET")], None), ([("regex", "code ^2"), ("regex", "agent curl")], "AND"),
                 ([("regex", "code ^404$"), ("regex", "method ^P")], "OR"), ([("regex", "host .*")], None), ([("regex", "nokey x")], None),
                 ([("regex", "log a")], None), ([("exclude", "log a")], None), ([("regex", "$k['sub'][1] /hello/i")], None), ([("regex", "log ^$")], None),
                 ([("exclude", "log a"), ("regex", "log b")], None), ([("exclude", "code ^2"), ("exclude", "method ^P")], "OR"),
                 ([("exclude", "code ^2"), ("exclude", "method ^G")], "AND"), ([("regex", "log é")], None)]
    cases, wants, what = [], [], []
    for data in (parsed, small, grp + small, small + b"\x92\x01", chunk, b""):
        for rules, op in rule_sets:
            cases.append(rf.grep_case(rules, op, data))
            wants.append(ob.Grep(rules, op).filter(data))
            what.append((rules, op, len(data)))
    _check(cases, wants, what)


def _rand_obj(rng, depth=0):
    t = rng.randrange(12 if depth < 3 else 8)
    if t == 0: return None
    if t == 1: return rng.choice([True, False])
    if t == 2: return rng.choice([0, 1, 127, 128, 255, 256, 65535, 65536, 2 ** 32, 2 ** 63, -1, -32, -33, -128, -129, -32769, -2 ** 31 - 1])
    if t == 3: return rng.choice([0.5, -1e10, 3.14])
    if t == 4: return synth.Raw(b"\xca" + struct.pack(">f", rng.random()))
    if t == 5: return synth.Raw(b"\xc4\x03abc")
    if t == 6: return synth.Raw(b"\xc7\x02\x05xy")
    if t < 9: return rng.choice(["", "x", "GET /a HTTP/1.1", "500", "\xc3\xa9", "y" * rng.choice([31, 32, 255, 256, 70000])])
    if t == 9: return [_rand_obj(rng, depth + 1) for _ in range(rng.randrange(0, 18))]
    return synth.KV([(rng.choice(["a", "b", "log", "code", 7, None, "k%d" % rng.randrange(20)]), _rand_obj(rng, depth + 1)) for _ in range(rng.randrange(0, 18))])


def test_structural_fuzz_against_the_real_plugins():
    """random record shapes (every msgpack family, nested containers, non-string keys, duplicate keys, legacy / V2 / float /
    integer timestamps, metadata, group markers) and byte-level corruptions"""
    rng = random.Random(77)
    pa = dict(regex=r"^(?<method>[A-Z]+) (?<path>[^ ]*)(?: (?<proto>.*))?$")
    cases, wants, what = [], [], []
    for trial in range(10):
        recs = []
        for i in range(300):
            body = synth.KV([(rng.choice(["log", "code", "a", "b", "nest", 5, "log"]), _rand_obj(rng)) for _ in range(rng.randrange(0, 7))])
            r = rng.random()
            if r < 0.55: rec = synth.v2_record(rng.randrange(2 ** 32), rng.randrange(10 ** 9), body, rng.choice([None, {}, {"m": [1, {"x": "y"}]}]))
            elif r < 0.7: rec = synth.legacy_record(rng.choice([5, 2 ** 31, 1.5, 1e9 + 0.25]), body)
            elif r < 0.75: rec = synth.mp([[synth.ext_ts(rng.choice([0xffffffff, 0xfffffffe]), 0), {}], body])
            elif r < 0.8: rec = synth.mp([[rng.randrange(10 ** 6), {}], body])
            elif r < 0.85: rec = synth.mp([[synth.ext_ts(7, 10 ** 9 + 5), {}], body])
            else: rec = synth.v2_record(1, 2, body)
            recs.append(rec)
        data = b"".join(recs)
        if trial % 3 == 1:
            k = rng.randrange(len(data))
            data = rng.choice([data[:k] + bytes([rng.randrange(256)]) + data[k + 1:], data[:k], data[:k] + b"\xc1" + data[k:]])
        for reserve, preserve in ((False, False), (True, True)):
            cases.append(rf.parser_case("log", [pa], data, reserve, preserve))
            wants.append(ob.FilterParser("log", [ob.Parser(**pa)], reserve, preserve).filter(data))
            what.append((trial, "parser", reserve, preserve))
        for rules, op in (([("regex", "log ^GET"), ("exclude", "code ^5")], None), ([("regex", "a x"), ("regex", "$nest['a'] y")], "OR"),
                          ([("exclude", "log HTTP"), ("exclude", "b ^$")], "AND")):
            cases.append(rf.grep_case(rules, op, data))
            wants.append(ob.Grep(rules, op).filter(data))
            what.append((trial, rules, op))
    _check(cases, wants, what)
