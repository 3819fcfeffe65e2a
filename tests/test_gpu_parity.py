"""GPU parity: the HIP path (through the C ABI of include/flb_gpu.h) against the CPU oracle on the
same inputs -- bit-exact output bytes and return codes."""
import os, random, struct
import numpy as np
import pytest
import oracle_binding as ob
import synth
import flbamd_loader

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
APACHE2 = r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$'
APACHE = r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^\"]*?)(?: +\S*)?)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>[^\"]*)")?$'
TF = "%d/%b/%Y:%H:%M:%S %z"


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


def both_parser(g, data, key, pargs_list, reserve=False, preserve=False):
    op = [ob.Parser(**a) for a in pargs_list]
    gp = [g.Parser(**a) for a in pargs_list]
    ro, oo = ob.FilterParser(key, op, reserve, preserve).filter(data)
    f = g.FilterParser(key, gp, reserve, preserve)
    rg, og = f.filter(data)
    f.close()
    for p in gp:
        p.close()
    return (ro, oo), (rg, og)


def both_grep(g, data, rules, op=None):
    ro, oo = ob.Grep(rules, op).filter(data)
    f = g.FilterGrep(rules, op)
    rg, og = f.filter(data)
    f.close()
    return (ro, oo), (rg, og)


def first_diff(a, b):
    if a is None or b is None:
        return "one side None"
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return "byte %d: oracle %r gpu %r (len %d vs %d)" % (i, a[max(0, i - 20):i + 20], b[max(0, i - 20):i + 20], len(a), len(b))
    return "length %d vs %d" % (len(a), len(b))


def test_parser_apache2_synthetic(g):
    data, off, ep = synth.apache_records(20000)
    o, q = both_parser(g, bytes(data), "log", [dict(regex=APACHE2, time_fmt=TF, time_key="time")])
    assert o[0] == q[0] == ob.MODIFIED
    assert o[1] == q[1], first_diff(o[1], q[1])


def test_parser_apache_fixture_400(g):
    data = open(os.path.join(HERE, "golden", "apache_400.mp"), "rb").read()
    for rx in (APACHE2, APACHE):
        o, q = both_parser(g, data, "log", [dict(regex=rx, time_fmt=TF, time_key="time")])
        assert o[0] == q[0] and o[1] == q[1], first_diff(o[1], q[1])


def test_grep_on_parser_output(g):
    data, off, ep = synth.apache_records(20000)
    o, q = both_parser(g, bytes(data), "log", [dict(regex=APACHE2, time_fmt=TF, time_key="time")])
    parsed = o[1]
    for rules, op in [([("regex", r"code ^5\d\d$")], None), ([("exclude", "method GET")], None),
                      ([("regex", "code ^2"), ("regex", "agent curl")], "AND"),
                      ([("regex", "code ^404$"), ("regex", "method ^P")], "OR"),
                      ([("regex", "host .*")], None), ([("regex", "nokey x")], None)]:
        a, b = both_grep(g, parsed, rules, op)
        assert a[0] == b[0], (rules, a[0], b[0])
        assert a[1] == b[1], (rules, first_diff(a[1], b[1]))


def _rec(body, sec=1, nsec=0, meta=None):
    return synth.v2_record(sec, nsec, body, meta)


def test_grep_edge_cases(g):
    recs = [_rec({"log": "aaa"}), _rec({"log": "bbb"}), _rec({"log": "abc", "x": "1"}), _rec({"other": "aaa"}),
            _rec({"log": 5}), _rec(synth.KV([("log", "zzz"), ("log", "aaa")])), synth.legacy_record(5, {"log": "aaa"}),
            synth.legacy_record(1.5, {"log": "xaax"}), _rec({"k": {"sub": ["x", "HELLO"]}}), _rec({"log": "héllo wörld"}),
            _rec({"log": ""})]
    grp = synth.mp([[synth.ext_ts(0xffffffff, 0), {}], {"g": 1}])
    data = b"".join(recs)
    cases = [([("regex", "log a")], None), ([("exclude", "log a")], None), ([("regex", "log .*")], None),
             ([("regex", "log nomatch")], None), ([("regex", "$k['sub'][1] /hello/i")], None),
             ([("regex", "log ^$")], None), ([("regex", "log é")], None), ([("regex", "log [^a-z ]")], None),
             ([("exclude", "log a"), ("regex", "log b")], None)]
    for rules, op in cases:
        for d in (data, grp + data, data + b"\x92\x01", recs[0], grp + recs[0] + grp):
            a, b = both_grep(g, d, rules, op)
            assert a == b, (rules, d[:40], a[0], b[0], first_diff(a[1], b[1]))


def test_grep_names_of_every_length(g):
    """round 6: the one-pass kernel (glane_kernels.inc) compares a record's names with the rules' as masked dwords -- names of 1 .. 33
    bytes (33: the three launches take the filter), names that differ in their last byte only, the last of two entries of a name, a
    nested name; against the oracle under every Logical_Op"""
    names = ["k" * n for n in range(1, 34)] + ["direction", "directioN", "namespace_name", "namespace_nam3", "abcdefgh", "abcdefgX", "abcdefghi"]
    recs = []
    for i, nm in enumerate(names):
        recs.append(_rec({nm: "hit%d" % (i % 3), "pad": "x" * (i % 7)}))
        recs.append(_rec(synth.KV([(nm, "first"), ("z", 1), (nm, "hit1")])))
        recs.append(_rec({"outer": {nm: "hit2"}, nm[:-1] + "?": "hit0"}))
    data = b"".join(recs)
    for nm in names:
        for rules, op in (([("regex", "%s ^hit1$" % nm)], None), ([("exclude", "%s hit[02]" % nm), ("regex", "pad x")], None),
                          ([("regex", "%s hit1" % nm), ("regex", "$outer['%s'] hit2" % nm)], "OR"), ([("exclude", "%s t1$" % nm), ("exclude", "pad ^$")], "AND")):
            a, b = both_grep(g, data, rules, op)
            assert a == b, (nm, rules, op, a[0], b[0], first_diff(a[1], b[1]))


def test_filter_parser_semantics(g):
    ty = dict(regex=r"^(?<INT>[^ ]+) (?<FLOAT>[^ ]+) (?<BOOL>[^ ]+) (?<STRING>.+)$", types="INT:integer BOOL:bool STRING:string")
    src = _rec({"data": "100 0.5 true x", "extra": "y"}) + _rec({"data": "-7 1 nope zz zz", "n": {"a": [1, 2.5, None, True]}}) + _rec({"nodata": 1})
    for reserve, preserve in [(False, False), (True, False), (True, True), (False, True)]:
        o, q = both_parser(g, src, "data", [ty], reserve, preserve)
        assert o == q, (reserve, preserve, first_diff(o[1], q[1]))
    pt = dict(regex=r"^(?<time>[^ ]+) (?<msg>.*)$", time_fmt="%Y-%m-%dT%H:%M:%S.%L", time_key="time")
    src = _rec({"log": "2017-11-01T22:25:21.648 hello"}) + _rec({"log": "2017-11-01T22:25:21 nofrac"}) + _rec({"log": "garbage here"})
    for keep in (False, True):
        o, q = both_parser(g, src, "log", [dict(pt, time_keep=keep)])
        assert o == q, first_diff(o[1], q[1])
    # map32 / int canonicalisation of unparsed bodies and metadata
    raw = b"\x92\x92" + synth.ext_ts(7, 9).b + b"\xdf\x00\x00\x00\x01\xa1m\xd0\x05" + b"\xdf\x00\x00\x00\x01\xa3log\xd9\x03abc"
    o, q = both_parser(g, raw + raw, "log", [pt])
    assert o == q, first_diff(o[1], q[1])
    # several parsers, skip_empty on/off, record accessor key, duplicate keys
    pa = dict(regex=r"^(?<a>\d+)$"); pb = dict(regex=r"^(?<b>\w+)$")
    o, q = both_parser(g, _rec({"log": "abc"}) + _rec({"log": "123"}) + _rec({"log": "!!"}), "log", [pa, pb])
    assert o == q, first_diff(o[1], q[1])
    for se in (True, False):
        o, q = both_parser(g, _rec({"log": " yy"}) + _rec({"log": "xx "}), "log", [dict(regex=r"^(?<a>x*) (?<b>y*)$", skip_empty=se)])
        assert o == q, first_diff(o[1], q[1])
    o, q = both_parser(g, _rec({"k": {"in": "42"}}) + _rec({"k": "17"}), "$k['in']", [pa], True, True)
    assert o == q, first_diff(o[1], q[1])
    dup = _rec(synth.KV([("log", "12"), ("z", 1), ("log", "34")])) + _rec(synth.KV([("log", "12"), ("log", "xx")]))
    for reserve, preserve in [(False, False), (True, False), (False, True)]:
        o, q = both_parser(g, dup, "log", [pa], reserve, preserve)
        assert o == q, first_diff(o[1], q[1])
    # legacy records, group markers, float/int timestamps, bad tail
    grp = synth.mp([[synth.ext_ts(0xffffffff, 0), {}], {"g": 1}])
    mix = grp + synth.legacy_record(5, {"log": "77"}) + synth.legacy_record(1.25, {"log": "x"}) + _rec({"log": "9"}, 3, 4, {"m": 1})
    o, q = both_parser(g, mix, "log", [pa])
    assert o == q, first_diff(o[1], q[1])
    o, q = both_parser(g, mix + b"\x93\x01\x02\x03" + _rec({"log": "5"}), "log", [pa])
    assert o == q, first_diff(o[1], q[1])
    o, q = both_parser(g, mix + b"\x92\x01", "log", [pa])
    assert o == q, first_diff(o[1], q[1])


def test_time_text_fixed_layout_and_interpreter_agree(g):
    """k_parser_finish reads fixed-width time texts at fixed offsets and leaves everything else to the
    strptime interpreter: both must give flb_parser_time_lookup's answer (src/flb_parser.c:1899-2040,
    src/flb_strptime.c) for texts on either side of every condition of the fast path."""
    rng = random.Random(21)
    texts = ["10/Oct/2000:13:55:36 -0700", "01/jan/1970:00:00:00 +0000", "31/DEC/9999:23:59:60 +1430", "7/Oct/2000:13:55:36 -0700",
             "07/October/2000:13:55:36 -0700", "07/Octo/2000:13:55:36 -0700", "07/May/2000:13:55:36 -0700", "07/Mayo/2000:13:55:36 +0100",
             "07/Mar/2000:13:55:36 +0100", "07/Marc/200:13:55:36 +0100", "45/Oct/2000:13:55:36 -0700", "32/Oct/2000:13:55:36 -0700",
             "00/Oct/2000:13:55:36 -0700", "10/Oct/2000:24:55:36 -0700", "10/Oct/2000:23:60:36 -0700", "10/Oct/2000:23:59:61 -0700",
             "10/Oct/2000:23:59:59  -0700", "10/Oct/2000:23:59:59\t-0700", "10/Oct/2000:23:59:59-0700", "10/Oct/2000:23:59:59 -07:00",
             "10/Oct/2000:23:59:59 -07", "10/Oct/2000:23:59:59 -070", "10/Oct/2000:23:59:59 Z", "10/Oct/2000:23:59:59 GMT",
             "10/Oct/2000:23:59:59 -07x0", "10/Oct/2000:23:59:59 +9999", "10/Oct/20000:23:59:59 +0000", "10/Oct/0000:23:59:59 +0000",
             "10/Oct/2000:23:59:59 *0700", "1O/Oct/2000:23:59:59 +0700", "10/Oct/2000:23:59:5 +0700", "10/Oct/2000:23:59:59 +0700 trailing",
             "10/0ct/2000:23:59:59 +0700", "29/Feb/2001:12:00:00 +0000", "31/Apr/2024:12:00:00 +0000", "", "x"]
    for _ in range(300):
        t = list(rng.choice(texts[:3]))
        for _ in range(rng.randrange(1, 3)):
            t[rng.randrange(len(t))] = rng.choice("0123456789/: +-OctZ\x00")
        texts.append("".join(t))
    recs = b"".join(_rec({"log": "h - u [%s] \"GET /x HTTP/1.1\" 200 5" % t}) for t in texts)
    for extra in (dict(), dict(time_keep=True), dict(time_strict=False)):
        o, q = both_parser(g, recs, "log", [dict(regex=APACHE2, time_fmt=TF, time_key="time", **extra)])
        assert o == q, (extra, first_diff(o[1], q[1]))
    # a format without %z (fixed Time_Offset), numeric month, seconds right before the end
    t2 = ["2017-11-01 22:25:21", "2017-11-1 22:25:21", "2017-13-01 22:25:21", "2017-00-01 22:25:21", "2017-11-01  22:25:21",
          "2017-11-01 22:25:2", "2017-11-01 22:25:211", "201-11-01 22:25:21", "2017-11-01T22:25:21", "0000-01-01 00:00:00"]
    recs = b"".join(_rec({"log": "%s|msg" % t}) for t in t2)
    for off in (None, "+0530", "-0100"):
        o, q = both_parser(g, recs, "log", [dict(regex=r"^(?<time>[^|]*)\|(?<m>.*)$", time_fmt="%Y-%m-%d %H:%M:%S", time_key="time", time_offset=off)])
        assert o == q, (off, first_diff(o[1], q[1]))


def test_map16_header_width(g):
    names = ["g%02d" % i for i in range(16)]
    rx = "^" + " ".join("(?<%s>[a-z]*)" % n for n in names) + "$"
    line = " ".join(["ab"] * 15 + [""])
    o, q = both_parser(g, _rec({"log": line}) + _rec({"log": " ".join(["ab"] * 16)}), "log", [dict(regex=rx)])
    assert o == q, first_diff(o[1], q[1])


def test_random_mutations_parity(g):
    """fuzz: mutated apache lines (ASCII and UTF-8 insertions) through parser then grep"""
    rng = random.Random(11)
    data, off, ep = synth.apache_records(4000)
    lines = [bytes(data[int(off[i]) + 21:int(off[i + 1])]) for i in range(4000)]
    recs = []
    for ln in lines:
        m = bytearray(ln)
        for _ in range(rng.randint(0, 3)):
            if len(m) < 2:
                break
            k = rng.randrange(len(m))
            op = rng.random()
            if op < 0.4: m[k] = rng.choice(b' "[]\n-x')
            elif op < 0.7: del m[k]
            else: m = m[:max(k, 1)]
        # multi-byte characters are inserted last and only between characters, so the text stays
        # well-formed UTF-8 (malformed sequences are a documented deviation)
        chars = list(m.decode("ascii"))
        for _ in range(rng.randint(0, 2)):
            chars.insert(rng.randrange(len(chars) + 1), rng.choice(["é", "日本", "ü", "😀"]))
        recs.append(_rec({"log": "".join(chars).encode()}, rng.randrange(2**31), rng.randrange(10**9)))
    blob = b"".join(recs)
    o, q = both_parser(g, blob, "log", [dict(regex=APACHE2, time_fmt=TF, time_key="time"), dict(regex=APACHE, time_fmt=TF, time_key="time")])
    assert o == q, first_diff(o[1], q[1])
    a, b = both_grep(g, o[1], [("regex", r"code ^[45]"), ("exclude", "agent curl")])
    assert a == b, first_diff(a[1], b[1])


def test_device_level_chain_matches_host_level(g):
    data, off, ep = synth.apache_records(5000)
    L = g.lib()
    d_data = L.flbgpu_dev_alloc(len(data)); d_off = L.flbgpu_dev_alloc(off.nbytes)
    L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, len(data)); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
    p = g.Parser(APACHE2, time_fmt=TF, time_key="time")
    fp = g.FilterParser("log", [p]); fg = g.FilterGrep([("regex", r"code ^5\d\d$")])
    ch = g.DevChunk(d_data, d_off, 5000, len(data))
    r1, o1 = fp.filter_dev(ch)
    assert r1 == g.MODIFIED and fp.counts() == (5000, 5000)
    r2, o2 = fg.filter_dev(o1)
    assert r2 == g.MODIFIED
    out = np.empty(o2.bytes, dtype=np.uint8)
    L.flbgpu_memcpy_d2h(out.ctypes.data, o2.data, o2.bytes)
    po = ob.Parser(APACHE2, time_fmt=TF, time_key="time")
    _, want1 = ob.FilterParser("log", [po]).filter(bytes(data))
    _, want2 = ob.Grep([("regex", r"code ^5\d\d$")]).filter(want1)
    assert bytes(out) == want2
    assert fg.counts() == (5000, ob.count_records(want2))
    L.flbgpu_dev_free(d_data); L.flbgpu_dev_free(d_off)


def oracle_chain(filters, data):
    """flb_filter_do (src/flb_filter.c:121-325) with oracle filters"""
    work, modified = data, False
    for f in filters:
        r = f.filter(work)
        if isinstance(r, tuple):
            r, out = r
        else:
            out = b""                      # log_to_metrics: MODIFIED only with discard_logs (empty output)
        if r != ob.MODIFIED:
            continue
        work, modified = out, True
        if len(out) == 0:
            break
    return (ob.MODIFIED, work) if modified else (ob.NOTOUCH, None)


def test_filter_chain_matches_flb_filter_do(g):
    data, off, ep = synth.apache_records(30000)
    data = bytes(data)
    pa = dict(regex=APACHE2, time_fmt=TF, time_key="time")
    cases = [
        [("parser", "log"), ("grep", [("regex", r"code ^5\d\d$")])],
        [("grep", [("regex", "log HTTP")]), ("parser", "log"), ("grep", [("exclude", "method GET")])],   # first filter keeps everything: NOTOUCH
        [("parser", "log"), ("grep", [("regex", "code ^9")]), ("grep", [("regex", "code .")])],          # everything dropped: chain stops
        [("parser", "log"), ("l2m", dict(metric_mode="counter", props=[("label_field", "code")])), ("grep", [("regex", "code ^2")])],
        [("parser", "log"), ("l2m", dict(metric_mode="counter", props=[("label_field", "code")], discard_logs=True)), ("grep", [("regex", "code ^2")])],
        [("grep", [("regex", "nokey x")])],
    ]
    for spec in cases:
        of, gf, keep = [], [], []
        for kind, arg in spec:
            if kind == "parser":
                op, gp = ob.Parser(**pa), g.Parser(**pa)
                keep.append(gp)
                of.append(ob.FilterParser(arg, [op])); gf.append(g.FilterParser(arg, [gp]))
            elif kind == "grep":
                of.append(ob.Grep(arg)); gf.append(g.FilterGrep(arg))
            else:
                of.append(ob.L2M(**arg)); gf.append(g.FilterLogToMetrics(**arg))
        for payload in (data, data + b"\xc1trailing"):
            want = oracle_chain(of, payload)
            ch = g.FilterChain(gf)
            got = ch.filter(payload)
            assert got[0] == want[0], (spec, ch.last_stats())
            assert got[1] == want[1], (spec, first_diff(want[1], got[1]))
        for o_, g_ in zip(of, gf):
            if isinstance(o_, ob.L2M):
                assert [(x["labels"], x["value"]) for x in g_.snapshot()] == [(x["labels"], x["value"]) for x in o_.snapshot()[2]]
        for f in gf:
            f.close()
        for p in keep:
            p.close()


def test_types_float_and_friends(g):
    # flb_parser_typecast (src/flb_parser.c:2067-2164): atof on the field text -- incl. literals whose
    # rounding needs the exact path (rewritten by k_parser_emit_exact), junk, hex floats, inf/nan
    rx = r"^(?<i>[^ ]*) (?<f>[^ ]*) (?<b>[^ ]*) (?<h>[^ ]*) (?<s>.*)$"
    types = "i:integer f:float b:bool h:hex s:string"
    floats = ["1.5", "0.1", "-2.5e3", "abc", "", "1e400", "-1e-400", "0x1.8p1", "inf", "nan", "9007199254740993",
              "2.4703282292062328e-324", "1.7976931348623157e308", "123456789012345678901234567890.5", "  7", "1,5", "12abc",
              "0.30000000000000004", "8.41e21", "1e23", "3.141592653589793238462643383279502884197", "+.5", ".", "-"]
    rng = random.Random(12)
    recs = []
    for i in range(3000):
        f = rng.choice(floats) if rng.random() < 0.7 else repr(rng.uniform(-1, 1) * 10.0 ** rng.randrange(-30, 30))
        if rng.random() < 0.1: f = "%d.%d" % (rng.randrange(10), rng.getrandbits(90))
        line = "%s %s %s %s tail %d" % (rng.choice(["12", "-7", "x", "99999999999"]), f, rng.choice(["true", "FALSE", "maybe"]),
                                        rng.choice(["ff", "0x10", "zz"]), i)
        recs.append(synth.v2_record(1700000000 + i, 0, {"log": line, "other": i}))
    data = b"".join(recs)
    for reserve, preserve in ((False, False), (True, True)):
        o, q = both_parser(g, data, "log", [dict(regex=rx, types=types, skip_empty=False)], reserve, preserve)
        assert o[0] == q[0] == ob.MODIFIED
        assert o[1] == q[1], first_diff(o[1], q[1])


def _rand_obj(rng, depth=0):
    t = rng.randrange(12 if depth < 3 else 8)
    if t == 0: return None
    if t == 1: return rng.choice([True, False])
    if t == 2: return rng.choice([0, 1, 127, 128, 255, 256, 65535, 65536, 2 ** 32, 2 ** 63, -1, -32, -33, -128, -129, -32769, -2 ** 31 - 1])
    if t == 3: return rng.choice([0.5, -1e10, 3.14])
    if t == 4: return synth.Raw(b"\xca" + struct.pack(">f", rng.random()))
    if t == 5: return synth.Raw(b"\xc4\x03abc")                                  # bin
    if t == 6: return synth.Raw(b"\xc7\x02\x05xy")                               # ext
    if t < 9: return rng.choice(["", "x", "GET /a HTTP/1.1", "500", "é", "y" * rng.choice([31, 32, 255, 256, 70000])])
    if t == 9: return [_rand_obj(rng, depth + 1) for _ in range(rng.randrange(0, 18))]
    return synth.KV([(rng.choice(["a", "b", "log", "code", 7, None, "k%d" % rng.randrange(20)]), _rand_obj(rng, depth + 1)) for _ in range(rng.randrange(0, 18))])


def test_structural_fuzz_all_filters(g):
    """random record shapes (every msgpack family, nested containers, non-string keys, duplicate keys,
    legacy / V2 / float / integer timestamps, metadata, group markers) and byte-level corruptions,
    through filter_parser, filter_grep and filter_log_to_metrics against the oracle"""
    rng = random.Random(2026)
    pa = dict(regex=r"^(?<method>[A-Z]+) (?<path>[^ ]*)(?: (?<proto>.*))?$")
    for trial in range(12):
        recs = []
        for i in range(400):
            body = synth.KV([(rng.choice(["log", "code", "a", "b", "nest", 5, "log"]), _rand_obj(rng)) for _ in range(rng.randrange(0, 7))])
            r = rng.random()
            if r < 0.55: rec = synth.v2_record(rng.randrange(2 ** 32), rng.randrange(10 ** 9), body, rng.choice([None, {}, {"m": [1, {"x": "y"}]}]))
            elif r < 0.7: rec = synth.legacy_record(rng.choice([5, 2 ** 31, 1.5, 1e9 + 0.25]), body)
            elif r < 0.75: rec = synth.mp([[synth.ext_ts(rng.choice([0xffffffff, 0xfffffffe]), 0), {}], body])
            elif r < 0.8: rec = synth.mp([[rng.randrange(10 ** 6), {}], body])
            elif r < 0.85: rec = synth.mp([[synth.ext_ts(7, 10 ** 9 + 5), {}], body])              # invalid nsec
            else: rec = synth.v2_record(1, 2, body)
            recs.append(rec)
        data = b"".join(recs)
        if trial % 3 == 1:
            # corrupt one byte / truncate / splice junk somewhere: everything before it must still agree
            k = rng.randrange(len(data))
            data = rng.choice([data[:k] + bytes([rng.randrange(256)]) + data[k + 1:], data[:k], data[:k] + b"\xc1" + data[k:]])
        for reserve, preserve in ((False, False), (True, True)):
            o, q = both_parser(g, data, "log", [pa], reserve, preserve)
            assert o == q, (trial, first_diff(o[1], q[1]))
        for rules, op in (([("regex", "log ^GET"), ("exclude", "code ^5")], None), ([("regex", "a x"), ("regex", "$nest['a'] y")], "OR"),
                          ([("exclude", "log HTTP"), ("exclude", "b ^$")], "AND")):
            a, b = both_grep(g, data, rules, op)
            assert a == b, (trial, rules, first_diff(a[1], b[1]))
        props = [("label_field", "code"), ("add_label", "n $nest['a']"), ("exclude", "log ^POST")]
        for mode, vf in (("counter", None), ("histogram", "a"), ("gauge", "b")):
            om = ob.L2M(mode, props, value_field=vf); gm = g.FilterLogToMetrics(mode, props, value_field=vf)
            assert om.filter(data) == gm.filter(data)[0]
            want, got = om.snapshot()[2], gm.snapshot()
            assert [x["labels"] for x in got] == [x["labels"] for x in want], (trial, mode)
            for x, y in zip(got, want):
                assert x["buckets"] == y["buckets"] and x["count"] == y["count"], (trial, mode)
                assert struct.pack("<d", x["value"]) == struct.pack("<d", y["value"]) or (x["value"] != x["value"] and y["value"] != y["value"])
                # the exact sum rounded once against the reference's sequential f64 sum: the bound of tests/test_l2m_gpu.py
                # ((n - 1) roundings of the running sum); the fuzzed values are few and of one magnitude per series or cancel to
                # sums far above the roundings
                tol = max(x["count"] - 1, 0) * 2.0 ** -52 * max(abs(x["sum"]), abs(y["sum"]), 1e-300) * 4 if x["sum"] == x["sum"] and y["sum"] == y["sum"] else 0
                assert x["sum"] == y["sum"] or (x["sum"] != x["sum"] and y["sum"] != y["sum"]) or abs(x["sum"] - y["sum"]) <= tol, (x["sum"], y["sum"], x["count"])
            gm.close()


def test_parser_do_scalar_entry_point(g):
    # flb_parser_do (src/flb_parser.c:1784-1806): msgpack map, parsed time and the "last byte consumed"
    # return value (end of the last named group that took part) -- tests/internal/parser_regex.c style
    cases = [
        (dict(regex=r"^(?<a>\d+) (?<b>[a-z]+)(?: (?<c>.+))?$"), [b"12 abc", b"12 abc tail here", b"x", b"7 z "]),
        (dict(regex=APACHE2, time_fmt=TF, time_key="time"),
         [b'10.0.0.1 - bob [10/Oct/2000:13:55:36 -0700] "GET /a HTTP/1.0" 200 2326 "http://r" "agent x"', b"garbage"]),
        (dict(regex=r"(?<k>[a-z]+)=(?<v>\d+)", types="v:integer"), [b"..foo=42;;", b"nothing here"]),
    ]
    for pargs, inputs in cases:
        po, pg = ob.Parser(**pargs), g.Parser(**pargs)
        for s in inputs:
            want, got = po.do(s), pg.do(s)
            assert got == want, (pargs["regex"][:30], s, got, want)
        pg.close()


def test_full_size_properties_10M(g):
    """BASELINE configs[1] at full size (10 M records, 2.77 GB): properties that do not need a 10 M-record
    oracle run -- (0) the device indexer reproduces the record offsets from the raw bytes; (a) filtering is linear over concatenation: the two halves filtered separately give
    the bytes of the whole; (b) a random sample of output rows equals the oracle's output for the same
    input rows, byte for byte; (c) three independent kernels agree on the 5xx count (grep's kept
    records, log_to_metrics' counter by code, the sample)."""
    import hashlib
    n = 10_000_000
    data, off, ep = synth.apache_records(n)
    L = g.lib()
    nbytes = int(data.nbytes)
    d_data = L.flbgpu_dev_alloc(nbytes); d_off = L.flbgpu_dev_alloc(off.nbytes)
    assert d_data and d_off
    L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, nbytes); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
    p = g.Parser(APACHE2, time_fmt=TF, time_key="time")
    fp = g.FilterParser("log", [p]); fg = g.FilterGrep([("regex", r"code ^5\d\d$")])

    def run(chunk):
        r1, o1 = fp.filter_dev(chunk)
        assert r1 == g.MODIFIED
        parsed = np.empty(int(o1.bytes), dtype=np.uint8)
        L.flbgpu_memcpy_d2h(parsed.ctypes.data, o1.data, int(o1.bytes))
        poff = np.empty(int(o1.n) + 1, dtype=np.uint64)
        L.flbgpu_memcpy_d2h(poff.ctypes.data, o1.row_off, poff.nbytes)
        r2, o2 = fg.filter_dev(o1)
        assert r2 == g.MODIFIED
        kept = np.empty(int(o2.bytes), dtype=np.uint8)
        L.flbgpu_memcpy_d2h(kept.ctypes.data, o2.data, int(o2.bytes))
        return parsed, poff, kept, fg.counts()[1]

    parsed, poff, kept, nkept = run(g.DevChunk(d_data, d_off, n, nbytes))
    assert len(poff) == n + 1 and int(poff[-1]) == len(parsed)
    # (0) the device record indexer finds, from the raw bytes, exactly the boundaries the generator wrote
    ix = g.Indexer()
    ich, consumed = ix.index_dev(d_data, nbytes)
    assert int(ich.n) == n and consumed == nbytes, (int(ich.n), consumed, ix.stats())
    ioff = np.empty(n + 1, dtype=np.uint64)
    L.flbgpu_memcpy_d2h(ioff.ctypes.data, ich.row_off, ioff.nbytes)
    assert np.array_equal(ioff, np.asarray(off, dtype=np.uint64))
    del ix
    # (a) halves: row offsets of the second half are rebased on the device copy of the same bytes
    h = n // 2
    off2 = (off[h:] - off[h]).astype(np.uint64)
    d_off2 = L.flbgpu_dev_alloc(off2.nbytes)
    L.flbgpu_memcpy_h2d(d_off2, off2.ctypes.data, off2.nbytes)
    pa, _, ka, na = run(g.DevChunk(d_data, d_off, h, int(off[h])))
    pb, _, kb, nb = run(g.DevChunk(d_data + int(off[h]), d_off2, n - h, nbytes - int(off[h])))
    sha = lambda *arrs: hashlib.sha256(b"".join(memoryview(a) for a in arrs)).hexdigest()
    assert sha(pa, pb) == sha(parsed) and sha(ka, kb) == sha(kept) and na + nb == nkept
    # (b) sample against the oracle
    rng = random.Random(5)
    idx = sorted(rng.sample(range(n), 4000))
    sample_in = b"".join(bytes(data[int(off[i]):int(off[i + 1])]) for i in idx)
    po = ob.Parser(APACHE2, time_fmt=TF, time_key="time")
    _, want = ob.FilterParser("log", [po]).filter(sample_in)
    got = b"".join(bytes(parsed[int(poff[i]):int(poff[i + 1])]) for i in idx)
    assert got == want, first_diff(want, got)
    _, want_kept = ob.Grep([("regex", r"code ^5\d\d$")]).filter(want)
    # (c) 5xx counts: grep vs log_to_metrics vs the sample's proportion
    fm = g.FilterLogToMetrics("counter", [("label_field", "code")])
    r, o1 = fp.filter_dev(g.DevChunk(d_data, d_off, n, nbytes))
    fm.filter_dev(o1)
    snap = fm.snapshot()
    assert sum(x["value"] for x in snap) == n
    assert sum(x["value"] for x in snap if x["labels"][0].startswith(b"5")) == nkept
    assert abs(ob.count_records(want_kept) / 4000 - nkept / n) < 0.02
    for f in (fp, fg, fm):
        f.close()
    p.close()
    for d in (d_data, d_off, d_off2):
        L.flbgpu_dev_free(d)


def _chain_both(g, blob, pargs, rules, op=None):
    """flb_filter_do over [filter_parser, filter_grep]: the device chain (fused pair when the configuration allows)
    against the oracle's two filters run one after the other."""
    po = ob.Parser(**pargs)
    r1, o1 = ob.FilterParser("log", [po]).filter(blob)
    cur = o1 if r1 == ob.MODIFIED else blob
    r2, o2 = ob.Grep(rules, op).filter(cur)
    want = (ob.MODIFIED, o2) if r2 == ob.MODIFIED else ((ob.MODIFIED, o1) if r1 == ob.MODIFIED else (ob.NOTOUCH, None))
    pg = g.Parser(**pargs)
    fp = g.FilterParser("log", [pg]); fg = g.FilterGrep(rules, op)
    ch = g.FilterChain([fp, fg])
    got = ch.filter(blob)
    stats = ch.last_stats()
    fp.close(); fg.close(); pg.close()
    return want, got, stats, (r1, o1, r2, o2)


def test_fused_parser_grep_pair(g):
    """the one-pass evaluation of [filter_parser, filter_grep] (fused_kernels.inc) is flb_filter_do over the two"""
    rng = random.Random(21)
    data, off, ep = synth.apache_records(6000)
    recs = []
    for i in range(6000):
        m = bytearray(data[int(off[i]) + 21:int(off[i + 1])])
        t = rng.random()
        if t < 0.15:                                   # lines the parser refuses / parses differently
            k = rng.randrange(len(m))
            m[k] = rng.choice(b' "[]\nx')
        elif t < 0.2:
            m = m[:rng.randrange(1, len(m))]
        body = {"log": bytes(m)}
        if t > 0.97: body = {"other": b"x", "code": b"503"}            # no Key_Name: unparsed, grep sees the original keys
        if 0.95 < t <= 0.97: body = {"log": 5}
        sec = rng.randrange(2**31)
        if t < 0.01: sec = 0xffffffff                                   # group marker
        recs.append(synth.mp([[synth.ext_ts(sec, 0 if sec == 0xffffffff else rng.randrange(10**9)), {}], body]))
    blob = b"".join(recs)
    base = dict(regex=APACHE2, time_fmt=TF, time_key="time")
    cases = [
        (base, [("regex", r"code ^5\d\d$")], None),
        (base, [("exclude", "method GET")], None),
        (base, [("regex", "code ^2"), ("regex", "agent curl")], "AND"),
        (base, [("regex", "code ^404$"), ("regex", "method ^P"), ("regex", "nokey x")], "OR"),
        (base, [("exclude", "code ^2"), ("exclude", "code ^3")], "OR"),
        (base, [("regex", "nokey x")], None),                           # nothing kept
        (base, [("regex", "log .")], None),                             # key only in unparsed records
        (base, [("regex", "time 2")], None),                            # the time field is consumed (Time_Keep off): key absent
        (dict(base, time_keep=True), [("regex", "time ^1")], None),
        (dict(base, skip_empty=False), [("regex", "user ^-$"), ("exclude", "referer ^$")], None),
        (dict(regex=r"^(?<x>\S+) (?<x>\S+) (?<y>.*)$"), [("regex", r"x ^-$")], None),          # duplicate name: the LAST entry decides
        (dict(regex=r"(?<w>[a-z]+)/(?<v>\d)"), [("regex", "w ^(http|curl)$")], None),       # not start-anchored: reverse pass
        (base, [("regex", "host ."), ("exclude", "nokey x")], None),    # keeps every parsed record: grep is NOTOUCH
    ]
    for pargs, rules, op in cases:
        want, got, stats, parts = _chain_both(g, blob, pargs, rules, op)
        assert got[0] == want[0], (rules, op, got[0], want[0])
        assert got[1] == want[1], (rules, op, first_diff(want[1], got[1]))
        r1, o1, r2, o2 = parts
        if r1 == ob.MODIFIED:
            assert stats[0]["out_records"] == ob.count_records(o1) and stats[0]["out_bytes"] == len(o1), (rules, stats)
        if r2 == ob.MODIFIED:
            assert stats[1]["out_records"] == ob.count_records(o2) and stats[1]["out_bytes"] == len(o2), (rules, stats)
    # a malformed record ends the parser's loop: everything from it on is missing
    broken = blob[:len(blob) // 2] + b"\x93\x01\x02\x03" + blob[len(blob) // 2:]
    want, got, stats, parts = _chain_both(g, broken, base, [("regex", r"code ^5\d\d$")], None)
    assert got == want, first_diff(want[1], got[1])


def test_grep_32_rules_baseline_config2(g):
    """BASELINE configs[2]: the 32-pattern set of bench.py (16 Regex rules in OR mode, then 16 Exclude rules in OR
    mode: two filter_grep instances, grep.c:90-98) on NDJSON-shaped records, every rule / both instances / the
    chain against the oracle."""
    import json as _json
    from bench import GREP32_REGEX, GREP32_EXCLUDE
    rng = random.Random(77)
    recs = []
    for i in range(20000):
        d = {"time": "2026-09-21T10:%02d:%02d.%03dZ" % (rng.randrange(60), rng.randrange(60), rng.randrange(1000)),
             "level": rng.choice(["info", "warn", "error", "debug", "nothing", " info"]),
             "msg": "request %d finished %s" % (rng.randrange(10 ** 6), rng.choice(["ok", "timeout", "refused", "ok again"])),
             "code": rng.randrange(200, 600), "latency": round(rng.random() * 100, 3),
             "svc": {"name": rng.choice(["api", "db", "cache", ""]), "pod": "pod-%d" % rng.randrange(1000)},
             "path": "/v%d/items/%d?x=%d" % (rng.choice([1, 1, 1, 2]), rng.randrange(10 ** 5), rng.randrange(100)), "bytes": rng.randrange(10 ** 6)}
        if rng.random() < 0.03: d.pop("msg")
        if rng.random() < 0.03: d["svc"] = "db"                       # not a map: the sub-key rules see no STR
        recs.append(_rec({k: (v.encode() if isinstance(v, str) else ({kk: vv.encode() for kk, vv in v.items()} if isinstance(v, dict) else v))
                          for k, v in d.items()}, 5, i))
    blob = b"".join(recs)
    assert len(GREP32_REGEX) == 16 and len(GREP32_EXCLUDE) == 16
    for rule in GREP32_REGEX + GREP32_EXCLUDE:                       # each of the 32 patterns on its own
        a, b = both_grep(g, blob, [rule])
        assert a == b, (rule, a[0], b[0], first_diff(a[1], b[1]))
    a1, b1 = both_grep(g, blob, GREP32_REGEX, "OR")
    assert a1 == b1 and a1[0] == ob.MODIFIED, first_diff(a1[1], b1[1])
    a2, b2 = both_grep(g, a1[1], GREP32_EXCLUDE, "OR")
    assert a2 == b2 and a2[0] == ob.MODIFIED, first_diff(a2[1], b2[1])
    assert 0 < ob.count_records(a2[1]) < ob.count_records(a1[1]) < 20000
    f1 = g.FilterGrep(GREP32_REGEX, "OR"); f2 = g.FilterGrep(GREP32_EXCLUDE, "OR")
    got = g.FilterChain([f1, f2]).filter(blob)
    f1.close(); f2.close()
    assert got == a2
    # the same 32 in AND mode and as one legacy list
    for rules, op in ((GREP32_REGEX, "AND"), (GREP32_EXCLUDE, "AND"), (GREP32_REGEX[:3] + GREP32_EXCLUDE, None)):
        a, b = both_grep(g, blob, rules, op)
        assert a == b, (op, first_diff(a[1], b[1]))
