"""Rules / parsers that are not regular expressions (look-around, atomic groups, possessive repeats, back-references) do not abort
start-up: the device does everything but the search of that pattern, the product's backtracking matcher (csrc/rxbt.inc) answers on
the host (flbgpu.cpp "host rules").  Checked against the REAL Onigmo's decisions on the same values (oracle/_ref/libonig_ref.so travels
with the snapshot) and against the oracle's filters run with an equivalent regular expression."""
import ctypes, os, random, sys
import numpy as np
import pytest
import oracle_binding as ob
import synth
import flbamd_loader
import rxdiff

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(rxdiff.load_ref() is None, reason="oracle/_ref/libonig_ref.so not built")
APACHE2 = r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$'
TF = "%d/%b/%Y:%H:%M:%S %z"


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


WORDS = ["GET /health 200", "GET /ping 200", "POST /api/v1/items 201", "error: disk full", "warn: retry 3", "user=alice id=alice", "user=bob id=carol",
         "'quoted' text", '"double" quoted"', "abcabc", "abcabd", "price 100 USD", "price 100USD", "tail\n", "", "é=é", "x" * 300, "aaa", "aab", "#comment", "k=v"]


def records(n, seed, keys=("log",), extra=True):
    rng = random.Random(seed)
    recs, vals = [], []
    for i in range(n):
        body = {}
        v = {}
        for k in keys:
            t = rng.choice(WORDS) + (" %d" % rng.randrange(1000) if rng.random() < 0.5 else "")
            body[k] = t
            v[k] = t
        if extra and rng.random() < 0.2:
            body["n"] = i
        if rng.random() < 0.05:
            body[keys[0]] = i                     # not a string: no rule matches it
            v[keys[0]] = None
        recs.append(synth.v2_record(1700000000 + i, i % 1000, body))
        vals.append(v)
    return recs, vals


def ref_match(ref, pat, value):
    if value is None:
        return False
    eng = rxdiff.RefRegex(ref, pat.encode() if isinstance(pat, str) else pat)
    assert eng.ok
    return eng.search(value.encode() if isinstance(value, str) else value) is not None


HOST_PATTERNS = [r"^(?!.*(?:health|ping)).*\d$", r"(?<=user=)(\w+) id=\1", r"(?>a+)b", r"^(['\"]).*\1", r"\d+(?! ?USD)\b", r"(abc)\1", r"error(?=:)|warn(?=:)", r"a++b",
                 # round 5: the absent operator and subexpression calls
                 r"^(?~ \d)$", r"/(?~/)/(?~/)$", r"(?<w>[a-z]+)=\g<w>", r"^(?<q>'(?:[^']|\g<q>)*')"]


@needs_ref
@pytest.mark.parametrize("pat", HOST_PATTERNS)
def test_grep_host_rule_against_the_real_engine(g, pat):
    ref = rxdiff.load_ref()
    recs, vals = records(6000, 5)
    blob = b"".join(recs)
    for kind in ("regex", "exclude"):
        f = g.FilterGrep([(kind, "log " + pat)])
        assert f.host_rules()["rules"] == 1
        r, out = f.filter(blob)
        keep = [ref_match(ref, pat, v["log"]) == (kind == "regex") for v in vals]
        want = b"".join(x for x, k in zip(recs, keep) if k)
        if all(keep):
            assert r == ob.NOTOUCH
        else:
            assert r == ob.MODIFIED and out == want
        st = f.host_rules()
        assert st["values"] > 0 and st["budget_over"] == 0
        # the device-level call: no host copy of the chunk at hand, the values come back from the device
        data = np.frombuffer(blob, dtype=np.uint8)
        n, off, cons = g.index_host(blob)
        L = g.lib()
        offs = np.array(off, dtype=np.uint64)
        d_data = L.flbgpu_dev_alloc(data.nbytes); d_off = L.flbgpu_dev_alloc(offs.nbytes)
        L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_off, offs.ctypes.data, offs.nbytes)
        chunk = g.DevChunk(d_data, d_off, n, data.nbytes)
        r2, o2 = f.filter_dev(chunk)
        if all(keep):
            assert r2 == ob.NOTOUCH
        else:
            got = np.empty(int(o2.bytes), dtype=np.uint8)
            L.flbgpu_memcpy_d2h(got.ctypes.data, o2.data, int(o2.bytes))
            assert r2 == ob.MODIFIED and got.tobytes() == want
        L.flbgpu_dev_free(d_data); L.flbgpu_dev_free(d_off)
        f.close()


@needs_ref
@pytest.mark.parametrize("op", [None, "and", "or"])
def test_host_and_device_rules_together(g, op):
    """a host rule between device rules, every Logical_Op: the loop's order and its early ends are the device's"""
    ref = rxdiff.load_ref()
    recs, vals = records(5000, 8, keys=("log", "msg"))
    blob = b"".join(recs)
    rules = [("regex", r"log \d"), ("regex", r"msg ^(?!.*(?:health|ping))"), ("regex", r"log [a-z]")] if op else \
            [("exclude", r"log ^#"), ("regex", r"msg (?<![a-z])\d+$"), ("exclude", r"log USD"), ("regex", r"log (\w)\1")]
    f = g.FilterGrep(rules, op)
    assert f.host_rules()["rules"] == (1 if op else 2)
    r, out = f.filter(blob)

    def rule(i, v):
        kind, kv = rules[i]
        key, pat = kv.split(" ", 1)
        return ref_match(ref, pat, v[key])
    keep = []
    for v in vals:
        if op is None:
            k = True
            for i, (kind, _) in enumerate(rules):
                m = rule(i, v)
                if not m:
                    if kind == "regex":
                        k = False
                        break
                else:
                    k = kind != "exclude"
                    break
            keep.append(k)
        else:
            found = False
            for i in range(len(rules)):
                found = rule(i, v)
                if (op == "or" and found) or (op == "and" and not found):
                    break
            keep.append(found)
    want = b"".join(x for x, k in zip(recs, keep) if k)
    assert r == ob.MODIFIED and out == want
    f.close()


def test_host_parser_equals_an_equivalent_regular_one(g):
    """the same captures by construction: a look-ahead that only repeats what the next atom demands, an atomic group / a possessive
    repeat where nothing could be given back anyway -- against the oracle's filter_parser with the plain pattern"""
    data, off, ep = synth.apache_records(20000)
    blob = bytes(data)
    host_rx = APACHE2.replace(r"^(?<host>[^ ]*) ", r"^(?=[^ ]* )(?<host>[^ ]*+) ").replace(r'(?<code>[^ ]*) ', r'(?<code>(?>[^ ]*)) ')
    assert host_rx != APACHE2
    p = g.Parser(host_rx, time_fmt=TF, time_key="time")
    f = g.FilterParser("log", [p])
    assert f.host_rules()["rules"] == 1
    r, out = f.filter(blob)
    ro, oo = ob.FilterParser("log", [ob.Parser(regex=APACHE2, time_fmt=TF, time_key="time")]).filter(blob)
    assert r == ro == ob.MODIFIED and out == oo
    st = f.host_rules()
    assert st["values"] == 20000 and st["unhandled"] == 0 and st["budget_over"] == 0
    # ... and in a chain in front of a device grep
    ch = g.FilterChain([f, g.FilterGrep([("regex", r"code ^5\d\d$")])])
    r2, o2 = ch.filter(blob)
    r3, o3 = ob.Grep([("regex", r"code ^5\d\d$")]).filter(oo)
    assert r2 == r3 == ob.MODIFIED and o2 == o3


def test_host_parser_that_rejects_lines(g):
    recs, vals = records(8000, 21)
    blob = b"".join(recs)
    p = g.Parser(r"^(?!#)(?<key>[^=]+)=(?<val>.*)$")
    f = g.FilterParser("log", [p], True, False)
    r, out = f.filter(blob)
    ro, oo = ob.FilterParser("log", [ob.Parser(regex=r"^(?<key>[^#=][^=]*)=(?<val>.*)$")], True, False).filter(blob)
    assert r == ro and out == oo


# (host form, the same language as a regular expression): pairs for lists of parsers
HOST_KV = (r"^(?!#)(?<key>[^=]+)=(?<val>.*)$", r"^(?<key>[^#=][^=]*)=(?<val>.*)$")
HOST_REQ = (r"^(?=[A-Z]+ /)(?<method>[A-Z]++) (?<path>\S+) (?<code>\d+)", r"^(?<method>[A-Z]+) (?<path>/\S*) (?<code>\d+)")
HOST_NUM = (r"(?<word>[a-z]+(?>:)) (?<rest>.*)$", r"(?<word>[a-z]+:) (?<rest>.*)$")
DEV_ALL = r"^(?<all>.+)$"
DEV_PRICE = r"^price (?<amount>\d+) ?(?<cur>[A-Z]+)"


@pytest.mark.parametrize("order", [
    ["KV", "ALL"], ["ALL", "KV"],                        # a host parser first / last of two
    ["PRICE", "KV", "ALL"],                              # ... between two device parsers
    ["REQ", "KV", "NUM", "PRICE"],                       # three host parsers, a device parser behind them: rows nobody takes stay as they are
    ["PRICE", "REQ", "ALL", "KV"],                       # a host parser behind a device parser that takes everything: never asked
])
@pytest.mark.parametrize("reserve", [False, True])
def test_host_parsers_in_a_list(g, order, reserve):
    """plugins/filter_parser/filter_parser.c:286-323: the list is tried in order on every value, the first parser that takes it wins.  A
    host parser's answers are computed before the list's kernel runs, which reads them in the parser's turn -- against the oracle's
    filter_parser over the same list with the equivalent regular expressions"""
    recs, vals = records(6000, 31)
    blob = b"".join(recs)
    host = {"KV": HOST_KV, "REQ": HOST_REQ, "NUM": HOST_NUM}
    dev = {"ALL": DEV_ALL, "PRICE": DEV_PRICE}
    gp = [g.Parser(host[k][0]) if k in host else g.Parser(dev[k]) for k in order]
    op = [ob.Parser(regex=host[k][1]) if k in host else ob.Parser(regex=dev[k]) for k in order]
    f = g.FilterParser("log", gp, reserve, False)
    assert f.host_rules()["rules"] == sum(1 for k in order if k in host)
    r, out = f.filter(blob)
    ro, oo = ob.FilterParser("log", op, reserve, False).filter(blob)
    assert r == ro and out == oo, g.last_error()
    assert f.host_rules()["unhandled"] == 0
    # ... the same list on a device-resident chunk in front of a device grep
    ch = g.FilterChain([f, g.FilterGrep([("exclude", "key ^user$")])])
    r2, o2 = ch.filter(blob)
    r3, o3 = ob.Grep([("exclude", "key ^user$")]).filter(oo if ro == ob.MODIFIED else blob)
    if r3 == ob.MODIFIED:
        assert r2 == ob.MODIFIED and o2 == o3
    else:
        assert (r2, o2) == (ro, oo if ro == ob.MODIFIED else None)
    f.close()


def test_a_list_takes_four_host_parsers(g):
    hp = [g.Parser(r"^(?!%s)(?<key>[^=]+)=(?<val>.*)$" % c) for c in "#;!%~"]
    with pytest.raises(ValueError, match="a list takes up to 4"):
        g.FilterParser("log", hp)
    assert g.FilterParser("log", hp[:4]).host_rules()["rules"] == 4
    assert g.FilterParser("log", hp[:1]).host_rules()["rules"] == 1


def test_refused_when_asked_to(g):
    os.environ["FLBGPU_NO_HOST_RULES"] = "1"
    try:
        with pytest.raises(Exception):
            g.FilterGrep([("regex", r"log a(?=b)")])
        with pytest.raises(Exception):
            g.Parser(r"^(?<a>x)(?=y)")
    finally:
        del os.environ["FLBGPU_NO_HOST_RULES"]
    with pytest.raises(Exception):
        g.FilterGrep([("regex", r"log a\xffb(?=c)")])        # what the host's matcher does not take either (a raw byte escape) is still refused


def l2m_same(a, b, mode="counter"):
    assert [s["labels"] for s in a] == [s["labels"] for s in b]
    for x, y in zip(a, b):
        if mode == "histogram":
            assert x["buckets"] == y["buckets"] and x["count"] == y["count"] and x["sum"] == y["sum"], (x, y)      # (integer observations: the sequential sum is exact)
        else:
            assert x["value"] == y["value"], (x, y)


L2M_EQUIV = [(r"^(?!#).", r"^[^#]"), (r"(?>a+)b", r"a+b"), (r"error(?=:)|warn(?=:)", r"(error|warn):"), (r"price \d++(?= ?USD)", r"price \d+ ?USD")]


@pytest.mark.parametrize("host,plain", L2M_EQUIV)
def test_log_to_metrics_host_rule_equals_an_equivalent_regular_one(g, host, plain):
    """filter_log_to_metrics: the rules run as a hidden filter_grep in front of the metric kernels (l2m.cpp l2m_gate) -- against the
    oracle's filter with a regular expression that decides the same"""
    chunks = [b"".join(records(5000, s, keys=("log", "msg"))[0]) for s in (31, 32)]
    for mode, vf, extra in (("counter", None, [("exclude", "msg USD")]), ("histogram", "n", [("regex", "msg .")]), ("gauge", "n", [])):
        props = lambda p: [("regex", "log " + p)] + extra + [("label_field", "msg"), ("add_label", "app demo")]
        f = g.FilterLogToMetrics(mode, props(host), value_field=vf)
        o = ob.L2M(mode, props(plain), value_field=vf)
        assert f.host_rules()["rules"] == 1
        for c in chunks:
            ro = o.filter(c)
            rg, out = f.filter(c)
            assert ro == rg == ob.NOTOUCH
        okeys, obounds, osn = o.snapshot()
        assert okeys == f.label_keys and len(osn) > 10
        l2m_same(f.snapshot(), osn, mode)
        assert f.host_rules()["values"] > 0 and f.host_rules()["budget_over"] == 0
        f.close()


@needs_ref
@pytest.mark.parametrize("kind", ["regex", "exclude"])
def test_log_to_metrics_host_rule_against_the_real_engine(g, kind):
    """a back-reference: no regular equivalent -- the oracle's filter without rules over the records the REAL engine keeps"""
    ref = rxdiff.load_ref()
    pat = r"(?<=user=)(\w+) id=\1"
    recs, vals = records(6000, 41)
    keep = [ref_match(ref, pat, v["log"]) == (kind == "regex") for v in vals]
    assert 0 < sum(keep) < len(keep)
    f = g.FilterLogToMetrics("counter", [(kind, "log " + pat), ("label_field", "log")], discard_logs=True)
    o = ob.L2M("counter", [("label_field", "log")], discard_logs=True)
    rg, out = f.filter(b"".join(recs))
    ro = o.filter(b"".join(x for x, k in zip(recs, keep) if k))
    assert rg == ro == ob.MODIFIED and out == b""
    l2m_same(f.snapshot(), o.snapshot()[2])
    # nothing passes: nothing counted, the series stay
    f2 = g.FilterLogToMetrics("counter", [("regex", r"log (zz)\1(?!z)"), ("label_field", "log")])
    assert f2.filter(b"".join(recs))[0] == ob.NOTOUCH and f2.snapshot() == []
    f.close(); f2.close()


def test_log_to_metrics_host_rule_in_a_chain_on_a_device_chunk_and_without_the_ahead_launch(g):
    recs, vals = records(7000, 51, keys=("log", "msg"))
    blob = b"".join(recs)
    host, plain = r"(?>a+)b|price(?= )", r"a+b|price "
    props = lambda p: [("regex", "log " + p), ("label_field", "msg")]
    pre = [("exclude", "msg ^#")]
    ro, oo = ob.Grep(pre).filter(blob)
    assert ro == ob.MODIFIED
    o = ob.L2M("counter", props(plain))
    assert o.filter(oo) == ob.NOTOUCH
    want = o.snapshot()[2]
    assert len(want) > 5
    # a device grep in front: the gate reads the values from a copy back of the grep's output
    fm = g.FilterLogToMetrics("counter", props(host))
    ch = g.FilterChain([g.FilterGrep(pre), fm])
    r, out = ch.filter(blob)
    assert r == ob.MODIFIED and out == oo
    l2m_same(fm.snapshot(), want)
    # the device-level call
    o2 = ob.L2M("counter", props(plain))
    assert o2.filter(blob) == ob.NOTOUCH
    f2 = g.FilterLogToMetrics("counter", props(host))
    data = np.frombuffer(blob, dtype=np.uint8)
    n, off, cons = g.index_host(blob)
    L = g.lib()
    offs = np.array(off, dtype=np.uint64)
    d_data = L.flbgpu_dev_alloc(data.nbytes); d_off = L.flbgpu_dev_alloc(offs.nbytes)
    L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_off, offs.ctypes.data, offs.nbytes)
    r2, _ = f2.filter_dev(g.DevChunk(d_data, d_off, n, data.nbytes))
    assert r2 == ob.NOTOUCH
    l2m_same(f2.snapshot(), o2.snapshot()[2])
    L.flbgpu_dev_free(d_data); L.flbgpu_dev_free(d_off)
    # the host-level call the usual way (a wait per counter)
    os.environ["FLBGPU_NO_SPEC"] = "1"
    try:
        f3 = g.FilterLogToMetrics("counter", props(host))
        assert f3.filter(blob)[0] == ob.NOTOUCH
        l2m_same(f3.snapshot(), o2.snapshot()[2])
    finally:
        del os.environ["FLBGPU_NO_SPEC"]


@pytest.mark.parametrize("reserve,preserve", [(False, False), (True, True)])
def test_host_parser_with_types_time_keep_and_the_reserve_options(g, reserve, preserve):
    """everything behind the capture search is the device's as always: Types casts, Time_Keep, Reserve_Data / Preserve_Key"""
    data, off, ep = synth.apache_records(6000)
    blob = bytes(data)
    host_rx = APACHE2.replace(r"^(?<host>[^ ]*) ", r"^(?=[^ ]* )(?<host>[^ ]*+) ")
    types = "code:integer size:integer"
    p = g.Parser(host_rx, time_fmt=TF, time_key="time", time_keep=True, types=types)
    f = g.FilterParser("log", [p], reserve, preserve)
    assert f.host_rules()["rules"] == 1
    r, out = f.filter(blob)
    ro, oo = ob.FilterParser("log", [ob.Parser(regex=APACHE2, time_fmt=TF, time_key="time", time_keep=True, types=types)], reserve, preserve).filter(blob)
    assert r == ro == ob.MODIFIED and out == oo
    assert f.host_rules()["unhandled"] == 0
