"""The reference engines' known answers ON THE DEVICE.

tests/golden/regex_kat.json (16.7 k answers of the real Onigmo, ill-formed UTF-8 included) and
tests/golden/strptime_kat.json (5.5 k answers of the real flb_strptime.c) are pushed through the HIP walkers
-- k_parser_rx / k_parser_generic (rx_reverse, rx_forward, u8_symbol), k_grep_match (dfa_match + the UTF-8
tables), k_parser_finish (time_fast, d_strptime) -- as filter_parser / filter_grep runs over one record per
case, and compared byte for byte with the oracle, which tests/test_oracle.py and tests/test_strptime_pin.py
pin on the same files: device == oracle == reference on every case.  Plus the boundary cases of the two
structural limits the kernels restate: msgpack-c's 32 open containers and the 64-bit mask of parsed
Key_Name entries."""
import base64, json, os, struct
import pytest
import oracle_binding as ob
import synth
import flbamd_loader

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


def _rec(body, sec=1, nsec=2):
    return synth.mp([[synth.ext_ts(sec, nsec), {}], body])


def first_diff(a, b):
    if a is None or b is None:
        return "one side None"
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return "byte %d: oracle %r gpu %r (len %d vs %d)" % (i, a[max(0, i - 20):i + 20], b[max(0, i - 20):i + 20], len(a), len(b))
    return "length %d vs %d" % (len(a), len(b))


def test_regex_kat_on_device(g):
    kat = json.load(open(os.path.join(HERE, "golden", "regex_kat.json")))
    n_pat = n_parser = n_cases = 0
    for ent in kat:
        pat = base64.b64decode(ent["pattern"])
        if not ent["compiles"] or pat.startswith(b"/") or b"\x00" in pat:
            continue                                    # (a leading '/' is flb_regex's option syntax, not the pattern's)
        # documented deviations of the table compiler: non-ASCII members of POSIX brackets, \b next
        # to non-ASCII characters
        dev = any(t in pat for t in (rb'[:', rb'\b', rb'\B'))
        subjects = [base64.b64decode(c[0]) for c in ent["cases"]]
        if dev:
            subjects = [s for s in subjects if all(c < 0x80 for c in s)]
        want = {base64.b64decode(c[0]): c[1] for c in ent["cases"]}
        blob = b"".join(_rec({"log": s}, 7, i) for i, s in enumerate(subjects))
        rules = [("regex", b"log " + pat)]
        try:
            fo = ob.Grep(rules)
        except ValueError:
            continue
        try:
            fg = g.FilterGrep(rules)
        except ValueError:
            continue                                    # refused at create (look-around, atomic groups, ...): fails loudly
        a, b = fo.filter(blob), fg.filter(blob)
        fg.close()
        assert a == b, (pat, first_diff(a[1], b[1]))
        # the oracle's decision per case is the KAT's (keeps exactly the matching subjects)
        kept = sum(1 for s in subjects if want[s] is not None)
        if a[0] == ob.MODIFIED:
            assert ob.count_records(a[1]) == kept, pat
        else:
            assert kept == len(subjects), pat
        n_pat += 1
        n_cases += len(subjects)
        if ent["names"]:
            # capture spans: the parsed record holds the text of every named group
            for skip_empty in (True, False):
                po = ob.Parser(pat, skip_empty=skip_empty)
                pg = g.Parser(pat, skip_empty=skip_empty)
                fpg = g.FilterParser("log", [pg])
                x, y = ob.FilterParser("log", [po]).filter(blob), fpg.filter(blob)
                fpg.close(); pg.close()
                assert x == y, (pat, skip_empty, first_diff(x[1], y[1]))
            n_parser += 1
    assert n_pat > 150 and n_parser > 25 and n_cases > 12000, (n_pat, n_parser, n_cases)


def test_strptime_kat_on_device(g):
    kat = json.load(open(os.path.join(HERE, "golden", "strptime_kat.json")))
    by_fmt = {}
    for c in kat["cases"]:
        by_fmt.setdefault(c["fmt"], []).append(c["text"].encode("latin-1") if isinstance(c["text"], str) else bytes(c["text"]))
    n_fmt = n_cases = refused = 0
    rx = r"\A(?<t>(?m:.*))\z"
    now = 1709164800 + 86399                          # 2024-02-29 23:59:59 UTC: what year-less formats read from the clock
    g.set_time_now(now); ob.set_time_now(now)
    for fmt, texts in by_fmt.items():
        blob = b"".join(_rec({"log": t}, 11, i) for i, t in enumerate(texts))
        for strict in (True, False):
            for keep in (False, True):
                kw = dict(regex=rx, time_fmt=fmt, time_key="t", time_strict=strict, time_keep=keep)
                try:
                    po = ob.Parser(**kw)
                except ValueError:
                    continue
                try:
                    pg = g.Parser(**kw)
                except ValueError:
                    refused += 1                        # (a directive the device does not take: refused at create, never a silent difference)
                    continue
                fpg = g.FilterParser("log", [pg])
                x, y = ob.FilterParser("log", [po]).filter(blob), fpg.filter(blob)
                fpg.close(); pg.close()
                assert x == y, (fmt, strict, keep, first_diff(x[1], y[1]))
        n_fmt += 1
        n_cases += len(texts)
    g.set_time_now(0); ob.set_time_now(0)
    assert n_fmt > 20 and n_cases > 5000, (n_fmt, n_cases, refused)


def test_yearless_time_formats(g):
    """Time_Format without a year (conf/parsers.conf syslog-rfc3164-local, src/flb_parser.c:922-941,1945-2001): the
    current year is read in front of the text, month and day default to today's, texts of 58..63 bytes fail."""
    SYSLOG = r'^\<(?<pri>[0-9]+)\>(?<time>[^ ]* {1,2}[^ ]* [^ ]*) (?<host>[^ ]*) (?<ident>[a-zA-Z0-9_\/\.\-]*)(?:\[(?<pid>[0-9]+)\])?(?:[^\:]*\:)? *(?<message>.*)$'
    lines = [b"<34>Oct 11 22:14:15 mymachine su[123]: 'su root' failed", b"<13>Feb  5 17:32:18 10.0.0.99 app: hello", b"<13>Feb 29 00:00:00 h a: leap",
             b"<13>Feb 30 00:00:00 h a: no such day", b"<1>  Mar 1 01:02:03 h a: leading blanks", b"<1>Dec 31 23:59:60 h a: leap second", b"<1>junk h a: x",
             b"<1>Jan 1 1:2:3 h a: short"]
    blob = b"".join(_rec({"log": l}, 3, i) for i, l in enumerate(lines))
    for now in (1709164800 + 5, 1735689599, 1735689600, 951782400):
        g.set_time_now(now); ob.set_time_now(now)
        for strict in (True, False):
            for fmt in ("%b %d %H:%M:%S", "%b %e %T", "%H:%M:%S", "%d %b", "%b %d %H:%M:%S.%L"):
                kw = dict(regex=SYSLOG, time_fmt=fmt, time_key="time", time_strict=strict)
                po = ob.Parser(**kw); pg = g.Parser(**kw)
                fp = g.FilterParser("log", [pg])
                x, y = ob.FilterParser("log", [po]).filter(blob), fp.filter(blob)
                fp.close(); pg.close()
                assert x == y, (now, strict, fmt, first_diff(x[1], y[1]))
        # the 64-byte buffer rule: text lengths around 57 / 58 / 63 / 64
        rx = r"^(?<time>.*)$"
        texts = [b"Oct 11 22:14:15" + b" " * k for k in (0, 40, 41, 42, 43, 47, 48, 49, 60, 200)]
        tb = b"".join(_rec({"log": t}, 3, i) for i, t in enumerate(texts))
        kw = dict(regex=rx, time_fmt="%b %d %H:%M:%S", time_key="time", time_keep=True)
        po = ob.Parser(**kw); pg = g.Parser(**kw)
        fp = g.FilterParser("log", [pg])
        x, y = ob.FilterParser("log", [po]).filter(tb), fp.filter(tb)
        fp.close(); pg.close()
        assert x == y, (now, first_diff(x[1], y[1]))
    g.set_time_now(0); ob.set_time_now(0)


def _nest(depth, leaf=b"\x01"):
    """`depth` arrays around one leaf: 91 91 ... 01"""
    return b"\x91" * depth + leaf


def test_msgpack_32_open_containers(g):
    """lib/msgpack-c unpack_define.h:25 + unpack_template.h:139-146: an array / map header met while 32
    containers are open fails the object; the decoder loop ends at that record."""
    def event(body_raw):
        return b"\x92\x92\xd7\x00" + struct.pack(">II", 5, 6) + b"\x80" + body_raw
    good = _rec({"log": b"x 500 y", "k": 1})
    recs = []
    for d in (29, 30, 31, 32, 33):
        # value nested d deep inside root array + body map: the innermost header is met with 2 + d - 1 open
        body = b"\x82\xa3log\xa7x 500 y\xa1v" + _nest(d)
        recs.append((d, event(body)))
    for d, r in recs:
        blob = good + r + good
        for rules in ([("regex", "log 500")], [("exclude", "log 404")]):
            a = ob.Grep(rules).filter(blob); f = g.FilterGrep(rules); b = f.filter(blob); f.close()
            assert a == b, (d, rules, a[0], b[0], first_diff(a[1], b[1]))
        po = ob.Parser(r"^(?<a>[^ ]*) (?<code>[^ ]*)", skip_empty=True); pg = g.Parser(r"^(?<a>[^ ]*) (?<code>[^ ]*)", skip_empty=True)
        for reserve in (False, True):
            fp = g.FilterParser("log", [pg], reserve_data=reserve)
            x, y = ob.FilterParser("log", [po], reserve).filter(blob), fp.filter(blob)
            fp.close()
            assert x == y, (d, reserve, x[0], y[0], first_diff(x[1], y[1]))
        pg.close()
        fm = g.FilterLogToMetrics("counter", [("regex", "log 500")]); om = ob.L2M("counter", [("regex", "log 500")])
        assert fm.filter(blob)[0] == om.filter(blob)
        assert [(s["labels"], s["value"]) for s in fm.snapshot()] == [(s["labels"], s["value"]) for s in om.snapshot()[2]], d
        fm.close()
    # deep metadata and an empty container at the limit
    for d in (30, 31, 32):
        meta = b"\x81\xa1m" + _nest(d - 1, b"\x90")          # ends in an EMPTY array: still counted (the test precedes the shortcut)
        r = b"\x92\x92\xd7\x00" + struct.pack(">II", 5, 6) + meta + b"\x81\xa3log\xa7x 500 y"
        blob = good + r + good
        a = ob.Grep([("regex", "log 500")]).filter(blob); f = g.FilterGrep([("regex", "log 500")]); b = f.filter(blob); f.close()
        assert a == b, ("meta", d, a[0], b[0])
        po = ob.Parser(r"^(?<a>[^ ]*)"); pg = g.Parser(r"^(?<a>[^ ]*)")
        fp = g.FilterParser("log", [pg])
        x, y = ob.FilterParser("log", [po]).filter(blob), fp.filter(blob)
        fp.close(); pg.close()
        assert x == y, ("meta", d, x[0], y[0], first_diff(x[1], y[1]))


def test_parsed_key_at_body_index_64_and_up(g):
    """filter_parser.c:311-319 nulls append_arr[i] for ANY index: Reserve_Data On + Preserve_Key Off on wide records."""
    rx = r"^(?<a>[^ ]*) (?<b>.*)$"
    recs = []
    for pos in ([70], [64], [63, 64], [10, 70, 90], [66, 67, 68, 69], [100]):
        kv = []
        for i in range(110):
            if i in pos:
                kv.append((b"log", b"w%d tail %d" % (i, i)))
            else:
                kv.append((b"k%03d" % i, i))
        body = b"\xde" + struct.pack(">H", len(kv)) + b"".join(synth.mp(k) + synth.mp(v) for k, v in kv)
        recs.append(b"\x92\x92\xd7\x00" + struct.pack(">II", 9, 9) + b"\x80" + body)
    # a Key_Name entry the parser refuses stays (no space in the value)
    kv = [(b"k%03d" % i, i) for i in range(80)] + [(b"log", b"nospace"), (b"log", b"ok yes")]
    recs.append(b"\x92\x92\xd7\x00" + struct.pack(">II", 9, 9) + b"\x80\xde" + struct.pack(">H", len(kv)) + b"".join(synth.mp(k) + synth.mp(v) for k, v in kv))
    blob = b"".join(recs)
    for reserve, preserve in ((True, False), (True, True), (False, True), (False, False)):
        po = ob.Parser(rx); pg = g.Parser(rx)
        fp = g.FilterParser("log", [pg], reserve_data=reserve, preserve_key=preserve)
        x, y = ob.FilterParser("log", [po], reserve, preserve).filter(blob), fp.filter(blob)
        fp.close(); pg.close()
        assert x == y, (reserve, preserve, first_diff(x[1], y[1]))


def test_parser_do_whole_value_group_named_k(g):
    """flbgpu_parser_do reports success from the record's parsed flag, not from the shape of the output"""
    p = g.Parser(r"^(?<k>.*)$")
    o = ob.Parser(r"^(?<k>.*)$")
    for s in (b"hello world", b"", b"x" * 40):
        a, b = o.do(s), p.do(s)
        assert a[0] == b[0] and a[1] == b[1], (s, a, b)
    p.close()


def test_illformed_utf8_through_stock_patterns(g):
    """Latin-1 bytes, lone leads, cut sequences inside apache lines: the reference parses them (a byte that
    starts no well-formed sequence is a one-byte character, lib/onigmo/regenc.c:54-67), so must the device."""
    import random
    from test_gpu_parity import APACHE2, APACHE, TF
    rng = random.Random(5)
    data, off, ep = synth.apache_records(3000)
    frag = [b"\xe9", b"\xc3", b"\xa9", b"\xe2\x82", b"\xf0\x9f\x98", b"\xff", b"\xc0\x80", b"\xed\xa0\x80", b"\xc3\xa9", b"\xe2\x82\xac", b"\xf4\x90\x80\x80", b"\x80"]
    recs = []
    for i in range(3000):
        m = bytearray(data[int(off[i]) + 21:int(off[i + 1])])
        for _ in range(rng.randint(1, 3)):
            k = rng.randrange(len(m) + 1)
            m[k:k] = rng.choice(frag)
        if rng.random() < 0.2:
            m = m[:rng.randrange(1, len(m))] + rng.choice(frag[:6])       # cut by the end of the text
        recs.append(_rec({"log": bytes(m)}, rng.randrange(2**31), rng.randrange(10**9)))
    blob = b"".join(recs)
    from test_gpu_parity import both_parser, both_grep
    o, q = both_parser(g, blob, "log", [dict(regex=APACHE2, time_fmt=TF, time_key="time"), dict(regex=APACHE, time_fmt=TF, time_key="time")])
    assert o == q, first_diff(o[1], q[1])
    assert ob.count_records(o[1]) == 3000
    for rules in ([("regex", r"agent [^ ]+ \S+"), ("exclude", "user .")], [("regex", "log \u00e9"), ("regex", 'log "[^"]*\u20ac')]):
        a, b = both_grep(g, blob, rules, "OR" if len(rules) == 2 and rules[0][0] == rules[1][0] else None)
        assert a == b, (rules, first_diff(a[1], b[1]))


def test_wide_reverse_tables_on_device(g, monkeypatch):
    """Stock parsers `envoy` / `ambassador` (conf/parsers.conf:102, conf/parsers_ambassador.conf:4): their UTF-8 capture
    automaton has more than 0x7FF0 states and is walked from 32-bit tables in HBM (rx.hpp `wide`; kdev.inc
    rx_reverse<wide>, rx_resolve_multi).  Lines with ill-formed UTF-8 go that way; byte-identical to the oracle.  Then
    the same walkers under every utf8 set of the apache case (FLBGPU_RX_FORCE_WIDE)."""
    import stock_wide
    from test_gpu_parity import both_parser, both_grep, APACHE2, TF
    for pat, prefix, tkey in ((stock_wide.ENVOY, b"", "start_time"), (stock_wide.AMBASSADOR, b"ACCESS ", None)):
        blob = b"".join(_rec({"log": s}, 100 + i, i) for i, s in enumerate(stock_wide.lines(4000, 23, prefix)))
        pa = dict(regex=pat)
        if tkey:
            pa.update(time_fmt=stock_wide.ENVOY_TIME_FMT, time_key=tkey)
        o, q = both_parser(g, blob, "log", [pa], reserve=True)
        assert o == q, first_diff(o[1], q[1])
        assert ob.count_records(o[1]) == 4000
    monkeypatch.setenv("FLBGPU_RX_FORCE_WIDE", "1")
    import random
    rng = random.Random(9)
    data, off, ep = synth.apache_records(2000)
    recs = []
    for i in range(2000):
        m = bytearray(data[int(off[i]) + 21:int(off[i + 1])])
        for _ in range(rng.randint(1, 3)):
            k = rng.randrange(len(m) + 1)
            m[k:k] = rng.choice(stock_wide.FRAG[:12])
        recs.append(_rec({"log": bytes(m)}, i, i))
    blob = b"".join(recs)
    o, q = both_parser(g, blob, "log", [dict(regex=APACHE2, time_fmt=TF, time_key="time")])
    assert o == q, first_diff(o[1], q[1])
    for rules in ([("regex", "log é"), ("regex", 'log "[^"]*€')], [("exclude", "log [à-ÿ]{1,2} ")]):
        a, b = both_grep(g, blob, rules, "OR" if len(rules) == 2 else None)
        assert a == b, (rules, first_diff(a[1], b[1]))
