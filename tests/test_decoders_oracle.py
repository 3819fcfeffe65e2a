"""Decode_Field / Decode_Field_As (src/flb_parser_decoder.c): the oracle restatement (oracle/oflb.c decoder_do, dec_backend,
unescape_plain, mysql_unquote) against the reference's OWN objects -- flb_unescape.c (oracle/_ref/libunescape_ref.so) for the
two string functions, and flb_parser_decoder.c + the four parser formats inside filter_parser's cb_filter
(oracle/_ref/ref_filters) for the rule engine: same chunks, identical return codes and bytes.
TEST INFRASTRUCTURE for the next step of the path (the device side of the decoders is not built yet: the product refuses
parsers with decoders, plugin/filter_gpu_plugins.c:266,390)."""
import ctypes, os, random, sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_binding as ob
import ref_filters as rf
import synth

REF_UNESC = os.path.join(HERE, "..", "oracle", "_ref", "libunescape_ref.so")

STR_CASES = [b"plain", b"a\\nb\\tc\\\\d\\ae\\bf\\vg\\fh\\ri", b"\\q\\1\\\"x\\'", b"end\\", b"\\", b"\\\\", b"", b"a\\\\\\n", b"\\0\\Z\\%\\_", b"'it\\'s'",
             b"\xc3\xa9\\n\xff", b"tab\\there", b"\\\\\\", b"a\\"]


def _call(fn, s):
    out = ctypes.create_string_buffer(2 * len(s) + 16)
    n = fn(s, len(s), out)
    return out.raw[:n]


@pytest.mark.skipif(not os.path.exists(REF_UNESC), reason="oracle/_ref/libunescape_ref.so not built (needs /root/reference)")
def test_string_backends_match_the_real_flb_unescape():
    ref = ctypes.CDLL(REF_UNESC)
    L = ob.lib()
    for f in (L.oflb_unescape_plain, L.oflb_mysql_unquote):
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    ref.flb_unescape_string.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
    ref.flb_mysql_unquote_string.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]

    def ref_call(fn, s):
        buf = ctypes.create_string_buffer(2 * len(s) + 16)
        p = ctypes.c_char_p(ctypes.addressof(buf))
        n = fn(s, len(s), ctypes.byref(p))            # (s is NUL-terminated like the sds the decoder hands over)
        return buf.raw[:n]
    rng = random.Random(23)
    cases = list(STR_CASES)
    alphabet = [b"\\", b"n", b"t", b"a", b"b", b"v", b"f", b"r", b"0", b"Z", b"'", b'"', b"x", b" ", b"\xc3\xa9", b"q"]
    for _ in range(5000):
        cases.append(b"".join(rng.choice(alphabet) for _ in range(rng.randrange(0, 16))))
    for s in cases:
        assert _call(L.oflb_unescape_plain, s) == ref_call(ref.flb_unescape_string, s), s
        assert _call(L.oflb_mysql_unquote, s) == ref_call(ref.flb_mysql_unquote_string, s), s


def test_string_backends_known_answers():
    # (travels: the GPU box has no /root/reference)
    L = ob.lib()
    for f in (L.oflb_unescape_plain, L.oflb_mysql_unquote):
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    assert _call(L.oflb_unescape_plain, b"a\\nb\\tc\\\\d\\qe") == b"a\nb\tc\\dqe"          # an unknown escape loses its backslash
    assert _call(L.oflb_unescape_plain, b"end\\") == b"end\x00"                               # the trailing backslash copies the sds NUL
    assert _call(L.oflb_mysql_unquote, b"a\\nb\\0c\\Zd\\qe\\") == b"a\nb\x00c\x1ad\\qe\\"


DECODER_SETS = [
    [(True, "escaped", "log")],
    [(True, "escaped_utf8", "log")],
    [(True, "mysql_quoted", "log")],
    [(True, "json", "log")],
    [(False, "json", "log")],
    [(True, "escaped_utf8", "log", "do_next"), (True, "json", "log")],                        # the docker parser of conf/parsers.conf:53-60
    [(True, "json", "log", "try_next"), (True, "escaped", "log")],
    [(False, "json", "log", "do_next"), (True, "escaped", "log"), (True, "mysql_quoted", "other")],
    [(False, "json", "log"), (False, "json", "other")],
    [(True, "escaped", "nokey")],
    [(False, "escaped", "log", "try_next"), (False, "json", "log", "do_next"), (True, "json", "log")],
]
VALUES = [b'{"a": 1, "b": {"c": [1, 2, "x"]}, "s": "t\\u00e9"}', b'  {"k": "v"}', b'[1, 2]', b'{"a": 1} {"b": 2}', b'{"broken": ', b"plain text",
          b'{\\"a\\": \\"b\\\\n\\"}', b"'quoted \\' mysql\\n'", b'"dq \\" \\Z"', b"x", b"", b"tab\\there\\\\", b"trail\\", b'{"time": "2020-01-02T03:04:05.5", "z": 9}',
          b'{"log": "again", "other": "in"}', b"\\u00e9\\uD83D\\uDE00\\x41", b'{"a": 1}trailing', b'{}', b'{"n": null, "t": true, "f": 1.5e3}']
TFMT = "%Y-%m-%dT%H:%M:%S.%L"


def _chunks():
    rng = random.Random(31)
    recs = []
    for i in range(400):
        v, w = rng.choice(VALUES), rng.choice(VALUES)
        kind = i % 4
        if kind == 0:        # for the regex parser: two captured fields
            line = b"<" + v + b"> <" + w + b">"
        elif kind == 1:      # Format json: the record text is a JSON object whose fields hold the values as strings
            import json
            line = json.dumps({"log": v.decode("latin-1"), "n": i, "other": w.decode("latin-1"), "time": "2021-05-06T07:08:09.25"}).encode("latin-1")
        elif kind == 2:      # logfmt
            line = b'log="' + v.replace(b'"', b"'") + b'" other=' + (w.split(b" ")[0] or b"e") + b" n=%d" % i
        else:                # ltsv
            line = b"log:" + v.replace(b"\t", b" ") + b"\tother:" + w.replace(b"\t", b" ") + b"\tn:%d" % i
        recs.append(synth.v2_record(100 + i, i, {"msg": line, "keep": i}))
    return b"".join(recs)


PARSERS = [dict(regex=r"^<(?<log>.*)> <(?<other>.*)>$"), dict(format="json", time_fmt=TFMT, time_key="time"), dict(format="json"),
           dict(format="logfmt"), dict(format="ltsv", types="n:integer")]


@pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built (no reference tree)")
def test_decoders_against_the_real_parser_decoder():
    data = _chunks()
    cases, want = [], []
    for decs in DECODER_SETS:
        for pa in PARSERS:
            for reserve, preserve in ((False, False), (True, True)):
                p = dict(pa, decoders=decs)
                cases.append(rf.parser_case("msg", [p], data, reserve, preserve))
                want.append(ob.FilterParser("msg", [ob.Parser(**p)], reserve, preserve).filter(data))
    got = rf.run(cases)
    changed = 0
    plain = {}
    for (ret, out), (wret, wout), i in zip(got, want, range(len(got))):
        ctx = (DECODER_SETS[i // (len(PARSERS) * 2)], PARSERS[(i // 2) % len(PARSERS)], i % 2)
        assert ret == wret, ctx
        assert out == (wout or b""), ctx
        key = (i // 2) % len(PARSERS), i % 2
        plain.setdefault(key, ob.FilterParser("msg", [ob.Parser(**PARSERS[key[0]])], bool(key[1]), bool(key[1])).filter(data)[1])
        changed += out != plain[key]
    assert changed > len(got) * 0.6              # the decoders did something in most configurations


def test_decoders_golden_digests():
    """the reference's answers, committed (tests/golden/gen_decoders_kat.py): the check that travels"""
    import hashlib, json
    kat = json.load(open(os.path.join(HERE, "golden", "decoders_kat.json")))
    data = _chunks()
    assert hashlib.sha256(data).hexdigest() == kat["chunk_sha256"]
    assert len(kat["cases"]) == len(DECODER_SETS) * len(PARSERS) * 2
    for c in kat["cases"]:
        di, pi, rp = c["case"]
        p = dict(PARSERS[pi], decoders=DECODER_SETS[di])
        ret, out = ob.FilterParser("msg", [ob.Parser(**p)], bool(rp), bool(rp)).filter(data)
        assert ret == c["ret"] and hashlib.sha256(out or b"").hexdigest() == c["sha256"], (DECODER_SETS[di], PARSERS[pi], rp)


def test_device_ready_string_backends_match_the_oracle():
    """csrc/dec.hpp (the decoders' string backends as the kernels will run them: one template for host and device) executed on
    the host through flbgpu_dec_simulate, against the oracle functions the test above pins on the reference's flb_unescape.c;
    the size-only call (what a size pass does) agrees with the writing call"""
    import flbamd_loader
    G = flbamd_loader.load().lib()
    G.flbgpu_dec_simulate.restype = ctypes.c_int64
    G.flbgpu_dec_simulate.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    L = ob.lib()
    for f in (L.oflb_unescape_plain, L.oflb_mysql_unquote):
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]

    def oracle_mysql_quoted(s):                      # decode_mysql_quoted (src/flb_parser_decoder.c:114-147) over the pinned unquote
        if len(s) >= 2 and ((s[:1] == b"'" and s[-1:] == b"'") or (s[:1] == b'"' and s[-1:] == b'"')):
            return _call(L.oflb_mysql_unquote, s[1:-1])
        return s
    rng = random.Random(29)
    cases = list(STR_CASES) + [b"'a\\nb'", b'"q\\"', b"'", b"''", b'"x', b"'\\'"]
    alphabet = [b"\\", b"n", b"t", b"a", b"b", b"v", b"f", b"r", b"0", b"Z", b"'", b'"', b"x", b" ", b"\xc3\xa9", b"q", b"\x00"]
    for _ in range(5000):
        cases.append(b"".join(rng.choice(alphabet) for _ in range(rng.randrange(0, 16))))
    for s in cases:
        for backend, want in ((1, _call(L.oflb_unescape_plain, s)), (3, oracle_mysql_quoted(s))):
            out = ctypes.create_string_buffer(2 * len(s) + 16)
            n = G.flbgpu_dec_simulate(backend, s, len(s), out, len(out))
            assert n == len(want) and out.raw[:n] == want, (backend, s)
            assert G.flbgpu_dec_simulate(backend, s, len(s), None, 0) == n
    assert G.flbgpu_dec_simulate(0, b"{}", 2, None, 0) == -1
    # escaped_utf8 on the 6 k answers of the real flb_unescape.c (tests/golden/unescape_kat.json) and the corner list
    import json
    from test_kv_oracle import ESC_CASES
    L.oflb_unescape_utf8.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    kat = [(bytes.fromhex(c["in"]), bytes.fromhex(c["out"])) for c in json.load(open(os.path.join(HERE, "golden", "unescape_kat.json")))["cases"]]
    kat += [(s, _call(L.oflb_unescape_utf8, s)) for s in ESC_CASES]
    for s, want in kat:
        out = ctypes.create_string_buffer(2 * len(s) + 16)
        n = G.flbgpu_dec_simulate(2, s, len(s), out, len(out))
        assert n == len(want) and out.raw[:n] == want, s
        n2 = G.flbgpu_dec_simulate(102, s, len(s), out, len(out))
        assert out.raw[:n2] == want.split(b"\x00")[0], s


@pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built (no reference tree)")
def test_random_decoder_sets_against_the_real_parser_decoder():
    """random rule lists (1-4 rules over four keys, every backend, both rule types, try_next / do_next / an unknown action word)
    on random parser formats: the rule bookkeeping of flb_parser_decoder_do beyond the hand-written sets (a run of 300
    configurations of the same loop: no difference)"""
    data = _chunks()
    rng = random.Random(77)
    cases, want, meta = [], [], []
    for _ in range(60):
        rules = []
        for _ in range(rng.randint(1, 4)):
            r = [rng.random() < 0.6, rng.choice(["json", "escaped", "escaped_utf8", "mysql_quoted"]), rng.choice(["log", "log", "other", "nokey", "n"])]
            a = rng.choice([None, None, "try_next", "do_next", "bogus"])
            if a:
                r.append(a)
            rules.append(tuple(r))
        pa = rng.choice(PARSERS)
        rp = rng.random() < 0.5
        p = dict(pa, decoders=rules)
        cases.append(rf.parser_case("msg", [p], data, rp, rp))
        want.append(ob.FilterParser("msg", [ob.Parser(**p)], rp, rp).filter(data))
        meta.append((rules, pa, rp))
    for (ret, out), (wret, wout), m in zip(rf.run(cases), want, meta):
        assert ret == wret and out == (wout or b""), m
