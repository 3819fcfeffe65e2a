"""Format logfmt / ltsv parsers: the oracle restatement (oracle/oflb.c kv_walk, kv_unescape_utf8) against
the reference's own test expectations (tests/internal/parser_logfmt.c, parser_ltsv.c) and -- for the
escape decoder of logfmt's quoted values -- against the REAL src/flb_unescape.c compiled from the
reference (oracle/_ref/libunescape_ref.so)."""
import ctypes
import os
import random

import msgpack
import pytest

import oracle_binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
REF_UNESC = os.path.join(HERE, "..", "oracle", "_ref", "libunescape_ref.so")
TFMT = "%Y-%m-%dT%H:%M:%S.%L"


def do(p, s):
    r, out, t = p.do(s)
    return r, (msgpack.unpackb(out, raw=True, strict_map_key=False) if out is not None else None), t


def test_logfmt_reference_vectors():
    # tests/internal/parser_logfmt.c: test_basic / test_time_key / test_time_keep
    s = b'str="text" int=100 double=1.23 bool=true'
    r, m, t = do(ob.Parser(format="logfmt"), s)
    assert r == len(s) and m == {b"str": b"text", b"int": b"100", b"double": b"1.23", b"bool": b"true"} and t == (0, 0)
    s2 = s + b" time=2022-10-31T12:00:01.123"
    r, m, t = do(ob.Parser(format="logfmt", time_fmt=TFMT, time_key="time"), s2)
    assert m == {b"str": b"text", b"int": b"100", b"double": b"1.23", b"bool": b"true"} and t == (1667217601, 123000000)
    r, m, t = do(ob.Parser(format="logfmt", time_fmt=TFMT, time_key="time", time_keep=True), s2)
    assert m[b"time"] == b"2022-10-31T12:00:01.123" and len(m) == 5 and t == (1667217601, 123000000)


def test_logfmt_types_reference_vector():
    # tests/internal/parser_logfmt.c:322-392 test_types: int:hex on "int=100" -> 256, everything else a string
    s = b'str="text" int=100 double=1.23 bool=true'
    r, m, t = do(ob.Parser(format="logfmt", types="int:hex"), s)
    assert r == len(s) and m == {b"str": b"text", b"int": 256, b"double": b"1.23", b"bool": b"true"}
    # with Types every pair goes through flb_parser_typecast on the raw text: no `true` for a bare key, no unescaping
    r, m, t = do(ob.Parser(format="logfmt", types="n:integer f:float b:bool"), b'bare n=12x f=1e2 b=TRUE q="a\\nb" e= b=maybe')
    assert m == {b"bare": b"", b"n": 12, b"f": 100.0, b"b": b"maybe", b"q": b"a\\nb", b"e": b""}
    r, m, t = do(ob.Parser(format="ltsv", types="size:integer ok:bool"), b"size:512\tok:false\thost:h")
    assert m == {b"size": 512, b"ok": False, b"host": b"h"}


def test_ltsv_reference_vectors():
    # tests/internal/parser_ltsv.c: test_basic / test_time_key / test_time_keep / the json_str field
    s = b"str:text\tint:100\tdouble:1.23\tbool:true"
    r, m, t = do(ob.Parser(format="ltsv"), s)
    assert r == len(s) and m == {b"str": b"text", b"int": b"100", b"double": b"1.23", b"bool": b"true"}
    s2 = s + b"\ttime:2022-10-31T12:00:01.123"
    r, m, t = do(ob.Parser(format="ltsv", time_fmt=TFMT, time_key="time"), s2)
    assert m == {b"str": b"text", b"int": b"100", b"double": b"1.23", b"bool": b"true"} and t == (1667217601, 123000000)
    r, m, t = do(ob.Parser(format="ltsv", time_fmt=TFMT, time_key="time", time_keep=True), s2)
    assert m[b"time"] == b"2022-10-31T12:00:01.123" and len(m) == 5 and t == (1667217601, 123000000)
    js = b'json_str:{"str":"text", "int":100, "double":1.23, "bool":true}'
    r, m, t = do(ob.Parser(format="ltsv"), js)
    assert m == {b"json_str": js[9:]}


def test_logfmt_grammar_corners():
    p = ob.Parser(format="logfmt")
    assert do(p, b"")[0] == -1 and do(p, b'  ="= ')[0] == -1                      # nothing to pack
    assert do(p, b"bare")[1] == {b"bare": True}
    assert do(p, b"k= j=")[1] == {b"k": True, b"j": True}
    assert do(p, b'k="" j="x')[1] == {b"k": b"", b"j": b"x"}                        # empty quoted string; unterminated quote
    assert do(p, b'k="a\\"b" z=1')[1] == {b"k": b'a"b', b"z": b"1"}
    assert do(p, b'k="a\\')[1] == {b"k": b"a\\"}                                    # backslash at the very end
    r, m, _ = do(p, b"a=1\nb=2")
    assert (r, m) == (4, {b"a": b"1"})                                              # a newline right behind a pair ends the record
    r, m, _ = do(p, b"a=1 \nb=2")
    assert m == {b"a": b"1", b"b": b"2"}                                            # ... but not behind other garbage
    r, m, _ = do(p, b"a=1\r\nb=2")
    assert (r, m) == (5, {b"a": b"1"})
    assert do(p, b"a=1\rb=2")[0] == 4
    assert do(p, b"k=v=w x")[1] == {b"k": b"v", b"w": True, b"x": True}
    assert do(p, b"k\x80\xff=\x01v")[1] == {b"k\x80\xff": True, b"v": True}
    nb = ob.Parser(format="logfmt", no_bare_keys=True)
    assert do(nb, b"a=1 bare b=2")[0] == -1 and do(nb, b"a=1 b= c=3")[1] == {b"a": b"1", b"b": True, b"c": b"3"}
    # time: every pair with the key goes through the lookup, a strict failure fails the parser, only-time = nothing packed
    pt = ob.Parser(format="logfmt", time_fmt="%Y-%m-%d", time_key="t")
    assert do(pt, b"t=2020-01-02 a=1 t=2021-03-04")[2] == (1614816000, 0)
    assert do(pt, b"t=garbage a=1")[0] == -1 and do(pt, b"t=2020-01-02")[0] == -1
    assert do(pt, b"t= a=1")[1] == {b"t": True, b"a": b"1"}                         # empty value: not a time
    assert do(ob.Parser(format="logfmt", time_fmt="%Y-%m-%d", time_key="t", time_strict=False), b"t=20x a=1")[1] == {b"a": b"1"}


def test_ltsv_grammar_corners():
    p = ob.Parser(format="ltsv")
    assert do(p, b"")[0] == -1 and do(p, b"nolabel")[0] == -1 and do(p, b":v\ta:1")[1] == {b"a": b"1"}
    assert do(p, b"a:\tb:x y:z")[1] == {b"a": b"", b"b": b"x y:z"}
    assert do(p, b"a:1\t\tb:2")[1] == {b"a": b"1"}                                  # empty field between tabs: the walk stops
    assert do(p, b"a:1\nb:2")[0] == 4 and do(p, b"a:1\r\nb:2")[0] == 5
    assert do(p, b"a b:1")[0] == -1 and do(p, b"a:1\tb")[1] == {b"a": b"1"}
    assert do(p, b"a:x\x00y\tb:2")[1] == {b"a": b"x"}


ESC_CASES = [b"plain", b"a\\nb\\tc\\\\d\\\"e\\'f\\/g\\bh\\fi\\rj", b"\\v\\a\\e\\z\\1\\12\\123\\1234\\777\\8", b"\\x41\\x4\\xzz\\x",
             b"\\u0041\\u00e9\\u20ac\\uD83D\\uDE00", b"\\uD83D", b"\\uD83D\\u0041", b"\\uDE00x", b"\\u12", b"\\uZZ", b"\\u", b"\\uD83D\\u12",
             b"\\uD83D\\uZ", b"\\U0001F600\\U41\\U\\UFFFFFFFF\\U00110000", b"\\0tail", b"a\\u0000tail", b"\x80\xff\\\xe9", b"end\\", b"\\",
             b"a\x00b", b"\\uD83D\\", b"\\uD83D\\u", b"\\uD83D\\uDE0"]


def _oracle_unescape(s):
    out = ctypes.create_string_buffer(len(s) + 8)
    n = ob.lib().oflb_unescape_utf8(s, len(s), out)
    return out.raw[:n]


@pytest.mark.skipif(not os.path.exists(REF_UNESC), reason="oracle/_ref/libunescape_ref.so not built (needs /root/reference)")
def test_unescape_matches_the_real_reference():
    ref = ctypes.CDLL(REF_UNESC)
    ref.flb_unescape_string_utf8.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    rng = random.Random(17)
    cases = list(ESC_CASES)
    alphabet = [b"\\", b"u", b"U", b"x", b"D", b"8", b"3", b"d", b"c", b"0", b"1", b"7", b"9", b"f", b"n", b'"', b"a", b"z", b"\xc3", b"\xa9", b"\x00", b" "]
    for _ in range(4000):
        cases.append(b"".join(rng.choice(alphabet) for _ in range(rng.randrange(0, 24))))
    for s in cases:
        want = ctypes.create_string_buffer(len(s) + 8)
        n = ref.flb_unescape_string_utf8(s, len(s), want)
        assert _oracle_unescape(s) == want.raw[:n], s


def test_unescape_golden_vectors():
    """answers of the real flb_unescape.c, committed (tests/golden/gen_unescape_kat.py): the check that travels"""
    import json
    kat = json.load(open(os.path.join(HERE, "golden", "unescape_kat.json")))
    assert len(kat["cases"]) > 6000
    for c in kat["cases"]:
        s = bytes.fromhex(c["in"])
        assert _oracle_unescape(s) == bytes.fromhex(c["out"]), s


def test_unescape_known_answers():
    # kept as goldens so that the GPU box (no /root/reference) still checks the restatement
    assert _oracle_unescape(b"a\\nb\\u00e9\\uD83D\\uDE00\\x41\\101") == "a\nb\u00e9\U0001F600AA".encode()
    assert _oracle_unescape(b"\\uDE00|\\u12|\\uD83Dx") == "\ufffd|\ufffd|\ufffdx".encode()
    assert _oracle_unescape(b"\x80\\q\\") == b"\x80q\\"
    # the packed value is cut at the first NUL (strlen of the decoded copy)
    assert do(ob.Parser(format="logfmt"), b'k="ab\\0cd" z="\\u0000"')[1] == {b"k": b"ab", b"z": b""}
