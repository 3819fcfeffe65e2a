"""the product's SQL front end (csrc/sp.cpp, no device needed) against the oracle's parse of the same query, which is pinned on
the reference's grammar through ref_sp (test_sp_oracle.py)"""
import ctypes
import os
import struct
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import flbamd_loader
import osp
import sp_synth


@pytest.fixture(scope="module")
def L():
    lib = flbamd_loader.load().lib()
    lib.flbgpu_sp_parse_check.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    return lib


def describe(L, sql):
    buf = ctypes.create_string_buffer(4096)
    if L.flbgpu_sp_parse_check(sql.encode(), buf, 4096) != 0:
        return None
    return buf.value.decode()


def _keyname(name, sub):
    return name + "".join("['%s']" % s for s in (sub or []))


def _leaf(c):
    k = c.a[0]
    if k == "key":
        return "K:" + _keyname(c.a[1], c.a[2])
    if k == "func":
        return "T" if c.a[1] == "time" else "C:" + _keyname(c.a[2].a[1], c.a[2].a[2])
    t, v = c.a[1], c.a[2]
    if t == "int":
        return "I:%d" % v
    if t == "float":
        return "F:%016x" % struct.unpack("<Q", struct.pack("<d", v))[0]
    if t == "str":
        return "S:" + v.encode().hex()
    if t == "bool":
        return "B:%d" % int(v)
    return "N"


def _postfix(c, out):
    op, l, r = c.a[1], c.a[2], c.a[3]
    if op == "PAR":
        return _postfix(l, out)
    if op in ("EQ", "LT", "LTE", "GT", "GTE"):
        out.append("%s(%s,%s)" % (op, _leaf(l), _leaf(r)))
    elif op == "NOT":
        _postfix(l, out)
        out.append("NOT")
    elif op == "OR" and (l is None or r is None):            # "condition: key" / "condition: value"
        out.append("TRUTH(%s)" % _leaf(l if l is not None else r))
    else:
        _postfix(l, out)
        _postfix(r, out)
        out.append(op)


def oracle_describe(sql):
    q = osp.parse(sql)
    keys = "|".join("%s:%d:%d" % (k.out_name, k.func, -1 if k.gb is None else k.gb) for k in q.keys)
    gb = "|".join(_keyname(n, s) for n, s in q.gb_keys)
    ops = []
    if q.cond is not None:
        _postfix(q.cond, ops)
    win = "2:%d:%d" % (q.window_size, q.advance_by) if q.window == "hopping" else "%d:%d" % (1 if q.window == "tumbling" else 0, q.window_size)
    return "keys=%s;gb=%s;window=%s;source=%d:%s;stream=%s;where=%s" % (
        keys, gb, win, 1 if q.source_type == "tag" else 0, q.source, q.stream_name or "", " ".join(ops))


QUERIES = sp_synth.QUERIES + sp_synth.HOPPING_QUERIES + sp_synth.SELECT_QUERIES + [
    "select a.b['x']['y'] as k, count(*) , Avg(v) from tag:'t.*' window tumbling (2 minute) where not a = 1 and b <> 'it''s' or c group by a.b['x']['y'];",
    "SELECT COUNT(*) FROM STREAM:s WHERE (a > 1.5 OR NOT (b IS NOT NULL)) AND 'x' OR true AND -3 AND @record.time() >= 10;",
    "SELECT SUM(x['a']), MIN(x['a']), MAX(y) AS top FROM STREAM:s WHERE @record.contains(x['a']) != false GROUP BY z;".replace("GROUP BY z", ""),
    "SELECT k, COUNT(k) FROM STREAM:s WINDOW TUMBLING (1 HOUR) WHERE k = 0.1 GROUP BY k;",
]


def test_front_end_matches_the_oracle(L):
    for q in QUERIES:
        got = describe(L, q)
        assert got is not None, q
        assert got == oracle_describe(q), q


def test_front_end_refuses_what_the_reference_refuses(L):
    bad = ["SELECT id, MIN(id) FROM STREAM:FLB;", "SELECT * FROM STREAM:FLB WHERE;", "SELECT COUNT(*) FROM STREAM:FLB GROUP BY;",
           "SELECT COUNT(*) FROM STREAM:FLB", "SELECT COUNT() FROM STREAM:FLB;", "SELECT COUNT(*) FROM STREAM:s WHERE time > 3;",
           "SELECT COUNT(*) FROM STREAM:s WHERE a = 99999999999;", "SELECT COUNT(*) FROM s;", "SELECT COUNT(*) FROM STREAM:s WHERE a == 1;",
           # outside this path (refused loudly rather than answered differently)
           "SELECT a, * FROM STREAM:s;", "SELECT *, * FROM STREAM:s;", "SELECT NOW(), * FROM STREAM:s;", "SELECT NOW FROM STREAM:s;", "SELECT COUNT(*) FROM STREAM:s WINDOW HOPPING (5 SECOND, ADVANCE BY 5 SECOND);", "SELECT COUNT(*) FROM STREAM:s WINDOW HOPPING (5 SECOND);",
           "SELECT TIMESERIES_FORECAST(a, 10) FROM STREAM:s;", "SELECT NOW(), a, COUNT(*) FROM STREAM:s;",
           "CREATE SNAPSHOT s AS SELECT * FROM STREAM:x LIMIT 5;"]
    for q in bad:
        assert describe(L, q) is None, q
        with pytest.raises((osp.ParseError, osp.Unsupported)):
            osp.parse(q)
