"""osp.py -- TEST INFRASTRUCTURE: a CPU restatement of the reference's stream processor for aggregate queries
(SELECT ... COUNT/SUM/AVG/MIN/MAX ... FROM ... [WINDOW TUMBLING (n SECOND)] [WHERE ...] [GROUP BY ...];) and for SELECTs
without aggregation functions (flb_sp.c:1607-1850 sp_process_data -> Task._do_select: keys, aliases, `*`).

Follows, record by record and in arrival order (so also the order-dependent parts: float sums, first-seen group order,
the I64 -> F64 switch of a sum at the first non-zero float):
  parser/sql.l + sql.y (tokens, grammar, bison's shift preference for AND / OR / NOT)      -> parse()
  parser/flb_sp_parser.c:120-290 flb_sp_key_create (aliases "AVG(k)", "k['a']['b']")        -> Key.out_name
  flb_sp.c:201-262 sp_cmd_aggregated_keys (select key <-> GROUP BY key mapping)             -> Query.__init__
  flb_sp.c:264-357 string_to_number, :361-401 object_to_number                             -> string_to_number / object_to_number
  flb_sp_key.c:54-231 flb_sp_key_to_value / subkey_to_value                                 -> key_to_value
  flb_sp.c:783-1009 numerical_comp / value_to_bool / logical_operation, :1011-1159 reduce_expression -> Cond.eval
  flb_sp.c:1280-1429 sp_process_aggregate_data, :1435-1601 sp_process_data_aggr             -> Task.do
  flb_sp_aggregate_func.c:50-197 add / calc of SUM AVG COUNT MIN MAX                         -> Task._add / Task._package
  flb_sp.c:1161-1278 package_results (msgpack_pack_float: AVG and float sums leave as *float32*)
  flb_sp_window.c:26-50 flb_sp_window_prune (DEFAULT / TUMBLING)
Pinned on the reference itself: tests/test_sp_oracle.py runs the same queries over the same chunks through
oracle/_ref/ref_sp (the reference's own sources compiled in place) and wants identical bytes.

Not restated (Unsupported is raised, the product refuses the same queries): TIMESERIES_FORECAST, snapshots; a GROUP BY column whose values mix number / string classes in one window (the reference's rb-tree
comparator is not an order there: flb_sp_groupby.c:77 "Sides have different types -> -1", and it rewrites nodes in place :37-44).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file."""
import decimal
import math
import re
import struct

import msgpack

FLB_SP_AVG, FLB_SP_SUM, FLB_SP_COUNT, FLB_SP_MIN, FLB_SP_MAX = 1, 2, 3, 4, 5
FUNC_NAMES = {1: "AVG", 2: "SUM", 3: "COUNT", 4: "MIN", 5: "MAX"}
KEYWORDS = {"CREATE", "FLUSH", "STREAM", "SNAPSHOT", "WITH", "SELECT", "AS", "FROM", "WHERE", "AND", "OR", "NOT", "WINDOW", "LIMIT", "IS",
            "NULL", "SUM", "AVG", "COUNT", "MIN", "MAX", "TIMESERIES_FORECAST", "CONTAINS", "TIME", "TUMBLING", "HOPPING", "HOUR", "MINUTE",
            "SECOND", "NOW", "UNIX_TIMESTAMP", "RECORD_TAG", "RECORD_TIME"}


class Unsupported(Exception):
    pass


class ParseError(Exception):
    pass


# ------------------------------------------------------------------------------------------ sql.l
_TOK = re.compile(r"""[ \t\n]+
 |(?P<kw2>GROUP\ BY|ADVANCE\ BY|STREAM:|TAG:|@RECORD)
 |(?P<ident>[_A-Za-z][A-Za-z0-9_.]*)
 |(?P<float>(?:-?[1-9][0-9]*|0)\.[0-9]+)
 |(?P<int>-?[1-9][0-9]*|0)
 |(?P<str>'(?:[^']|'')*')
 |(?P<op>!=|<>|<=|>=|<|>)
 |(?P<ch>[*,=()\[\].;])
 |(?P<bad>.)""", re.X | re.I | re.S)


def _f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def tokenize(sql):
    out = []
    for m in _TOK.finditer(sql):
        if m.lastgroup is None:
            continue
        k, v = m.lastgroup, m.group(m.lastgroup)
        if k == "kw2":
            out.append(("kw", v.upper()))
        elif k == "ident":
            u = v.upper()
            if u in KEYWORDS:
                out.append(("kw", u))
            elif u in ("TRUE", "FALSE"):
                out.append(("bool", u == "TRUE"))
            else:
                out.append(("ident", v))
        elif k == "float":
            out.append(("float", _f32(float(v))))          # yylval->fval is a C float
        elif k == "int":
            i = int(v)
            if not -2 ** 31 <= i < 2 ** 31:
                raise Unsupported("integer literal outside int (atoi)")
            out.append(("int", i))
        elif k == "str":
            out.append(("str", v[1:-1].replace("''", "'")))
        elif k == "op":
            out.append(("op", "!=" if v == "<>" else v))
        elif k == "ch":
            out.append(("ch", v))
        else:
            raise ParseError("bad input character %r" % v)
    out.append(("eof", None))
    return out


TFUNC_NAMES = {"NOW": "NOW()", "UNIX_TIMESTAMP": "UNIX_TIMESTAMP()", "RECORD_TAG": "RECORD_TAG()", "RECORD_TIME": "RECORD_TIME()"}


class Key:
    def __init__(self, func, name, subkeys, alias, tfunc=None):
        self.func, self.name, self.subkeys = func, name, subkeys       # func 0 = plain key; name None = '*'
        self.alias = alias
        self.tfunc = tfunc                                             # NOW / UNIX_TIMESTAMP / RECORD_TAG / RECORD_TIME
        self.gb = None
        if tfunc is not None:
            self.out_name = alias if alias is not None else TFUNC_NAMES[tfunc]
        elif alias is not None:
            self.out_name = alias
        elif subkeys:
            base = name + "".join("['%s']" % s for s in subkeys)
            self.out_name = "%s(%s)" % (FUNC_NAMES[func], base) if func else base
        elif func:
            self.out_name = "%s(%s)" % (FUNC_NAMES[func], name if name is not None else "*")
        else:
            self.out_name = name if name is not None else "*"


class Cond:
    """expression node: ('key', name, subkeys) ('val', type, v) ('func', 'contains'|'time', param) ('op', OP, left, right)"""

    def __init__(self, *a):
        self.a = a


class Query:
    def __init__(self):
        self.keys, self.gb_keys, self.cond = [], [], None
        self.window, self.window_size, self.advance_by = "default", 0, 0
        self.source_type, self.source, self.stream_name, self.props, self.limit = None, None, None, [], 0
        self.select_only = False

    def finish(self):
        aggr = not_aggr = 0
        for k in self.keys:
            if k.tfunc is not None:
                continue                                                # neither kind (flb_sp.c:243-245)
            if k.func:
                aggr += 1
                continue
            mapped = False
            for i, (gname, gsub) in enumerate(self.gb_keys):
                if k.name is None:
                    break
                # flb_sds_cmp(key->name, gb_key->name, len(gb_key->name)): equal lengths, equal bytes
                if k.name == gname and (k.subkeys or []) == (gsub or []):
                    k.gb = i
                    mapped = True
                    break
            if not mapped:
                not_aggr += 1
        if aggr == 0:
            # sp_cmd_aggregated_keys returns 0: flb_sp_task_create (flb_sp.c:491-508) leaves aggregate_keys off, the task's window type
            # stays DEFAULT and flb_sp_do takes the sp_process_data branch; GROUP BY is never looked at
            self.select_only = True
            self.window, self.gb_keys = "default", []
            for k in self.keys:
                k.gb = None
            return
        if not_aggr > 0:
            raise ParseError("aggregated query cannot include the aggregated keys")


class _P:
    def __init__(self, sql):
        self.t = tokenize(sql)
        self.i = 0

    def cur(self):
        return self.t[self.i]

    def adv(self):
        self.i += 1

    def is_(self, kind, v=None):
        k, x = self.t[self.i]
        return k == kind and (v is None or x == v)

    def eat(self, kind, v=None):
        if self.is_(kind, v):
            x = self.t[self.i][1]
            self.i += 1
            return x if x is not None else True
        return None

    def need(self, kind, v=None):
        x = self.eat(kind, v)
        if x is None:
            raise ParseError("expected %s %s at token %d" % (kind, v or "", self.i))
        return x

    def subkeys(self):
        out = []
        while self.eat("ch", "["):
            out.append(self.need("str"))
            self.need("ch", "]")
        return out or None

    def alias(self):
        if self.eat("kw", "AS"):
            return self.need("ident")
        return None

    def record_key(self, q):
        if self.eat("ch", "*"):
            if q.keys:
                raise ParseError("wildcard after keys")                 # flb_sp_key_create, flb_sp_parser.c:172-184
            q.keys.append(Key(0, None, None, None))
            return
        if self.is_("ident"):
            name = self.need("ident")
            sub = self.subkeys()
            q.keys.append(Key(0, name, sub, self.alias()))
            return
        if self.is_("kw"):
            f = self.cur()[1]
            code = {"AVG": 1, "SUM": 2, "COUNT": 3, "MIN": 4, "MAX": 5}.get(f)
            if code is None:
                if f in TFUNC_NAMES:
                    # sql.y: time_record_func '(' ')' key_alias
                    self.adv()
                    self.need("ch", "(")
                    self.need("ch", ")")
                    q.keys.append(Key(0, None, None, self.alias(), tfunc=f))
                    return
                if f == "TIMESERIES_FORECAST":
                    raise Unsupported(f)
                raise ParseError("unexpected " + f)
            self.adv()
            self.need("ch", "(")
            if code == FLB_SP_COUNT and self.eat("ch", "*"):
                self.need("ch", ")")
                q.keys.append(Key(code, None, None, self.alias()))
                return
            name = self.need("ident")
            sub = self.subkeys()
            self.need("ch", ")")
            q.keys.append(Key(code, name, sub, self.alias()))
            return
        raise ParseError("bad select key")

    def key(self):
        name = self.need("ident")
        return Cond("key", name, self.subkeys())

    def is_value(self):
        return self.cur()[0] in ("int", "float", "str", "bool")

    def value(self):
        k, v = self.cur()
        if k not in ("int", "float", "str", "bool"):
            raise ParseError("value expected")
        self.adv()
        return Cond("val", k, v)

    def primary(self):
        if self.eat("ch", "("):
            e = self.condition()
            self.need("ch", ")")
            return Cond("op", "PAR", e, None)
        if self.is_value():
            return Cond("op", "OR", None, self.value())
        plain = False
        if self.eat("kw", "@RECORD"):
            self.need("ch", ".")
            if self.eat("kw", "CONTAINS"):
                self.need("ch", "(")
                k = self.key()
                self.need("ch", ")")
                left = Cond("func", "contains", k)
            elif self.eat("kw", "TIME"):
                self.need("ch", "(")
                self.need("ch", ")")
                left = Cond("func", "time", None)
            else:
                raise ParseError("record function")
        else:
            left = self.key()
            plain = True
        if plain and self.eat("kw", "IS"):
            neg = self.eat("kw", "NOT")
            self.need("kw", "NULL")
            c = Cond("op", "EQ", left, Cond("val", "null", None))
            return Cond("op", "NOT", c, None) if neg else c
        op = None
        if self.is_("ch", "="):
            op = "EQ"
        elif self.is_("op"):
            op = {"!=": "NEQ", "<": "LT", "<=": "LTE", ">": "GT", ">=": "GTE"}[self.cur()[1]]
        if op is None:
            if plain:
                return Cond("op", "OR", left, None)
            return Cond("op", "EQ", left, Cond("val", "bool", True))
        self.adv()
        v = self.value()
        if op == "NEQ":
            return Cond("op", "NOT", Cond("op", "EQ", left, v), None)
        return Cond("op", op, left, v)

    def condition(self):
        if self.eat("kw", "NOT"):
            return Cond("op", "NOT", self.condition(), None)
        left = self.primary()
        if self.is_("kw", "AND") or self.is_("kw", "OR"):
            op = self.cur()[1]
            self.adv()
            return Cond("op", op, left, self.condition())
        return left

    def time_unit(self):
        for w, m in (("SECOND", 1), ("MINUTE", 60), ("HOUR", 3600)):
            if self.eat("kw", w):
                return m
        raise ParseError("time unit")

    def select(self, q):
        self.need("kw", "SELECT")
        self.record_key(q)
        while self.eat("ch", ","):
            self.record_key(q)
        self.need("kw", "FROM")
        if self.eat("kw", "STREAM:"):
            q.source_type, q.source = "stream", self.need("ident")
        elif self.eat("kw", "TAG:"):
            q.source_type, q.source = "tag", self.need("str")
        else:
            raise ParseError("source")
        if self.eat("kw", "WINDOW"):
            if self.eat("kw", "TUMBLING"):
                self.need("ch", "(")
                n = self.need("int")
                q.window, q.window_size = "tumbling", n * self.time_unit()
                self.need("ch", ")")
            elif self.eat("kw", "HOPPING"):
                # sql.y:275-278: HOPPING '(' INTEGER time ',' ADVANCE_BY INTEGER time ')'; flb_sp_cmd_window's -1 for
                # advance_by >= size is not looked at by the grammar action (the product refuses those queries)
                self.need("ch", "(")
                n = self.need("int")
                size = n * self.time_unit()
                self.need("ch", ",")
                self.need("kw", "ADVANCE BY")
                a = self.need("int")
                q.window, q.window_size, q.advance_by = "hopping", size, a * self.time_unit()
                self.need("ch", ")")
                if q.advance_by >= q.window_size:
                    raise Unsupported("HOPPING window that advances by its size or more")
            else:
                raise ParseError("window")
        if self.eat("kw", "WHERE"):
            q.cond = self.condition()
        if self.eat("kw", "GROUP BY"):
            while True:
                name = self.need("ident")
                q.gb_keys.append((name, self.subkeys()))
                if not self.eat("ch", ","):
                    break
        if self.eat("kw", "LIMIT"):
            q.limit = self.need("int")
        self.need("ch", ";")


def parse(sql):
    p = _P(sql)
    q = Query()
    if p.eat("kw", "CREATE"):
        if not p.eat("kw", "STREAM"):
            raise Unsupported("snapshots")
        q.stream_name = p.need("ident")
        if p.eat("kw", "WITH"):
            p.need("ch", "(")
            while True:
                k = p.need("ident")
                p.need("ch", "=")
                q.props.append((k, p.need("str")))
                if not p.eat("ch", ","):
                    break
            p.need("ch", ")")
        p.need("kw", "AS")
    p.select(q)
    if not p.is_("eof"):
        raise ParseError("trailing input")
    q.finish()
    return q


# ------------------------------------------------------------------------------------------ values
class Map:
    """a msgpack map with its entries in wire order (duplicate keys stay)"""

    def __init__(self, pairs):
        self.pairs = pairs


def _b(s):
    return s.encode("utf-8", "surrogateescape") if isinstance(s, str) else s


def decode_chunk(buf):
    u = msgpack.Unpacker(raw=False, unicode_errors="surrogateescape", object_pairs_hook=Map, strict_map_key=False)
    u.feed(bytes(buf))
    out = []
    for rec in u:
        head, body = rec[0], rec[1]
        ts = head[0] if isinstance(head, list) else head
        out.append((ts, body))
    return out


def time_to_double(ts):
    if isinstance(ts, msgpack.ExtType):
        sec, nsec = struct.unpack(">II", ts.data[:8])
        return float(sec) + float(nsec) / 1000000000.0
    if isinstance(ts, float):
        return ts
    return float(ts)


_DEC = re.compile(rb"[ \t\n\v\f\r]*([+-]?)(?:(0[xX](?:[0-9a-fA-F]+\.?[0-9a-fA-F]*|\.[0-9a-fA-F]+)(?:[pP][+-]?[0-9]+)?)"
                  rb"|((?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?)|([iI][nN][fF](?:[iI][nN][iI][tT][yY])?)|([nN][aA][nN]))")
_LDBL_MAX = decimal.Decimal("1.18973149535723176502e4932")
_LDBL_MIN = decimal.Decimal("3.36210314311209350626e-4932")
_INT = re.compile(rb"[ \t\n\v\f\r]*([+-]?[0-9]+)")


def string_to_number(s):
    """flb_sp.c:264-357: ('i', int) / ('f', float) / None.  One '.' -> strtold (the value then narrows to double), none -> strtoll"""
    cut = s.find(b"\0")
    dots = s.count(b".")
    if cut >= 0:
        s = s[:cut]
    if dots > 1:
        return None
    if dots == 1:
        m = _DEC.match(s)
        if not m:
            return None
        sign = -1.0 if m.group(1) == b"-" else 1.0
        if m.group(2):
            h = m.group(2).decode()
            if "p" not in h.lower():
                h += "p0"
            try:
                d = float.fromhex(h)
            except OverflowError:
                return None
        elif m.group(3):
            # strtold: ERANGE only outside the x87 long double range; inside it a value binary64 cannot hold narrows to inf / 0
            dec = decimal.Decimal(m.group(3).decode())
            if dec != 0 and not (_LDBL_MIN <= dec <= _LDBL_MAX):
                return None
            d = float(m.group(3))
        elif m.group(4):
            d = math.inf
        else:
            d = math.nan
        return ("f", sign * d)
    m = _INT.match(s)
    if not m:
        return None
    i = int(m.group(1))
    if not -2 ** 63 <= i < 2 ** 63:
        return None                             # ERANGE
    return ("i", i)


def _i64(v):
    return v - 2 ** 64 if v >= 2 ** 63 else v


def object_to_number(o, conv):
    if isinstance(o, bool):
        return None
    if isinstance(o, int):
        return ("i", _i64(o))
    if isinstance(o, float):
        return ("f", o)
    if isinstance(o, str) and conv:
        b = _b(o)
        if len(b) > 19:
            return None
        return string_to_number(b)
    return None


class NoValue:
    pass


NOVALUE = NoValue()


def _sp_value(o):
    """flb_sp_key.c:54-101: the object when the stream processor has a value class for it, NOVALUE otherwise"""
    if isinstance(o, (bool, int, float, str, Map)) or o is None:
        return o
    return NOVALUE


def key_to_value(name, m, subkeys):
    nb = name.encode()
    for k, v in m.pairs:
        # (flb_sp_key_to_value compares key.via.str without looking at key.type -- :189 -- and a bin key shares that layout: the
        # first entry called `name` wins, string or binary; the sub-key levels do check the type, :104)
        if not isinstance(k, (str, bytes)) or _b(k) != nb:
            continue
        if isinstance(v, Map) and subkeys is not None:
            cur, matched, found = v, 0, False
            for want in subkeys:
                if not isinstance(cur, Map):
                    break
                found = False
                for k2, v2 in cur.pairs:
                    if isinstance(k2, str) and _b(k2) == want.encode():
                        found, cur = True, v2
                        matched += 1
                        break
                if matched == len(subkeys):
                    break
            if not found or (matched > 0 and matched != len(subkeys)):
                return NOVALUE
            return _sp_value(cur)
        return _sp_value(v)
    return NOVALUE


# ------------------------------------------------------------------------------------------ raw objects (sp_process_data re-packs them)
def _raw_tok(b, o):
    """one msgpack head at b[o] -> (kind, value, next): scalars carry their value and `next` is their end; str / bin / ext carry
    (body offset, length[, type]) and `next` is their end; array / map carry the count and `next` is the first child"""
    c = b[o]
    if c <= 0x7f:
        return "uint", c, o + 1
    if c >= 0xe0:
        return "int", c - 256, o + 1
    if 0x80 <= c <= 0x8f:
        return "map", c & 15, o + 1
    if 0x90 <= c <= 0x9f:
        return "array", c & 15, o + 1
    if 0xa0 <= c <= 0xbf:
        n = c & 31
        if o + 1 + n > len(b):
            raise IndexError
        return "str", (o + 1, n), o + 1 + n
    if c == 0xc0:
        return "nil", None, o + 1
    if c == 0xc1:
        raise ValueError("0xc1")
    if c in (0xc2, 0xc3):
        return "bool", c == 0xc3, o + 1
    def be(n, at=o + 1):
        if at + n > len(b):
            raise IndexError
        return int.from_bytes(b[at:at + n], "big")
    if c in (0xc4, 0xc5, 0xc6, 0xd9, 0xda, 0xdb):
        w = {0xc4: 1, 0xc5: 2, 0xc6: 4, 0xd9: 1, 0xda: 2, 0xdb: 4}[c]
        n = be(w)
        if o + 1 + w + n > len(b):
            raise IndexError
        return ("bin" if c <= 0xc6 else "str"), (o + 1 + w, n), o + 1 + w + n
    if c in (0xc7, 0xc8, 0xc9):
        w = {0xc7: 1, 0xc8: 2, 0xc9: 4}[c]
        n = be(w)
        t = be(1, o + 1 + w)
        if o + 2 + w + n > len(b):
            raise IndexError
        return "ext", (o + 2 + w, n, t - 256 if t > 127 else t), o + 2 + w + n
    if c == 0xca:
        be(4)
        return "f32", struct.unpack(">f", b[o + 1:o + 5])[0], o + 5
    if c == 0xcb:
        be(8)
        return "f64", struct.unpack(">d", b[o + 1:o + 9])[0], o + 9
    if 0xcc <= c <= 0xcf:
        w = 1 << (c - 0xcc)
        return "uint", be(w), o + 1 + w
    if 0xd0 <= c <= 0xd3:
        w = 1 << (c - 0xd0)
        v = be(w)
        if v >= 1 << (8 * w - 1):
            v -= 1 << (8 * w)
        return ("uint" if v >= 0 else "int"), v, o + 1 + w
    if 0xd4 <= c <= 0xd8:
        n = 1 << (c - 0xd4)
        t = be(1)
        if o + 2 + n > len(b):
            raise IndexError
        return "ext", (o + 2, n, t - 256 if t > 127 else t), o + 2 + n
    w = 2 if c in (0xdc, 0xde) else 4
    return ("array" if c <= 0xdd else "map"), be(w), o + 1 + w


def _raw_skip(b, o):
    kind, v, nxt = _raw_tok(b, o)
    if kind in ("array", "map"):
        for _ in range(v * (2 if kind == "map" else 1)):
            nxt = _raw_skip(b, nxt)
    return nxt


def _raw_repack(b, o):
    """msgpack_pack_object (lib/msgpack-c/src/objectc.c:39-126) of the object at b[o] -> (bytes, end): every head in its shortest
    form, non-negative integers in the unsigned family, float32 stays float32"""
    kind, v, nxt = _raw_tok(b, o)
    if kind in ("uint", "int"):
        return _pack_int(v), nxt
    if kind == "str":
        return _pack_str(b[v[0]:v[0] + v[1]]), nxt
    if kind == "bin":
        n = v[1]
        h = b"\xc4" + bytes([n]) if n < 256 else b"\xc5" + struct.pack(">H", n) if n < 65536 else b"\xc6" + struct.pack(">I", n)
        return h + b[v[0]:v[0] + n], nxt
    if kind == "ext":
        n, t = v[1], v[2] & 0xFF
        if n in (1, 2, 4, 8, 16):
            h = bytes([0xd4 + (1, 2, 4, 8, 16).index(n), t])
        elif n < 256:
            h = b"\xc7" + bytes([n, t])
        elif n < 65536:
            h = b"\xc8" + struct.pack(">H", n) + bytes([t])
        else:
            h = b"\xc9" + struct.pack(">I", n) + bytes([t])
        return h + b[v[0]:v[0] + n], nxt
    if kind in ("array", "map"):
        fix, w16, w32 = (0x90, 0xdc, 0xdd) if kind == "array" else (0x80, 0xde, 0xdf)
        out = bytearray(bytes([fix | v]) if v < 16 else bytes([w16]) + struct.pack(">H", v) if v < 65536 else bytes([w32]) + struct.pack(">I", v))
        for _ in range(v * (2 if kind == "map" else 1)):
            piece, nxt = _raw_repack(b, nxt)
            out += piece
        return bytes(out), nxt
    return b[o:nxt], nxt                                               # nil, bool, f32, f64: as they came


def _raw_key_to_value(rec, pairs, name, subkeys):
    """flb_sp_key_to_value (flb_sp_key.c:54-231) on the raw record: the offset of the value object of the FIRST entry called
    `name` (sub-keys walked through nested maps, all of them or nothing), None where the reference answers NULL -- no such
    key, or a value the stream processor has no class for (array, bin, ext)"""
    for ks, vs, _ve in pairs:
        kk, kv, _n = _raw_tok(rec, ks)
        if kk not in ("str", "bin") or rec[kv[0]:kv[0] + kv[1]] != name:      # (no type check up here: flb_sp_key.c:189)
            continue
        cur = vs
        kind, cnt, nxt = _raw_tok(rec, cur)
        if kind == "map" and subkeys is not None:
            matched, found = 0, False
            for want in subkeys:
                kind, cnt, nxt = _raw_tok(rec, cur)
                if kind != "map":
                    break
                found = False
                for _ in range(cnt):
                    k2, v2, _e = _raw_tok(rec, nxt)
                    val = _raw_skip(rec, nxt)
                    if k2 == "str" and rec[v2[0]:v2[0] + v2[1]] == want.encode():
                        found, cur = True, val
                        matched += 1
                        break
                    nxt = _raw_skip(rec, val)
                if matched == len(subkeys):
                    break
            if not found or (matched > 0 and matched != len(subkeys)):
                return None
        kind = _raw_tok(rec, cur)[0]
        return None if kind in ("array", "bin", "ext") else cur
    return None


# ------------------------------------------------------------------------------------------ WHERE
def _strncmp(a, b, n):
    a, b = a[:n], b[:n]
    for i in range(n):
        ca = a[i] if i < len(a) else 0
        cb = b[i] if i < len(b) else 0
        if ca != cb:
            return ca - cb
        if ca == 0:
            return 0
    return 0


def _typed(o):
    """(type, value) of an expression value"""
    if o is None:
        return ("null", None)
    if isinstance(o, bool):
        return ("bool", o)
    if isinstance(o, int):
        return ("int", _i64(o))
    if isinstance(o, float):
        return ("float", o)
    if isinstance(o, str):
        return ("str", _b(o))
    if isinstance(o, Map):
        return ("bool", True)
    raise AssertionError(o)


def _reduce(c, ts, m):
    """flb_sp.c:1011-1159; None stands for the NULL pointer"""
    kind = c.a[0]
    if kind == "val":
        t, v = c.a[1], c.a[2]
        return (t, v.encode() if t == "str" else v)
    if kind == "key":
        v = key_to_value(c.a[1], m, c.a[2])
        return None if v is NOVALUE else _typed(v)
    if kind == "func":
        if c.a[1] == "contains":
            p = _reduce(c.a[2], ts, m)
            return None if p is None else ("bool", True)
        return ("float", time_to_double(ts))
    op, l, r = c.a[1], c.a[2], c.a[3]
    left = _reduce(l, ts, m) if l is not None else None
    right = _reduce(r, ts, m) if r is not None else None
    if op == "PAR":
        return ("bool", False if left is None else bool(left[1]) if left[0] == "bool" else False)
    if op in ("NOT", "AND", "OR"):
        lv, rv = _to_bool(left), _to_bool(right)
        return ("bool", (not lv) if op == "NOT" else (lv and rv) if op == "AND" else (lv or rv))
    # numerical_comp
    if left is None or right is None:
        return ("bool", False)
    if left[0] == "str" and right[0] != "str":
        n = string_to_number(left[1])
        if n is not None:
            left = ("float", n[1]) if n[0] == "f" else ("int", n[1])
    if left[0] == "int" and right[0] == "float":
        left = ("float", float(left[1]))
    elif left[0] == "float" and right[0] == "int":
        right = ("float", float(right[1]))
    if left[0] != right[0]:
        return ("bool", False)
    t, a, b = left[0], left[1], right[1]
    if op == "EQ":
        if t == "null":
            return ("bool", True)
        if t == "str":
            return ("bool", len(a) == len(b) and _strncmp(a, b, len(a)) == 0)
        return ("bool", a == b)
    if t in ("int", "float"):
        return ("bool", {"LT": a < b, "LTE": a <= b, "GT": a > b, "GTE": a >= b}[op])
    if t == "str":
        c_ = _strncmp(a, b, len(a))
        return ("bool", {"LT": c_ < 0, "LTE": c_ <= 0, "GT": c_ > 0, "GTE": c_ >= 0}[op])
    return ("bool", False)


def _to_bool(v):
    if v is None:
        return False
    t, x = v
    if t == "bool":
        return bool(x)
    if t in ("int", "float"):
        return x > 0
    if t == "str":
        return True
    return False


# ------------------------------------------------------------------------------------------ aggregation
class _Num:
    __slots__ = ("type", "ops", "i64", "f64", "boolean", "string")

    def __init__(self):
        self.type, self.ops, self.i64, self.f64, self.boolean, self.string = "i", 0, 0, 0.0, False, None


def _wrap(i):
    return (i + 2 ** 63) % 2 ** 64 - 2 ** 63


def _tfunc_value(tfunc, tag, now):
    """the value half of flb_sp_func_time.c:39-83 / flb_sp_func_record.c:39-63; `now` stands for time(NULL) and, for RECORD_TIME
    of a packaged aggregate, for the package time"""
    import time as _time
    if tfunc == "NOW":
        return _pack_str(_time.strftime("%Y-%m-%d %H:%M:%S", _time.localtime(now[0])).encode())
    if tfunc == "UNIX_TIMESTAMP":
        return _pack_int(now[0])
    if tfunc == "RECORD_TAG":
        return _pack_str(tag)
    return b"\xcb" + struct.pack(">d", float(now[0]) + float(now[1]) / 1000000000.0)


class Task:
    def __init__(self, sql, str_conv=True, now=(1, 0), tag=b"t"):
        self.q = parse(sql) if isinstance(sql, str) else sql
        self.conv = str_conv
        self.now, self.tag = now, tag                                  # time(NULL) and the tag of the chunks (time / record functions)
        self._reset()

    def _reset(self):
        self.groups = {}          # canonical key -> node
        self.order = []
        self.records = 0
        self.col_class = [None] * len(self.q.gb_keys)
        self.slots = []           # HOPPING: task->window.hopping_slot, oldest first

    def _group(self, m):
        q = self.q
        if not q.gb_keys:
            if not self.order:
                node = {"records": 1, "nums": [_Num() for _ in q.keys], "gb": None, "canon": ()}
                self.order.append(node)
            else:
                node = self.order[0]
                node["records"] += 1
            return node
        gb = [_Num() for _ in q.gb_keys]
        found = 0
        for k, _v in m.pairs:
            for gi, (gname, gsub) in enumerate(q.gb_keys):
                if not isinstance(k, (str, bytes)) or _b(k) != gname.encode():      # (no type check in this loop: flb_sp.c:1317-1326)
                    continue
                v = key_to_value(gname, m, gsub)
                if v is NOVALUE:
                    continue
                found += 1
                n = object_to_number(v, self.conv)
                g = gb[gi]
                if n is None:
                    if isinstance(v, str):
                        g.type, g.string = "s", _b(v)
                    elif isinstance(v, bool):
                        g.type, g.i64 = "i", int(v)
                elif n[0] == "i":
                    g.type, g.i64 = "i", n[1]
                else:
                    g.type, g.f64 = "f", n[1]
        if found < len(q.gb_keys):
            return None
        canon = []
        for gi, g in enumerate(gb):
            cls = "s" if g.type == "s" else "f" if g.type == "f" else "i"
            if self.col_class[gi] is None:
                self.col_class[gi] = cls
            elif self.col_class[gi] != cls:
                raise Unsupported("GROUP BY column %d mixes value classes (%s, %s)" % (gi, self.col_class[gi], cls))
            if cls == "f" and g.f64 != g.f64:
                raise Unsupported("NaN group key")
            canon.append((cls, g.string if cls == "s" else g.i64 if cls == "i" else (g.f64 + 0.0)))
            if cls == "s" and b"\0" in g.string:
                raise Unsupported("NUL in a string group key (strcmp)")
            if cls == "s" and self.q.window == "hopping":
                # a slot's nodes share the sds of a string key with the window's node (memcpy of groupby_nums / nums,
                # flb_sp.c:1900,1985) and both destroy it: the reference binary dies with a double free
                raise Unsupported("string GROUP BY key in a HOPPING window (the reference frees the key twice)")
        canon = tuple(canon)
        node = self.groups.get(canon)
        if node is None:
            node = {"records": 1, "nums": [_Num() for _ in q.keys], "gb": gb, "canon": canon}
            self.groups[canon] = node
            self.order.append(node)
        else:
            node["records"] += 1
        return node

    def _do_select(self, chunk):
        """sp_process_data (flb_sp.c:1607-1850): per record WHERE, then [the record's first element re-packed, a map of the
        selected pairs]; the map header keeps the width msgpack_pack_map chose for the INCOMING size, the count of what was
        written is patched in afterwards (:1801-1815 -- a fixmap byte takes more than 15 as it comes); a record none of
        whose keys exist leaves nothing (:1796-1799).  Returns (records that passed WHERE, bytes) -- (0, b"") when none did."""
        q = self.q
        b = bytes(chunk)
        out = bytearray()
        records = 0
        off = 0
        while off < len(b):
            try:
                end = _raw_skip(b, off)
            except (IndexError, struct.error, ValueError):
                break                                                    # msgpack_unpack_next stops at what does not decode
            rec = b[off:end]
            off = end
            kind, cnt, p = _raw_tok(rec, 0)
            if kind != "array" or cnt != 2:
                raise Unsupported("record is not [time, map]")
            e0_end = _raw_skip(rec, p)
            mkind, map_size, mp = _raw_tok(rec, e0_end)
            if mkind != "map":
                raise Unsupported("record body is not a map")
            (ts, m), = decode_chunk(rec)
            if q.cond is not None:
                r = _reduce(q.cond, ts, m)
                if r is None or not r[1]:
                    continue
            records += 1
            pairs = []
            for _ in range(map_size):
                ks = mp
                vs = _raw_skip(rec, ks)
                mp = _raw_skip(rec, vs)
                pairs.append((ks, vs, mp))
            body = bytearray()
            entries = 0
            for ck in q.keys:
                if ck.tfunc is not None:
                    # flb_sp_func_time / flb_sp_func_record (:1732-1746): one entry whatever the record holds
                    body += _pack_str(ck.out_name.encode())
                    if ck.tfunc == "RECORD_TIME":
                        body += b"\xcb" + struct.pack(">d", time_to_double(ts))
                    else:
                        body += _tfunc_value(ck.tfunc, self.tag, self.now)
                    entries += 1
                    continue
                for ks, vs, ve in pairs:
                    kk, kv, _n = _raw_tok(rec, ks)
                    if kk != "str":
                        continue
                    if ck.name is None:
                        body += _raw_repack(rec, ks)[0] + _raw_repack(rec, vs)[0]
                        entries += 1
                        continue
                    if rec[kv[0]:kv[0] + kv[1]] != ck.name.encode():
                        continue
                    # flb_sp_key_create gives a key with sub-keys and no alias the alias "k['a']['b']" (flb_sp_parser.c:206-222)
                    body += _pack_str(ck.out_name.encode()) if (ck.alias is not None or ck.subkeys) else _raw_repack(rec, ks)[0]
                    v = _raw_key_to_value(rec, pairs, ck.name.encode(), ck.subkeys)
                    if v is not None:
                        body += _raw_repack(rec, v)[0]                   # (a NULL value: the key stays alone in the map)
                    entries += 1
            if entries == 0:
                continue
            out += b"\x92" + _raw_repack(rec, p)[0]
            if map_size < 16:
                out += bytes([0x80 | (entries & 0xFF)])
            elif map_size < 65536:
                out += b"\xde" + struct.pack(">H", entries & 0xFFFF)
            else:
                out += b"\xdf" + struct.pack(">I", entries & 0xFFFFFFFF)
            out += body
        if records == 0:
            return 0, b""
        return records, bytes(out)

    def do(self, chunk):
        q = self.q
        if q.select_only:
            return self._do_select(chunk)
        for ts, m in decode_chunk(chunk):
            if not isinstance(m, Map):
                raise Unsupported("record body is not a map")
            if q.cond is not None:
                r = _reduce(q.cond, ts, m)
                if r is None or not r[1]:
                    continue
            node = self._group(m)
            if node is None:
                continue
            self.records += 1
            nums = node["nums"]
            for k, _v in m.pairs:
                if not isinstance(k, str):
                    continue
                kb = _b(k)
                for ki, ck in enumerate(q.keys):
                    if ck.name is None or ck.name.encode() != kb:
                        continue
                    v = key_to_value(ck.name, m, ck.subkeys)
                    if v is NOVALUE:
                        continue
                    num = nums[ki]
                    if ck.func:
                        n = object_to_number(v, self.conv)
                        if n is None:
                            continue
                        ival, dval = (n[1], 0.0) if n[0] == "i" else (0, n[1])
                        if dval != 0.0 and num.type == "i":
                            num.type, num.f64 = "f", float(num.i64)
                        self._add(ck.func, num, ival, dval)
                    else:
                        if isinstance(v, bool):
                            num.type, num.boolean = "b", v
                        elif isinstance(v, int):
                            num.type, num.i64 = "i", _i64(v)
                        elif isinstance(v, float):
                            num.type, num.f64 = "f", v
                        elif isinstance(v, str):
                            if q.window == "hopping":
                                # (also when str_conv turns the value into a number for the GROUP BY column: nums[] keeps the sds)
                                raise Unsupported("string value under a plain select key in a HOPPING window (the reference frees it twice)")
                            num.type = "s"
                            if num.string is None:
                                num.string = _b(v)
        out = b""
        if q.window == "default":
            out = self._package()
            self._prune()
        return self.records if q.window != "default" else self._last_records, out

    @staticmethod
    def _add(func, num, ival, dval):
        if func in (FLB_SP_AVG, FLB_SP_SUM):
            if num.type == "i":
                num.i64 = _wrap(num.i64 + ival)
            else:
                num.f64 += dval if dval != 0.0 else float(ival)
            num.ops += 1
        elif func in (FLB_SP_MIN, FLB_SP_MAX):
            less = (lambda a, b: a > b) if func == FLB_SP_MIN else (lambda a, b: a < b)
            if num.type == "i":
                if num.ops == 0 or less(num.i64, ival):
                    num.i64 = ival
                    num.ops += 1
            else:
                x = dval if dval != 0.0 else float(ival)
                if num.ops == 0 or less(num.f64, x):
                    num.f64 = x
                    num.ops += 1

    def _package(self, now=(1, 0)):
        self._last_records = self.records
        p = msgpack.Packer(use_single_float=True, use_bin_type=False)
        out = []
        for node in self.order:
            rec = bytearray(b"\x92\xd7\x00" + struct.pack(">II", now[0], now[1]))
            n = len(self.q.keys)
            rec += p.pack_map_header(n)
            for ki, ck in enumerate(self.q.keys):
                rec += _pack_str(ck.out_name.encode())
                if ck.tfunc is not None:
                    rec += _tfunc_value(ck.tfunc, self.tag, now)           # flb_sp.c:1199-1206 (time(NULL) == the package time here)
                    continue
                num = node["nums"][ki]
                if ck.gb is not None and node["gb"] is not None:
                    num = node["gb"][ck.gb]
                if ck.func == 0:
                    if num.type == "i":
                        rec += _pack_int(num.i64)
                    elif num.type == "f":
                        rec += b"\xca" + struct.pack(">f", _to_f32(num.f64))
                    elif num.type == "s":
                        rec += _pack_str(num.string)
                    elif num.type == "b":
                        rec += b"\xc3" if num.boolean else b"\xc2"
                elif ck.func == FLB_SP_AVG:
                    s = float(num.i64) if num.type == "i" else num.f64
                    rec += b"\xca" + struct.pack(">f", _to_f32(s / node["records"]))
                elif ck.func == FLB_SP_COUNT:
                    rec += _pack_int(node["records"])
                else:
                    if num.type == "i":
                        rec += _pack_int(num.i64)
                    else:
                        rec += b"\xca" + struct.pack(">f", _to_f32(num.f64))
            out.append(bytes(rec))
        return b"".join(out)

    def _remove_sums(self, node_nums, prev_nums):
        """aggregate_func_remove[] (flb_sp_aggregate_func.c:207-221,348-355): AVG / SUM subtract by the type of the node that
        loses the values (aggregate_num is a struct: an I64-typed slot holds f64 == 0.0), COUNT / MIN / MAX are no-ops"""
        for ki, ck in enumerate(self.q.keys):
            if ck.func in (FLB_SP_AVG, FLB_SP_SUM):
                a, b = node_nums[ki], prev_nums[ki]
                if a.type == "i":
                    a.i64 = _wrap(a.i64 - b.i64)
                elif a.type == "f":
                    a.f64 -= b.f64

    def hop(self):
        """sp_process_hopping_slot (flb_sp.c:1852-2004): the hop timer (every ADVANCE BY) closes a slot = what the window
        gained since the previous slot, as a clone of every live node minus the slots still in the list"""
        hs = {"nodes": {}, "records": 0}
        for node in self.order:
            c = {"records": node["records"], "nums": []}
            for n in node["nums"]:
                m = _Num()
                m.type, m.ops, m.i64, m.f64, m.boolean, m.string = n.type, n.ops, n.i64, n.f64, n.boolean, n.string
                c["nums"].append(m)
            for prev in self.slots:
                pn = prev["nodes"].get(node["canon"])
                if pn is not None:
                    c["records"] -= pn["records"]
                    self._remove_sums(c["nums"], pn["nums"])
            if c["records"] > 0:
                hs["nodes"][node["canon"]] = c
        hs["records"] = self.records - sum(p["records"] for p in self.slots)
        self.slots.append(hs)
        return 0

    def _prune(self):
        if self.q.window == "hopping":
            # flb_sp_window_prune, FLB_SP_WINDOW_HOPPING (flb_sp_window.c:57-104): the oldest slot leaves the window
            if not self.slots:
                return
            hs = self.slots[0]
            for node in list(self.order):
                pn = hs["nodes"].get(node["canon"])
                if pn is None:
                    continue
                if pn["records"] == node["records"]:
                    self.order.remove(node)
                    self.groups.pop(node["canon"], None)
                else:
                    node["records"] -= pn["records"]
                    self._remove_sums(node["nums"], pn["nums"])
            self.records -= hs["records"]
            self.slots.pop(0)
            return
        if self.records > 0:
            self._reset()

    def timer(self, now=(1, 0)):
        out = self._package(now) if self.records > 0 else b""
        self._prune()
        return out


def _to_f32(d):
    """(float) d as C does it: round to nearest even, overflow to inf"""
    try:
        return struct.unpack("<f", struct.pack("<f", d))[0]
    except OverflowError:
        return math.copysign(math.inf, d)


def _pack_str(b):
    n = len(b)
    if n < 32:
        return bytes([0xa0 | n]) + b
    if n < 256:
        return b"\xd9" + bytes([n]) + b
    if n < 65536:
        return b"\xda" + struct.pack(">H", n) + b
    return b"\xdb" + struct.pack(">I", n) + b


def _pack_int(v):
    """msgpack_pack_int64: the shortest encoding, unsigned families for v >= 0"""
    if v >= 0:
        if v < 128:
            return bytes([v])
        if v < 256:
            return b"\xcc" + bytes([v])
        if v < 65536:
            return b"\xcd" + struct.pack(">H", v)
        if v < 2 ** 32:
            return b"\xce" + struct.pack(">I", v)
        return b"\xcf" + struct.pack(">Q", v)
    if v >= -32:
        return struct.pack("b", v)
    if v >= -128:
        return b"\xd0" + struct.pack("b", v)
    if v >= -32768:
        return b"\xd1" + struct.pack(">h", v)
    if v >= -2 ** 31:
        return b"\xd2" + struct.pack(">i", v)
    return b"\xd3" + struct.pack(">q", v)
