"""JSON corpus + bindings for the JSON -> msgpack checks (test infrastructure)."""
import ctypes, os, random
from ctypes import c_char_p, c_int, c_size_t, c_void_p, POINTER, byref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bind(lib, name):
    f = getattr(lib, name)
    f.argtypes = [c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]
    f.restype = c_int
    return f


class Packer:
    """(ret, out bytes, root_type, records, consumed) like flb_pack_json_recs"""
    def __init__(self, path, fn, free_fn):
        self.lib = ctypes.CDLL(path)
        self.fn = _bind(self.lib, fn)
        self.free = getattr(self.lib, free_fn)
        self.free.argtypes = [c_void_p]
    def __call__(self, js):
        out = c_void_p(); sz = c_size_t(); rt = c_int(0); rec = c_int(0); cons = c_size_t(0)
        r = self.fn(js, len(js), byref(out), byref(sz), byref(rt), byref(rec), byref(cons))
        if r != 0:
            return (r, None, 0, 0, 0)
        data = ctypes.string_at(out, sz.value) if out.value else b""
        if out.value:
            self.free(out)
        return (0, data, rt.value, rec.value, cons.value)


def oracle():
    return Packer(os.path.join(ROOT, "oracle", "liboracle.so"), "ojson_pack", "oflb_free")


def reference():
    p = os.path.join(ROOT, "oracle", "_ref", "libyyjson_ref.so")
    return Packer(p, "ref_pack_json", "ref_free") if os.path.exists(p) else None


EDGE = [
    b'{}', b'[]', b'{"a":1}', b'[1,2,3]', b' \t\r\n{"a" : [ 1 , 2 ] , "b" : { } }\n', b'"str"', b'123', b'-0', b'0', b'-0.0', b'0.0',
    b'true', b'false', b'null', b'tru', b'nul', b'truex', b'1 2 3', b'{"a":1}{"b":2}', b'{"a":1}\n{"b":2}\n', b'{"a":1} x', b'x {"a":1}',
    b'[1,]', b'[,1]', b'{"a":1,}', b'{,}', b'{"a"}', b'{"a":}', b'{a:1}', b"{'a':1}", b'[1 2]', b'[1', b'{"a":1', b'', b'   ', b'\n',
    b'01', b'-', b'+1', b'.5', b'1.', b'1.e5', b'1e', b'1e+', b'1E5', b'1e-5', b'-1e400', b'1e400', b'1e-400', b'1e99999999999999999999',
    b'1e-99999999999999999999', b'0e99999999999999999999', b'18446744073709551615', b'18446744073709551616', b'9223372036854775807',
    b'9223372036854775808', b'-9223372036854775808', b'-9223372036854775809', b'123456789012345678901234567890', b'-123456789012345678901',
    b'1.7976931348623157e308', b'1.7976931348623159e308', b'4.9e-324', b'2.4703282292062327e-324', b'2.4703282292062328e-324',
    b'0.1', b'1e23', b'8.41e21', b'9007199254740993', b'9007199254740993.0', b'[1.5,2.5,-3.25e2]', b'12345678901234567890.5', b'100000000000000000000',
    b'1.0000000000000000000000000000000000000000000000001', b'0.000000000000000000000000000000000000000000000000000000001e60',
    b'"\\u0041\\u00e9\\u4e2d\\ud83d\\ude00"', b'"\\ud83d"', b'"\\ud83dx"', b'"\\ud83d\\u0041"', b'"\\ud83d\\uZZZZ"', b'"\\ud83d\\u12"', b'"\\ude00"',
    b'"\\u12"', b'"\\u12G4"', b'"\\u12\\"x"', b'"\\uZ"', b'"\\u"', b'"\\u12', b'"\\u12\x00"', b'"\\u0000"', b'"\\x41"', b'"\\a"', b'"\\',
    b'"a\nb"', b'"a\x01b"', b'"a\x00b"', b'"abc', b'"abc\x00', b'"\xff\xfe"', b'"\xc3\xa9\xe4\xb8\xad\xf0\x9f\x98\x80"', b'"\xed\xa0\x80"', b'"\xc0\xaf"',
    b'"\\"\\\\\\/\\b\\f\\n\\r\\t"', b'{"k\\n":"v\\u00e9","k\\n":2}', b'[[[[[[[[[[1]]]]]]]]]]', b'[' * 70 + b']' * 70, b'[' * 200 + b'1' + b']' * 200,
    b'{"a":{"b":{"c":[{"d":1},{"e":[]}]}}}', b'[{"a":1},{"a":2}]' * 3, b'["' + b'x' * 40 + b'"]', b'{"' + b'k' * 300 + b'":"' + b'v' * 70000 + b'"}',
    b'[' + b','.join(b'%d' % i for i in range(20)) + b']', b'{' + b','.join(b'"k%d":%d' % (i, i) for i in range(17)) + b'}',
    b'[' + b','.join(b'1' for i in range(70000)) + b']', b'\xef\xbb\xbf{"a":1}', b'{"a":1}\x00{"b":2}', b'[1]\x00', b'[1,\x002]', b'/* c */ 1', b'[1] // x',
    b'NaN', b'Infinity', b'-Infinity', b'[nan]', b'{"a":tru}', b'{"a":nulL}', b'[1,,2]', b'[1]]', b'{"a":1}}', b'}', b']', b',', b':',
    b'{"a":"b":"c"}', b'{"a",1}', b'[1:2]', b'{"a":[1,2}', b'["a":1]', b'\t\t1', b'1\t\t2', b'"a""b"', b'"a" "b"', b'1,2', b'[1],[2]',
]


def rand_value(rng, depth=0):
    t = rng.randrange(12 if depth < 5 else 8)
    if t == 0: return rng.choice(["true", "false", "null"])
    if t == 1: return str(rng.randrange(-10 ** rng.randrange(1, 22), 10 ** rng.randrange(1, 22)))
    if t == 2: return repr(rng.uniform(-1, 1) * 10.0 ** rng.randrange(-30, 30))
    if t == 3: return "%d.%de%d" % (rng.randrange(100), rng.getrandbits(rng.randrange(1, 90)), rng.randrange(-340, 320))
    if t < 8:
        parts = []
        for _ in range(rng.randrange(0, 12)):
            k = rng.randrange(14)
            if k < 7: parts.append(rng.choice("abcXYZ 0123_-:,{}[]"))
            elif k == 7: parts.append(rng.choice(['\\"', "\\\\", "\\/", "\\b", "\\f", "\\n", "\\r", "\\t"]))
            elif k == 8: parts.append("\\u%04x" % rng.choice([0x41, 0xe9, 0x4e2d, 0x0, 0x1f, 0xd83d, 0xde00, 0xffff, 0xd800, 0xdfff]))
            elif k == 9: parts.append("\\ud83d\\ude%02x" % rng.randrange(256))
            elif k == 10: parts.append(rng.choice(["é", "中", "😀"]))
            elif k == 11: parts.append(rng.choice(["\\u12", "\\uZ", "\\ud83d\\u", "\\ud83d\\uD", "\\x", "\x01", "\x7f"]))
            else: parts.append(rng.choice("abc"))
        return '"' + "".join(parts) + '"'
    ws = lambda: rng.choice(["", "", "", " ", "\n", "\t ", "\r\n"])
    if t < 10:
        return "[" + ws() + ("," + ws()).join(rand_value(rng, depth + 1) for _ in range(rng.randrange(0, 6))) + ws() + "]"
    items = []
    for _ in range(rng.randrange(0, 6)):
        items.append(rand_value(rng, 9).replace("true", '"t"') if False else '"%s"' % rng.choice(["a", "b", "key", "k\\n", "", "é"]) + ws() + ":" + ws() + rand_value(rng, depth + 1))
    return "{" + ws() + ("," + ws()).join(items) + ws() + "}"


def mutate(rng, b):
    b = bytearray(b)
    for _ in range(rng.randrange(1, 4)):
        if not b: break
        k = rng.randrange(4)
        i = rng.randrange(len(b))
        if k == 0: del b[i]
        elif k == 1: b.insert(i, rng.choice(b'{}[]",:\\ue01 \n\x00\xff'))
        elif k == 2: b[i] = rng.choice(b'{}[]",:\\u0e1.-+tfn \x00\x80')
        else: del b[i:]
    return bytes(b)


def corpus(seed, n):
    rng = random.Random(seed)
    out = list(EDGE)
    for _ in range(n):
        docs = [rand_value(rng).encode("utf-8", "surrogatepass") for _ in range(rng.choice([1, 1, 1, 2, 3]))]
        js = rng.choice([b"", b"\n", b" ", b"\r\n"]).join(docs) + rng.choice([b"", b"\n", b"  "])
        out.append(js)
        if rng.random() < 0.5:
            out.append(mutate(rng, js))
    return out
