"""Seeded corpus for the msgpack -> JSON output formatter (flb_pack_msgpack_to_json_format, src/flb_pack.c:1320-1600):
chunks of log events holding every msgpack type, the string shapes flb_utils_write_str distinguishes, the floats
the "%.1f" / "%.16g" split distinguishes, duplicate keys, metadata, group markers, and the configurations the
output plugins pass (format x date format x date key x escape_unicode x convert_nan_to_null)."""
import random, struct
from synth import mp, Raw, KV, ext_ts

FORMATS = (1, 2, 3)            # json, stream, lines (include/fluent-bit/flb_pack.h:57-60)
DATE_FORMATS = (0, 1, 2, 3, 4)  # double, iso8601, epoch, java_sql_timestamp, epoch_ms (:38-42)

ILL = [b"\x80", b"\xbf", b"\xc0\x80", b"\xc1\xbf", b"\xc3", b"\xe2\x82", b"\xe2\x28\xa1", b"\xed\xa0\x80", b"\xed\xbf\xbf",
       b"\xe0\x80\x80", b"\xe0\x9f\xbf", b"\xf0\x80\x80\x80", b"\xf0\x8f\xbf\xbf", b"\xf4\x90\x80\x80", b"\xf5\x80\x80\x80",
       b"\xf7\xbf\xbf\xbf", b"\xf8\x88\x80\x80\x80", b"\xfc\x84\x80\x80\x80\x80", b"\xfe", b"\xff", b"\xf0\x9f\x98", b"\xf0\x9f",
       b"\xc3\x28", b"\xe2\x82\x28", b"\xf0\x28\x8c\xbc", b"\xf0\x90\x28\xbc", b"\xf0\x28\x8c\x28", b"\xef\xbf\xbd", b"\xee\x83\x8e"]
GOOD = ["é", "ß", "€", "日本語", "😀", "𝄞", "\u07ff", "\u0800", "\uffff", "\U00010000", "\U0010ffff", "\u007f", "\u0080"]


def rand_string(r):
    k = r.randrange(10)
    if k == 0: return b""
    if k == 1: return bytes(r.randrange(0x20, 0x7f) for _ in range(r.randrange(1, 70)))
    if k == 2: return bytes(r.randrange(0, 0x80) for _ in range(r.randrange(1, 40)))
    if k == 3: return "".join(r.choice(GOOD + ["a", "b", " ", "\"", "\\", "\n", "/"]) for _ in range(r.randrange(1, 24))).encode()
    if k == 4: return bytes(r.randrange(256) for _ in range(r.randrange(1, 40)))
    if k == 5:
        parts = []
        for _ in range(r.randrange(1, 8)):
            parts.append(r.choice(ILL) if r.random() < 0.5 else r.choice(GOOD + ["x", "yz", "\t"]).encode())
        return b"".join(parts)
    if k == 6:   # an ill-formed piece at the very end / start (truncation rules look at the string length)
        return bytes(r.randrange(0x20, 0x7f) for _ in range(r.randrange(0, 20))) + r.choice(ILL)[:r.randrange(1, 4)]
    if k == 7:   # long plain runs around one special byte (the 16-byte blocks of the reference's scanner)
        return b"a" * r.randrange(0, 40) + r.choice([b"\"", b"\\", b"\x01", b"\x7f", "é".encode(), b"\xff", b"\n"]) + b"b" * r.randrange(0, 40)
    if k == 8: return r.choice(["GET /index.html HTTP/1.1", "192.168.0.1", "error: something \"quoted\"", "tab\tsep", "C:\\path\\file"]).encode()
    return bytes(r.choice([0x41, 0x22, 0x5c, 0x0a, 0xc3, 0xa9, 0xe2, 0x82, 0xac, 0xf0, 0x9f, 0x98, 0x80, 0x80, 0xff]) for _ in range(r.randrange(1, 30)))


FLOATS = [0.0, -0.0, 1.0, -1.0, 1.5, -2.25, 0.1, 1e15, 1e15 + 0.5, 1e16, 1e17, 1e22, 1e23, 123456789012345678.0, 9007199254740993.0,
          2.0 ** 53, 2.0 ** 62, 2.0 ** 63, -(2.0 ** 63), 2.0 ** 64, 1e300, 1.7976931348623157e308, 5e-324, 2.2250738585072014e-308,
          1e-4, 1e-5, 0.0001234567890123456, 0.00001234, 123456.7890123456789, 0.3, 2.675, 1e-7, 99999.99999999999, 999999999999999.9,
          9999999999999998.0, 9999999999999999.0, 0.9999999999999999, 0.99999999999999994, 1700000000.123456789, 4.35, 0.5, 1 / 3,
          float("inf"), float("-inf"), float("nan"), -float("nan"), 3.14159, 1e21, 1e-10, 123.456e-20, 8.5e18, 9.3e18]


def rand_float(r):
    k = r.randrange(6)
    if k == 0: return Raw(b"\xcb" + struct.pack(">d", r.choice(FLOATS)))
    if k == 1: return Raw(b"\xcb" + struct.pack(">Q", r.getrandbits(64)))
    if k == 2: return Raw(b"\xca" + struct.pack(">I", r.getrandbits(32)))
    if k == 3: return Raw(b"\xca" + struct.pack(">f", r.choice([0.1, 1.5, 3.0, 1e10, 16777216.0, 1e-3, 3.4e38])))
    if k == 4: return Raw(b"\xcb" + struct.pack(">d", r.uniform(-1, 1) * 10.0 ** r.randrange(-30, 30)))
    return Raw(b"\xcb" + struct.pack(">d", float(r.randrange(-10 ** 6, 10 ** 6)) / r.choice([1, 2, 4, 8, 10, 100, 1000])))


INTS = [0, 1, 127, 128, 255, 256, 65535, 65536, 2 ** 32 - 1, 2 ** 32, 2 ** 63 - 1, 2 ** 63, 2 ** 64 - 1, -1, -32, -33, -128, -129, -32768, -32769,
        -2 ** 31, -2 ** 31 - 1, -2 ** 63, 1700000000, 42]


def rand_value(r, depth=0):
    k = r.randrange(14 if depth < 4 else 10)
    if k == 0: return None
    if k == 1: return r.random() < 0.5
    if k in (2, 3): return r.choice(INTS) if r.random() < 0.6 else r.randrange(-10 ** 12, 10 ** 12)
    if k in (4, 5): return rand_float(r)
    if k in (6, 7, 8): return rand_string(r)
    if k == 9:
        if r.random() < 0.5:   # bin
            b = rand_string(r)
            return Raw((b"\xc4" + bytes([len(b)]) if len(b) < 256 else b"\xc5" + struct.pack(">H", len(b))) + b)
        n = r.choice([1, 2, 4, 8, 16, 3, 0, 20])   # ext
        body = bytes(r.randrange(256) for _ in range(n))
        fix = {1: 0xd4, 2: 0xd5, 4: 0xd6, 8: 0xd7, 16: 0xd8}
        t = struct.pack("b", r.randrange(-128, 128))
        return Raw((bytes([fix[n]]) + t if n in fix and r.random() < 0.8 else b"\xc7" + bytes([n]) + t) + body)
    if k in (10, 11):
        return [rand_value(r, depth + 1) for _ in range(r.randrange(0, 5))]
    return rand_map(r, depth + 1)


KEYS = [b"log", b"message", b"host", b"level", b"a", b"b", b"date", b"time", b"k\"q", "ключ".encode(), b"x" * 40, b""]


def rand_map(r, depth=0, big=False):
    items = []
    for _ in range(r.randrange(0, 24 if big else 6)):
        kk = r.random()
        if kk < 0.8: key = r.choice(KEYS)
        elif kk < 0.9: key = rand_string(r)
        else: key = r.choice([7, None, True, -3, Raw(b"\xc4\x01k")])     # non-string keys are formatted too (and never deduplicated)
        items.append((key, rand_value(r, depth)))
    return KV(items)


def rand_chunk(r, nrec=None):
    out = []
    nrec = r.randrange(0, 8) if nrec is None else nrec
    in_group = False
    for _ in range(nrec):
        k = r.random()
        body = rand_map(r, big=r.random() < 0.1)
        if k < 0.62:
            meta = rand_map(r, 1) if r.random() < 0.2 else KV([])
            out.append(mp([[ext_ts(r.choice([0, 1, 1700000000, 2 ** 31 - 1, 2 ** 32 - 3, r.randrange(0, 2 ** 32 - 2)]),
                                   r.choice([0, 1, 999, 1000, 123456789, 999999999, 999999, 500000000])), meta], body]))
        elif k < 0.72:
            out.append(mp([r.choice([0, 1, 1700000000, 253402300799, 253402300800, 2 ** 40, r.randrange(0, 2 ** 33)]), body]))           # legacy, integer time
        elif k < 0.80:
            out.append(mp([Raw(b"\xcb" + struct.pack(">d", r.choice([0.0, 1700000000.5, 1.25, 1700000000.123456789, 4294967296.75, 0.999999999]))), body]))
        elif k < 0.86:
            out.append(mp([[r.choice([5, 1700000000]), rand_map(r, 1)], body]))                                # v2 header, integer time
        elif k < 0.93:
            in_group = not in_group
            if in_group: out.append(mp([[ext_ts(0xffffffff, 0), rand_map(r, 1)], rand_map(r, 1) if r.random() < 0.8 else KV([])]))
            else: out.append(mp([[ext_ts(0xfffffffe, 0), KV([])], KV([])]))
        elif k < 0.96:
            out.append(mp([[Raw(b"\xcb" + struct.pack(">d", r.choice([-5.0, -1.0, -2.0, -1.5]))), KV([])], body]))   # negative times: markers / skipped
        elif k < 0.98:
            out.append(mp([r.choice([2 ** 64 - 1, 2 ** 64 - 2, 2 ** 64 - 7]), body]))
        else:
            out.append(r.choice([b"\xc1", mp([1, 2, 3]), mp([1, "x"]), mp("str"), mp([[ext_ts(1, 10 ** 9), {}], {}]), mp([[1], {}]), mp([[1, 2], {}])]))
    b = b"".join(out)
    if r.random() < 0.08 and len(b) > 2: b = b[:r.randrange(1, len(b))]
    return b


def rand_config(r):
    fmt = r.choice(FORMATS)
    df = r.choice(DATE_FORMATS)
    dk = r.choice([None, b"date", b"date", b"@timestamp", b"log", b"", "dé".encode()])
    return dict(json_format=fmt, date_format=df, date_key=dk, escape_unicode=r.randrange(2), nan_to_null=r.randrange(2))


def corpus(seed, n):
    r = random.Random(seed)
    cases = []
    for i in range(n):
        cases.append((rand_config(r), rand_chunk(r)))
    # deep nesting against msgpack-c's 32 open containers: body values at depths around the limit
    for depth in (28, 29, 30, 31, 32, 33):
        v = 1
        for _ in range(depth): v = [v]
        for fmt in FORMATS:
            cases.append((dict(json_format=fmt, date_format=0, date_key=b"date", escape_unicode=1, nan_to_null=0),
                          mp([[ext_ts(5, 0), {}], KV([(b"deep", v)])])))
            cases.append((dict(json_format=fmt, date_format=0, date_key=None, escape_unicode=1, nan_to_null=0),
                          mp([[ext_ts(5, 0), KV([(b"m", v)])], KV([(b"k", 1)])])))
    for fmt in FORMATS:
        cases.append((dict(json_format=fmt, date_format=0, date_key=b"date", escape_unicode=1, nan_to_null=0), b""))
    return cases
