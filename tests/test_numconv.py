"""csrc/numconv.hpp (host instantiation through the C ABI) against glibc: strtod, sscanf("%lf"),
printf("%f"), printf("%ld").  The kernels run the device instantiation of the same header
(tests/test_l2m_gpu.py::test_numconv_on_device compares the two)."""
import ctypes, math, random, struct
import flbamd_loader

g = flbamd_loader.load()
L = g.lib()
libc = ctypes.CDLL(None)
libc.strtod.restype = ctypes.c_double
libc.strtod.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p)]


def ours(s, mode, exact=1):
    out = ctypes.c_double(); cons = ctypes.c_int()
    st = L.flbgpu_nc_scan_double(s, len(s), mode, exact, ctypes.byref(out), ctypes.byref(cons))
    return st, out.value, cons.value


def same(a, b):
    return struct.pack("<d", a) == struct.pack("<d", b) or (math.isnan(a) and math.isnan(b))


def gen_cases(rng, n):
    out = []
    for _ in range(n):
        t = rng.randrange(8)
        if t == 0: out.append(repr(struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0]))
        elif t == 1: out.append("%.*e" % (rng.randrange(0, 30), rng.uniform(-1, 1) * 10.0 ** rng.randrange(-320, 308)))
        elif t == 2: out.append("%d.%de%d" % (rng.randrange(10 ** 9), rng.getrandbits(rng.randrange(1, 120)), rng.randrange(-340, 310)))
        elif t == 3:
            # exact decimal expansion of a midpoint between two doubles (the hard rounding cases)
            b = rng.getrandbits(63) % 0x7FE0000000000000
            lo = struct.unpack("<d", struct.pack("<Q", b))[0]; hi = struct.unpack("<d", struct.pack("<Q", b + 1))[0]
            from fractions import Fraction
            mid = (Fraction(lo) + Fraction(hi)) / 2
            if rng.random() < 0.5 and mid.denominator.bit_length() < 1200 and mid.numerator.bit_length() < 1200:
                # scale to an integer numerator over a power of ten
                k = max(mid.denominator.bit_length() - 1, 0)
                num = mid.numerator * 5 ** k
                s = str(num)
                s = (s[:-k] or "0") + "." + s[-k:].rjust(k, "0") if k else s
                out.append(s + rng.choice(["", "0", "1", "0000000000000000000001"]))
            else:
                out.append(repr(lo))
        elif t == 4: out.append("0x%x.%xp%d" % (rng.getrandbits(rng.randrange(1, 80)), rng.getrandbits(rng.randrange(1, 80)), rng.randrange(-1200, 1100)))
        elif t == 5: out.append("%d%s" % (rng.getrandbits(rng.randrange(1, 200)), rng.choice(["", "e-30", "e5", ".5"])))
        elif t == 6: out.append("0." + "0" * rng.randrange(0, 330) + str(rng.getrandbits(64)))
        else: out.append("".join(rng.choice("0123456789.eE+-xXpPinfatyINFANb \t()") for _ in range(rng.randrange(1, 10))))
    return out


def test_strtod_against_glibc():
    rng = random.Random(1)
    n_exact = 0
    for c in gen_cases(rng, 60000):
        b = c.encode()
        end = ctypes.c_char_p()
        buf = ctypes.create_string_buffer(b)
        ref = libc.strtod(buf, ctypes.byref(end))
        consumed = ctypes.cast(end, ctypes.c_void_p).value - ctypes.addressof(buf)
        st, v, cons = ours(b, 0)
        assert (st == 1) == (consumed > 0), c
        if st == 1:
            assert same(v, ref) and cons == consumed, (c, v, ref, cons, consumed)
        st2, v2, _ = ours(b, 0, exact=0)
        if st2 == 2: n_exact += 1
        else: assert st2 == st and (st != 1 or same(v2, ref)), c
    assert n_exact > 100            # the fuzz reaches the big-integer path


def test_sscanf_lf_against_glibc():
    rng = random.Random(2)
    cases = ["1e", "1e+", "0x", "0x1p", "0x.8", "0x.", "0xg", "1.", ".5", ".", "+.e1", "inf", "infinity", "infinit", "infx", "in",
             "nan", "nan(abc)", "nan(", "  12", "1_000", "", "-", "+", "-0", "1e400", "0x1P+", "-inf", "+nan", "1..2", "1e1e1",
             "1d5", "infinityx", "1e-", ".e5", "0.e", "5e 3", "0x.p1", "- 1", "i", "n", "na", "INF", "iNfInItY", "0X.", "12\x0034"]
    cases += gen_cases(rng, 30000)
    for c in cases:
        b = c.encode("latin1")
        ref = ctypes.c_double(-777.0)
        r = libc.sscanf(b, b"%lf", ctypes.byref(ref))
        st, v, _ = ours(b, 1)
        assert (st == 1) == (r == 1), (c, r, st)
        if r == 1:
            assert same(v, ref.value), (c, v, ref.value)


def test_printf_f_and_ld_against_glibc():
    rng = random.Random(3)
    vals = [0.0, -0.0, 1.5, 0.0000005, 0.0000015, 0.5000005, 0.9999995, 999999.9999995, 1e15 + 0.5, 1e22, 1e300, 1.7976931348623157e308,
            5e-324, 2.5e-7, 123456789.987654321, float("inf"), float("-inf"), float("nan"), 2.0 ** 63, 2.0 ** 64, 2.0 ** 53 + 2]
    for _ in range(40000):
        t = rng.randrange(4)
        if t == 0: vals.append(struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0])
        elif t == 1: vals.append(rng.randrange(-10 ** 7, 10 ** 7) / rng.randrange(1, 1000))
        elif t == 2: vals.append(rng.randrange(10 ** 8) / 1e6 + 0.0000005)
        else: vals.append(struct.unpack("<d", struct.pack("<Q", rng.getrandbits(52) | (rng.randrange(1023 - 40, 1023 + 80) << 52)))[0])
    buf = ctypes.create_string_buffer(400)
    ref = ctypes.create_string_buffer(400)
    for v in vals:
        n = L.flbgpu_nc_fmt_f6(v, buf, 251)
        libc.snprintf(ref, 252, b"%f", ctypes.c_double(v))
        want = ref.value
        assert buf.raw[:n] == want, (v, buf.raw[:n], want)
    for v in [0, 1, -1, 9, 10, 2 ** 63 - 1, -2 ** 63, 10 ** 18, -10 ** 18 + 1] + [rng.randrange(-2 ** 63, 2 ** 63) for _ in range(2000)]:
        n = L.flbgpu_nc_fmt_ld(v, buf)
        assert buf.raw[:n] == str(v).encode()


def test_json_double_against_glibc():
    """the float rule of the reference's JSON writer (src/flb_pack.c:1020-1034): "%.1f" when the value survives the
    round trip through long long (as x86-64 converts), else "%.16g"; both as glibc prints them"""
    rng = random.Random(5)
    vals = [0.0, -0.0, 1.0, -1.0, 1.5, 0.1, 1e15, 1e15 + 0.5, 1e16, 1e17, 1e22, 1e23, 2.0 ** 53, 2.0 ** 62, 2.0 ** 63, -(2.0 ** 63), 2.0 ** 64,
            9223372036854774784.0, 1e300, 1.7976931348623157e308, 5e-324, 2.2250738585072014e-308, 1e-4, 1e-5, 0.00001234, 9999999999999998.0,
            9999999999999999.0, 0.99999999999999994, 99999.99999999999, float("inf"), float("-inf"), float("nan"), -float("nan"), 1700000000.123456789]
    for _ in range(120000):
        t = rng.randrange(5)
        if t == 0: vals.append(struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0])
        elif t == 1: vals.append(rng.randrange(-10 ** 15, 10 ** 15) / 10 ** rng.randrange(0, 18))
        elif t == 2: vals.append(float(struct.unpack("<f", struct.pack("<I", rng.getrandbits(32)))[0]))
        elif t == 3: vals.append(1.6e9 + rng.randrange(4 * 10 ** 8) + rng.randrange(10 ** 9) / 1e9)
        else: vals.append(rng.uniform(-1, 1) * 10.0 ** rng.randrange(-320, 309))
    L.flbgpu_nc_fmt_json_double.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_char_p]
    buf = ctypes.create_string_buffer(64)
    ref = ctypes.create_string_buffer(400)
    for v in vals:
        n = L.flbgpu_nc_fmt_json_double(v, 0, buf)
        r = float(int(v)) if v == v and -2.0 ** 63 <= v < 2.0 ** 63 else -2.0 ** 63
        libc.snprintf(ref, 399, b"%.1f" if v == r else b"%.16g", ctypes.c_double(v))
        assert buf.raw[:n] == ref.value, (v, buf.raw[:n], ref.value)
    n = L.flbgpu_nc_fmt_json_double(float("nan"), 1, buf)
    assert buf.raw[:n] == b"null"
    n = L.flbgpu_nc_fmt_json_double(float("inf"), 1, buf)
    assert buf.raw[:n] == b"inf"
