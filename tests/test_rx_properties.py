"""Class intersection [x&&y] (round 4) and \\p{Name} / \\P{Name} / \\p{^Name} with the names that are the POSIX brackets' ctypes (round 4): the product's tables (executed on the
host) and the oracle's engine against the REAL Onigmo, every name in eight spellings -- outside brackets the NOT is a flag of the class,
inside the complement is added: what an ill-formed byte matches differs between the two --, on ASCII, UTF-8 and ill-formed texts.
General categories, scripts, binary properties and blocks (\\p{Lu}, \\p{Han}, \\p{Emoji}, \\p{In_Cyrillic}, \\p{Punct} = category P ...):
the members were probed from the real engine into unicode_props.inc (tools/gen_unicode_props.py); checked here the same way, the wide
classes through the NFA engine's walk on the host.  Unknown names stay refused (the ages, \\p{Age=6.0}, are in the table since round 5)."""
import ctypes, random, sys, os
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import flbamd_loader
import rxdiff

NAMES = ["Alpha", "Digit", "Alnum", "Upper", "Lower", "Space", "Blank", "Cntrl", "Graph", "Print", "XDigit", "Word", "ASCII"]
FORMS = [r"\p{%s}+", r"\P{%s}+", r"\p{^%s}+", r"[\p{%s}x]+", r"[^\p{%s}]+", r"[\P{%s}0]+", r"(?a)\p{%s}+x?", r"a\p{ %s }{2}"]


@pytest.mark.skipif(rxdiff.load_ref() is None, reason="oracle/_ref/libonig_ref.so not built (needs /root/reference)")
def test_property_names_against_the_real_engine():
    L = flbamd_loader.load().lib(); ref = rxdiff.load_ref(); orx = rxdiff.load_orx()
    L.flbgpu_rx_compile.restype = ctypes.c_void_p
    rng = random.Random(5)
    total = 0
    for nm in NAMES:
        for form in FORMS:
            p = (form % nm).encode()
            e = rxdiff.RefRegex(ref, p); err = ctypes.create_string_buffer(256)
            h = L.flbgpu_rx_compile(p, len(p), 0, 1, err, 256); o = rxdiff.OrxRegex(orx, p)
            assert e.ok and h and o.ok, (p, e.ok, err.value, o.ok)
            for k in range(120):
                s = rxdiff.rand_input(rng, p, 20, utf8=(k % 3 == 1)) if k % 3 != 2 else rxdiff.rand_input_illformed(rng, p, 16)
                if k % 7 == 0:
                    s += "É٣²¡« 　ＡKǅ́­ Ж".encode()
                beg = (ctypes.c_int * 40)(); end = (ctypes.c_int * 40)()
                n = L.flbgpu_rx_simulate_capture(ctypes.c_void_p(h), s, len(s), beg, end)
                got = None if n == -1 else [(beg[i], end[i]) for i in range(n)]
                want = e.search(s)
                if got != want or o.search(s) != want:
                    fl = ctypes.c_int()
                    assert L.flbgpu_rx_corner(ctypes.c_void_p(h), s, len(s), ctypes.byref(fl)) == 1, (p, s, got, want, o.search(s))
                total += 1
            L.flbgpu_rx_free(ctypes.c_void_p(h))
    assert total > 10000


UNI_NAMES = ["Han", "Lu", "L", "Greek", "Cyrillic", "Hiragana", "Katakana", "Nd", "P", "Punct", "S", "Sc", "Zs", "Cc", "Latin", "Arabic", "Hebrew", "Thai",
             "Hangul", "Emoji", "Any", "Assigned", "In_Basic_Latin", "InCyrillic", "Letter", "Uppercase_Letter", "M", "Mn", "White_Space", "Alphabetic",
             "Common", "Lo", "N", "Z", "C", "Cn", "Co", "Devanagari", "Math", "Hex_Digit", "ID_Start", "XID_Continue",
             "Age=1.1", "Age=3.0", "Age=6.0", "Age=10.0"]          # (round 5: the ages are in the table)
UNI_FORMS = [r"\p{%s}+", r"\P{%s}+", r"\p{^%s}+", r"[\p{%s}x]+", r"[^\p{%s}]+", r"[\P{%s}0]+", r"a\p{ %s }{2}", r"(?<w>\p{%s}+)-(?<r>.*)"]
UNI_POOL = "aZ09 _-$+<=>^`|~.,;:!?\t\n é ß Ж ж λ Σ 中 文 あ ア 한 ก ا ש ३ ٣ ² ½ € £ ∑ ≠ 😀 　 \u00a0 \u0301 \u200b \U00020000 \U000e0001 \ufffd \u0378".split(" ")


def _uni_text(rng):
    n = rng.randrange(0, 14)
    s = "".join(rng.choice(UNI_POOL) if rng.random() < 0.8 else
                chr(rng.choice([rng.randrange(0x80, 0x3000), rng.randrange(0x3000, 0xd7ff), rng.randrange(0xe000, 0x10ffff)])) for _ in range(n))
    b = s.encode()
    if rng.random() < 0.15 and b:
        k = rng.randrange(len(b))
        b = b[:k] + bytes([rng.choice([0x80, 0xff, 0xc3, 0xe4, 0xf0])]) + b[k:]
    return b


@pytest.mark.skipif(rxdiff.load_ref() is None, reason="oracle/_ref/libonig_ref.so not built (needs /root/reference)")
def test_unicode_properties_against_the_real_engine():
    """categories, scripts, binary properties, blocks: the oracle's engine and the product's (tables where they fit, else the NFA walk, both
    executed on the host) against the real Onigmo"""
    L = flbamd_loader.load().lib(); ref = rxdiff.load_ref(); orx = rxdiff.load_orx()
    L.flbgpu_rx_compile.restype = ctypes.c_void_p
    rng = random.Random(8)
    total = corners = 0
    for nm in UNI_NAMES:
        for form in UNI_FORMS:
            p = (form % nm).encode()
            e = rxdiff.RefRegex(ref, p); err = ctypes.create_string_buffer(256)
            h = L.flbgpu_rx_compile(p, len(p), 0, 1, err, 256); o = rxdiff.OrxRegex(orx, p)
            assert e.ok and h and o.ok, (p, e.ok, err.value, o.ok)
            for k in range(60):
                s = _uni_text(rng)
                beg = (ctypes.c_int * 40)(); end = (ctypes.c_int * 40)()
                n = L.flbgpu_rx_simulate_capture(ctypes.c_void_p(h), s, len(s), beg, end)
                got = None if n == -1 else [(beg[i], end[i]) for i in range(n)]
                want = e.search(s)
                assert o.search(s) == want, (p, s, o.search(s), want)
                if got != want:
                    fl = ctypes.c_int()
                    assert L.flbgpu_rx_corner(ctypes.c_void_p(h), s, len(s), ctypes.byref(fl)) == 1, (p, s, got, want)
                    corners += 1
                total += 1
            L.flbgpu_rx_free(ctypes.c_void_p(h))
    assert total > 20000 and corners * 50 < total, (total, corners)


@pytest.mark.skipif(rxdiff.load_ref() is None, reason="oracle/_ref/libonig_ref.so not built (needs /root/reference)")
def test_every_property_name_of_the_table_compiles_like_the_real_engine():
    """every name of unicode_props.inc: the real engine takes it, the product takes it, one member and one non-member agree"""
    import re
    L = flbamd_loader.load().lib(); ref = rxdiff.load_ref()
    L.flbgpu_rx_compile.restype = ctypes.c_void_p
    src = open(os.path.join(os.path.dirname(HERE), "fluent-bit_amd", "csrc", "unicode_props.inc")).read()
    names = re.findall(r'\{"([^"]+)", \d+\}', src)
    assert len(names) > 800
    for nm in names:
        p = (r"^\p{%s}$" % nm).encode()
        e = rxdiff.RefRegex(ref, p); err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(p, len(p), 0, 1, err, 256)
        assert e.ok and h, (nm, e.ok, err.value)
        for s in (b"a", b"0", b" ", "é".encode(), "中".encode(), "Ж".encode(), "😀".encode(), "\u0378".encode()):
            beg = (ctypes.c_int * 8)(); end = (ctypes.c_int * 8)()
            n = L.flbgpu_rx_simulate_capture(ctypes.c_void_p(h), s, len(s), beg, end)
            assert (n != -1) == (e.search(s) is not None), (nm, s)
        L.flbgpu_rx_free(ctypes.c_void_p(h))


def test_what_is_not_a_property_stays_refused():
    L = flbamd_loader.load().lib()
    L.flbgpu_rx_compile.restype = ctypes.c_void_p
    for p in [rb"\p{Age=99.0}", rb"\p{NoSuchProperty}", rb"\p{", rb"\pL", rb"[\p{Greekk}]"]:
        err = ctypes.create_string_buffer(256)
        assert not L.flbgpu_rx_compile(p, len(p), 0, 1, err, 256) and err.value, p


AND_PATTERNS = [rb"[a-z&&[^aeiou]]+", rb"[\w&&[^\d_]]+", rb"[a-z&&b-y&&[^m]]+x?", rb"[^a-z&&[^aeiou]]+", rb"[\x{80}-\x{2fff}&&[^\x{e9}]]+", rb"[[:alpha:]&&[^a-f]]+",
                rb"(?i)[a-z&&[^k]]+", rb"[a&&]b*", rb"[\D&&\w]+", rb"[\S&&[^\w]]+", rb"[a-c&&\p{Alpha}]+", rb"[^\W&&[^_]]+", rb"[&&a]", rb"[a-z&&]", rb"[&&]x",
                rb"^(?<k>[\w&&[^\d]]+)=(?<v>[^\s&&[^;]]*)"]


@pytest.mark.skipif(rxdiff.load_ref() is None, reason="oracle/_ref/libonig_ref.so not built (needs /root/reference)")
def test_class_intersection_against_the_real_engine():
    """[x&&y&&z] (regparse.c parse_char_class CC_AND / and_cclass): the product's tables on the host against the real Onigmo"""
    L = flbamd_loader.load().lib(); ref = rxdiff.load_ref()
    L.flbgpu_rx_compile.restype = ctypes.c_void_p
    rng = random.Random(9)
    total = 0
    for p in AND_PATTERNS:
        e = rxdiff.RefRegex(ref, p); err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(p, len(p), 0, 1, err, 256)
        assert e.ok and h, (p, e.ok, err.value)
        for k in range(400):
            s = rxdiff.rand_input(rng, p, 20, utf8=(k % 3 == 1)) if k % 3 != 2 else rxdiff.rand_input_illformed(rng, p, 16)
            if k % 5 == 0:
                s += "aeiouxyz_9 =;éÉ K€".encode()
            beg = (ctypes.c_int * 40)(); end = (ctypes.c_int * 40)()
            n = L.flbgpu_rx_simulate_capture(ctypes.c_void_p(h), s, len(s), beg, end)
            got = None if n == -1 else [(beg[i], end[i]) for i in range(n)]
            want = e.search(s)
            if got != want:
                fl = ctypes.c_int()
                assert L.flbgpu_rx_corner(ctypes.c_void_p(h), s, len(s), ctypes.byref(fl)) == 1, (p, s, got, want)
            total += 1
        L.flbgpu_rx_free(ctypes.c_void_p(h))
    assert total > 6000
