"""Class intersection [x&&y] (round 4) and \\p{Name} / \\P{Name} / \\p{^Name} with the names that are the POSIX brackets' ctypes (round 4): the product's tables (executed on the
host) and the oracle's engine against the REAL Onigmo, every name in eight spellings -- outside brackets the NOT is a flag of the class,
inside the complement is added: what an ill-formed byte matches differs between the two --, on ASCII, UTF-8 and ill-formed texts.
\\p{Punct} (Unicode category P: not the bracket's set), scripts, categories and ages stay refused."""
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


def test_what_is_not_a_posix_ctype_stays_refused():
    L = flbamd_loader.load().lib()
    L.flbgpu_rx_compile.restype = ctypes.c_void_p
    for p in [rb"\p{Punct}", rb"\p{Han}", rb"\p{Lu}", rb"\p{Age=6.0}", rb"\p{", rb"\pL", rb"[\p{Greek}]"]:
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
