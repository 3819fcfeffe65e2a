"""Logical_Op OR: filter_grep merges the rules that test one field into the alternation of their patterns, each inside an
inline option group (fluent-bit_amd/csrc/flbgpu.cpp, flbgpu_filter_grep_create).  The table compiler's answer for the
merged pattern must be "any of the parts matches" -- checked on the host execution of the tables against the oracle regex
(pinned on the reference's engine) for the 32 patterns of BASELINE configs[2] and for option-carrying patterns."""
import ctypes, os, random, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import flbamd_loader
from rxdiff import load_orx, OrxRegex, rand_input


def _split(pat):
    """src/flb_regex.c:60-152: /pat/imx"""
    if pat.startswith("/") and pat.rfind("/") > 0:
        last = pat.rfind("/")
        opts = pat[last + 1:]
        if all(c in "imx" for c in opts):
            return pat[1:last], opts
    return pat, ""


def _merged(pats):
    out = []
    for p in pats:
        inner, opts = _split(p)
        out.append("(?%s:%s%s)" % (opts, inner, "\n" if "x" in opts else ""))
    return "|".join(out)


def test_merged_alternation_equals_any_of_the_rules():
    from bench import GREP32_REGEX, GREP32_EXCLUDE
    L = flbamd_loader.load().lib()
    L.flbgpu_rx_compile.restype = ctypes.c_void_p
    L.flbgpu_rx_compile.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    L.flbgpu_rx_simulate_match.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    L.flbgpu_rx_free.argtypes = [ctypes.c_void_p]
    O = load_orx()
    rng = random.Random(5)
    groups = {}
    for kind, val in GREP32_REGEX + GREP32_EXCLUDE:
        field, pat = val.split(" ", 1)
        groups.setdefault((kind, field), []).append(pat)
    groups[("regex", "opts")] = ["/ab c/x", "/HELLO/i", "^x.y$", "/a.b/m", "tail$"]
    checked = 0
    for (kind, field), pats in groups.items():
        if len(pats) < 2:
            continue
        mp = _merged(pats).encode()
        h = L.flbgpu_rx_compile(mp, len(mp), 0, 0, ctypes.create_string_buffer(256), 256)
        assert h, (field, mp)
        parts = []
        for p in pats:
            inner, opts = _split(p)
            o = OrxRegex(O, inner.encode(), (1 if "i" in opts else 0) | (2 if "x" in opts else 0) | (4 if "m" in opts else 0))
            assert o.ok, p
            parts.append(o)
        texts = [b"", b"request 91 finished ok", b"/v1/items/12345?x=7", b"error", b"warn", b"info", b"db", b"cache", b"pod-1a", b"hello", b"x\ny", b"a\nb", b"abc",
                 b"request 1 finished", b"timeout while connecting", b"connection refused", b"/v2/users/9?x=12", b"555 finished", b"items/12"]
        texts += [rand_input(rng, mp, maxlen=30) for _ in range(400)]
        for t in texts:
            want = any(o.search(t) is not None for o in parts)
            got = L.flbgpu_rx_simulate_match(h, t, len(t)) == 1
            assert got == want, (field, mp, t)
            checked += 1
        L.flbgpu_rx_free(h)
    assert checked > 2000
