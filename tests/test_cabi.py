"""CPU-side checks of the product library (no GPU): the shared object loads, exports every
symbol include/flb_gpu.h declares, fails loudly without a device, and its regex table compiler
reproduces the real-Onigmo golden vectors when the tables are executed on the host."""
import base64, ctypes, json, os, re
import pytest
import flbamd_loader

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def g():
    return flbamd_loader.load()


def test_exports_match_header(g):
    import glob
    hdr = "".join(open(h).read() for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))))
    names = sorted(set(re.findall(r"\b(flbgpu_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25 and "flbgpu_dec_simulate" in names and "flbgpu_sp_hop" in names
    L = g.lib()
    for n in names:
        assert hasattr(L, n), "libflbgpu.so does not export %s" % n


def test_fails_loudly_without_gpu(g):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        g.init(0)
    assert "no CPU path" in g.last_error()


def test_unsupported_constructs_fail_at_create(g):
    # (look-around, atomic groups, possessive repeats and back-references are no longer among them: the host's backtracking matcher
    # answers such a pattern -- tests/test_host_rules_gpu.py; FLBGPU_NO_HOST_RULES=1 brings the refusal back)
    # (round 5: the absent operator (?~..) and subexpression calls \g<..> are the host matcher's too)
    # (round 5, later: control / meta / octal escapes are taken too; a raw byte above 0x7f stays refused -- tests/test_rxbt.py)
    for rx in [r"(?(1)a|b)", r"\g<1>", r"(?<=a+)b", r"a\xffb", r"a\200b", r"\p{NoSuchProperty}"]:
        with pytest.raises(ValueError) as ei:
            g.Parser("^(?<x>" + rx + ")$")
        assert "cannot compile regex" in str(ei.value), (rx, str(ei.value))       # (not: no device)
    os.environ["FLBGPU_NO_HOST_RULES"] = "1"
    try:
        for rx in [r"(a)\1", r"(?=a)b", r"(?<=a)b", r"(?>a+)b", r"a*+", r"(?~ab)", r"(?<y>a|b\g<y>)", r"\X", r"(?i)é", r"(?i)\p{Greek}"]:
            with pytest.raises(ValueError) as ei:
                g.Parser("^(?<x>" + rx + ")$")
            assert "cannot compile regex" in str(ei.value), (rx, str(ei.value))
    finally:
        del os.environ["FLBGPU_NO_HOST_RULES"]
    # (zone abbreviations, %Z, are taken since round 4 -- csrc/tz_abbr.inc, tests/test_kat_gpu.py)
    with pytest.raises(ValueError):
        g.Parser(r"^(?<time>.*)$", time_fmt="%Y %Q", time_key="time")              # no such directive
    with pytest.raises(ValueError):
        g.FilterGrep([("regex", "log a"), ("exclude", "log b")], "AND")


def test_regex_tables_against_golden(g):
    L = g.lib()
    kat = json.load(open(os.path.join(HERE, "golden", "regex_kat.json")))
    checked = 0
    for ent in kat:
        pat = base64.b64decode(ent["pattern"])
        if not ent["compiles"]:
            continue
        err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 256)
        if not h:
            continue        # constructs rejected by design (look-around, atomic, possessive, \Z, octal)
        buf = ctypes.create_string_buffer(4096)
        L.flbgpu_rx_names(h, buf, 4096)
        names = [[l.rsplit("=", 1)[0], int(l.rsplit("=", 1)[1])] for l in buf.value.decode().splitlines()]
        assert names == ent["names"], pat
        # every case -- ill-formed UTF-8, the non-ASCII members of POSIX brackets and \b / \B next to non-ASCII characters included
        # (deviations until round 3) -- must agree with the real engine
        for s64, want in ent["cases"]:
            s = base64.b64decode(s64)
            beg = (ctypes.c_int * 40)(); end = (ctypes.c_int * 40)()
            n = L.flbgpu_rx_simulate_capture(h, s, len(s), beg, end)
            got = None if n == -1 else [[beg[i], end[i]] for i in range(n)]
            assert got == want, (pat, s)
            assert L.flbgpu_rx_simulate_match(h, s, len(s)) == (0 if want is None else 1), (pat, s)
            checked += 1
        L.flbgpu_rx_free(h)
    assert checked > 15000


def _sim(L, h, s):
    beg = (ctypes.c_int * 40)(); end = (ctypes.c_int * 40)()
    n = L.flbgpu_rx_simulate_capture(h, s, len(s), beg, end)
    return None if n == -1 else [(beg[i], end[i]) for i in range(n)]


def test_wide_reverse_tables_stock_parsers(g):
    """`envoy` and `ambassador` (conf/parsers.conf:102, conf/parsers_ambassador.conf:4) compile since round 2: the
    UTF-8 capture automaton keeps 32-bit transitions (33 k / 96 k states), ambassador's match-only DFA is left out.
    The tables executed on the host against the real Onigmo (the oracle's engine where the reference is absent)."""
    import rxdiff, stock_wide
    L = g.lib()
    ref = rxdiff.load_ref()
    orx = rxdiff.load_orx()
    for pat, prefix in ((stock_wide.ENVOY, b""), (stock_wide.AMBASSADOR, b"ACCESS ")):
        pat = pat.encode()
        err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 256)
        assert h, err.value
        info = (ctypes.c_int * 16)()
        L.flbgpu_rx_info(h, info)
        assert info[2] <= 0x7FF0 < info[7], list(info)               # ascii narrow, utf8 wide
        eng = rxdiff.RefRegex(ref, pat) if ref else rxdiff.OrxRegex(orx, pat)
        assert eng.ok
        hit = 0
        for s in stock_wide.lines(1500, 11, prefix):
            want = eng.search(s)
            assert _sim(L, h, s) == want, (pat[:20], s)
            assert L.flbgpu_rx_simulate_match(h, s, len(s)) == (0 if want is None else 1)
            hit += want is not None
        assert 600 < hit < 1500, hit
        L.flbgpu_rx_free(h)
    # still over every budget: fails loudly at create
    with pytest.raises(ValueError):
        g.Parser(stock_wide.ENVOY.replace("(?<code>", "(?<response_code>").replace('"(?<upstream_host>[^ ]*)"',
                 '"(?<upstream_host>[^ ]*)" ' + " ".join("(?<f%d>[^ ]*)" % i for i in range(6)) + "$"))


def test_wide_reverse_tables_forced_on_golden(g, monkeypatch):
    """FLBGPU_RX_FORCE_WIDE makes every utf8 table set wide: the golden answers with bytes >= 0x80 through the 32-bit
    tables and the two-half checkpoints."""
    monkeypatch.setenv("FLBGPU_RX_FORCE_WIDE", "1")
    L = g.lib()
    kat = json.load(open(os.path.join(HERE, "golden", "regex_kat.json")))
    checked = 0
    for ent in kat:
        pat = base64.b64decode(ent["pattern"])
        if not ent["compiles"]:
            continue
        cases = [(base64.b64decode(a), w) for a, w in ent["cases"]]
        cases = [c for c in cases if any(b >= 0x80 for b in c[0])]
        if not cases:
            continue
        err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 256)
        if not h:
            continue
        info = (ctypes.c_int * 16)()
        L.flbgpu_rx_info(h, info)
        for s, want in cases:
            got = _sim(L, h, s)
            assert got == (None if want is None else [tuple(x) for x in want]), (pat, s)
            checked += 1
        L.flbgpu_rx_free(h)
    assert checked > 4000, checked


def test_index_host(g):
    import synth
    data, off, ep = synth.apache_records(100)
    n, o, consumed = g.index_host(bytes(data))
    assert n == 100 and consumed == len(data) and list(o) == list(off)
    n, o, consumed = g.index_host(bytes(data) + b"\x92\x01")
    assert n == 100 and consumed == len(data)
    n, o, consumed = g.index_host(bytes(data[:277]) + b"\xc1" + bytes(data[277:]))
    assert n == 1 and consumed == 277


def test_reference_side_binding_type_checks_against_reference_headers():
    """fluent-bit_amd/plugin/filter_gpu_plugins.c (the struct flb_filter_plugin shim of INTEGRATION.md)
    against the real fluent-bit headers; only where the reference tree is mounted."""
    import os, subprocess, pytest
    if not os.path.isdir("/root/reference/include/fluent-bit"):
        pytest.skip("reference tree not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([os.path.join(root, "fluent-bit_amd", "plugin", "check_syntax.sh")], capture_output=True, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_threaded_slab_copy(g):
    """the pageable <-> pinned copy of the host-level calls is split over helper threads above 1 MB: every byte arrives"""
    import random
    L = g.lib()
    L.flbgpu_diag_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    rng = random.Random(3)
    src = bytes(rng.getrandbits(8) for _ in range(1 << 16)) * 160
    for n in [(1 << 20) - 1, 1 << 20, (1 << 20) + 1, 1834560 * 4 + 2, 2621441, len(src) - 71] + [rng.randrange(1 << 20, len(src) - 64) for _ in range(20)]:
        off = rng.randrange(0, 64)
        dst = ctypes.create_string_buffer(n + 16)
        sbuf = ctypes.create_string_buffer(src, len(src))
        L.flbgpu_diag_copy(ctypes.addressof(dst) + 3, ctypes.addressof(sbuf) + off, n)
        assert dst.raw[3:3 + n] == src[off:off + n] and dst.raw[:3] == b"\0\0\0" and dst.raw[3 + n:3 + n + 8] == b"\0" * 8, n


def test_threaded_slab_copy_two_callers(g):
    """two threads in the slab copy at once (every filter has its own stream: concurrent host-level calls are intended):
    the pool has ONE job slot, the second caller copies by itself -- every byte of both arrives"""
    import random, threading
    L = g.lib()
    L.flbgpu_diag_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    rng = random.Random(11)
    srcs = [bytes(rng.getrandbits(8) for _ in range(1 << 14)) * 200 for _ in range(2)]
    bad = []

    def work(i):
        r = random.Random(100 + i)
        sbuf = ctypes.create_string_buffer(srcs[i], len(srcs[i]))
        for _ in range(40):
            n = r.randrange(1 << 20, len(srcs[i]) - 64)
            off = r.randrange(0, 64)
            dst = ctypes.create_string_buffer(n + 16)
            L.flbgpu_diag_copy(ctypes.addressof(dst) + 5, ctypes.addressof(sbuf) + off, n)
            if not (dst.raw[5:5 + n] == srcs[i][off:off + n] and dst.raw[5 + n:5 + n + 8] == b"\0" * 8):
                bad.append((i, n))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not bad, bad[:4]
