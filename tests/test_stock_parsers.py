"""Every regular expression the reference ships in conf/parsers*.conf (48 [PARSER] Regex lines + the [MULTILINE_PARSER] rules):
each must COMPILE on the product's regex front end -- through the byte tables, or through the bit-parallel NFA engine where those
give up (rx.hpp NfaSet) -- and answer like the real Onigmo (oracle/_ref/libonig_ref.so) on texts drawn from the pattern itself
(flbgpu_rx_sample), damaged copies of them (blanks, quotes, brackets, ill-formed UTF-8, cut lines, a second line behind a line
feed) and noise.  The list is not written by hand: the conf files are globbed when the reference is present, and the committed
fixture tests/golden/stock_parsers.json (tools/gen_stock_parsers.py; what the GPU box reads) must equal that parse.
The same texts run through the device kernels in tests/test_kat_gpu.py::test_stock_parsers_on_device."""
import ctypes, json, os, random, sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import flbamd_loader
import rxdiff

FIXTURE = os.path.join(HERE, "golden", "stock_parsers.json")
REF = "/root/reference"

FRAG = [b"\xe9", b"\xc3", b"\xa9", b"\xe2\x82", b"\xf0\x9f\x98", b"\xff", b"\xc0\x80", b"\xed\xa0\x80", b"\xc3\xa9", b"\xe2\x82\xac",
        b"\xf4\x90\x80\x80", b"\x80", b" ", b"  ", b'"', b'\\"', b"\\", b"]", b"[", b"\n", b":", b"-", b"\t", b"\xd0\x96", b"\xef\xbc\xa1",
        b"\xc2\xb2", b"\xe2\x84\xaa"]


def stock():
    return json.load(open(FIXTURE))


def inner(regex):
    """the pattern as flb_regex_create hands it to onig_new: /../ stripped (src/flb_regex.c:60-152; no stock pattern has flags)"""
    b = regex.encode("utf-8", "surrogateescape")
    if len(b) > 1 and b[:1] == b"/" and b[-1:] == b"/":
        b = b[1:-1]
    return b


def texts(L, pat, n, seed):
    """n texts for one pattern: samples, damaged samples, cut samples, two lines, noise"""
    rng = random.Random(seed)
    buf = ctypes.create_string_buffer(8192)
    out = []
    for i in range(n):
        k = L.flbgpu_rx_sample(pat, len(pat), 0, seed * 1000003 + i, buf, 8192)
        assert k >= 0
        s = buf.raw[:k]
        r = rng.random()
        if r < 0.35:
            pass
        elif r < 0.75:
            m = bytearray(s)
            for _ in range(rng.randint(1, 3)):
                q = rng.randrange(len(m) + 1)
                if rng.random() < 0.8:
                    m[q:q] = rng.choice(FRAG)
                elif q < len(m):
                    del m[q:q + rng.randint(1, 3)]
            s = bytes(m)
        elif r < 0.85:
            s = s[:rng.randrange(0, len(s) + 1)] + rng.choice(FRAG[:8])
        elif r < 0.93:
            s = s + b"\n" + s[:rng.randrange(0, len(s) + 1)]
        else:
            s = bytes(rng.choice(b" abc0159:-[]\"/.\n\xe9\xc3\xa9") for _ in range(rng.randint(0, 30)))
        out.append(s)
    out += [b"", b" ", b"\n"]
    return out


def compare_one(L, ref, pat, subjects, want_captures=1):
    eng = rxdiff.RefRegex(ref, pat)
    assert eng.ok, pat
    err = ctypes.create_string_buffer(512)
    h = L.flbgpu_rx_compile(pat, len(pat), 0, want_captures, err, 512)
    assert h, (pat, err.value)
    matched = 0
    try:
        for s in subjects:
            want = eng.search(s)
            if want_captures:
                beg = (ctypes.c_int * 64)(); end = (ctypes.c_int * 64)()
                n = L.flbgpu_rx_simulate_capture(h, s, len(s), beg, end)
                got = None if n == -1 else [(beg[i], end[i]) for i in range(n)]
                assert got == want, (pat, s, got, want)
            else:
                got = L.flbgpu_rx_simulate_match(h, s, len(s))
                assert got == (1 if want is not None else 0), (pat, s, got, want)
            matched += want is not None
        info = (ctypes.c_int * 6)()
        eng_bits = L.flbgpu_rx_engine(h, info, None, 0)
    finally:
        L.flbgpu_rx_free(h)
    return matched, eng_bits


@pytest.mark.skipif(rxdiff.load_ref() is None, reason="oracle/_ref/libonig_ref.so not built (needs /root/reference)")
def test_oracle_regex_on_the_stock_regexes():
    """the oracle's own engine (oracle/orx.c, what the -m gpu parity tests compare the device with) against the real one on the same texts"""
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    orx = rxdiff.load_orx()
    n = 0
    for k, it in enumerate(stock()):
        pat = inner(it["regex"])
        eng = rxdiff.RefRegex(ref, pat)
        o = rxdiff.OrxRegex(orx, pat)
        assert eng.ok and o.ok, (pat, o.err)
        for s in texts(L, pat, 40, 991 + k):
            assert o.search(s) == eng.search(s), (pat, s)
            n += 1
    assert n > 2000


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "conf")), reason="needs /root/reference")
def test_fixture_is_the_conf_files():
    import gen_stock_parsers
    live = json.loads(json.dumps(gen_stock_parsers.collect(REF)))
    assert live == stock()
    assert sum(1 for x in live if x["section"] == "PARSER") == 48


def test_every_stock_regex_compiles():
    L = flbamd_loader.load().lib()
    L.flbgpu_rx_compile.restype = ctypes.c_void_p
    engines = {}
    for it in stock():
        pat = inner(it["regex"])
        err = ctypes.create_string_buffer(512)
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 512)
        assert h, "%s:%d %s: %s" % (it["file"], it["line"], it["name"], err.value.decode())
        engines[it["name"]] = L.flbgpu_rx_engine(ctypes.c_void_p(h), None, None, 0)
        L.flbgpu_rx_free(ctypes.c_void_p(h))
    # the two the byte tables cannot hold take the second engine; the hot ones stay on the tables
    assert engines["istio-envoy-proxy"] == 3 and engines["http_statement"] & 1
    assert engines["apache2"] == 0 and engines["apache"] == 0 and engines["nginx"] == 0


@pytest.mark.skipif(rxdiff.load_ref() is None, reason="oracle/_ref/libonig_ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("force", ["", "1", "2"])
def test_stock_regexes_against_the_real_engine(force, monkeypatch):
    """force = "1": the NFA engine answers for every value with a byte >= 0x80, "2": for every value -- all 50 patterns through it"""
    if force:
        monkeypatch.setenv("FLBGPU_RX_FORCE_NFA", force)
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    total = matched = 0
    for k, it in enumerate(stock()):
        pat = inner(it["regex"])
        subj = texts(L, pat, 60 if not force else 40, 77 + k)
        m, bits = compare_one(L, ref, pat, subj)
        if force == "2":
            assert bits == 3
        total += len(subj); matched += m
        assert m >= 5, (it["name"], m)              # the sampler really produces lines the pattern takes
    assert total > 1500 and matched > 0.3 * total, (total, matched)
