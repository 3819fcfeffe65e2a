"""The second regex engine ON THE DEVICE (nfa_dev.inc: the bit-parallel walk over the character-level position automaton) and the
48 + 2 regular expressions of the reference's conf/parsers*.conf through the kernels.

* every stock regex as a filter_parser and as a filter_grep rule over texts drawn from the pattern itself (flbgpu_rx_sample) and
  damaged copies of them -- ill-formed UTF-8, cut lines, second lines --, byte for byte against the oracle (whose regex engine
  tests/test_stock_parsers.py pins on the real Onigmo over the same texts); `istio-envoy-proxy` and `http_statement` run on the NFA
  engine by themselves, and the whole list once more with FLBGPU_RX_FORCE_NFA=2 (no byte tables at all);
* the golden corpus of the real engine (tests/golden/regex_kat.json, 16.7 k answers) with the NFA engine behind every pattern;
* POSIX brackets with their Unicode members, (?a) / (?u) / (?d), on non-ASCII records."""
import base64, json, os, random, sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_binding as ob
import synth
import flbamd_loader
import test_stock_parsers as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


def _rec(body, sec=1, nsec=2):
    return synth.mp([[synth.ext_ts(sec, nsec), {}], body])


def first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return "byte %d: oracle %r gpu %r (len %d vs %d)" % (i, a[max(0, i - 20):i + 20], b[max(0, i - 20):i + 20], len(a), len(b))
    return "length %d vs %d" % (len(a), len(b))


def run_pattern(g, pat, subjects, parser=True):
    """filter_grep (Regex log <pat>) and, with named groups, filter_parser over one record per subject: device == oracle"""
    blob = b"".join(_rec({"log": s}, 7, i) for i, s in enumerate(subjects))
    rules = [("regex", b"log " + pat)]
    fo, fg = ob.Grep(rules), g.FilterGrep(rules)
    a, b = fo.filter(blob), fg.filter(blob)
    fg.close()
    assert a == b, (pat, first_diff(a[1], b[1]))
    kept = ob.count_records(a[1]) if a[0] == ob.MODIFIED else len(subjects)
    if parser and b"(?<" in pat:
        for skip_empty in (True, False):
            po = ob.Parser(pat, skip_empty=skip_empty)
            pg = g.Parser(pat, skip_empty=skip_empty)
            fpg = g.FilterParser("log", [pg])
            x, y = ob.FilterParser("log", [po]).filter(blob), fpg.filter(blob)
            fpg.close(); pg.close()
            assert x == y, (pat, skip_empty, first_diff(x[1], y[1]))
    return kept


@pytest.mark.parametrize("force", ["", "2"])
def test_stock_parsers_on_device(g, force, monkeypatch):
    if force:
        monkeypatch.setenv("FLBGPU_RX_FORCE_NFA", force)
    L = g.lib()
    total = kept = 0
    for k, it in enumerate(sp.stock()):
        pat = sp.inner(it["regex"])
        if pat.startswith(b"/"):
            continue
        subj = [s for s in sp.texts(L, pat, 48 if not force else 24, 4000 + k) if b"\x00" not in s]
        kept += run_pattern(g, pat, subj)
        total += len(subj)
    assert total > 1000 and kept > 0.25 * total, (total, kept)


@pytest.mark.parametrize("force", ["1", "2"])
def test_regex_kat_through_the_nfa_engine(g, force, monkeypatch):
    monkeypatch.setenv("FLBGPU_RX_FORCE_NFA", force)
    kat = json.load(open(os.path.join(HERE, "golden", "regex_kat.json")))
    n_pat = n_cases = 0
    for ent in kat:
        pat = base64.b64decode(ent["pattern"])
        if not ent["compiles"] or pat.startswith(b"/") or b"\x00" in pat:
            continue
        subjects = [base64.b64decode(c[0]) for c in ent["cases"]]
        want = {base64.b64decode(c[0]): c[1] for c in ent["cases"]}
        try:
            ob.Grep([("regex", b"log " + pat)])
            fg = g.FilterGrep([("regex", b"log " + pat)])
            fg.close()
        except ValueError:
            continue
        kept = run_pattern(g, pat, subjects, parser=bool(ent["names"]))
        assert kept == sum(1 for s in subjects if want[s] is not None), pat
        n_pat += 1
        n_cases += len(subjects)
    assert n_pat > 150 and n_cases > 12000, (n_pat, n_cases)


POSIX_PATTERNS = [
    rb"^(?<w>[[:alpha:]]+) (?<n>[[:digit:]]+)$",
    rb"(?<first>[[:upper:]][[:lower:]]*) (?<rest>.*)",
    rb"^(?<tok>[[:alnum:]_]+)(?<sep>[[:punct:]]+)(?<tail>[[:graph:]]*)",
    rb"(?<p>[[:print:]]+)$",
    rb"^(?<k>[[:word:]]+)=(?<v>[^[:space:]]*)",
    rb"(?a)^(?<w>[[:alpha:]]+)\b(?<r>.*)",
    rb"(?u)^(?<w>\w+)\s+(?<d>\d+)",
    rb"(?u:(?<w>\w+))-(?a:(?<x>\w+))",
    rb"^(?<a>(?a:\w+\b))(?<b>.*\b.)",
    rb"(?<x>.{0,40}x)$",
    rb"^(?<h>[[:xdigit:]]{2,})(?<s>[[:blank:]]*)(?<c>[[:cntrl:]]?)",
    # \p{..} with the POSIX bracket names (the NOT of an atom outside brackets is the class's flag)
    rb"^(?<w>\p{Alpha}+)(?<r>\P{Alpha}*)",
    rb"(?<u>\p{Upper}\p{Lower}*) (?<rest>[\p{Word}\P{ASCII}]*)",
    rb"^(?<t>\p{^Space}+)\p{Space}(?<d>\p{Digit}*)",
]
WORDS = ["abc", "Été", "Жук", "日本", "x", "naïve", "ǅ", "Ａ", "١٢٣", "42", "²", "_", "foo_bar", "\u212a", "ß", "«q»", "¡", "\u00a0", "\u3000", "0xFF", "dead", "-", "=",
         "😀", "e\u0301"]


def test_posix_brackets_and_group_options_on_device(g):
    rng = random.Random(5)
    subj = []
    for i in range(400):
        parts = [rng.choice(WORDS) for _ in range(rng.randint(1, 4))]
        s = rng.choice([" ", "  ", "=", "-", "\t"]).join(parts).encode()
        if i % 7 == 0:
            m = bytearray(s); q = rng.randrange(len(m) + 1); m[q:q] = rng.choice(sp.FRAG); s = bytes(m)
        subj.append(s)
    subj += [b"", b"abc 123", b"x" * 60, "Ünïcode 42".encode()]
    for pat in POSIX_PATTERNS:
        kept = run_pattern(g, pat, subj)
        assert kept > 0, pat


def test_regex_corners_are_counted_not_silent(g):
    """The two corners where the reference's own answer depends on its search optimizer (DESIGN.md "deviations"): a value that can
    meet one is COUNTED by the walkers (flbgpu_filter_regex_corners; the plugin shims warn) -- on the table engine and on the NFA
    engine, for filter_grep and filter_parser; clean values count nothing."""
    cases = [
        (rb"\bfoo", [b"x\x80foo", b"caf\xc3\xa9 foo", b"foo"], 1),              # a stray continuation byte and a word anchor
        (rb"^bar", [b"a\n\x80bar", b"a\nbar", b"\xc3\xa9\nbar"], 1),              # "\n" + continuation byte in front of a line anchor
        (rb"(?i)ks", [b"x\xe2\x84\xaas", b"xks", b"\xc3\xa9ks"], 1),            # the Kelvin sign under (?i)
        (rb"[[:alpha:]]+\b", [b"\xbfa b", b"\xc3\xa9a b"], 1),                    # the same through the NFA engine
        (rb"plain", [b"pl\x80ain", b"plain \xc3\xa9"], 0),                       # no anchor, no fold: nothing to count
    ]
    for pat, subjects, want in cases:
        blob = b"".join(_rec({"log": s}, 7, i) for i, s in enumerate(subjects))
        fg = g.FilterGrep([("regex", b"log " + pat)])
        fg.filter(blob)
        assert fg.regex_corners() == want, (pat, fg.regex_corners())
        fg.filter(blob)
        assert fg.regex_corners() == 2 * want                                  # cumulative
        fg.close()
        pg = g.Parser(b"(?<m>" + pat + b")")
        fp = g.FilterParser("log", [pg])
        fp.filter(blob)
        assert fp.regex_corners() == want, (pat, fp.regex_corners())
        fp.close(); pg.close()
