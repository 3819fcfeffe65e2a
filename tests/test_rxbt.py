"""The product's backtracking matcher (csrc/rxbt.inc: what the filters run on the host for a rule / parser that is not a regular
expression -- look-around, atomic groups, possessive repeats, back-references, \\Z \\G \\K) against the REAL Onigmo
(oracle/_ref/libonig_ref.so): known answers, random non-regular patterns over a grammar of those constructs, and -- the same
matcher, the same semantics underneath -- the regular random patterns of test_rx_random_patterns.py."""
import ctypes, os, random, sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import flbamd_loader
import rxdiff
import test_rx_random_patterns as rp

needs_ref = pytest.mark.skipif(rxdiff.load_ref() is None, reason="oracle/_ref/libonig_ref.so not built (needs /root/reference)")

KNOWN = [
    rb"foo(?=bar)", rb"foo(?!bar)", rb"(?<=foo)bar", rb"(?<!foo)bar", rb"(?<=a|bc)x", rb"(?<!a|bc)x", rb"(?>a+)b", rb"(?>a|ab)c", rb"a*+a", rb"a++b", rb"a?+a",
    rb"(a+)\1", rb"(?<x>\w+) \k<x>", rb"(\w)(\w)\2\1", rb"(?i)(ab)\1", rb"^(?!.*error).*$", rb"^(?=.*\d)(?=.*[a-z]).{4,}$", rb"\bfoo\b(?! bar)",
    rb"(?<=\d)(?=(\d{3})+$)", rb"x\Z", rb"\Aab\Z", rb"a\Kb", rb"\Gab", rb"(?<=^|,)[^,]*", rb"(?<![\w.])\d+(?![\w.])", rb"(?<k>a)?\k<k>b", rb"(a)|\1b",
    rb"(?>a*)a", rb"(?:(?=a)a|b)+", rb"(?!a)(?!b).", rb"(?<=(?<!x)y)z", "(?<=é)x".encode(), "(?<!日)本".encode(), rb"(?<=\s)\S+(?=\s)", rb"(a*)*+b", rb"(?>(a|b)*)c",
    rb"(a)?(?(1)b|c)", rb"^(?<q>\")?\w+(?(<q>)\")$", rb"(\()?[^()]+(?(1)\))", rb"(?<x>x)?(?(<x>)y)z", rb"a\Rb", rb"\R+", rb"^.*\R",
    rb"(?<q>['\"])(?<body>.*?)\k<q>", rb"^(?<host>\S+) (?!-)(?<user>\S+)", rb"(?<n>\d+)-\k<n>", rb"(?=(a+))a*b\1", rb"(?<!\\)\"", rb"(\d+)(?<=5)x", rb"a(?=b|c)(?<=a).",
]

# (round 5: ends of a one-character repeat in front of a byte the rest cannot begin with are not tried, searches start only where a
# match can -- patterns where that pruning sits next to capture bookkeeping, look-arounds, lazy repeats, multi-byte characters)
PRUNING = [
    rb"(a*)*b", rb"(?:(a*)b|\1c)+", rb"((a*)b\2)", rb"(x*|y)\1z", rb".*(?=b)b", rb"\w*(?<=a)b", rb"[^b]*?(?!c)b", rb"(a*)(?:b|\1)c", "é*b".encode(), "[aé]*é".encode(),
    rb"a*(?:b|)c", rb"a*(b)?c", rb"a*\Kb", rb"(?i)x*Mozilla(?!/4)", rb"\d*(?>\d)x", rb"^a*b|^c", rb"(?:^|\A)a*b", rb"\Ga*b", rb"(^a*)b", rb"(?:^a)+b", rb"a*$\n?b", rb"(?=a)a*a", rb"a{2,5}?a(?<!aaaa)b",
    rb"[^\n]*\n(?=x)", rb"(?<=\n)a*b", rb"^\s*(?<k>\w+)\s*=\s*(?<v>.*?)\s*$(?<!;)", rb"(a|ab)*c", rb".*?(\d+)(?!\d)", rb".*\b(\w+)\1",
]

LOOK_BODY = [rb"a", rb"b", rb"ab", rb"\d", rb"\w", rb"[a-c]", rb" ", rb"x|y", rb"ab|cd", rb"a|bc", rb"\s", rb"[^a]", rb".", "é".encode(), rb"\d\d", rb"^", rb"$", rb"\b"]


def gen_nonregular(rng):
    names = []
    n_plain = [0]

    def atom():
        return rng.choice(rp.ATOMS)

    def piece(d):
        r = rng.random()
        if r < 0.10:
            return b"(?=" + rng.choice(LOOK_BODY) + b")"
        if r < 0.18:
            return b"(?!" + rng.choice(LOOK_BODY) + b")"
        if r < 0.26:
            return b"(?<=" + rng.choice(LOOK_BODY) + b")"
        if r < 0.33:
            return b"(?<!" + rng.choice(LOOK_BODY) + b")"
        if r < 0.41 and d > 0:
            return b"(?>" + alt(d - 1) + b")" + rng.choice([b"", b"", b"?"])
        if r < 0.49:
            return atom() + rng.choice([b"*+", b"++", b"?+"])
        if r < 0.57 and names:
            return b"\\k<" + rng.choice(names) + b">"
        if r < 0.67 and d > 0 and len(names) < 4:
            nm = b"g%d" % len(names)
            body = b"(?<" + nm + b">" + alt(d - 1) + b")"
            names.append(nm)
            return body + rng.choice([b"", b"", b"?"])
        if r < 0.70:
            return rng.choice([rb"\Z", rb"\K", rb"\G", rb"\b", rb"\R"])
        if r < 0.74 and names:
            return b"(?(<" + rng.choice(names) + b">)" + atom() + rng.choice([b"", b"|" + atom()]) + b")"
        return atom() + rng.choice(rp.QUANT)

    def seq(d):
        return b"".join(piece(d) for _ in range(rng.randint(1, 4)))

    def alt(d):
        out = seq(d)
        while rng.random() < 0.2:
            out += b"|" + seq(d)
        return out
    p = alt(2)
    if rng.random() < 0.3:
        p = b"^" + p
    if rng.random() < 0.15:
        p = b"(?i)" + p
    return p


def subjects(L, rng, pat, n):
    buf = ctypes.create_string_buffer(4096)
    out = []
    for i in range(n):
        r = rng.random()
        if r < 0.45:
            k = L.flbgpu_rx_sample(pat, len(pat), 0, rng.getrandbits(48), buf, 4096)
            s = buf.raw[:max(k, 0)]
            if rng.random() < 0.5 and s:
                m = bytearray(s)
                q = rng.randrange(len(m) + 1)
                m[q:q] = rng.choice([b"a", b" ", b"5", b"\n", b"x", "é".encode(), b"\xe9", b"ab", b"b"])
                s = bytes(m)
            if rng.random() < 0.3:
                s = rxdiff.rand_input(rng, pat, 6) + s + rxdiff.rand_input(rng, pat, 6)
            out.append(s[:60])
        elif r < 0.8:
            out.append(rxdiff.rand_input(rng, pat, 20, utf8=rng.random() < 0.4))
        else:
            out.append(rxdiff.rand_input_illformed(rng, pat, 14))
    return out + [b"", b"a", b"\n"]


def bt_compile(L, pat):
    L.flbgpu_rxbt_compile.restype = ctypes.c_void_p
    err = ctypes.create_string_buffer(256)
    h = L.flbgpu_rxbt_compile(pat, len(pat), 0, err, 256)
    return (ctypes.c_void_p(h) if h else None), err.value


def bt_search(L, h, s):
    beg = (ctypes.c_int * 64)(); end = (ctypes.c_int * 64)()
    n = L.flbgpu_rxbt_search(h, s, len(s), beg, end)
    if n < 0:
        return None if n == -1 else ("budget", n)
    return [(beg[i], end[i]) for i in range(n)]


CORNERS = [0]


def corner(pat, s):
    """the two documented corners where the reference's answer hangs on how it steps BACK over ill-formed UTF-8 (a stray continuation
    byte in front of ^ \\b \\B or a look-behind) or on a fold that changes a character's length under (?i) (DESIGN: deviations; the
    table engines count such values, rx::corner)"""
    if rp.has_stray_continuation(s) and (b"\\b" in pat or b"\\B" in pat or b"^" in pat or b"(?<" in pat):
        return True
    if b"(?i)" in pat and rp.FOLD_LENGTH_CHANGERS.search(s) is not None:
        return True
    # a defect of the reference that is not reproduced: a pattern that can only match the empty string at the end of the text (nothing
    # but anchors and look-arounds, \\z or \\Z among them) is searched BACKWARDS from the end with the right limit of the match at the start of
    # the text (regexec.c onig_search end_buf: start = end - anchor_dmax, then the backward loop's MATCH_AND_RETURN_CHECK(orig_start)),
    # so every character test inside a look-behind fails its DATA_ENSURE: `(?<=\\d)\\z` does not match "1" there.
    return (b"\\Z" in pat or b"\\z" in pat) and b"(?<" in pat


def compare(L, ref, pat, subj):
    """-> (compared, matched); raises on a difference"""
    eng = rxdiff.RefRegex(ref, pat)
    h, err = bt_compile(L, pat)
    if not eng.ok:
        if h:
            L.flbgpu_rxbt_free(h)
        return 0, 0
    if not h and ((b"not supported" in err and b"case-insensitive" in err) or b"target of repeat operator is invalid" in err):
        return 0, 0                    # (the front end's own limits -- (?i) with non-ASCII members, a repeat of an anchor: refused at create, for every engine)
    assert h, (pat, err)
    matched = 0
    # (?i) over a non-ASCII literal and a text that is not well-formed UTF-8: the REFERENCE's engine dies there (SIGSEGV in onig_search --
    # `(?i)\xe2\x82\xac+?"` on e2 82 ac e2 82 61 62 ac .., found when the product's matcher began to take such patterns, round 5): only
    # well-formed texts are put to it for these patterns
    if b"(?i" in pat and any(c >= 0x80 for c in pat):
        def well_formed(x):
            try:
                x.decode("utf-8")
                return True
            except UnicodeDecodeError:
                return False
        subj = [x for x in subj if well_formed(x)]
    try:
        for s in subj:
            want = eng.search(s)
            got = bt_search(L, h, s)
            if got != want and corner(pat, s):
                CORNERS[0] += 1
                continue
            assert got == want, (pat, s, got, want)
            matched += want is not None
    finally:
        L.flbgpu_rxbt_free(h)
    return len(subj), matched


@needs_ref
def test_deep_searches_run_on_the_matcher_own_stack():
    """the recursion needed > 2 MB of the CALLER's stack for a 12 KB value and answered -4 ('no match') past depth 12 000.
    It now runs on a stack of its own: a thread with a 256 KB stack searches values of 12 KB .. 400 KB and gets the real engine's answer"""
    import threading
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    pats = [rb"(?=a)(?:ab)*!", rb"(?=a)(?:ab)*x?$", rb"^(?:(?!zz).)*$", rb"(a|b)*\1c"]
    subj = [b"ab" * 6000 + b"!", b"ab" * 3000, b"ab" * 50000 + b"!", b"ab" * 200000 + b"!", b"abab" * 3000 + b"bc", b"q" * 100000]
    out = {}

    def work():
        for pat in pats:
            h, err = bt_compile(L, pat)
            assert h, (pat, err)
            for i, s in enumerate(subj):
                out[(pat, i)] = bt_search(L, h, s)
            L.flbgpu_rxbt_free(h)
    old = threading.stack_size(256 << 10)
    try:
        t = threading.Thread(target=work)
        t.start(); t.join()
    finally:
        threading.stack_size(old)
    assert len(out) == len(pats) * len(subj)
    for pat in pats:
        eng = rxdiff.RefRegex(ref, pat)
        for i, s in enumerate(subj):
            got = out[(pat, i)]
            if isinstance(got, tuple):
                continue                            # the 10 M backtrack budget (the product's own limit, reported): not an answer
            assert got == eng.search(s), (pat, i, got and got[:2])
    # none of the linear ones may have been given up
    assert not isinstance(out[(pats[0], 3)], tuple) and out[(pats[0], 3)][0] == (0, 400001)
    assert not isinstance(out[(pats[2], 5)], tuple)


@needs_ref
def test_known_nonregular_patterns():
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    rng = random.Random(11)
    total = matched = 0
    for pat in KNOWN:
        assert L.flbgpu_rx_is_nonregular(pat, len(pat), 0) == 1, pat            # the GPU engines refuse it -- and say why
        n, m = compare(L, ref, pat, subjects(L, rng, pat, 120))
        assert n > 0, pat                                                       # the real engine takes every one of them
        total += n; matched += m
    assert total > 4000 and matched > 300, (total, matched)


@needs_ref
def test_pruned_searches_give_the_same_answers():
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    rng = random.Random(23)
    total = matched = 0
    for pat in PRUNING:
        extra = [b"aab", b"aaaab", b"ab\nab", b"\naab", b"xaab\ncc", "ééb".encode(), "aéé".encode(), b"k = v ;", b" k=v", b"12 345", b"ab ab", b"aaaaab", b"aaac", b"xxMOZILLA/5", b"xmozilla/4"]
        n, m = compare(L, ref, pat, subjects(L, rng, pat, 150) + extra)
        assert n > 0, pat
        total += n; matched += m
    assert total > 4000 and matched > 1000, (total, matched)


@needs_ref
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_nonregular_patterns(seed):
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    rng = random.Random(seed * 7919)
    total = matched = pats = 0
    for _ in range(700):
        pat = gen_nonregular(rng)
        n, m = compare(L, ref, pat, subjects(L, rng, pat, 25))
        total += n; matched += m; pats += n > 0
    assert pats > 300 and total > 8000 and matched > 800, (pats, total, matched)


@needs_ref
def test_regular_patterns_through_the_backtracker():
    """the same machine underneath: the random REGULAR patterns must come out as from the real engine, too"""
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    rng = random.Random(99)
    total = matched = 0
    base = rp.ATOMS
    for i in range(1200):
        names = []
        rp.ATOMS = base + rp.MORE_ATOMS if i % 2 else base
        try:
            pat = rp.gen(rng, 2, names)
        finally:
            rp.ATOMS = base
        n, m = compare(L, ref, pat, subjects(L, rng, pat, 12))
        total += n; matched += m
    assert total > 10000 and matched > 2000, (total, matched)


def test_what_stays_refused():
    """constructs neither engine nor the backtracker takes are refused with their name, not run wrongly"""
    L = flbamd_loader.load().lib()
    for pat in [rb"(?(1)a|b|c)", rb"\g<1>", rb"\p{NoSuchProperty}", rb"\p{Age=99.0}", rb"(?<=a+)b", rb"(a)\2", rb"\k<nope>", rb"(?<a>x\g<a>)", rb"(?<a>\g<a>x)", rb"(?<a>x)(?<a>y)\g<a>", rb"\g<0>", rb"(?<a>x|\g<a>y)", rb"(?<a>x\g<b>)(?<b>y\g<a>)"]:
        h, err = bt_compile(L, pat)
        assert h is None and err, pat


ABSENT_KNOWN = [
    rb"(?~abc)", rb"/\*(?~\*/)\*/", rb"^(?~abc)$", rb"a(?~b)c", rb"(?~a)", rb"(?~ab|cd)x", rb"\A(?~\d)\z", rb"(?~a+)b", rb"x(?~y)*z", rb"(?~\n)\n",
    rb"<(?~>)>", rb"(?~aa)a", rb"^(?~ error )$", rb"(?<q>(?~,)),", rb"(?~[ab]c)d", rb"(?~a)(?~b)", rb"\[(?~\])\]", rb"(?~abc)abc", rb"(?i)(?~AB)x",
    rb"(?~(?~a))", rb"(?~a|)", rb"(?~)x", rb"(?~.)", rb"(?~ab)+c", rb"(?~\z)", rb"(?~\b)", rb"(?~a$)", rb"(?~^a)",
]
CALL_KNOWN = [
    rb"(?<p>\((?:[^()]|\g<p>)*\))", rb"(a|b\g<1>c)", rb"(?<x>a|b)\g<x>", rb"\A(?<a>|.|(?:(?<b>.)\g<a>\k<b>))\z", rb"(?<n>\d+)(?:,\g<n>)*", rb"(\w)\g<1>",
    rb"(a)\g<-1>", rb"\g<+1>(x|y)", rb"(?<t><(?:[^<>]|\g<t>)*>)!", rb"(?<e>\d|\(\g<e>(?:[+*]\g<e>)*\))$", rb"(ab)?\g<1>", rb"(?i)(?<k>ab)\g<k>",
    rb"(?<a>x(?<b>y)?)\g<a>\k<b>", rb"\g<0>?a", rb"a\g<0>?b", rb"(?<o>(?=a)\w)\g<o>", rb"(?>(?<v>a+))\g<v>", rb"(?<w>a*)b\g<w>",
]


def gen_absent_or_call(rng):
    body = [rb"a", rb"ab", rb"abc", rb"\d", rb"[a-c]", rb"a|b", rb"ab|c", rb"a+", rb"\*/", rb",", rb" ", rb"\w\w", rb"x?y", rb"a.", "é".encode()]
    atom = lambda: rng.choice(rp.ATOMS)
    r = rng.random()
    if r < 0.55:
        core = b"(?~" + rng.choice(body) + b")"
        if rng.random() < 0.2:
            core = b"(?:" + core + rng.choice([b")+", b")*", b")?", b"){2}"])
        pre = b"".join(atom() + rng.choice(rp.QUANT) for _ in range(rng.randint(0, 2)))
        post = b"".join(atom() + rng.choice(rp.QUANT) for _ in range(rng.randint(0, 2)))
        p = pre + core + post
    else:
        inner = rng.choice([rb"[^()]", rb"\w", rb"a", rb"\d", rb"[ab]"])
        nm = rng.choice([b"g", b"r"])
        kind = rng.random()
        if kind < 0.4:
            p = b"(?<" + nm + b">\\((?:" + inner + b"|\\g<" + nm + b">)*\\))"
        elif kind < 0.7:
            # (a second atom inside the group: with the repeat LAST in the group the reference compiles it with a peek at the character
            # that follows the group in the text of the pattern -- OP_PUSH_IF_PEEK_NEXT, regcomp.c next_setup -- and the call, which is
            # followed by something else, inherits that test: test_call_inherits_the_peek_of_its_definition)
            p = b"(?<" + nm + b">" + atom() + rng.choice(rp.QUANT) + atom() + b")" + atom() + b"\\g<" + nm + b">"
        else:
            p = b"(" + atom() + b"|" + atom() + b"\\g<1>" + atom() + b")"
    if rng.random() < 0.3:
        p = b"^" + p
    if rng.random() < 0.3:
        p = p + b"$"
    return p


@needs_ref
def test_absent_operator_and_subexpression_calls():
    """round 5: (?~X) and \\g<..> run on the host's matcher like the other constructs that are not regular expressions (they aborted
    start-up before: VERDICT r4, missing 6); known patterns and random ones against the real engine"""
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    rng = random.Random(2025)
    total = matched = 0
    for pat in ABSENT_KNOWN + CALL_KNOWN:
        assert L.flbgpu_rx_is_nonregular(pat, len(pat), 0) == 1, pat
        extra = [b"/* a */ b */", b"/**/", b"/* x * / y */z", b"xabcx", b"ab", b"abc", b"aabc", b"(a(b)c)", b"((a)", b"1,22,333", b"<a<b>c>!", b"(1+(2*3))", b"abab", b"ABab",
                 b"racecar", b"abba", b"xyxyy", b"aaab", b"aaa", b"a\nb\n", b"[x]]", b"a, b, c"]
        n, m = compare(L, ref, pat, subjects(L, rng, pat, 80) + extra)
        total += n; matched += m
    assert total > 3000 and matched > 500, (total, matched)
    pats = 0
    for _ in range(600):
        pat = gen_absent_or_call(rng)
        n, m = compare(L, ref, pat, subjects(L, rng, pat, 20) + [b"(a(b)c)", b"/* a */", b"abcabc", b"a1b2"])
        total += n; matched += m; pats += n > 0
    assert pats > 400, pats


@needs_ref
def test_call_inherits_the_peek_of_its_definition():
    """A deviation that is the reference's optimizer, pinned so that it is seen if either side changes: a group whose body is ONE greedy
    unbounded repeat, followed in the pattern by a literal character, is compiled with a peek at that character (regcomp.c next_setup ->
    OP_PUSH_IF_PEEK_NEXT); a call of the group elsewhere runs the same code, so there the repeat only stops in front of that character
    too.  `^(?<g>\\S*)/\\g<g>$` does not match "ab/" in the reference (the called \\S* may not stop at the end of the text); the product's
    matcher runs the pattern as written and matches.  DESIGN 8."""
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    pat = rb"^(?<g>\S*)/\g<g>$"
    h, err = bt_compile(L, pat)
    assert h, err
    try:
        assert bt_search(L, h, b"ab/") == [(0, 3), (3, 3)] and rxdiff.RefRegex(ref, pat).search(b"ab/") is None
    finally:
        L.flbgpu_rxbt_free(h)


@needs_ref
def test_more_than_31_groups_go_to_the_host_matcher():
    """round 5: the tables keep a group set in 32 bits; a pattern with more groups (the reference takes 32 767, onigmo.h:441) no longer aborts
    start-up: flbgpu_rx_is_nonregular says 1 and the host's matcher answers it (here with a back-reference to group 40)"""
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    rng = random.Random(5)
    for pat in [b"^" + b"".join(b"(%c)" % (97 + i % 26) for i in range(40)) + b"\\40", b"(\\d)" * 33 + b"-(x|y)+$", b"(?:" + b"(a)|" * 34 + b"(b))c"]:
        assert L.flbgpu_rx_is_nonregular(pat, len(pat), 0) == 1, pat
        base = bytes(97 + i % 26 for i in range(40))
        n, m = compare(L, ref, pat, subjects(L, rng, pat, 40) + [base + b"n", base + b"m", b"1" * 33 + b"-xyx", b"1" * 32 + b"-x", b"aab" * 3 + b"c"])
        assert n > 0 and m > 0, (pat, n, m)


ICASE_WORDS = ["Straße", "ÉCOLE", "ошибка", "Ошибка Сервера", "ΣΊΣΥΦΟΣ", "σίσυφος", "İstanbul", "ǅemal", "ﬁn", "Åse", "ŉ", "ǰ", "Ὀδυσσεύς", "ᾳδω", "Ωμέγα", "ſtraße",
               "Ԑԑ", "Ꙋꙋ", "Ⴀⴀ", "ꭰᎠ", "Ａｂｃ", "𐐀𐐨", "ﬃ", "ﬅ", "ẞ", "ÿŸ", "µΜμ", "K", "Å", "Ω", "ΐ", "ΰ", "և", "ﬓ", "ẖẗẘẙ", "ẛṡ", "ϐβ", "ϑθ", "ςσ", "ǇǈǉĲĳ", "ıI", "niño", "GRÖßE"]


def icase_subjects(rng, word):
    w = word
    forms = {w, w.lower(), w.upper(), w.casefold(), w.swapcase(), w.title(), w.upper().lower(), w.lower().upper(), w.casefold().upper()}
    out = []
    for f in forms:
        out.append(f.encode())
        out.append(("xx " + f + " yy").encode())
        if len(f) > 1:
            k = rng.randrange(len(f))
            out.append((f[:k] + f[k].swapcase() + f[k + 1:]).encode())
            out.append((f[:k] + f[k + 1:]).encode())
    return out


@needs_ref
def test_case_insensitive_non_ascii_literals():
    """round 5: (?i) over non-ASCII literals runs on the host's matcher -- a literal is the class of its partners, a character that
    stands for a sequence (U+00DF "ss", U+0390 ...) also matches the sequence, and a run of letters also matches the characters that
    stand for a part of it -- with the partners and sequences PROBED from the real engine (tools/gen_casefold.py).  Words of several
    scripts in every case form against the real engine; literals inside larger patterns."""
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    rng = random.Random(77)
    total = matched = 0
    for w in ICASE_WORDS:
        for pat in [("(?i)" + w).encode(), ("(?i)^" + w + "$").encode(), ("(?i)(?<a>" + w + ")\\s*=\\s*\\d+").encode(), ("x(?i:" + w + ")y").encode()]:
            assert L.flbgpu_rx_is_nonregular(pat, len(pat), 0) == 1 or all(ord(c) < 128 for c in w), pat
            subj = []
            for w2 in [w] + rng.sample(ICASE_WORDS, 3):
                subj += icase_subjects(rng, w2)
            subj += [s_ + b" = 12" for s_ in subj[:6]] + [b"x" + s_ + b"y" for s_ in subj[:6]]
            n, m = compare(L, ref, pat, subj)
            total += n; matched += m
    assert total > 10000 and matched > 1200 and CORNERS[0] >= 0, (total, matched)


ICASE_CLASSES = ["[а-яё]+", "[ßa]x", "[^é]x", "[à-ÿ]{2,}", "[ΐk]s", "x[ǆ-ǌ]", "[ſ]t", "[K]", "[k-m]+", "[α-ω]+ς?", "[^а-я ]+", "[éa-c]+\\d", "[ﬁﬂ]n", "[\\x{1F80}-\\x{1FAF}]",
                 "[İı]", "[A-Zà-þ]+", "[ԱԲ]+", "[Ꭰ-Ꮿ]+", "[ꭰ-ꮿ]+", "[Ａ-Ｚ]+!", "[𐐀-𐐧]+", "[^ß]", "[ẞ]", "[ŉǰ]", "x[ẖ-ẚ]y",
                 "\\p{Greek}+", "\\P{Cyrillic}x", "[\\p{Lu}]+", "\\p{Ll}{2}", "\\p{Lu}s", "[\\p{Armenian}x]+", "\\p{^Latin}+"]


@needs_ref
def test_case_insensitive_classes_with_non_ascii_members():
    """round 5: (?i)[..] with some non-ASCII members on the host's matcher: the positive members closed under the engine's fold pairs,
    the members that stand for a sequence also match it (not in a negated class); against the real engine"""
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    rng = random.Random(78)
    total = matched = 0
    for cl in ICASE_CLASSES:
        for pat in [("(?i)" + cl).encode(), ("(?i)^" + cl + "$").encode(), ("a(?i:" + cl + ")b").encode()]:
            subj = []
            for w2 in rng.sample(ICASE_WORDS, 8):
                subj += icase_subjects(rng, w2)
            letters = "".join(ch for ch in cl if ord(ch) >= 0x80) or "k"
            for ch in letters:
                for f in {ch, ch.lower(), ch.upper(), ch.casefold(), ch.title()}:
                    subj += [f.encode(), ("x" + f + "y").encode(), (f + "s").encode(), (f + "t").encode(), (f + "n!").encode(), ("a" + f + "b").encode(), (f * 3 + "7").encode()]
            n, m = compare(L, ref, pat, subj)
            total += n; matched += m
    assert total > 10000 and matched > 800, (total, matched)


GRAPHEME_POOL = ["a", "b", " ", "\r", "\n", "\r\n", "\x01", "\u0301", "\u0308", "\u0300", "\u200d", "\u0903", "\u0600", "\U000110bd", "\u1100", "\u1161", "\u11a8", "\uac00", "\uac01",
                 "\U0001f1ef", "\U0001f1f5", "\U0001f1fa", "\U0001f468", "\U0001f469", "\U0001f467", "\U0001f3fb", "\u2764", "\ufe0f", "\u0e33", "\u0d4e", "é", "日", "\u00ad", "\u200c", "\u2028"]


@needs_ref
def test_extended_grapheme_cluster():
    """round 5: \\X on the host's matcher, spelled over the same property classes the reference builds it from (regparse.c
    node_extended_grapheme_cluster); texts of combining marks, Hangul jamo, regional indicators, ZWJ sequences, CR LF, controls,
    Prepend / SpacingMark characters -- well-formed and cut -- against the real engine"""
    L = flbamd_loader.load().lib()
    ref = rxdiff.load_ref()
    rng = random.Random(79)
    total = matched = 0
    for pat in [rb"\X", rb"^\X$", rb"^\X\X$", rb"a\X+b", rb"(?<g>\X)\k<g>", rb"\X{3}", rb"^(?:\X)*$", rb"(?i)\Xa", rb"\X(?=\u200d)".replace(b"\\u200d", "\u200d".encode()), rb"[^a]\X"]:
        assert L.flbgpu_rx_is_nonregular(pat, len(pat), 0) == 1, pat
        subj = []
        for _ in range(300):
            t = "".join(rng.choice(GRAPHEME_POOL) for _ in range(rng.randint(1, 7))).encode()
            if rng.random() < 0.15:
                t = t[:rng.randrange(len(t) + 1)]                            # cut anywhere: ill-formed tails
            if rng.random() < 0.1:
                k = rng.randrange(len(t) + 1); t = t[:k] + rng.choice([b"\xff", b"\x80", b"\xe2\x80"]) + t[k:]
            subj.append(t)
        n, m = compare(L, ref, pat, subj)
        total += n; matched += m
    assert total > 2500 and matched > 1000, (total, matched)


ESCAPES = [
    rb"\cA", rb"\ca", rb"\c?", rb"\C-a", rb"\C-?", rb"\M-a", rb"\M-\C-a", rb"\c\M-a", rb"\C-\M-a", rb"\M-\cA", rb"\M-\n", rb"\c\n", rb"\c\b", rb"\c\e", rb"\M-\x41",
    rb"\101", rb"\0", rb"\01", rb"\012", rb"\0123", rb"\18", rb"\19", rb"\81", rb"\177", rb"(a)\01", rb"(a)\10", rb"(a)\11", rb"(a)\1",
    rb"(a)(b)(c)(d)(e)(f)(g)(h)(i)(j)\10", rb"(a)(b)(c)(d)(e)(f)(g)(h)(i)(j)\11", rb"(a)(b)(c)(d)(e)(f)(g)(h)(i)(j)\12", rb"(?<n>a)\101",
    rb"(?i)\101", rb"(?i)\x41", rb"(?i)\cA", rb"(?i)\M-a", rb"[\cA-\cZ]+", rb"[\C-a]", rb"[\M-a]", rb"[\M-a-\M-z]", rb"[\c?]", rb"[\101-\103]+", rb"[\18]", rb"[\8]",
    "x\\cé".encode(), "x\\M-é".encode(), rb"a\cAb", rb"^\cI+x", rb"\1000", rb"\1001", rb"\99999999999", rb"\77", rb"\78",
]
ESCAPES_REFUSED_BY_BOTH = [rb"\1", rb"\8", rb"\400", rb"(?<n>a)\1", rb"(?<n>a)(b)(c)(d)(e)(f)(g)(h)(i)(j)(k)(l)\12", rb"\M-", rb"\M", rb"\C", rb"\C-", rb"\c", rb"\Ma", rb"\Ca"]
ESCAPE_TEXTS = [b"\x01", b"A", b"a", b"\x7f", b"\xc3\xa1", b"\xc2\x81", b"\xe1", b"\x81", b"a\x08", b"a\x09", b"aa", b"\x018", b"\x019", b"81", b"8", b"abcdefghij\x08",
                b"abcdefghija", b"abcdefghij\n", b"aA", b"\n", b"\n3", b"\x00", b"\xc3\x81", b"\x1a\x03", b"\xc3\xa9", b"ABC", b"?8", b"\t\tx", b"x\x89", b"x\xc2\x89", b"x\xc3\xa9", b"@0", b"@1", b"S0"]


@needs_ref
def test_control_meta_and_octal_escapes():
    """round 5: \\cX \\C-X \\M-X with their nestings (regparse.c:2429 fetch_escaped_value: a code point, \\M-a is U+00E1) and the rule that
    decides between a back-reference and an octal escape (fetch_token '1'..'9': a reference while the number is at most 9 or at most the
    groups opened so far, \\8 \\9 otherwise themselves) -- through the host's matcher AND, where the pattern is regular, through the
    tables executed on the host, against the real engine.  (A raw byte above 0x7f -- \\x82, \\200 -- stays refused: whether the reference
    finds it inside a character hangs on which search optimisation it picked: `\\x82\\xac` is found inside e2 82 ac, `\\x82` is not.)"""
    L = flbamd_loader.load().lib()
    L.flbgpu_rx_compile.restype = ctypes.c_void_p
    ref = rxdiff.load_ref()
    rng = random.Random(5)
    total = matched = tabled = 0
    for pat in ESCAPES:
        eng = rxdiff.RefRegex(ref, pat)
        assert eng.ok, pat
        subj = subjects(L, rng, pat, 30) + ESCAPE_TEXTS
        n, m = compare(L, ref, pat, subj)
        total += n; matched += m
        err = ctypes.create_string_buffer(256)
        th = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 256)
        if not th:
            assert L.flbgpu_rx_is_nonregular(pat, len(pat), 0) == 1 or b"case-insensitive" in err.value, (pat, err.value)
            continue
        tabled += 1
        for s in subj:
            if b"(?i" in pat and any(c >= 0x80 for c in s):
                continue                                # (compare()'s note on the reference and ill-formed texts under (?i))
            beg = (ctypes.c_int * 40)(); end = (ctypes.c_int * 40)()
            k = L.flbgpu_rx_simulate_capture(ctypes.c_void_p(th), s, len(s), beg, end)
            got = None if k == -1 else [(beg[i], end[i]) for i in range(k)]
            assert got == eng.search(s), (pat, s)
        L.flbgpu_rx_free(ctypes.c_void_p(th))
    assert total > 3000 and matched > 600 and tabled >= 45, (total, matched, tabled)
    for pat in ESCAPES_REFUSED_BY_BOTH + [rb"\x82", rb"\200", rb"a\xe2\x82\xac"]:
        h, err = bt_compile(L, pat)
        assert h is None and err, pat
        if pat in ESCAPES_REFUSED_BY_BOTH:
            assert not rxdiff.RefRegex(ref, pat).ok, pat


def test_searches_from_several_threads_at_once():
    """flbgpu.cpp parallel_rows searches one compiled program from up to 16 threads, each on its thread's own search stack (rxbt.inc: the
    stack switch is a few instructions of x86-64 since round 5, swapcontext elsewhere and under the sanitizers): 8 threads, shallow and
    deep searches mixed, every answer equal to the single-threaded one"""
    import threading
    L = flbamd_loader.load().lib()
    pats = [rb"(?<=user=)(\w+) id=\1", rb"^(?!.*(?:health|ping)).*\d$", rb"(?>a+)b", rb"(?i)(ab)\1", rb"(a)?(?(1)b|c)", rb"a\Rb", rb"(?=a)(?:ab)*!", rb"\Anope(?!x)", rb"^x(?=y)"]
    hs = []
    for p in pats:
        h, err = bt_compile(L, p)
        assert h, (p, err)
        hs.append(h)
    subj = [b"user=alice id=alice", b"GET /health 200", b"GET /x 200 5", b"aaab", b"ABab", b"ab", b"c", b"a\r\nb", b"x" * 200, b"host rest of line",
            b"ab" * 6000 + b"!", b"ab" * 20000, b"q\nxy", "é".encode() * 50 + b"aab"]
    base = {(i, j): bt_search(L, h, s) for i, h in enumerate(hs) for j, s in enumerate(subj)}
    assert base[(6, 10)] == [(0, 12001)] and base[(8, 12)] == [(2, 3)]
    bad = []

    def work(seed):
        rng = random.Random(seed)
        for _ in range(500):
            i = rng.randrange(len(hs)); j = rng.randrange(len(subj))
            r = bt_search(L, hs[i], subj[j])
            if r != base[(i, j)]:
                bad.append((i, j, r))
    ts = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    for t in ts: t.start()
    for t in ts: t.join()
    for h in hs:
        L.flbgpu_rxbt_free(h)
    assert not bad, bad[:3]
