"""The compact forward tables of the single-pass tile kernel (fluent-bit_amd/csrc/fx.cpp: one column per byte class,
MATCH / dead ends / multi-candidate cells as plain steps into an absorbing row, the end of the text as a sentinel
byte) executed on the host with the kernel's rules, against the answers of the real engine
(tests/golden/regex_kat.json, generated from the reference's Onigmo).  Whenever the walk settles a text its spans must
be the engine's; it may decline (the kernel then runs the reverse pass + classic walk), but not on the plain matching
ASCII lines the tables exist for."""
import base64
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import flbamd_loader


def _lib():
    L = flbamd_loader.load().lib()
    L.flbgpu_rx_compile.restype = ctypes.c_void_p
    L.flbgpu_rx_compile.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    L.flbgpu_rx_simulate_fx.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.flbgpu_rx_free.argtypes = [ctypes.c_void_p]
    return L


def test_fx_tables_against_golden():
    L = _lib()
    kat = json.load(open(os.path.join(HERE, "golden", "regex_kat.json")))
    settled = declined_match = matches_at0 = patterns = 0
    for ent in kat:
        pat = base64.b64decode(ent["pattern"])
        if not ent["compiles"] or not ent["names"] or not (pat.startswith(b"^") or pat.startswith(b"\\A")):
            continue
        err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 256)
        if not h:
            continue
        named = sorted({g for _, g in ent["names"]})
        used = False
        for s64, want in ent["cases"]:
            s = base64.b64decode(s64)
            beg = (ctypes.c_int * 40)(); end = (ctypes.c_int * 40)()
            n = L.flbgpu_rx_simulate_fx(h, s, len(s), beg, end)
            if n == -4:
                break
            used = True
            ascii_only = all(c < 0x80 for c in s)
            if want is not None and want[0][0] == 0 and ascii_only:
                matches_at0 += 1
                if n < 0:
                    declined_match += 1
            if n >= 0:
                # the walk settled the text: a match that starts at 0, with the engine's spans for the named groups
                assert want is not None and want[0] == [0, end[0]], (pat, s, want, end[0])
                for g in named:
                    assert [beg[g], end[g]] == want[g], (pat, s, g, [beg[g], end[g]], want[g])
                settled += 1
        patterns += 1 if used else 0
        L.flbgpu_rx_free(h)
    assert patterns >= 10 and settled > 50, (patterns, settled)
    # (the corpus is full of deliberately ambiguous toy patterns: declining is the expected answer for many of them;
    # the log-line tests below check that the texts the tables exist for are settled)


def _check(L, h, named, s, want, stats):
    beg = (ctypes.c_int * 40)(); end = (ctypes.c_int * 40)()
    n = L.flbgpu_rx_simulate_fx(h, s, len(s), beg, end)
    assert n != -4
    if want is not None and want[0][0] == 0 and all(c < 0x80 for c in s):
        stats["at0"] += 1
        stats["declined"] += 1 if n < 0 else 0
    if n >= 0:
        assert want is not None and list(want[0]) == [0, end[0]], (s, want, end[0])
        for g in named:
            assert [beg[g], end[g]] == list(want[g]), (s, g, [beg[g], end[g]], want[g])
        stats["settled"] += 1


def test_fx_tables_on_the_reference_fixture():
    """the engine's spans on 400 lines of the reference's apache_10k.mp (tests/golden/apache_400_spans.json)"""
    sys.path.insert(0, HERE)
    import synth
    L = _lib()
    data = open(os.path.join(HERE, "golden", "apache_400.mp"), "rb").read()
    spans = json.load(open(os.path.join(HERE, "golden", "apache_400_spans.json")))
    recs = synth.unpack_all(data)
    from rxdiff import PATTERNS
    for name, pat in (("apache2", PATTERNS[0]), ("apache", PATTERNS[1])):
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, ctypes.create_string_buffer(256), 256)
        assert h
        stats = {"at0": 0, "declined": 0, "settled": 0}
        named = None
        for rec, want in zip(recs, spans[name]):
            line = rec[1][1][0][1]
            if named is None:
                named = [g for g in range(1, len(want))]
            beg = (ctypes.c_int * 40)(); end = (ctypes.c_int * 40)()
            if L.flbgpu_rx_simulate_fx(h, line, len(line), beg, end) == -4:
                break
            _check(L, h, named, line, want, stats)
        else:
            # apache2 is decided by the byte / the next byte everywhere; "apache" ((?<path>[^\"]*?)(?: +\S*)?) needs the
            # reverse states, which the forward walk from boundary 0 does not have: every line is declined
            if name == "apache2":
                assert stats["settled"] == 400 and stats["declined"] == 0, (name, stats)
        L.flbgpu_rx_free(h)


def test_fx_tables_against_the_oracle_on_generated_lines():
    """start-anchored parser patterns on generated log lines and their mutations, against oracle/orx.c (itself pinned on
    the real engine): settled texts carry the oracle's spans, and matching ASCII lines are settled"""
    import random
    sys.path.insert(0, HERE)
    import synth
    from rxdiff import PATTERNS, load_orx, OrxRegex, rand_input
    L = _lib()
    O = load_orx()
    rng = random.Random(20260921)
    data, off, _ = synth.apache_records(300)
    blob = bytes(data)
    lines = [blob[int(off[i]) + 21:int(off[i + 1])] for i in range(300)]
    lines += [b'[Tue Mar 05 10:11:12.123 2024] [core:error] [pid 35708] [client 72.15.99.187] File does not exist: /favicon.ico',
              b'Feb  5 10:11:12 host-1 sshd[4242]: Accepted publickey for root', b'<34>Oct 11 22:14:15 mymachine su: failed for lonvick',
              b'2024-03-05T10:11:12.123456789Z stdout F hello world', b'12 3.5 true some text', b'a b 2024-01-01', b'']
    total = {"at0": 0, "declined": 0, "settled": 0}
    used = 0
    for pat in PATTERNS:
        if not (pat.startswith(b"^") or pat.startswith(b"\\A")) or b"(?<" not in pat:
            continue
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, ctypes.create_string_buffer(256), 256)
        o = OrxRegex(O, pat)
        if not h or not o.ok:
            continue
        beg = (ctypes.c_int * 40)(); end = (ctypes.c_int * 40)()
        if L.flbgpu_rx_simulate_fx(h, b"x", 1, beg, end) == -4:
            L.flbgpu_rx_free(h)
            continue
        used += 1
        named = sorted({g for _, g in o.names()})
        texts = list(lines)
        for ln in lines[:120]:
            m = bytearray(ln)
            for _ in range(rng.randint(1, 3)):
                if not m:
                    break
                k = rng.randrange(len(m))
                r = rng.random()
                if r < 0.3:
                    del m[k]
                elif r < 0.6:
                    m[k] = rng.choice(b' "[]-:/\n\xe9\xffa0')
                elif r < 0.8:
                    m.insert(k, rng.choice(b' "]x\n'))
                else:
                    del m[k:]
            texts.append(bytes(m))
        texts += [rand_input(rng, pat, maxlen=40) for _ in range(200)]
        mine = {"at0": 0, "declined": 0, "settled": 0}
        for s in texts:
            want = o.search(s)
            _check(L, h, named, s, want, mine)
        for k in total:
            total[k] += mine[k]
        if pat == PATTERNS[0]:
            # apache2 (conf/parsers.conf:8-13) is decided by the byte / the next byte everywhere: nothing is declined.
            # Patterns that need the reverse states (lazy loops in front of optional groups, ...) decline; the filter
            # notices the rate and goes back to the phase kernels for them.
            assert mine["declined"] == 0 and mine["settled"] >= 300, mine
        L.flbgpu_rx_free(h)
    assert used >= 6 and total["settled"] > 1500, (used, total)


def test_fx_tables_on_random_patterns():
    """random anchored patterns with named groups (the generator of tests/test_rx_random_patterns.py): whenever the compact
    forward walk settles a text, its spans are the real engine's (a 7-minute run of the same loop: 1.5 M texts, 224 k settled,
    no difference)"""
    import random
    import pytest
    import rxdiff
    import test_rx_random_patterns as T
    ref = rxdiff.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref/libonig_ref.so not built (needs /root/reference)")
    L = _lib()
    rng = random.Random(0xF0F0)
    settled = patterns = 0
    for _ in range(1200):
        names = []
        pat = T.gen(rng, 2, names)
        if not names or b"(" in pat.replace(b"(?<", b"").replace(b"(?:", b""):
            continue
        if not pat.startswith(b"^"):
            pat = b"^" + pat
        eng = rxdiff.RefRegex(ref, pat)
        if not eng.ok:
            continue
        err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 256)
        if not h:
            continue
        named = sorted({g for _, g in eng.names()})
        patterns += 1
        for k in range(12):
            s = rxdiff.rand_input(rng, pat, 24, utf8=(k % 4 == 0))
            want = eng.search(s)
            beg = (ctypes.c_int * 40)(); end = (ctypes.c_int * 40)()
            n = L.flbgpu_rx_simulate_fx(h, s, len(s), beg, end)
            if n == -4:
                break
            if n >= 0:
                assert want is not None and list(want[0]) == [0, end[0]], (pat, s, want, end[0])
                for g in named:
                    assert [beg[g], end[g]] == list(want[g]), (pat, s, g, [beg[g], end[g]], want[g])
                settled += 1
        L.flbgpu_rx_free(h)
    assert patterns > 200 and settled > 300, (patterns, settled)


def test_pair_cells_give_the_single_steps_answers():
    """fx.cpp build_fx(pair = true) -- a cell per (row, class of byte j, class of byte j + 1), what k_parser_reg<PAIR2> walks two
    positions at a time -- must answer like the single-step tables: same return, same spans, on the golden corpus, on apache lines
    with every kind of damage, and on random texts over the patterns' own alphabets"""
    import random
    L = _lib()
    L.flbgpu_rx_simulate_fx2.argtypes = L.flbgpu_rx_simulate_fx.argtypes
    L.flbgpu_rx_simulate_fx_walk_all.argtypes = L.flbgpu_rx_simulate_fx.argtypes
    kat = json.load(open(os.path.join(HERE, "golden", "regex_kat.json")))
    rng = random.Random(5)
    compared = with_pairs = 0

    def both(h, s):
        nonlocal compared
        b1 = (ctypes.c_int * 40)(); e1 = (ctypes.c_int * 40)(); b2 = (ctypes.c_int * 40)(); e2 = (ctypes.c_int * 40)()
        n1 = L.flbgpu_rx_simulate_fx_walk_all(h, s, len(s), b1, e1)      # (every position walked: the pair tables carry no tail)
        n2 = L.flbgpu_rx_simulate_fx2(h, s, len(s), b2, e2)
        if n2 == -4:
            return False                                  # the pair tables do not fit: the kernel keeps the single steps
        assert n1 == n2, (s, n1, n2)
        if n1 >= 0:
            assert list(b1[:n1 + 1]) == list(b2[:n1 + 1]) and list(e1[:n1 + 1]) == list(e2[:n1 + 1]), (s, list(b1[:n1 + 1]), list(b2[:n1 + 1]))
        compared += 1
        return True

    for ent in kat:
        pat = base64.b64decode(ent["pattern"])
        if not ent["compiles"] or not ent["names"] or not (pat.startswith(b"^") or pat.startswith(b"\\A")):
            continue
        err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 256)
        if not h:
            continue
        ok = True
        texts = [base64.b64decode(s64) for s64, _ in ent["cases"]]
        alphabet = sorted(set(b"".join(texts) + pat)) or [97]
        for s in texts:
            ok = ok and both(h, s)
            if not ok:
                break
        for _ in range(200 if ok else 0):
            base = bytearray(rng.choice(texts)) if texts and rng.random() < 0.7 else bytearray()
            for _ in range(rng.randrange(0, 6)):
                if base and rng.random() < 0.5:
                    base[rng.randrange(len(base))] = rng.choice(alphabet)
                else:
                    base.insert(rng.randrange(len(base) + 1), rng.choice(alphabet))
            both(h, bytes(base))
        with_pairs += 1 if ok else 0
        L.flbgpu_rx_free(h)
    assert with_pairs >= 10 and compared > 2500, (with_pairs, compared)

    from bench import APACHE2
    import synth
    import numpy as np
    err = ctypes.create_string_buffer(256)
    h = L.flbgpu_rx_compile(APACHE2.encode(), len(APACHE2), 0, 1, err, 256)
    data, off, _ = synth.apache_records(600)
    ev = np.asarray(data).reshape(600, 277)
    n0 = compared
    for i in range(600):
        line = bytes(ev[i, 21:])
        assert both(h, line)
        for _ in range(6):
            b = bytearray(line)
            k = rng.randrange(4)
            if k == 0:
                b = b[:rng.randrange(len(b) + 1)]
            elif k == 1:
                b[rng.randrange(len(b))] = rng.choice(b' "[]\\x-\xc3\xff\n')
            elif k == 2:
                del b[rng.randrange(len(b))]
            else:
                b.insert(rng.randrange(len(b)), rng.choice(b' "[]'))
            both(h, bytes(b))
    L.flbgpu_rx_free(h)
    assert compared - n0 > 4000


def test_tail_skip_equals_the_full_walk():
    """dev.hpp DevFx::tail_min -- the rows of `(?<message>.*)$` and the like sit last in the table; a lane standing there at a multiple
    of 16 skips to the end of its text: no kill byte in what it skips, then the last byte and the end-of-text column decide.  The
    host execution of the skipping walk (the earliest exit a lane can take) against the walk over every position: same return and
    spans, or -2 (the complete algorithm decides, as for a byte >= 0x80) when a kill byte sits in the skipped part -- on the stock
    parsers that have a tail, with their sample lines cut, stretched, damaged and with kill bytes thrown in."""
    import random
    import re
    L = _lib()
    L.flbgpu_rx_simulate_fx_walk_all.argtypes = L.flbgpu_rx_simulate_fx.argtypes
    L.flbgpu_rx_fx_tail.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    PATS = {
        "apache2": (rb'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$',
                    [b'192.168.2.20 - - [29/Jul/2015:10:27:10 -0300] "GET /cgi-bin/try/ HTTP/1.0" 200 3395 "http://example.com/a" "Mozilla/5.0 (X11; Linux x86_64) AppleWebKit/537.36 (KHTML, like Gecko) Chrome/44.0"',
                     b'127.0.0.1 - frank [10/Oct/2000:13:55:36 -0700] "GET /apache_pb.gif HTTP/1.0" 200 2326']),
        "apache_error": (rb'^\[[^ ]* (?<time>[^\]]*)\] \[(?<level>[^\]]*)\](?: \[pid (?<pid>[^\]]*)\])?( \[client (?<client>[^\]]*)\])? (?<message>.*)$',
                         [b'[Wed Oct 11 14:32:52 2000] [error] [pid 1234] [client 127.0.0.1] client denied by server configuration: /export/home/live/ap/htdocs/test and more text to make it long']),
        "syslog-rfc3164-local": (rb'^\<(?<pri>[0-9]+)\>(?<time>[^ ]* {1,2}[^ ]* [^ ]*) (?<ident>[a-zA-Z0-9_\/\.\-]*)(?:\[(?<pid>[0-9]+)\])?(?:[^\:]*\:)? *(?<message>.*)$',
                                 [b'<34>Oct 11 22:14:15 su[123]: pam_unix(su:session): session opened for user root by (uid=0) and it goes on and on for a while']),
        "cri": (rb'^(?<time>[^ ]+) (?<stream>stdout|stderr) (?<logtag>[^ ]*) (?<message>.*)$',
                [b'2020-10-10T00:10:00.333333333Z stdout F Hello Fluent Bit, this is a fairly long container log line that keeps going']),
        "message_only": (rb'^(?<level>[A-Z]+) (?<message>.*)$', [b'INFO ' + b'x' * 200, b'WARN short']),
    }
    rng = random.Random(0x7A11)
    compared = skipped = declined = 0
    for name, (pat, samples) in PATS.items():
        err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 256)
        assert h, (name, err.value)
        nk = ctypes.c_int(); kill = ctypes.create_string_buffer(4); first = ctypes.c_int()
        rows = L.flbgpu_rx_fx_tail(h, None, 0, ctypes.byref(nk), kill, ctypes.byref(first))
        assert rows >= 1 and list(kill.raw[:nk.value]) == [10], (name, rows, list(kill.raw[:nk.value]))
        texts = []
        for smp in samples:
            texts.append(smp)
            for _ in range(400):
                t = bytearray(smp)
                r = rng.random()
                if r < 0.25:
                    t = t[:rng.randrange(len(t) + 1)]
                elif r < 0.5:
                    t += bytes(rng.choice(b'abc "\\]x') for _ in range(rng.randrange(120)))
                elif r < 0.7:
                    for _ in range(rng.randrange(1, 4)):
                        t[rng.randrange(len(t))] = rng.choice(b'\n"\xe9 ]:\t\x00')
                elif r < 0.8:
                    t += rng.choice([b'"', b'\n', b'"\n', b'\n\n', b' ', b'\r\n'])
                else:
                    k = rng.randrange(len(t))
                    t = t[:k] + bytes(rng.choice(b'ab "') for _ in range(rng.randrange(40))) + t[k:]
                texts.append(bytes(t[:271]))
        for s in texts:
            b1 = (ctypes.c_int * 40)(); e1 = (ctypes.c_int * 40)(); b2 = (ctypes.c_int * 40)(); e2 = (ctypes.c_int * 40)()
            n1 = L.flbgpu_rx_simulate_fx(h, s, len(s), b1, e1)
            n2 = L.flbgpu_rx_simulate_fx_walk_all(h, s, len(s), b2, e2)
            L.flbgpu_rx_fx_tail(h, s, len(s), ctypes.byref(nk), kill, ctypes.byref(first))
            compared += 1
            if first.value >= 0:
                skipped += 1
            if n1 != n2:
                # only ever the decline, and only with a kill byte behind the point the lane left the walk at
                assert n1 == -2 and first.value >= 0 and any(c == 10 or c >= 0x80 for c in s[first.value:]), (name, s, n1, n2, first.value)
                declined += 1
                continue
            if n1 >= 0:
                assert list(b1[:n1 + 1]) == list(b2[:n1 + 1]) and list(e1[:n1 + 1]) == list(e2[:n1 + 1]), (name, s)
        L.flbgpu_rx_free(h)
    assert compared > 2500 and skipped > 700 and declined > 20, (compared, skipped, declined)


def test_stock_parsers_with_a_tail():
    """which of the reference's stock regex parsers (conf/parsers.conf, restated here) get a tail in their compact tables, and with
    which kill bytes: the line feed alone (`.` does not take it; bytes >= 0x80 poison the walk whatever the row)"""
    L = _lib()
    L.flbgpu_rx_fx_tail.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    pats = {
        "apache2": rb'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$',
        "apache_error": rb'^\[[^ ]* (?<time>[^\]]*)\] \[(?<level>[^\]]*)\](?: \[pid (?<pid>[^\]]*)\])?( \[client (?<client>[^\]]*)\])? (?<message>.*)$',
        "syslog-rfc5424": rb'^\<(?<pri>[0-9]{1,5})\>1 (?<time>[^ ]+) (?<host>[^ ]+) (?<ident>[^ ]+) (?<pid>[-0-9]+) (?<msgid>[^ ]+) (?<extradata>(\[(.*?)\]|-)) (?<message>.+)$',
        "cri": rb'^(?<time>[^ ]+) (?<stream>stdout|stderr) (?<logtag>[^ ]*) (?<message>.*)$',
        "nginx": rb'^(?<remote>[^ ]*) (?<host>[^ ]*) (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^\"]*?)(?: +\S*)?)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>[^\"]*)")',
        "fields_only": rb'^(?<a>[^ ]*) (?<b>[^ ]*) (?<c>[^ ]*)$',
    }
    # kill bytes: the line feed (`.` does not take it); a last field `[^ ]*$` also dies on a space -- such a line is no match either way
    want_tail = {"apache2": [10], "apache_error": [10], "syslog-rfc5424": [10], "cri": [10], "nginx": None, "fields_only": [10, 32]}
    for name, pat in pats.items():
        err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 256)
        assert h, (name, err.value)
        nk = ctypes.c_int(); kill = ctypes.create_string_buffer(4); first = ctypes.c_int()
        rows = L.flbgpu_rx_fx_tail(h, None, 0, ctypes.byref(nk), kill, ctypes.byref(first))
        assert (rows > 0) == (want_tail[name] is not None), (name, rows)
        if rows > 0:
            assert sorted(kill.raw[:nk.value]) == want_tail[name], (name, list(kill.raw[:nk.value]))
        L.flbgpu_rx_free(h)


def test_fx3_tables_give_the_single_steps_answers():
    """fx.cpp build_fx3 -- 8-byte cells, two unconditional capture writes per step (positions j - 1 and j), look-ahead cells turned
    into pending rows, double writes carried into the next step: what k_parser_reg<.., FX3> walks without a single test inside the
    step -- must answer like the tables with special entries: same return, same spans; it may hand a text on (-1) where those settle
    it (a look-ahead that resolves to a double write), rarely.  Golden corpus, every stock parser on texts drawn from its own
    pattern, apache lines with every kind of damage."""
    import random
    import test_stock_parsers as sp
    L = _lib()
    L.flbgpu_rx_simulate_fx_walk_all.argtypes = L.flbgpu_rx_simulate_fx.argtypes
    L.flbgpu_rx_simulate_fx3.argtypes = list(L.flbgpu_rx_simulate_fx.argtypes) + [ctypes.POINTER(ctypes.c_int)]
    rng = random.Random(11)
    stats = {"compared": 0, "handed_on": 0, "patterns": 0}

    def both(h, s):
        b1 = (ctypes.c_int * 40)(); e1 = (ctypes.c_int * 40)(); b2 = (ctypes.c_int * 40)(); e2 = (ctypes.c_int * 40)()
        n1 = L.flbgpu_rx_simulate_fx_walk_all(h, s, len(s), b1, e1)
        n3 = L.flbgpu_rx_simulate_fx3(h, s, len(s), b2, e2, None)
        if n3 == -4 or n1 == -4:
            return False
        # fx4 (two positions per cell, the same cells composed): exactly fx3's answer wherever it exists
        b4 = (ctypes.c_int * 40)(); e4 = (ctypes.c_int * 40)(); inf = (ctypes.c_int * 2)(-1, 0)
        n4 = L.flbgpu_rx_simulate_fx3(h, s, len(s), b4, e4, inf)
        if n4 != -4:
            assert n4 == n3, (s, n3, n4)
            if n3 >= 0:
                assert list(b4[:n3 + 1]) == list(b2[:n3 + 1]) and list(e4[:n3 + 1]) == list(e2[:n3 + 1]), (s, list(b4[:n3 + 1]), list(b2[:n3 + 1]))
            stats["pairs"] = stats.get("pairs", 0) + 1
        # fx5 (round 5: the pairs with THREE write ports -- positions j - 1, j, j + 1; build_fx3 pairs = 2): exists only for a pattern
        # none of whose cells needs two writes at one position, and is then exactly fx4
        b5 = (ctypes.c_int * 40)(); e5 = (ctypes.c_int * 40)(); inf5 = (ctypes.c_int * 2)(-2, 0)
        n5 = L.flbgpu_rx_simulate_fx3(h, s, len(s), b5, e5, inf5)
        if n5 != -4:
            # (-1: a cell with two writes at one position sends the three-port walk to the absorbing row with the FAIL slot -- the record
            # takes the complete algorithm, whatever the four-port walk says of it, a later byte >= 0x80 included)
            assert n4 != -4 and (n5 == n4 or n5 == -1), (s, n4, n5)
            if n5 != n4:
                stats["ports3_handed_on"] = stats.get("ports3_handed_on", 0) + 1
            if n5 >= 0:
                assert list(b5[:n5 + 1]) == list(b4[:n5 + 1]) and list(e5[:n5 + 1]) == list(e4[:n5 + 1]), (s, list(b5[:n5 + 1]), list(b4[:n5 + 1]))
            stats["ports3"] = stats.get("ports3", 0) + 1
        if n3 == -1 and n1 != -1:
            stats["handed_on"] += 1
            return True
        assert n1 == n3, (s, n1, n3)
        if n1 >= 0:
            assert list(b1[:n1 + 1]) == list(b2[:n1 + 1]) and list(e1[:n1 + 1]) == list(e2[:n1 + 1]), (s, list(b1[:n1 + 1]), list(b2[:n1 + 1]), list(e1[:n1 + 1]), list(e2[:n1 + 1]))
        stats["compared"] += 1
        return True

    kat = json.load(open(os.path.join(HERE, "golden", "regex_kat.json")))
    for ent in kat:
        pat = base64.b64decode(ent["pattern"])
        if not ent["compiles"] or not ent["names"] or not (pat.startswith(b"^") or pat.startswith(b"\\A")):
            continue
        err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 256)
        if not h:
            continue
        texts = [base64.b64decode(s64) for s64, _ in ent["cases"]]
        alphabet = sorted(set(b"".join(texts) + pat)) or [97]
        ok = all(both(h, s) for s in texts)
        for _ in range(150 if ok else 0):
            base = bytearray(rng.choice(texts)) if texts and rng.random() < 0.7 else bytearray()
            for _ in range(rng.randrange(0, 6)):
                if base and rng.random() < 0.5:
                    base[rng.randrange(len(base))] = rng.choice(alphabet)
                else:
                    base.insert(rng.randrange(len(base) + 1), rng.choice(alphabet))
            both(h, bytes(base))
        stats["patterns"] += 1 if ok else 0
        L.flbgpu_rx_free(h)
    n_kat = stats["compared"]
    assert stats["patterns"] >= 10 and n_kat > 2000, stats
    for k, it in enumerate(sp.stock()):
        pat = sp.inner(it["regex"])
        if not pat.startswith(b"^") or it["section"] != "PARSER":
            continue
        err = ctypes.create_string_buffer(256)
        h = L.flbgpu_rx_compile(pat, len(pat), 0, 1, err, 256)
        assert h
        for s in sp.texts(L, pat, 60, 300 + k):
            if not both(h, s):
                break
        L.flbgpu_rx_free(h)
    assert stats["compared"] - n_kat > 800, stats
    assert stats["handed_on"] * 50 < stats["compared"], stats
    assert stats.get("pairs", 0) > 1500, stats                     # the two-position tables fit for most of these patterns
    assert stats.get("ports3", 0) > 600, stats                     # ... and the three-port form exists for a good part of them (apache2 among them)
    print(stats)
