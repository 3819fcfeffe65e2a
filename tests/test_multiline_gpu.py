"""The multiline core on the device (csrc/ml.cpp, ml_kernels.inc) against the oracle (oracle/oml.c, pinned on the reference's own
src/multiline/*.c by tests/test_multiline_oracle.py): the vectors of the reference's unit test, random parsers / texts / read
boundaries, the state a stream carries from read to read, and a buffer of a few hundred thousand lines."""
import json, os, random
import pytest

import flbamd_loader
import ml_synth
from test_multiline_oracle import VECTORS, ELASTIC_RULES, vector_text, contents, oracle_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


def device_run(g, cfg, frames, skip_empty_lines=False, final_flush=False):
    p = g.MultilineParser(rules=cfg.get("rules"), builtin=cfg.get("builtin"), type=cfg.get("type", "regex"), match_string=cfg.get("match_string"),
                          negate=cfg.get("negate", False), key_content=cfg.get("key_content"), buffer_limit=cfg.get("buffer_limit_bytes", -1))
    s = p.stream()
    out, n = b"", 0
    try:
        for sec, nsec, text in frames:
            o, r = s.append(text, sec, nsec, skip_empty_lines)
            out += o; n += r
        if final_flush:
            o, r = s.flush(1900000000, 3)
            out += o; n += r
        return out, n, s.state() + (s.truncations(),)
    finally:
        s.close(); p.close()


def first_diff(a, b):
    n = min(len(a), len(b))
    i = next((k for k in range(n) if a[k] != b[k]), n)
    return i, len(a), len(b), a[max(0, i - 60):i + 40], b[max(0, i - 60):i + 40]


@pytest.mark.parametrize("name", ["java", "ruby", "python", "go", "elastic"])
def test_reference_vectors(g, name):
    cfg = {"rules": ELASTIC_RULES} if name == "elastic" else {"builtin": name}
    text = vector_text(name)
    want = [o.encode("latin-1") for o in VECTORS[name]["output"]]
    for frames in ([(1700000000, 1, text)], [(1700000000 + i, i, text[i:i + 97]) for i in range(0, len(text), 97)]):
        out, n, _ = device_run(g, cfg, frames, final_flush=True)
        assert contents(out) == want
        assert n == len(want)
        assert out == oracle_run(cfg, frames, final_flush=True, clock_of_the_call=True)[0]


@pytest.mark.parametrize("seed", range(8))
def test_random_cases(g, seed):
    rng = random.Random(5200 + seed)
    for _ in range(60):
        cfg, frames, kw = ml_synth.random_case(rng)
        if rng.random() < 0.3:
            cfg["buffer_limit_bytes"] = rng.choice([0, 1, 8, 40, 200, 1000])
        want, n, trunc = oracle_run(cfg, frames, clock_of_the_call=True, **kw)
        got, gn, st = device_run(g, cfg, frames, **kw)
        assert got == want, (cfg, frames, kw, first_diff(want, got))
        assert gn == n
        assert st[2] == trunc, (cfg, frames, kw)


def test_rule_by_rule_walk(g, monkeypatch):
    """the product automaton off (FLBGPU_ML_NO_PRODUCT=1, read when the parser is initialised): every rule's own DFA per line"""
    monkeypatch.setenv("FLBGPU_ML_NO_PRODUCT", "1")
    rng = random.Random(6100)
    for _ in range(40):
        cfg, frames, kw = ml_synth.random_case(rng)
        want, n, _ = oracle_run(cfg, frames, clock_of_the_call=True, **kw)
        got, gn, _ = device_run(g, cfg, frames, **kw)
        assert got == want and gn == n, (cfg, frames, kw)


def test_builtin_products(g):
    for name in ml_synth.BUILTINS:
        p = g.MultilineParser(builtin=name)
        states, classes, live = p.product()
        assert 0 < live <= states and classes > 1, (name, states, classes, live)
        p.close()


def test_the_runtime_tests_parser_with_a_negative_lookahead(g):
    """tests/runtime/data/tail/parsers_multiline.conf `multiline-regex`: the continuation rule is "a line that does not start like a
    first line" -- ^(?!A).* -- which the tables answer with A's own automaton (ml.cpp split_leading_lookahead)"""
    first = r"\[\d{4}-\d{2}-\d{2} \d{2}:\d{2}:\d{2},\d{3}\]"
    rules = [("start_state", "/^%s/" % first, "cont"), ("cont", "/^(?!%s).*/" % first, "cont")]
    rng = random.Random(11)
    lines = []
    for i in range(3000):
        r = rng.random()
        if r < 0.3:
            lines.append(b"[2021-03-%02d 10:%02d:%02d,%03d] request %d" % (rng.randrange(1, 29), rng.randrange(60), rng.randrange(60), rng.randrange(1000), i))
        elif r < 0.35:
            lines.append(b"[2021-03-01 10:00:00.123] almost a first line")
        else:
            lines.append(rng.choice([b"  at x.y.Z(A.java:%d)" % i, b"Caused by: q", b"", b"[", b"\xc3\xa9t\xc3\xa9 [2021"]))
    text = b"\n".join(lines) + b"\n"
    frames = [(50, 1, text[:7777]), (60, 2, text[7777:])]
    want, n, _ = oracle_run({"rules": rules}, frames, final_flush=True, clock_of_the_call=True)
    got, gn, _ = device_run(g, {"rules": rules}, frames, final_flush=True)
    assert got == want and gn == n and n > 500


def test_stream_state_is_carried(g):
    rules = [("start_state", r"/^\d+ start/", "cont"), ("cont", r"/^\s+/", "cont")]
    text = b"1 start\n  a\n\n  b\nnope\n  c\n2 start\n"
    for cut in range(len(text) + 1):
        frames = [(100, 5, text[:cut]), (200, 6, text[cut:])]
        want, n, _ = oracle_run({"rules": rules}, frames, clock_of_the_call=True)
        got, gn, state = device_run(g, {"rules": rules}, frames)
        assert got == want and gn == n, cut
        assert state[:2] == (0, len(b"2 start"))                # rule 0 holds the stream; its start line waits in the buffer


def test_a_large_buffer(g):
    rng = random.Random(77)
    text = ml_synth.random_text(rng, 300000, ml_synth.SEED_LINES["java"], long_line=0.0005)
    cfg = {"builtin": "java"}
    cuts = [0, len(text) // 3 + 11, 2 * len(text) // 3 + 5, len(text)]
    frames = [(1700000000 + i, 9 * i, text[cuts[i]:cuts[i + 1]]) for i in range(3)]
    want, n, _ = oracle_run(cfg, frames, final_flush=True, clock_of_the_call=True)
    got, gn, _ = device_run(g, cfg, frames, final_flush=True)
    assert gn == n and n > 50000
    assert got == want, first_diff(want, got)


def test_refusals(g):
    with pytest.raises(ValueError):
        g.MultilineParser(rules=[("cont", r"/^\s/", "cont")])       # the first rule must hold a start_state
    with pytest.raises(ValueError):
        g.MultilineParser(rules=[("start_state", r"/^a/", "nowhere")])


def test_truncation_rounds(g):
    """many groups over the limit in one buffer: every truncating continuation resets the state for what follows"""
    rules = [("start_state", r"/^a/", "c"), ("c", r"/^b/", "c")]
    rng = random.Random(3)
    lines = []
    for _ in range(400):
        lines.append(b"a" * rng.randrange(1, 30))
        for _ in range(rng.randrange(0, 12)):
            lines.append(rng.choice([b"b" * rng.randrange(1, 25), b"", b"x", b"bb"]))
    text = b"\n".join(lines) + b"\n"
    for limit in (16, 33, 64):
        cfg = {"rules": rules, "buffer_limit_bytes": limit}
        frames = [(10, 1, text[:len(text) // 2]), (20, 2, text[len(text) // 2:])]
        want, n, trunc = oracle_run(cfg, frames, final_flush=True, clock_of_the_call=True)
        got, gn, st = device_run(g, cfg, frames, final_flush=True)
        assert trunc > 5
        assert got == want, first_diff(want, got)
        assert gn == n and st[2] == trunc


def test_long_lines_long_groups_and_two_streams(g):
    """lines and carried buffers larger than the emit pass' staging area (copied by the whole wave), groups that stay open across
    reads, and two streams of one parser fed alternately (each carries its own state)"""
    rules = [("start_state", r"/^START/", "c"), ("c", r"/^\s/", "c")]
    rng = random.Random(21)

    def text_of(seed):
        r = random.Random(seed)
        lines = []
        for i in range(300):
            k = r.random()
            if k < 0.2:
                lines.append(b"START %d " % i + bytes(r.choice(b"abcdefg ") for _ in range(r.choice([5, 40, 30000, 120000]))))
            elif k < 0.9:
                lines.append(b"  cont " + bytes(r.choice(b"xyz \t") for _ in range(r.choice([0, 3, 70, 25000]))))
            else:
                lines.append(b"other " + b"q" * r.choice([1, 20000]))
        return b"\n".join(lines) + b"\n"

    cfg = {"rules": rules, "buffer_limit_bytes": 0}          # no limit: groups of megabytes stay whole
    ta, tb = text_of(1), text_of(2)
    cuts_a = sorted(rng.randrange(len(ta)) for _ in range(6))
    cuts_b = sorted(rng.randrange(len(tb)) for _ in range(6))
    fa = [(100 + i, i, ta[a:b]) for i, (a, b) in enumerate(zip([0] + cuts_a, cuts_a + [len(ta)]))]
    fb = [(500 + i, i, tb[a:b]) for i, (a, b) in enumerate(zip([0] + cuts_b, cuts_b + [len(tb)]))]
    want_a = oracle_run(cfg, fa, final_flush=True, clock_of_the_call=True)
    want_b = oracle_run(cfg, fb, final_flush=True, clock_of_the_call=True)
    p = g.MultilineParser(rules=rules, buffer_limit=0)
    sa, sb = p.stream(), p.stream()
    got_a = got_b = b""
    na = nb = 0
    for (x, y) in zip(fa, fb):
        o, r = sa.append(x[2], x[0], x[1]); got_a += o; na += r
        o, r = sb.append(y[2], y[0], y[1]); got_b += o; nb += r
    o, r = sa.flush(1900000000, 3); got_a += o; na += r
    o, r = sb.flush(1900000000, 3); got_b += o; nb += r
    sa.close(); sb.close(); p.close()
    assert (got_a, na) == want_a[:2], first_diff(want_a[0], got_a)
    assert (got_b, nb) == want_b[:2], first_diff(want_b[0], got_b)
    assert max(len(x) for x in contents(got_a)) > 100000


CRI_SAMPLE = (b"2021-05-17T17:35:01.184675702Z stdout F [DEBUG] 1 start multiline - \n"
              b"2021-05-17T17:35:01.184747208Z stdout P partial one \n2021-05-17T17:35:01.184747209Z stderr P err part \n"
              b"2021-05-17T17:35:01.184747210Z stdout F and the end\nnot a cri line\n2021-05-17T17:35:02.1Z stderr F err end\n"
              b"2021-05-17T17:35:03.1Z stdout P dangling\n")
DOCKER_SAMPLE = (b'{"log":"one, ","stream":"stdout","time":"2021-02-01T01:40:03.53413Z"}\n{"log":"two\\n","stream":"stdout","time":"2021-02-01T01:40:03.53414Z"}\n'
                 b'plain\n{"log":"x","stream":"stderr","time":"2021-02-01T01:40:03.5Z"}\n')


@pytest.mark.parametrize("name,text", [("cri", CRI_SAMPLE), ("docker", DOCKER_SAMPLE)])
def test_cri_and_docker_samples(g, name, text):
    """the built-in parsers with a parser in front: lines parsed on the device, one buffer per stream, the first line's map re-packed"""
    for cut in range(0, len(text) + 1, 7):
        frames = [(100, 5, text[:cut]), (200, 6, text[cut:])]
        want, n, _ = oracle_run({"builtin": name}, frames, final_flush=True, clock_of_the_call=True)
        got, gn, _ = device_run(g, {"builtin": name}, frames, final_flush=True)
        assert got == want, (cut, first_diff(want, got))
        assert gn == n


@pytest.mark.parametrize("seed", range(4))
def test_cri_and_docker_random(g, seed):
    rng = random.Random(9100 + seed)
    for _ in range(50):
        cfg, frames, kw = ml_synth.random_sub_case(rng, bad_times=False)
        want, n, _ = oracle_run(cfg, frames, clock_of_the_call=True, **kw)
        got, gn, _ = device_run(g, cfg, frames, **kw)
        assert got == want, (cfg, frames, kw, first_diff(want, got))
        assert gn == n


def test_cri_refuses_what_it_cannot_reproduce(g):
    p = g.MultilineParser(builtin="cri")
    s = p.stream()
    with pytest.raises(RuntimeError):            # the parser takes the line, its time is no time: the reference drops the record and keeps the bytes
        s.append(b"garbage-time stdout F x\n", 100, 5)
    s.close(); p.close()


def test_cri_large_and_custom_parser_in_front(g):
    """200 k containerd lines in three reads; and a [MULTILINE_PARSER] of type endswith with its own `parser` (Format json), key_content
    message, key_group svc: flbgpu_ml_parser_set_subparser"""
    rng = random.Random(31)
    text = ml_synth.cri_text(rng, 200000, damage=0.01, bad_times=False)
    cuts = [0, len(text) // 3 + 5, 2 * len(text) // 3 + 17, len(text)]
    frames = [(1700000000 + i, 7 * i, text[cuts[i]:cuts[i + 1]]) for i in range(3)]
    want, n, _ = oracle_run({"builtin": "cri"}, frames, final_flush=True, clock_of_the_call=True)
    got, gn, _ = device_run(g, {"builtin": "cri"}, frames, final_flush=True)
    assert gn == n and n > 50000
    assert got == want, first_diff(want, got)

    import json
    lines = []
    for i in range(3000):
        d = {"svc": rng.choice(["a", "b"]), "message": rng.choice(["part ", "more ", "done;", ";", ""]), "n": i}
        if rng.random() < 0.1:
            d.pop("svc")
        lines.append(json.dumps(d).encode() if rng.random() < 0.95 else b"not json")
    text = b"\n".join(lines) + b"\n"
    frames = [(50, 1, text[:len(text) // 2]), (60, 2, text[len(text) // 2:])]
    sub_o = dict(regex=None, time_fmt=None, time_key=None, skip_empty=True)
    mo = ob_multiline(type="endswith", match_string=";", key_content="message", subparser=sub_o, key_group="svc")
    want = b""; n = 0
    for sec, nsec, t in frames:
        mo_set_now(mo, sec, nsec)
        o, r, _ = mo.append(t, sec, nsec); want += o; n += r
    mo_set_now(mo, 1900000000, 3)
    o, r, _ = mo.flush(); want += o; n += r
    pj = g.Parser(format="json")
    p = g.MultilineParser(type="endswith", match_string=";", key_content="message", subparser=pj, key_group="svc")
    s = p.stream()
    got = b""; gn = 0
    for sec, nsec, t in frames:
        o, r = s.append(t, sec, nsec); got += o; gn += r
    o, r = s.flush(1900000000, 3); got += o; gn += r
    s.close(); p.close(); pj.close()
    assert (got, gn) == (want, n), first_diff(want, got)


def ob_multiline(**kw):
    import oracle_binding as ob
    return ob.Multiline(**kw)


def mo_set_now(m, sec, nsec):
    import oracle_binding as ob
    ob.lib().oml_set_now.argtypes = [ob.c_void_p, ob.c_int64, ob.c_int64]
    ob.lib().oml_set_now(m.h, sec, nsec)


def list_run(g, names, frames, skip_empty_lines=False, final_flush=False):
    ps = [g.MultilineParser(builtin=nm) for nm in names]
    ml = g.MultilineList(ps)
    out, n = b"", 0
    try:
        for sec, nsec, text in frames:
            o, r = ml.append(text, sec, nsec, skip_empty_lines)
            out += o; n += r
        if final_flush:
            o, r = ml.flush(1900000000, 3)
            out += o; n += r
        return out, n, ml.lru
    finally:
        ml.close()
        for p in ps:
            p.close()


@pytest.mark.parametrize("seed", range(3))
def test_parser_lists(g, seed):
    """in_tail's `multiline.parser docker, cri` (both orders): flb_ml_append_text's loop over the parser instances.  Files of one
    runtime -- every line that is taken at all is taken by one parser, damaged lines leave alone -- equal the oracle's list (which
    equals the reference's, test_multiline_oracle.py::test_parser_lists_against_the_reference) at every read boundary"""
    rng = random.Random(9300 + seed)
    ran = 0
    for _ in range(40):
        names = rng.choice([["docker", "cri"], ["cri", "docker"]])
        n = rng.randrange(0, 70)
        text = ml_synth.cri_text(rng, n, bad_times=False) if rng.random() < 0.5 else ml_synth.docker_text(rng, n)
        if rng.random() < 0.2 and text:
            text = text[:-1]
        frames = ml_synth.frames_of(rng, text)
        kw = dict(skip_empty_lines=rng.random() < 0.4, final_flush=rng.random() < 0.7)
        cfg = {"builtin": ", ".join(names)}
        want, wn, _ = oracle_run(cfg, frames, clock_of_the_call=True, **kw)
        got, gn, lru = list_run(g, names, frames, **kw)
        assert got == want, (names, frames, kw, first_diff(want, got))
        assert gn == wn
        ran += 1
    assert ran == 40


def test_parser_list_refuses_a_split_read(g):
    """lines of both runtimes in one read: the reference hands them to different parsers line by line -- the device path says no"""
    rng = random.Random(5)
    a = ml_synth.cri_text(rng, 20, bad_times=False, damage=0.0)
    b = ml_synth.docker_text(rng, 20)
    ps = [g.MultilineParser(builtin="docker"), g.MultilineParser(builtin="cri")]
    ml = g.MultilineList(ps)
    with pytest.raises(RuntimeError):
        ml.append(a + b, 100, 5)
    # nothing moved: the same list still takes a clean file, and remembers who took it
    out, n = ml.append(a, 100, 5)
    assert ml.lru == 1 and n > 0
    want, wn, _ = oracle_run({"builtin": "docker, cri"}, [(100, 5, a)], clock_of_the_call=True)
    assert out == want and n == wn
    ml.close()
    for p in ps:
        p.close()
    with pytest.raises(ValueError):
        g.MultilineList([g.MultilineParser(builtin="java")])          # no parser in front: taking a line depends on the parser's state
