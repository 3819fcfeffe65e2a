"""The multiline restatement (oracle/oml.c) pinned on the reference: the vectors of the reference's own unit test
(tests/golden/multiline_vectors.json, written by tools/gen_ml_golden.py from tests/internal/multiline.c) and, where oracle/_ref/ref_filters
is built, the reference's own src/multiline/*.c run on the same frames (kind 5 of oracle/ref_filters_shim.c)."""
import json, os, random
import msgpack
import pytest

import oracle_binding as ob
import ref_filters as rf
import ml_synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VECTORS = json.load(open(os.path.join(ROOT, "tests", "golden", "multiline_vectors.json")))
ELASTIC_RULES = [("start_state", r"/^\[/", "elastic_cont"), ("elastic_cont", r"/^\s+/", "elastic_cont")]


def vector_text(name):
    return b"".join(l.encode("latin-1").rstrip(b"\n") + b"\n" for l in VECTORS[name]["input"])


def contents(records, key=b"log"):
    u = msgpack.Unpacker(raw=True)
    u.feed(records)
    return [r[1][key] for r in u]


def oracle_run(cfg, frames, skip_empty_lines=False, final_flush=False, clock_of_the_call=False, flush_time=(1900000000, 3)):
    """clock_of_the_call: flb_time_get() (a group flushed before any time was registered) answers the time of the frame being appended"""
    names = [x.strip() for x in cfg["builtin"].split(",")] if cfg.get("builtin") and "," in cfg["builtin"] else None
    if names:
        # in_tail's `multiline.parser a, b`: one instance per name on the same stream (flb_ml.c:671-760)
        m = ob.Multiline(builtin=names[0], key_content=cfg.get("key_content"))
        for nm in names[1:]:
            m.chain(ob.Multiline(builtin=nm, key_content=cfg.get("key_content")))
    else:
        m = ob.Multiline(rules=cfg.get("rules"), builtin=cfg.get("builtin"), type=cfg.get("type", "regex"), match_string=cfg.get("match_string"),
                         negate=cfg.get("negate", False), key_content=cfg.get("key_content"), buffer_limit=cfg.get("buffer_limit_bytes", -1))
    ob.lib().oml_set_now.argtypes = [ob.c_void_p, ob.c_int64, ob.c_int64]
    ob.lib().oml_set_now(m.h, 1600000000, 77)
    out, n, trunc = b"", 0, 0
    for sec, nsec, text in frames:
        if clock_of_the_call:
            ob.lib().oml_set_now(m.h, sec, nsec)
        o, r, t = m.append(text, sec, nsec, skip_empty_lines)
        out += o; n += r; trunc += t
    if final_flush:
        if clock_of_the_call:
            ob.lib().oml_set_now(m.h, *flush_time)
        o, r, t = m.flush()
        out += o; n += r
    return out, n, trunc


def ref_case(cfg, frames, skip_empty_lines=False, final_flush=False):
    kw = {k: v for k, v in cfg.items() if k != "buffer_limit_bytes"}
    if "buffer_limit_bytes" in cfg:
        kw["buffer_limit"] = str(cfg["buffer_limit_bytes"])
    return rf.ml_case(frames, skip_empty_lines=skip_empty_lines, final_flush=final_flush, **kw)


@pytest.mark.parametrize("name", ["java", "ruby", "python", "go", "elastic"])
def test_reference_vectors(name):
    cfg = {"rules": ELASTIC_RULES} if name == "elastic" else {"builtin": name}
    out, n, _ = oracle_run(cfg, [(1700000000, 1, vector_text(name))], final_flush=True)
    want = [o.encode("latin-1") for o in VECTORS[name]["output"]]
    assert contents(out) == want
    assert n == len(want)


def test_record_layout_and_times():
    rules = [("start_state", r"/^\d+ start/", "cont"), ("cont", r"/^\s+/", "cont")]
    text = b"1 start\n  a\n\n  b\nnope\n  c\n2 start\n"
    out, n, _ = oracle_run({"rules": rules}, [(100, 5, text[:11]), (200, 6, text[11:])], final_flush=True)
    u = msgpack.Unpacker(raw=True); u.feed(out)
    recs = list(u)
    assert [r[1][b"log"] for r in recs] == [b"1 start\n  a\n", b"  b\n", b"nope\n", b"  c\n", b"2 start\n"]
    assert out.startswith(bytes.fromhex("9292d700000000c800000006df00000000") + b"\x81\xa3log\xac1 start\n  a\n")
    # the empty line is taken by nobody: it flushes the group and would leave alone, but an empty buffer makes no record;
    # rule_to_state survives the flush, so "  b" continues into a fresh buffer


@pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built (needs /root/reference)")
def test_against_the_reference_vectors_and_layout():
    cases, wants = [], []
    for name in ["java", "ruby", "python", "go", "elastic"]:
        cfg = {"rules": ELASTIC_RULES} if name == "elastic" else {"builtin": name}
        for kc in (None, "log"):
            c = dict(cfg, key_content=kc) if kc else cfg
            fr = [(1700000000, 1, vector_text(name))]
            cases.append(ref_case(c, fr, final_flush=True))
            wants.append(oracle_run(c, fr, final_flush=True))
    for (ret, out), (want, n, _) in zip(rf.run(cases), wants):
        assert ret == n and out == want


@pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(6))
def test_random_cases_against_the_reference(seed):
    rng = random.Random(4100 + seed)
    cases, wants, descr = [], [], []
    for _ in range(120):
        cfg, frames, kw = ml_synth.random_case(rng)
        if rng.random() < 0.25:
            cfg["buffer_limit_bytes"] = rng.choice([0, 1, 8, 40, 200, 1000])
        cases.append(ref_case(cfg, frames, **kw))
        wants.append(oracle_run(cfg, frames, **kw))
        descr.append((cfg, frames, kw))
    for (ret, out), (want, n, _), d in zip(rf.run(cases), wants, descr):
        assert out == want, d
        assert ret == n, d


@pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(4))
def test_cri_and_docker_against_the_reference(seed):
    """the built-in parsers with a parser in front (cri: regex, docker: json): key_content / key_pattern / key_group from the parsed
    map, one buffer per stream, the first line's map re-packed at the flush, refused lines flushing every group"""
    rng = random.Random(8800 + seed)
    cases, wants, descr = [], [], []
    for _ in range(100):
        cfg, frames, kw = ml_synth.random_sub_case(rng)
        cases.append(ref_case(cfg, frames, **kw))
        wants.append(oracle_run(cfg, frames, **kw))
        descr.append((cfg, frames, kw))
    for (ret, out), (want, n, _), d in zip(rf.run(cases), wants, descr):
        assert out == want, d
        assert ret == n, d


@pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(3))
def test_parser_lists_against_the_reference(seed):
    """in_tail's `multiline.parser docker, cri` (and the other order, and java behind them): flb_ml_append_text tries the parser that took
    the stream's last line first, then the others in order; a line nobody takes flushes every parser's groups and leaves alone through the
    FIRST parser's default group (flb_ml.c:671-760).  Homogeneous files, and files that switch format half way or line by line."""
    rng = random.Random(9900 + seed)
    cases, wants, descr = [], [], []
    for _ in range(80):
        lst = rng.choice(["docker, cri", "cri, docker", "docker,cri", "cri, docker, java", "java, cri"])
        n = rng.randrange(0, 60)
        kind = rng.random()
        if kind < 0.35:
            text = ml_synth.cri_text(rng, n, bad_times=False)
        elif kind < 0.7:
            text = ml_synth.docker_text(rng, n)
        elif kind < 0.85:
            text = ml_synth.cri_text(rng, n // 2, bad_times=False) + ml_synth.docker_text(rng, n - n // 2)
        else:
            a = ml_synth.cri_text(rng, n, bad_times=False).split(b"\n")
            b = ml_synth.docker_text(rng, n).split(b"\n")
            mix = [rng.choice([x, y]) for x, y in zip(a, b)]
            text = b"\n".join(mix) + b"\n"
        cfg = {"builtin": lst}
        frames = ml_synth.frames_of(rng, text)
        kw = dict(skip_empty_lines=rng.random() < 0.4, final_flush=rng.random() < 0.7)
        cases.append(ref_case(cfg, frames, **kw))
        wants.append(oracle_run(cfg, frames, **kw))
        descr.append((cfg, frames, kw))
    produced = 0
    for (ret, out), (want, n, _), d in zip(rf.run(cases), wants, descr):
        assert out == want, d
        assert ret == n, d
        produced += n
    assert produced > 300
