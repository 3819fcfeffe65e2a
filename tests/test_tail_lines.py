"""in_tail's line packing: the oracle's restatement (oracle/oflb.c oflb_tail_process: process_content's loop + the record
layout of flb_tail_file_pack_line) against the reference's REAL event encoder driven through the same call sequence
(oracle/_ref/ref_filters kind 4) on the CPU, and the device path (tail_kernels.inc, through the C ABI) against the oracle."""
import os, random, sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_binding as ob
import ref_filters as rf


def _texts(seed, n=40):
    rng = random.Random(seed)
    out = [b"", b"\n", b"\r\n", b"no newline at all", b"\0\0\0", b"\0\0abc\n", b"a\n\n\r\nb\r\n", b"x\r\r\n", b"\r\n\r\n", b"last\nline without newline",
           b"long " + b"y" * 70000 + b"\nshort\n", b"\xff\xfe\n\xc3\xa9\n", b"tab\there\n \n"]
    for _ in range(n):
        parts = []
        for _ in range(rng.randrange(0, 400)):
            r = rng.random()
            if r < 0.1: line = b""
            elif r < 0.15: line = b"\r"
            elif r < 0.8: line = bytes(rng.choice(b"abcdefghij 0123456789\"[]/-.:") for _ in range(rng.randrange(1, 300)))
            else: line = bytes(rng.randrange(1, 256) for _ in range(rng.randrange(1, 40))).replace(b"\n", b" ")
            parts.append(line + rng.choice([b"\n", b"\n", b"\r\n"]))
        t = b"".join(parts)
        if rng.random() < 0.5: t += b"partial line"
        if rng.random() < 0.2: t = b"\0" * rng.randrange(1, 200) + t
        out.append(t)
    return out


CONFIGS = [dict(), dict(skip_empty_lines=False), dict(key="message", path_key="file", path="/var/log/app/a.log"),
           dict(offset_key="offset", stream_offset=123456789012), dict(path_key="p", path="", offset_key="o", stream_offset=250, skip_empty_lines=False)]


@pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built (no reference tree)")
def test_oracle_against_the_real_encoder():
    cases, wants = [], []
    for t in _texts(1):
        for c in CONFIGS:
            cases.append(rf.tail_case(t, sec=1700000000, nsec=123456789, **c))
            wants.append(ob.tail_process(t, sec=1700000000, nsec=123456789, **c))
    for res, want in zip(rf.run(cases), wants):
        assert rf.tail_result(res) == want


@pytest.mark.gpu
def test_device_against_the_oracle():
    import flbamd_loader
    g = flbamd_loader.load()
    g.init(0)
    for c in CONFIGS:
        so = c.get("stream_offset", 0)
        kw = {k: v for k, v in c.items() if k != "stream_offset"}
        t = g.TailLines(**kw)
        for text in _texts(2, 25):
            want = ob.tail_process(text, sec=1700000000, nsec=5, **c)
            got = t.process(text, stream_offset=so, sec=1700000000, nsec=5)
            assert got[0] == want[0] and got[2] == want[2], (c, len(text), got[0], want[0], got[2], want[2])
            assert got[1] == want[1], (c, len(text))
        t.close()


@pytest.mark.gpu
def test_device_lines_feed_the_filters():
    """text -> events on the device -> filter_parser + filter_grep without leaving HBM: the chain's output equals the oracle's"""
    import flbamd_loader, synth
    from bench import APACHE2, TIME_FMT, GREP_RULE
    g = flbamd_loader.load()
    g.init(0)
    data, off, _ = synth.apache_records(5000)
    blob = bytes(data)
    text = b"".join(blob[int(off[i]) + 21:int(off[i + 1])] + b"\n" for i in range(5000))
    L = g.lib()
    d = L.flbgpu_dev_alloc(len(text)); L.flbgpu_memcpy_h2d(d, text, len(text))
    t = g.TailLines()
    lines, chunk, processed = t.process_dev(d, len(text), sec=7, nsec=8)
    assert lines == 5000 and processed == len(text)
    p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
    fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE])
    r, o = g.FilterChain([fp, fg]).filter_dev(chunk)
    import ctypes
    host = (ctypes.c_uint8 * int(o.bytes))()
    L.flbgpu_memcpy_d2h(host, ctypes.c_void_p(o.data), int(o.bytes))
    ev = ob.tail_process(text, sec=7, nsec=8)[1]
    w1 = ob.FilterParser("log", [ob.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")]).filter(ev)[1]
    w2 = ob.Grep([GREP_RULE]).filter(w1)[1]
    assert bytes(host) == w2
