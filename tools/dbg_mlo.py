#!/usr/bin/env python3
import sys, os, random, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, ml_synth, synth, oracle_binding as ob
g = flbamd_loader.load(); g.init(0); L = g.lib()
rng = random.Random(int(sys.argv[1]))
CRI = r"^(?<time>.+?) (?<stream>stdout|stderr) (?<_p>F|P) (?<log>.*)$"
def rows(ch):
    off = np.empty(int(ch.n) + 1, dtype=np.uint64); L.flbgpu_memcpy_d2h(off.ctypes.data, ch.row_off, off.nbytes)
    buf = ctypes.create_string_buffer(max(1, int(ch.bytes))); L.flbgpu_memcpy_d2h(buf, ch.data, int(ch.bytes))
    return off, buf.raw
for it in range(50):
    cfg, frames, kw = ml_synth.random_sub_case(rng, bad_times=False)
    if cfg["builtin"] != "cri": continue
    pend = b""
    for sec, nsec, t in frames:
        text = pend + t
        pg = g.Parser(CRI, time_fmt="%Y-%m-%dT%H:%M:%S.%L%z", time_key="time", time_keep=True, time_strict=False, skip_empty=False)
        fg = g.FilterParser("log", [pg])
        tl = g.TailLines(skip_empty_lines=kw["skip_empty_lines"])
        d = L.flbgpu_dev_alloc(len(text) + 16); L.flbgpu_memcpy_h2d(d, text, len(text))
        lines, T, proc = tl.process_dev(d, len(text), sec=sec, nsec=nsec)
        pend = text[proc:]
        if int(T.n) == 0: continue
        r, P = fg.filter_dev(T)
        toff, tb = rows(T); poff, pb = rows(P)
        for i in range(int(T.n)):
            if toff[i + 1] > toff[i] and poff[i + 1] == poff[i]:
                print("case", it, "row", i, "of", int(T.n), "ret", r, "P.n", int(P.n), "T row", tb[toff[i]:toff[i + 1]][:160])
                print("stats", fg.counts() if hasattr(fg, "counts") else None, g.last_error())
                sys.exit(0)
print("no empty parsed row")
