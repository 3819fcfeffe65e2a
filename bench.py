#!/usr/bin/env python3
"""bench.py -- records/sec through filter_parser(apache2) -> filter_grep on MI355X.

One "step" = one flb_filter_do (src/flb_filter.c:121-325) over one batch of synthetic input that is already
resident in HBM: the chunk (10 M seeded apache-combined lines of 256 B wrapped as 277 B V2 log events,
BASELINE.json configs[1]) goes through the chain [filter_parser (conf/parsers.conf 'apache2', Key_Name log),
filter_grep (Regex code ^5\\d\\d$)] -- flbgpu_filter_chain_run_dev, which runs the two as a pair: the rules
are evaluated on the capture spans and only the kept records are written (fused_kernels.inc; SURVEY 8(d)
"fused": 552 + 275 x keep algorithmic bytes per record).

    python bench.py --gpus N --steps K --warmup W [--records R]

For N > 1 the driver launches one rank per GPU with torch.distributed.run; every rank filters its
own shard (records are independent: no data-path collective, weak scaling), the timed region is
bracketed by barrier + synchronize and the MAX over ranks is reported.  Rank 0 prints one JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

APACHE2 = (r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" '
           r'(?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$')
TIME_FMT = "%d/%b/%Y:%H:%M:%S %z"
GREP_RULE = ("regex", r"code ^5\d\d$")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# BASELINE configs[2]: 32 patterns on NDJSON-derived records = 16 Regex rules in OR mode, then an
# Exclude-only set of 16 (two filter_grep instances: AND / OR need one rule type, grep.c:90-98); literal-heavy
# with four class / quantifier patterns in each set
GREP32_REGEX = [("regex", r) for r in (
    "level ^(error|warn)$", "msg timeout", "msg refused", "$svc['name'] db", "path ^/v1/items/1", "msg request 9", "level debug", "path x=7",
    "msg finished ok$", "$svc['name'] ^cache$", "path /items/[0-9]{5}", r"msg ^request \d+ finished", "level ^i", "path [?]x=[0-9]$",
    "msg 00 finished", r"path ^/v1/\w+/\d*0[?]")]
GREP32_EXCLUDE = [("exclude", r) for r in (
    "msg request 1", "level ^warn$", "path x=1$", "$svc['pod'] ^pod-1", "msg 77", "path /items/4", "level nothing", "msg never",
    "path ^/v2", "$svc['name'] ^$", r"path x=\d\d$", "msg [5-6]{3} finished", r"level ^\s", "path items/[1-2]{2}", "msg ok ", "$svc['name'] b$")]


PMC_FILE = os.path.join("profiles", "r2e_pmc_hbm_bench_10M.json")


def recorded_traffic(kernel, n):
    """HBM bytes per launch of `kernel` from the committed PMC pass of this same command (rocprofv3 cannot run
    inside the timed process): FETCH_SIZE + WRITE_SIZE (KiB) scaled by the factors tools/calib_counters.py
    measured for this access pattern on a kernel with a known byte count (profiles/r2_counter_calibration.json).
    None when the workload differs from the profiled one."""
    path = os.path.join(ROOT, PMC_FILE)
    if n != 10_000_000 or not os.path.exists(path):
        return None, None
    try:
        ks = json.load(open(path))["kernels"]
        base = lambda k: k.split("::")[-1].split("<")[0].split("(")[0]
        hit = [v for k, v in ks.items() if base(k) == kernel or (kernel.endswith("k_scan") and base(k).startswith("k_scan_"))]
        if not hit:
            return None, None
        if len(hit) > 1:                                  # the scan is three small kernels
            hit = [{c: sum(h.get(c, 0) for h in hit) for c in ("FETCH_SIZE", "WRITE_SIZE")}]
        # profiles/r2_counter_calibration.json (tools/calib_counters.py, kernels with a known HBM byte count): on this
        # GPU FETCH_SIZE x 1024 is HALF the bytes fetched for every read shape measured -- 16 B/lane and 4 B/lane
        # coalesced, a whole 128 B line per lane, 16 B of a line per lane (which pulls the whole line) -- and
        # WRITE_SIZE x 1024 is the bytes written.
        cal = {}
        cpath = os.path.join(ROOT, "profiles", "r2_counter_calibration.json")
        if os.path.exists(cpath):
            cal = json.load(open(cpath))
        pattern = {"k_parser_locate": "coalesced16", "k_grep_match": "coalesced16", "k_gather": "coalesced16",
                   "k_parser_rx": "lane_line128", "k_parser_finish": "column4", "k_parser_emit": "lane_line128", "k_pg_emit": "lane_line128",
                   "k_parser_reg": "per_lane16", "k_parser_tile": "coalesced16"}.get(kernel, "column4")
        ff = float(cal.get("fetch_factor", {}).get(pattern, 2.0))
        wf = float(cal.get("write_factor", {}).get("write16", 1.0))
        b = hit[0].get("FETCH_SIZE", 0) * 1024 * ff + hit[0].get("WRITE_SIZE", 0) * 1024 * wf
        return int(b), PMC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x %.2f, WRITE x %.2f: profiles/r2_counter_calibration.json)" % (ff, wf)
    except Exception:
        return None, None


def cpu_nproc_leg(sample, nproc, seconds=6.0):
    """N independent processes (one oracle filter pair each, its own shard of `sample`), wall-clock records/s of
    the lot -- the baseline SURVEY 8(d) asks for next to the single thread.  Runs before the GPU is touched."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    blob, off = sample

    def work(i):
        import oracle_binding as ob
        n = len(off) - 1
        lo, hi = n * i // nproc, n * (i + 1) // nproc
        part = blob[int(off[lo]):int(off[hi])]
        po = ob.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
        fo = ob.FilterParser("log", [po]); go = ob.Grep([GREP_RULE])
        done = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            r, parsed = fo.filter(part)
            go.filter(parsed)
            done += hi - lo
        q.put((done, time.perf_counter() - t0))

    ps = [ctx.Process(target=work, args=(i,)) for i in range(nproc)]
    t0 = time.perf_counter()
    for p_ in ps: p_.start()
    res = [q.get() for _ in ps]
    for p_ in ps: p_.join()
    wall = time.perf_counter() - t0
    return sum(r[0] for r in res) / wall


def measure_secondary(g, torch, dist, rank, world, parsed_chunk, n, args, raw_chunk=None):
    """filter_log_to_metrics on the parsed chunk (BASELINE configs[3] shape: counter + histogram, partial
    aggregates all-reduced over RCCL when N > 1) and NDJSON -> msgpack events -> 32-rule filter_grep
    (configs[2] shape).  Reported next to the headline number, never folded into it."""
    import json as _json
    import random
    out = {}
    steps = 3
    # -- record boundaries of the raw input chunk found on the device (what the decoder loop of every
    #    cb_filter does first); the headline step takes them as part of the device-resident chunk format
    if raw_chunk is not None:
        ix = g.Indexer()
        ch, consumed = ix.index_dev(raw_chunk.data, int(raw_chunk.bytes))
        assert int(ch.n) == n and consumed == int(raw_chunk.bytes), (int(ch.n), consumed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ix.index_dev(raw_chunk.data, int(raw_chunk.bytes))
        dt = (time.perf_counter() - t0) / steps
        out["record_indexer"] = {"records_per_s_per_gpu": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                                 "chunk_GBps": round(int(raw_chunk.bytes) / dt / 1e9, 1), **ix.stats()}
        del ix
    # -- log_to_metrics
    for name, mode, props, vf in (("l2m_counter", "counter", [("label_field", "method"), ("label_field", "code")], None),
                                  ("l2m_histogram", "histogram", [("label_field", "code")], "size")):
        f = g.FilterLogToMetrics(mode, props, value_field=vf)
        f.set_index_base(rank << 40)
        f.filter_dev(parsed_chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            f.filter_dev(parsed_chunk)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        e = {"records_per_s_per_gpu": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 3)}
        if dist is not None:
            # the collective behind the C ABI (flbgpu_l2m_all_reduce: RCCL all-gather of the label tuples, all-reduce
            # MAX / SUM of the rows); the communicator's id travels over the process group that launched the ranks
            if "rccl" not in out:
                def exchange(raw):
                    box = [raw]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
                out["rccl"] = g.RcclComm(world, rank, exchange)
            t0 = time.perf_counter()
            keys, rows = g.l2m_all_reduce_rccl(f, out["rccl"])
            e["all_reduce_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
            e["rccl_ranks"] = world
            snap = f.snapshot((keys, rows))
        else:
            snap = f.snapshot()
        e["series"] = len(snap)
        e["observations"] = int(sum(x["value"] for x in snap)) if mode == "counter" else int(sum(x["count"] for x in snap))
        out[name] = e
        f.close()
    # -- NDJSON lines -> events -> grep with 32 rules
    rng = random.Random(7 + rank)
    base = []
    for i in range(4096):
        d = {"time": "2026-09-21T10:%02d:%02d.%03dZ" % (rng.randrange(60), rng.randrange(60), rng.randrange(1000)),
             "level": rng.choice(["info", "warn", "error", "debug"]),
             "msg": "request %d finished %s" % (rng.randrange(10 ** 6), rng.choice(["ok", "timeout", "refused"])),
             "code": rng.randrange(200, 600), "latency": round(rng.random() * 100, 3),
             "svc": {"name": rng.choice(["api", "db", "cache"]), "pod": "pod-%d" % rng.randrange(1000)},
             "path": "/v1/items/%d?x=%d" % (rng.randrange(10 ** 5), rng.randrange(100)), "bytes": rng.randrange(10 ** 6)}
        base.append(_json.dumps(d).encode() + b"\n")
    nl = min(n, 4_000_000)
    data = b"".join(base) * ((nl + len(base) - 1) // len(base))
    off = g.split_lines(data)
    nl = len(off) - 1
    L = g.lib()
    d_data = L.flbgpu_dev_alloc(len(data) + 16); d_off = L.flbgpu_dev_alloc(off.nbytes)
    L.flbgpu_memcpy_h2d(d_data, data, len(data)); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
    chunk = g.DevChunk(d_data, d_off, nl, len(data))
    pk = g.JsonPacker()
    fg1 = g.FilterGrep(GREP32_REGEX, "OR"); fg2 = g.FilterGrep(GREP32_EXCLUDE, "OR")
    ch32 = g.FilterChain([fg1, fg2])
    ev = pk.run_dev(chunk, events=True, ts=(1, 0))
    ch32.filter_dev(ev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ev = pk.run_dev(chunk, events=True, ts=(1, 0))
    torch.cuda.synchronize()
    dt_j = (time.perf_counter() - t0) / steps
    fg1.profile(True); fg2.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        r32, o32 = ch32.filter_dev(ev)
    torch.cuda.synchronize()
    dt_g = (time.perf_counter() - t0) / steps
    p32 = fg1.profile_read()
    st32 = ch32.last_stats()
    out["ndjson_to_events"] = {"lines_per_s_per_gpu": round(nl / dt_j, 1), "ms_per_step": round(dt_j * 1e3, 3), "lines": nl,
                               "text_bytes": len(data), "text_GBps": round(len(data) / dt_j / 1e9, 2)}
    ev_bytes = int(ev.bytes)
    gm = p32.get("k_grep_match", (0, 1))
    out["grep_32_rules"] = {"records_per_s_per_gpu": round(nl / dt_g, 1), "ms_per_step": round(dt_g * 1e3, 3),
                            "rules": "16 Regex (OR) then 16 Exclude (OR): two filter_grep instances chained",
                            "kept_after_regex": int(st32[0]["out_records"]), "kept": int(st32[1]["out_records"]), "event_bytes": ev_bytes,
                            "roofline": {"kernel": "k_grep_match (first instance)", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                         "achieved": round(ev_bytes / (gm[0] / max(gm[1], 1) / 1e3) / 1e9, 1) if gm[0] else None,
                                         "frac": round(ev_bytes / (gm[0] / max(gm[1], 1) / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if gm[0] else None}}
    fg = fg1
    # -- msgpack -> JSON lines of the parsed chunk (flb_pack_msgpack_to_json_format: what out_stdout / out_http / out_kafka
    #    call on every flushed chunk), text left in HBM
    jf_ = g.JsonFormatter("lines", "double", b"date")
    jf_.format_dev(parsed_chunk)
    jf_.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rcj, oj = jf_.format_dev(parsed_chunk)
    torch.cuda.synchronize()
    dt_f = (time.perf_counter() - t0) / steps
    pj = jf_.profile_read()
    in_b, out_b = int(parsed_chunk.bytes), int(oj.bytes)
    ke = pj.get("k_fmt_emit", (0, 1))
    ke_ms = ke[0] / max(ke[1], 1)
    out["msgpack_to_json"] = {"records_per_s_per_gpu": round(n / dt_f, 1), "ms_per_step": round(dt_f * 1e3, 3), "format": "lines, date double, escape_unicode on",
                              "msgpack_bytes": in_b, "json_bytes": out_b,
                              "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in pj.items()},
                              "roofline": {"kernel": "k_fmt_emit", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                           "achieved": round((in_b + out_b) / (ke_ms / 1e3) / 1e9, 1) if ke_ms else None,
                                           "frac": round((in_b + out_b) / (ke_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if ke_ms else None,
                                           "step": {"achieved": round((2 * in_b + out_b) / dt_f / 1e9, 1), "note": "size pass reads the chunk once more"}}}
    if rank == 0 and world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding as ob_
        import numpy as _np
        # CPU leg: the oracle's formatter on the first 200 k parsed records (copied back from HBM)
        offs = _np.zeros(200_001 if n > 200_000 else n + 1, dtype=_np.uint64)
        L.flbgpu_memcpy_d2h(offs.ctypes.data, parsed_chunk.row_off, offs.nbytes)
        nb = int(offs[-1])
        hb = ctypes.create_string_buffer(nb)
        L.flbgpu_memcpy_d2h(hb, parsed_chunk.data, nb)
        t0 = time.perf_counter()
        ref = ob_.msgpack_to_json_format(hb.raw, 3, 0, b"date", 1, 0)
        cdt = time.perf_counter() - t0
        out["msgpack_to_json"]["cpu_port_records_per_s"] = round((len(offs) - 1) / cdt, 1)
        got = ctypes.create_string_buffer(len(ref))
        L.flbgpu_memcpy_d2h(got, oj.data, len(ref))
        out["msgpack_to_json"]["matches_oracle_prefix"] = bool(got.raw == ref)
    jf_.close()
    if rank == 0 and world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding as ob
        import jsonfuzz as jf
        o = jf.oracle()
        sample = data[: int(off[200_000])] if nl > 200_000 else data
        t0 = time.perf_counter()
        r = o(sample)
        cdt = time.perf_counter() - t0
        out["ndjson_to_events"]["cpu_port_lines_per_s"] = round(r[3] / cdt, 1)
    # -- in_tail in front of the path: a file buffer (the same apache lines, '\n' terminated) cut into log events on the device
    #    (process_content + flb_tail_file_pack_line), and the headline pair run on THAT chunk (in_tail's 32-bit map headers)
    try:
        import numpy as np
        m = min(n, 4_000_000)
        # the `log` values of the first m events (277 B = 21 B framing + 256 B line) with a newline behind each
        ev = np.zeros((m, 277), dtype=np.uint8)
        L.flbgpu_memcpy_d2h(ev.ctypes.data, raw_chunk.data, m * 277)
        txt = np.empty((m, 257), dtype=np.uint8)
        txt[:, :256] = ev[:, 21:]; txt[:, 256] = 10
        d_txt = L.flbgpu_dev_alloc(txt.nbytes); L.flbgpu_memcpy_h2d(d_txt, txt.ctypes.data, txt.nbytes)
        tl = g.TailLines()
        lines_, tchunk, proc_ = tl.process_dev(d_txt, txt.nbytes, sec=1700000000, nsec=0)
        assert lines_ == m and proc_ == txt.nbytes, (lines_, proc_)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tl.process_dev(d_txt, txt.nbytes, sec=1700000000, nsec=0)
        dt_t = (time.perf_counter() - t0) / steps
        lines_, tchunk, proc_ = tl.process_dev(d_txt, txt.nbytes, sec=1700000000, nsec=0)
        p2 = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
        f2 = g.FilterParser("log", [p2]); g2 = g.FilterGrep([GREP_RULE])
        ch2 = g.FilterChain([f2, g2])
        ch2.filter_dev(tchunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r_, o_ = ch2.filter_dev(tchunk)
        dt_c = (time.perf_counter() - t0) / steps
        out["tail_lines"] = {"lines_per_s_per_gpu": round(m / dt_t, 1), "ms_per_step": round(dt_t * 1e3, 3), "text_bytes": int(txt.nbytes),
                             "event_bytes": int(tchunk.bytes), "GBps_in_plus_out": round((txt.nbytes + int(tchunk.bytes)) / dt_t / 1e9, 1),
                             "then_parser_grep": {"records_per_s_per_gpu": round(m / dt_c, 1), "ms_per_step": round(dt_c * 1e3, 3), "kept": int(ch2.last_stats()[1]["out_records"]),
                                                  "layout": "in_tail's records: map32 metadata / body headers (299 B events)"}}
        f2.close(); g2.close(); tl.close(); L.flbgpu_dev_free(d_txt)
    except Exception as e:
        out["tail_lines"] = {"error": repr(e)[:300]}
    # -- flb_sp (BASELINE configs[4] shape): GROUP BY status, AVG(latency) over a tumbling window; the chunk is resident in HBM, the
    #    window's partial aggregates are exchanged over RCCL when N > 1 (one all-gather of KB-sized group states per timer)
    try:
        import sp_synth
        m = min(n, 4_000_000)
        sdata, soff = sp_synth.config4_chunk(m, seed=0x5ca1e + rank)
        d_sd = L.flbgpu_dev_alloc(sdata.nbytes + 16); d_so = L.flbgpu_dev_alloc(soff.nbytes)
        L.flbgpu_memcpy_h2d(d_sd, sdata.ctypes.data, sdata.nbytes); L.flbgpu_memcpy_h2d(d_so, soff.ctypes.data, soff.nbytes)
        sch = g.DevChunk(d_sd, d_so, m, sdata.nbytes)
        st_ = g.StreamTask(sp_synth.CONFIG4_SQL)
        st_.set_index_base(rank << 40)
        st_.do_dev(sch); st_.timer()
        st_.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            st_.do_dev(sch)
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / steps
        prof_s = st_.profile(False)
        e = {"records_per_s_per_gpu": round(m / dt_s, 1), "ms_per_step": round(dt_s * 1e3, 3), "query": sp_synth.CONFIG4_SQL,
             "chunk_bytes": int(sdata.nbytes), "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in prof_s.items()}}
        ke_ = e["kernel_ms"].get("k_sp_extract") or 0
        if ke_:
            a_ = sdata.nbytes / (ke_ / 1e3) / 1e9
            e["roofline"] = {"kernel": "k_sp_extract", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": round(a_, 1),
                             "frac": round(a_ / HBM_PEAK_GBS, 4), "note": "61 B records: the kernel is bound by per-record work, not by bytes"}
        if dist is not None and "rccl" in out:
            t0 = time.perf_counter()
            merged = st_.timer_all_reduce(out["rccl"])
            e["all_reduce_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
            e["rccl_ranks"] = world
        else:
            merged = st_.timer()
        import msgpack as _mp
        u_ = _mp.Unpacker(raw=True, strict_map_key=False)
        u_.feed(merged)
        rows_ = [r_[1] for r_ in u_]
        e["groups"] = len(rows_)
        e["window_records"] = int(sum(r_[b"COUNT(*)"] for r_ in rows_))
        if rank == 0:
            import ref_sp
            if ref_sp.available():
                k_ = min(m, 1_000_000)
                sample = sdata[: int(soff[k_])].tobytes()
                rr = ref_sp.RefSp(sp_synth.CONFIG4_SQL)
                t0 = time.perf_counter()
                rr.do(sample); want_ = rr.timer()
                dt_r = time.perf_counter() - t0
                rr.close()
                t2_ = g.StreamTask(sp_synth.CONFIG4_SQL)
                t2_.do(sample); got_ = t2_.timer(); t2_.close()
                e["cpu_baseline"] = {"value": round(k_ / dt_r, 1), "unit": "records/s", "cores": 1, "kind": "reference",
                                     "sample": "%d records through the reference's own flb_sp (oracle/_ref/ref_sp)" % k_, "identical_output": got_ == want_}
        out["flb_sp_group_by"] = e
        st_.close(); L.flbgpu_dev_free(d_sd); L.flbgpu_dev_free(d_so)
    except Exception as e:
        out["flb_sp_group_by"] = {"error": repr(e)[:300]}
    fg1.close(); fg2.close(); pk.close()
    L.flbgpu_dev_free(d_data); L.flbgpu_dev_free(d_off)
    if "rccl" in out:
        out.pop("rccl").close()
        out["rccl_ranks"] = world
    return out


def measure_cpu(data, off, n, args):
    """cpu_baseline: the oracle's filter_parser(apache2) + filter_grep, one thread, on a bounded sample of the same
    workload; plus the reference's own regex engine on the same lines (oracle/_ref/libonig_ref.so, the real Onigmo:
    the regex half of the reference's cost) and the N-process leg."""
    import numpy as np
    import oracle_binding as ob
    ns = min(args.cpu_sample, n)
    sample = bytes(data[: int(off[ns])])
    po = ob.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
    fo = ob.FilterParser("log", [po])
    go = ob.Grep([GREP_RULE])
    t0 = time.perf_counter()
    r, parsed = fo.filter(sample)
    r2_, kept = go.filter(parsed)
    cdt = time.perf_counter() - t0
    cpu = {"value": round(ns / cdt, 1), "unit": "records/s", "cores": 1, "kind": "port",
           "sample": "first %d records of the same seeded workload through oracle filter_parser(apache2)+filter_grep, "
                     "single thread (%d host cores present)" % (ns, os.cpu_count())}
    try:
        from rxdiff import load_ref, RefRegex
        R = load_ref()
        if R is not None and hasattr(R, "ref_onig_bench"):
            # the reference's own regex engine (Onigmo 6.2.0 built from its sources) on the same lines: rows = the
            # `log` values (21 B of event framing in front of each 256 B line)
            m = min(ns, 1_000_000)
            vals = np.ascontiguousarray(np.asarray(data[: int(off[m])]).reshape(m, -1)[:, 21:]) if (int(off[m]) % m == 0) else None
            if vals is not None:
                rows = np.arange(m + 1, dtype=np.int64) * vals.shape[1]
                R.ref_onig_bench.restype = ctypes.c_double
                R.ref_onig_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
                rr = RefRegex(R, APACHE2.encode())
                matched = ctypes.c_longlong(0)
                sec = R.ref_onig_bench(rr.reg, vals.ctypes.data, rows.ctypes.data, m, ctypes.byref(matched))
                cpu["reference_onigmo"] = {"lines_per_s": round(m / sec, 1), "lines": m, "matched": int(matched.value), "cores": 1,
                                           "what": "onig_search with a region, the real Onigmo 6.2.0 compiled from the reference's sources, apache2 pattern, same lines"}
    except Exception as e:
        cpu["reference_onigmo_error"] = repr(e)[:200]
    ref_ok = False
    try:
        # the reference's OWN plugins: cb_filter of filter_parser + filter_grep (plugins/filter_parser/filter_parser.c,
        # plugins/filter_grep/grep.c and everything under them, oracle/_ref/ref_filters built by oracle/Makefile from the
        # reference's sources), flb_filter_do's loop over the two, timed inside the driver (clock_gettime around the loop)
        import ref_filters as rf
        if rf.available():
            m = min(ns, 2_000_000)
            secs, rin, rkept = rf.bench_result(rf.run([rf.bench_pair_case("log", dict(regex=APACHE2, time_fmt=TIME_FMT, time_key="time"), [GREP_RULE],
                                                                           bytes(data[: int(off[m])]), 1)], timeout=900)[0])
            assert rin == m, (rin, m)
            cpu["port"] = {"value": cpu["value"], "sample": cpu["sample"]}
            cpu.update({"value": round(rin / secs, 1), "kind": "reference", "kept": int(rkept),
                        "sample": "first %d records of the same seeded workload through the reference's own cb_filter of filter_parser(apache2) and "
                                  "filter_grep (compiled from its sources: oracle/_ref/ref_filters), single thread (%d host cores present)" % (m, os.cpu_count())})
            ref_ok = True
    except Exception as e:
        cpu["reference_error"] = repr(e)[:200]
    try:
        nproc = os.cpu_count() or 1
        m = min(n, 2_000_000)
        if ref_ok:
            import ref_filters as rf
            from concurrent.futures import ThreadPoolExecutor
            per = max(2000, m // nproc)
            shards = [bytes(data[int(off[i * per % max(1, m - per)]): int(off[i * per % max(1, m - per) + per])]) for i in range(nproc)]
            iters = max(1, int(6.0 * cpu["value"] / per))               # ~6 s of work per process at the single-thread rate
            def one(sh):
                return rf.bench_result(rf.run([rf.bench_pair_case("log", dict(regex=APACHE2, time_fmt=TIME_FMT, time_key="time"), [GREP_RULE], sh, iters)], timeout=900)[0])
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=nproc) as ex:
                res = list(ex.map(one, shards))
            wall = time.perf_counter() - t0
            v = sum(r[1] for r in res) * iters / wall
            cpu["nproc"] = {"processes": nproc, "value": round(v, 1), "unit": "records/s", "kind": "reference",
                            "note": "%d independent processes of the reference's filter pair, %d records x %d passes each, wall-clock aggregate "
                                    "(process start-up included)" % (nproc, per, iters)}
        else:
            v = cpu_nproc_leg((bytes(data[: int(off[m])]), np.array(off[: m + 1])), nproc)
            cpu["nproc"] = {"processes": nproc, "value": round(v, 1), "unit": "records/s", "kind": "port",
                            "note": "%d independent processes, each the oracle pair on its own shard for ~6 s, wall-clock aggregate" % nproc}
    except Exception as e:
        cpu["nproc_error"] = repr(e)[:200]
    return cpu


def measure_host_level(g, data, off, n):
    """what one cb_filter / flb_filter_do call sees from host memory (PCIe inclusive; never `value`): the chain on
    an engine-sized chunk (~2 MB, what the engine appends at a time) and on a 28 MB chunk"""
    out = {}
    p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
    fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE])
    ch = g.FilterChain([fp, fg])
    for name, nrec in (("chunk_2MB", 7000), ("chunk_28MB", 100000)):
        nrec = min(nrec, n)
        blob = bytes(data[: int(off[nrec])])
        ch.filter(blob)
        reps = 20 if nrec < 50000 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ch.filter(blob)
        dt = (time.perf_counter() - t0) / reps
        out[name] = {"records": nrec, "bytes": len(blob), "ms_per_call": round(dt * 1e3, 3), "records_per_s": round(nrec / dt, 1),
                     "in_GBps": round(len(blob) / dt / 1e9, 2)}
    fp.close(); fg.close(); p.close()
    return out


def main():
    # stdout carries exactly ONE line (the JSON): libraries that chat on fd 1 (RCCL prints its own path there)
    # are pointed at stderr for the whole run, the line goes to the saved descriptor at the very end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--records", type=int, default=10_000_000, help="records per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=5_000_000, help="records timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the log_to_metrics / JSON side measurements")
    args = ap.parse_args()

    import flbamd_loader
    import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # ---- synthetic shard for this rank (seeded; per-GPU work is fixed => weak scaling)
    n = args.records
    t0 = time.time()
    data, off, ep = synth.apache_records(n, seed=synth.SEED + rank)
    gen_s = time.time() - t0
    in_bytes = int(data.nbytes)

    # ---- CPU baselines first (rank 0, N = 1), before this process owns a GPU context: the forked workers of the
    #      N-process leg only run the oracle
    cpu = None
    if not args.no_cpu and world == 1 and rank == 0:
        cpu = measure_cpu(data, off, n, args)

    import torch
    dist = None
    if world > 1 or os.environ.get("FLBGPU_BENCH_FORCE_DIST"):       # (forced: exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    g = flbamd_loader.load()
    g.init(local_rank if world > 1 else 0)
    L = g.lib()

    d_data = L.flbgpu_dev_alloc(in_bytes)
    d_off = L.flbgpu_dev_alloc(off.nbytes)
    assert d_data and d_off, g.last_error()
    L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, in_bytes)
    L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
    chunk = g.DevChunk(d_data, d_off, n, in_bytes)

    parser = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time", name="apache2")
    fparser = g.FilterParser("log", [parser])
    fgrep = g.FilterGrep([GREP_RULE])

    chain = g.FilterChain([fparser, fgrep])

    def step():
        # flb_filter_do over the two filters (flbgpu_filter_chain_run_dev): the pair path of fused_kernels.inc
        r2, o2 = chain.filter_dev(chunk)
        assert r2 == g.MODIFIED, g.last_error()
        return o2, r2

    for _ in range(args.warmup):
        step()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    fparser.profile(True)
    fgrep.profile(True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        o2, r2 = step()
    sync_all()
    dt = time.perf_counter() - t0
    prof = dict(fparser.profile_read())
    prof.update({"grep:" + k if k == "k_scan" else k: v for k, v in fgrep.profile_read().items()})
    fparser.profile(False); fgrep.profile(False)
    st = chain.last_stats()
    parsed_bytes = int(st[0]["out_bytes"])             # what filter_parser alone would emit (counted, not written)
    kept_bytes = int(o2.bytes)
    kept_records = int(st[1]["out_records"])
    fused = "k_pg_emit" in prof
    # the parsed chunk itself, for the side measurements below (one unfused filter_parser run, untimed)
    r1, o1 = fparser.filter_dev(chunk)
    assert r1 == g.MODIFIED and int(o1.bytes) == parsed_bytes, (g.last_error(), int(o1.bytes), parsed_bytes)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- side measurements (never part of `value`): the other rows of the hot-path scope table
    secondary = None
    if not args.no_secondary:
        try:
            secondary = measure_secondary(g, torch, dist, rank, world, o1, n, args, raw_chunk=chunk)
            if rank == 0 and world == 1:
                secondary["host_level"] = measure_host_level(g, data, off, n)
        except Exception as e:                      # the headline line must survive a failure here
            secondary = {"error": repr(e)[:300]}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_records = n * world * args.steps
    value = total_records / dt
    # dominant kernel and its algorithmic bytes (DESIGN.md "Measurement")
    dom = max(prof.items(), key=lambda kv: kv[1][0])[0] if prof else None
    value_bytes = in_bytes - 21 * n                      # the `log` values (277 B event = 21 B framing + 256 B line)
    alg_bytes_per_launch = {
        "k_parser_reg": in_bytes,                        # the single pass: every chunk byte once (header + value)
        "k_parser_tile": in_bytes,
        "k_parser_locate": in_bytes,                     # reads every chunk byte once
        "k_parser_rx": value_bytes,                      # the capture program consumes each value byte once
        "k_parser_finish": 8 * n,                        # time field + sizes
        "k_parser_emit": value_bytes + parsed_bytes,     # re-reads the values, writes the output once
        "k_pg_emit": 2 * kept_bytes,                     # kept records: their field bytes in, the parsed records out
        "k_grep_match": parsed_bytes,
        "k_gather": 2 * kept_bytes,
    }
    roof = None
    if dom:
        ms, launches = prof[dom]
        avg_s = ms / 1e3 / max(launches, 1)
        ach = alg_bytes_per_launch.get(dom, in_bytes) / avg_s / 1e9
        traffic, traffic_src = recorded_traffic(dom, n)
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(avg_s * 1e3, 4), "launches": int(launches),
                "algorithmic_bytes_per_launch": int(alg_bytes_per_launch.get(dom, in_bytes))}
        # the whole step against the same roof: SURVEY 8(d) wire-format bytes of the fused pair = input once + what
        # filter_parser emits + what filter_grep keeps (552 + 275 x keep B/record), over the step's kernel time
        step_bytes = in_bytes + parsed_bytes + kept_bytes
        kern_ms = sum(v[0] for v in prof.values()) / args.steps
        tr = [recorded_traffic(k, n)[0] for k in prof]
        roof["step"] = {"algorithmic_bytes": int(step_bytes), "kernel_ms": round(kern_ms, 3), "achieved": round(step_bytes / (kern_ms / 1e3) / 1e9, 1),
                        "frac": round(step_bytes / (kern_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
                        "traffic": int(sum(tr)) if tr and all(t is not None for t in tr) else None}
    kernels = {k: {"total_ms": round(v[0], 3), "launches": int(v[1])} for k, v in prof.items()}

    if isinstance(secondary, dict) and "record_indexer" in secondary:
        # what the step would sustain if it were handed raw bytes and had to find the rows first
        step_s = dt / args.steps
        secondary["record_indexer"]["headline_step_plus_indexing_records_per_s_per_gpu"] = round(
            n / (step_s + secondary["record_indexer"]["ms_per_step"] / 1e3), 1)
    line = {
        "metric": "log records/sec (256B apache-combined lines) through parser+grep",
        "value": round(value, 1), "unit": "records/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "filter_parser(conf/parsers.conf apache2, Key_Name log) -> filter_grep(Regex code ^5\\d\\d$) "
                               "on %d x 256B apache-combined lines per GPU (277B V2 events), flb_filter_do on device, %s" % (n, "fused pair: one pass over the chunk (event decode, capture program, time, rules on the capture spans), only the kept records written" if fused else "unfused"),
                   "records_per_gpu": n, "in_bytes": in_bytes, "parsed_bytes": parsed_bytes, "kept_records": int(kept_records),
                   "row_offsets": "part of the device-resident chunk (every filter's output carries them); finding them from "
                                  "raw bytes is secondary.record_indexer",
                   "seed": synth.SEED, "parallelism": "shard%d" % world, "gen_seconds": round(gen_s, 1)},
        "roofline": roof, "cpu_baseline": cpu, "kernels": kernels, "secondary": secondary,
        "rccl_ranks": world if dist is not None else 0,
    }
    if dist is not None:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()
