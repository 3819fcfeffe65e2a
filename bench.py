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
# with four class / quantifier patterns in each set.  Data and rules live in tests/ndjson_synth.py (the parity
# tests use the same): the first instance keeps about two thirds of the lines, the second under half of those.
from ndjson_synth import GREP32_REGEX, GREP32_EXCLUDE          # noqa: E402  (tests/ is on sys.path)


PMC_FILE = os.path.join("profiles", "r6_pmc_hbm_bench_10M.json")


def _sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


MIXED_LENS = [80, 120, 160, 256, 400, 600]


def mixed_shape_records():
    """the pool of secondary.mixed_shapes: 60 000 events whose lines are 80 .. 600 bytes long, 10 % with a four-key body (the value is
    looked up among other keys), 1 % legacy [ts, map] events"""
    import random as _rnd, re as _re
    import numpy as np
    import synth as _synth
    rng_ = _rnd.Random(0x51ab)
    lens_ = MIXED_LENS
    import re as _re
    cut_ = _re.compile(rb'^(\S+ - \S+ \[[^\]]+\] "\S+ /)(\S*)( HTTP/1.1" \d+ \d+)( ".*)$')
    pools = []
    for ll in lens_:
        d_, o_, _e = _synth.apache_records(10000, line_len=max(ll, 256), seed=0xF1B17 + ll)
        rec = int(o_[1] - o_[0])
        arr = np.asarray(d_).reshape(10000, rec)
        full = [bytes(arr[i, rec - max(ll, 256):]) for i in range(10000)]
        if ll < 256:                   # shorter lines: the request path cut, no referer / agent (the pattern's optional tail)
            short = []
            for ln in full:
                m_ = cut_.match(ln)
                keep = max(0, ll - len(m_.group(1)) - len(m_.group(3)))
                short.append(m_.group(1) + m_.group(2)[:keep] + m_.group(3))
            full = short
        pools.append(full)
    recs_ = []
    for i in range(60000):
        line = pools[rng_.randrange(len(lens_))][rng_.randrange(10000)]
        r_ = rng_.random()
        sec_ = 1700000000 + i
        if r_ < 0.01:
            recs_.append(_synth.legacy_record(_synth.ext_ts(sec_, 5), {"log": line}))
        elif r_ < 0.11:
            body = [("stream", "stdout"), ("log", line), ("pod", "api-%d" % rng_.randrange(100)), ("n", rng_.randrange(1000))]
            rng_.shuffle(body)
            recs_.append(_synth.v2_record(sec_, 7, dict(body)))
        else:
            recs_.append(_synth.v2_record(sec_, 7, {"log": line}))

    return recs_


def kernel_source_sha():
    """identity of the headline kernels' sources: the PMC summary is only quoted while it was taken from this build"""
    import hashlib
    h = hashlib.sha256()
    for f in ("tile_kernels.inc", "fused_kernels.inc", "kdev.inc", "nfa_dev.inc", "dev.hpp", "fx.cpp"):
        try:
            h.update(open(os.path.join(ROOT, "fluent-bit_amd", "csrc", f), "rb").read())
        except OSError:
            return None
    return h.hexdigest()[:16]


def pmc_pass(n, steps=5, warmup=1, timeout=900):
    """--pmc: HBM traffic of this very build, measured now -- the headline step once more under rocprofv3, one pass per counter
    (FETCH_SIZE, WRITE_SIZE: separate passes, kernel trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), in child
    processes.  Returns {kernel base name: {"FETCH_SIZE": KiB, "WRITE_SIZE": KiB}} averaged per launch, or (None, reason)."""
    import csv, glob, shutil, subprocess, tempfile, collections
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    tmp = tempfile.mkdtemp(prefix="flbgpu_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--no-cpu", "--no-secondary", "--no-pmc", "--steps", str(steps), "--warmup", str(warmup), "--records", str(n)]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s pass failed (rc %d)" % (ctr, r.returncode)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    out[row["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    except Exception as e:
        return None, repr(e)[:160]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()}, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (bench.py --pmc; FETCH x 2, WRITE x 1)"


def recorded_traffic(kernel, n):
    """HBM bytes per launch of `kernel` from the committed PMC pass of this same command (rocprofv3 cannot run inside
    the timed process): FETCH_SIZE + WRITE_SIZE (KiB), FETCH doubled as /opt/skills/guides/MI355X_MICROARCH.md's HBM
    section prescribes for gfx950 (and profiles/r2_counter_calibration.json confirmed on this pool for every read
    shape measured).  (None, reason) when the workload differs from the profiled one OR the file was recorded from
    other kernel sources than the ones in this tree (tools/profile_bench.sh stamps `kernel_source_sha`)."""
    path = os.path.join(ROOT, PMC_FILE)
    if n != 10_000_000:
        return None, "traffic is recorded for the 10 M-record workload only"
    if not os.path.exists(path):
        return None, PMC_FILE + " absent"
    try:
        doc = json.load(open(path))
        if doc.get("kernel_source_sha") != kernel_source_sha():
            return None, "%s was recorded from other kernel sources (%s, tree has %s): not quoted" % (PMC_FILE, doc.get("kernel_source_sha"), kernel_source_sha())
        ks = doc["kernels"]
        base = lambda k: k.split("::")[-1].split("<")[0].split("(")[0]
        hit = [v for k, v in ks.items() if base(k) == kernel or (kernel.endswith("k_scan") and base(k).startswith("k_scan_"))]
        if not hit:
            return None, "kernel not in " + PMC_FILE
        if len(hit) > 1:                                  # the scan is three small kernels
            hit = [{c: sum(h.get(c, 0) for h in hit) for c in ("FETCH_SIZE", "WRITE_SIZE")}]
        b = hit[0].get("FETCH_SIZE", 0) * 1024 * 2.0 + hit[0].get("WRITE_SIZE", 0) * 1024 * 1.0
        return int(b), PMC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x 2, WRITE x 1)"
    except Exception as e:
        return None, repr(e)[:120]


def usable_cores():
    """cores this process may really use: the scheduler affinity mask cut by the cgroup CPU quota (os.cpu_count() is
    the machine's, not the lease's)"""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    n = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return n, {"os_cpu_count": os.cpu_count(), "sched_affinity": aff, "cgroup_quota_cores": quota}


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


def measure_config2(g, torch, L, rank, world, args):
    """BASELINE configs[2] as stated: 100 M NDJSON lines on one GPU -> log events (flb_pack_json) -> filter_grep with
    32 patterns (two instances: 16 Regex in OR mode, then 16 Exclude in OR mode).  The lines are 1 M distinct seeded lines
    (tests/ndjson_synth.py) laid out ten times as one resident chunk of 10 M lines (2.2 GB of text -- far beyond the 256 MB of
    Infinity Cache); one "pass" = that chunk through JSON -> events -> grep -> grep, and the 100 M lines are ten passes.  Every
    stage is priced against the HBM roof on its wire-format bytes: text + events; events in + kept out, twice."""
    import numpy as np
    import ndjson_synth as ns
    out = {}
    total = int(args.ndjson_lines)
    per_chunk = min(total, 10_000_000)
    nbase = min(per_chunk, 1_000_000)
    t0 = time.time()
    base = ns.lines(nbase, seed=7 + rank)
    gen_s = time.time() - t0
    reps = max(1, per_chunk // nbase)
    per_chunk = reps * nbase
    passes = max(1, total // per_chunk)
    text = b"".join(base)
    blen = len(text)
    boff = np.zeros(nbase + 1, dtype=np.uint64)
    np.cumsum(np.fromiter((len(x) for x in base), dtype=np.uint64, count=nbase), out=boff[1:])
    off = (boff[:-1][None, :] + (np.arange(reps, dtype=np.uint64) * np.uint64(blen))[:, None]).reshape(-1)
    off = np.ascontiguousarray(np.concatenate([off, np.array([reps * blen], dtype=np.uint64)]))
    d_data = L.flbgpu_dev_alloc(reps * blen + 16); d_off = L.flbgpu_dev_alloc(off.nbytes)
    assert d_data and d_off, g.last_error()
    for r in range(reps):
        L.flbgpu_memcpy_h2d(d_data + r * blen, text, blen)
    L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
    nl, tbytes = per_chunk, reps * blen
    chunk = g.DevChunk(d_data, d_off, nl, tbytes)
    pk = g.JsonPacker()
    fg1 = g.FilterGrep(GREP32_REGEX, "OR"); fg2 = g.FilterGrep(GREP32_EXCLUDE, "OR")
    ev = pk.run_dev(chunk, events=True, ts=(1, 0))
    r1, k1 = fg1.filter_dev(ev)
    r2, k2 = fg2.filter_dev(k1)
    assert r1 == g.MODIFIED and r2 == g.MODIFIED, (r1, r2, g.last_error())
    torch.cuda.synchronize()
    t_j = t_1 = t_2 = 0.0
    fg1.profile(True); fg2.profile(True)
    for _ in range(passes):
        t0 = time.perf_counter()
        ev = pk.run_dev(chunk, events=True, ts=(1, 0))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        r1, k1 = fg1.filter_dev(ev)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        r2, k2 = fg2.filter_dev(k1)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        t_j += t1 - t0; t_1 += t2 - t1; t_2 += t3 - t2
    p1, p2 = fg1.profile_read(), fg2.profile_read()
    fg1.profile(False); fg2.profile(False)
    # the two instances as flb_filter_do runs them: one chain call (flbgpu_filter_chain_run_dev decides both in one pass of the events)
    chn = g.FilterChain([fg1, fg2])
    rc, kc = chn.filter_dev(ev)
    assert rc == g.MODIFIED and int(kc.bytes) == int(k2.bytes), (rc, int(kc.bytes), int(k2.bytes), g.last_error())
    torch.cuda.synchronize()
    fg1.profile(True)
    t_c = 0.0
    for _ in range(passes):
        t0 = time.perf_counter()
        rc, kc = chn.filter_dev(ev)
        torch.cuda.synchronize()
        t_c += time.perf_counter() - t0
    pc = fg1.profile_read()
    fg1.profile(False)
    chain_stats = chn.last_stats()
    kc_n = min(nl, 100_000)
    kc_off = np.zeros(kc_n + 1, dtype=np.uint64)
    L.flbgpu_memcpy_d2h(kc_off.ctypes.data, kc.row_off, kc_off.nbytes)
    kc_buf = ctypes.create_string_buffer(max(int(kc_off[-1]), 1))
    L.flbgpu_memcpy_d2h(kc_buf, kc.data, int(kc_off[-1]))
    kc_sample = kc_buf.raw[: int(kc_off[-1])]
    # (an instance's output stands until its next call: the one-by-one outputs again, for the parity sample below)
    r1, k1 = fg1.filter_dev(ev)
    r2, k2 = fg2.filter_dev(k1)
    torch.cuda.synchronize()
    ev_b, k1_b, k2_b = int(ev.bytes), int(k1.bytes), int(k2.bytes)
    n1, n2 = int(fg1.counts()[1]), int(fg2.counts()[1])
    lines = nl * passes

    def stage(name, sec, nbytes, extra=None):
        gbs = nbytes * passes / sec / 1e9
        d = {"ms_per_10M_lines": round(sec / passes * 1e3 * (10_000_000 / nl), 3), "algorithmic_bytes_per_pass": int(nbytes),
             "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}}
        d.update(extra or {})
        return name, d

    st = dict([stage("json_to_events", t_j, tbytes + ev_b, {"lines_per_s": round(lines / t_j, 1), "text_bytes": tbytes, "event_bytes": ev_b}),
               stage("grep_regex_or_16", t_1, ev_b + k1_b, {"records_per_s": round(lines / t_1, 1), "kept": n1, "keep_ratio": round(n1 / nl, 4),
                                                            "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in p1.items()}}),
               stage("grep_exclude_or_16", t_2, k1_b + k2_b, {"records_per_s": round(n1 * passes / t_2, 1), "kept": n2, "keep_ratio_of_input": round(n2 / max(n1, 1), 4),
                                                              "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in p2.items()}})])
    st.update([stage("grep_chain_2x16", t_c, ev_b + k2_b, {"records_per_s": round(lines / t_c, 1), "kept": int(chain_stats[1]["out_records"]),
                                                         "what": "both instances in one flbgpu_filter_chain_run_dev call: events in once, kept records out once",
                                                         "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in pc.items()},
                                                         "stats": chain_stats})])
    sep_s = t_j + t_1 + t_2
    tot_s = t_j + t_c
    tot_b = (tbytes + ev_b) + (ev_b + k2_b)
    e = {"lines": lines, "distinct_lines": nbase, "lines_per_chunk": nl, "passes": passes, "gen_seconds": round(gen_s, 1),
         "rules": "16 Regex (Logical_Op OR) then 16 Exclude (Logical_Op OR): two filter_grep instances chained; tests/ndjson_synth.py",
         "seconds_total": round(tot_s, 4), "lines_per_s_per_gpu": round(lines / tot_s, 1),
         "seconds_total_instances_called_one_by_one": round(sep_s, 4), "stages": st,
         "roofline": {"what": "whole step: JSON -> events, then the chain of the two grep instances; wire-format bytes of both calls (in once + out once)",
                      "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_pass": int(tot_b),
                      "achieved": round(tot_b * passes / tot_s / 1e9, 1), "frac": round(tot_b * passes / tot_s / 1e9 / HBM_PEAK_GBS, 4)}}
    if rank == 0 and world == 1 and not args.no_cpu:
        # parity on a sample, through the reference's own code: the first 100 k lines -> the oracle's flb_pack_json (pinned on the
        # real yyjson) must give the device's events; those events through the reference's own cb_filter of filter_grep
        # (oracle/_ref/ref_filters), both instances, must give the device's kept records for the same rows
        try:
            import jsonfuzz as jf
            import ref_filters as rf
            m = min(nl, 100_000)
            offs = {}
            for nm, chv in (("ev", ev), ("k1", k1), ("k2", k2)):
                o_ = np.zeros(m + 1, dtype=np.uint64)
                L.flbgpu_memcpy_d2h(o_.ctypes.data, chv.row_off, o_.nbytes)
                hb = ctypes.create_string_buffer(max(int(o_[-1]), 1))
                L.flbgpu_memcpy_d2h(hb, chv.data, int(o_[-1]))
                offs[nm] = hb.raw[: int(o_[-1])]
            par = {"sample_lines": m}
            # (one oracle call per line: the event wrapper 92 92 d7 00 <sec> <nsec> 80 <map> is put around each object here)
            import struct as _st
            o_ = jf.oracle()
            m_ev = min(m, 20_000)
            head = b"\x92\x92\xd7\x00" + _st.pack(">II", 1, 0) + b"\x80"
            t0 = time.perf_counter()
            want_ev = b"".join(head + o_(ln)[1] for ln in base[:m_ev])
            cdt = time.perf_counter() - t0
            ev_off = np.zeros(m_ev + 1, dtype=np.uint64)
            L.flbgpu_memcpy_d2h(ev_off.ctypes.data, ev.row_off, ev_off.nbytes)
            par["events_match_oracle"] = bool(want_ev == offs["ev"][: int(ev_off[-1])])
            par["events_sample_lines"] = m_ev
            e["stages"]["json_to_events"]["cpu_port_lines_per_s"] = round(m_ev / cdt, 1)
            if rf.available():
                t0 = time.perf_counter()
                res = rf.run([rf.grep_case(GREP32_REGEX, "OR", offs["ev"])], timeout=600)
                w1 = res[0][1] if res[0][0] == rf.MODIFIED else offs["ev"]
                res2 = rf.run([rf.grep_case(GREP32_EXCLUDE, "OR", w1)], timeout=600)
                w2 = res2[0][1] if res2[0][0] == rf.MODIFIED else w1
                cdt = time.perf_counter() - t0
                par["grep_regex_or_matches_reference"] = bool(w1 == offs["k1"])
                par["grep_exclude_or_matches_reference"] = bool(w2 == offs["k2"])
                par["grep_chain_matches_reference"] = bool(w2 == kc_sample)
                par["reference_records_per_s_both_instances"] = round(m / cdt, 1)
                par["kind"] = "reference (oracle/_ref/ref_filters: the reference's own cb_filter of filter_grep)"
            e["parity_sample"] = par
        except Exception as ex:
            e["parity_sample"] = {"error": repr(ex)[:300]}
    out["config2_ndjson_grep32"] = e
    # the events chunk stays for the log_to_metrics histogram on float values (measured ULP distance to the real cmetrics)
    out["_events_chunk"] = (ev, nbase, reps, base)
    out["_cleanup"] = (pk, fg1, fg2, d_data, d_off)
    return out


def measure_l2m_float(g, torch, L, evc, rank, world, args):
    import numpy as np
    ev, nbase, reps, base = evc
    nrec = int(ev.n)
    props = [("label_field", "level")]
    # both sum orders, each timed and each held against the real cmetrics: "reference" (the C ABI's default since round 6: cmetrics' own
    # sequential f64 sum, k_l2m_seqsum) and "exact" (fixed-point digits rounded once)
    snaps, ms = {}, {}
    for so in ("reference", "exact"):
        f = g.FilterLogToMetrics("histogram", props, value_field="latency")
        f.set_sum_order(so == "reference")
        f.filter_dev(ev)
        torch.cuda.synchronize()
        snaps[so] = f.snapshot()
        f.close()
        f = g.FilterLogToMetrics("histogram", props, value_field="latency")
        f.set_sum_order(so == "reference")
        t0 = time.perf_counter()
        for _ in range(3):
            f.filter_dev(ev)
        torch.cuda.synchronize()
        ms[so] = (time.perf_counter() - t0) / 3
        f.close()
    snap, dt = snaps["reference"], ms["reference"]
    e = {"sum_order": "reference (the default: the histogram sum as cmetrics adds it up, one f64 addition per observation in record order)",
         "records_per_s_per_gpu": round(nrec / dt, 1), "ms_per_step": round(dt * 1e3, 3), "ms_per_step_sum_order_exact": round(ms["exact"] * 1e3, 3),
         "observations": int(sum(x["count"] for x in snap)),
         "series": len(snap), "value_field": "latency (msgpack float64 from the NDJSON text)", "label": "level"}
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libcmetrics_ref.so")
    if rank == 0 and world == 1 and not args.no_cpu and os.path.exists(ref_so):
        import json as _json
        import struct as _st
        R = ctypes.CDLL(ref_so)
        R.refcmt_new.restype = ctypes.c_void_p
        R.refcmt_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        R.refcmt_update_many.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p)]
        R.refcmt_nseries.argtypes = [ctypes.c_void_p]; R.refcmt_nbuckets.argtypes = [ctypes.c_void_p]
        R.refcmt_label.restype = ctypes.c_char_p; R.refcmt_label.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        R.refcmt_sum.restype = ctypes.c_double; R.refcmt_sum.argtypes = [ctypes.c_void_p, ctypes.c_int]
        R.refcmt_bucket.restype = ctypes.c_uint64; R.refcmt_bucket.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        R.refcmt_count.restype = ctypes.c_uint64; R.refcmt_count.argtypes = [ctypes.c_void_p, ctypes.c_int]
        R.refcmt_free.argtypes = [ctypes.c_void_p]
        table, idx1, val1 = [], np.empty(nbase, dtype=np.int32), np.empty(nbase, dtype=np.float64)
        pos = {}
        for i, ln in enumerate(base):
            d = _json.loads(ln)
            lv = d["level"].encode()
            if lv not in pos:
                pos[lv] = len(table); table.append(lv)
            idx1[i] = pos[lv]; val1[i] = d["latency"]
        idx, vals = np.ascontiguousarray(np.tile(idx1, reps)), np.ascontiguousarray(np.tile(val1, reps))
        keys = (ctypes.c_char_p * 1)(b"level")
        tab = (ctypes.c_char_p * len(table))(*table)
        h = R.refcmt_new(2, 1, keys, -1, None)
        t0 = time.perf_counter()
        rc = R.refcmt_update_many(h, len(vals), vals.ctypes.data, idx.ctypes.data, tab)
        cdt = time.perf_counter() - t0
        nb = R.refcmt_nbuckets(h)
        bits = lambda x: _st.unpack("<q", _st.pack("<d", x))[0]
        worst, worst_exact, structure = 0, 0, rc == 0 and R.refcmt_nseries(h) == len(snap)
        rel = 0.0
        for si in range(R.refcmt_nseries(h)):
            want_l = R.refcmt_label(h, si, 0)
            got = snap[si] if si < len(snap) else None
            if got is None or got["labels"] != (want_l,):
                structure = False
                continue
            structure = structure and got["count"] == R.refcmt_count(h, si) and list(got["buckets"]) == [R.refcmt_bucket(h, si, b) for b in range(nb + 1)]
            ws = R.refcmt_sum(h, si)
            worst = max(worst, abs(bits(ws) - bits(got["sum"])))
            worst_exact = max(worst_exact, abs(bits(ws) - bits(snaps["exact"][si]["sum"])))
            rel = max(rel, abs(ws - snaps["exact"][si]["sum"]) / abs(ws) if ws else 0.0)
        R.refcmt_free(h)
        e["vs_cmetrics"] = {"kind": "reference (oracle/_ref/libcmetrics_ref.so: cmt_histogram_observe in record order)", "observations": int(len(vals)),
                            "series_labels_buckets_counts_identical": bool(structure), "max_ulp_vs_cmetrics": int(worst),
                            "sum_order_exact_max_ulp_vs_cmetrics": int(worst_exact), "sum_order_exact_max_rel_err": rel,
                            "note": "sum_order reference = cmetrics' own bits (0 ULP); sum_order exact = the true sum rounded once -- cmetrics' sequential additions drift from it as n grows",
                            "cmetrics_observations_per_s": round(len(vals) / cdt, 1)}
    return e


def measure_secondary(g, torch, dist, rank, world, parsed_chunk, n, args, raw_chunk=None, shared_gpu=False):
    """filter_log_to_metrics on the parsed chunk (BASELINE configs[3] shape: counter + histogram, partial
    aggregates all-reduced over RCCL when N > 1) and NDJSON -> msgpack events -> 32-rule filter_grep
    (configs[2] shape).  Reported next to the headline number, never folded into it."""
    import json as _json
    import random
    out = {}
    steps = 3
    L = g.lib()
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
    # -- log_to_metrics (BASELINE configs[3]: counter + histogram over 1 B records sharded across the GPUs: every rank runs its share
    #    -- l2m_records / world, as passes over its resident parsed chunk -- and the partial aggregates are all-reduced once per flush)
    import numpy as np
    share = max(n, int(args.l2m_records) // max(world, 1) if world > 1 else int(args.l2m_records) // 8)
    passes = max(1, (share + n - 1) // n)

    def reduce_l2m(f):
        """one flush: the ranks' partial aggregates merged (RCCL inside libflbgpu.so; gloo through the same row algebra when
        the ranks share a GPU)"""
        if dist is None:
            return None, None
        if shared_gpu:
            t0 = time.perf_counter()
            kr = g.l2m_all_reduce(f, dist, device="cpu")
            if getattr(f, "sum_order", 0) == 2 and f.mode == 2:
                g.l2m_chain(kr[0], dist, f, device="cpu")              # (the chain the C entry point runs over RCCL, here over gloo)
            return kr, time.perf_counter() - t0
        if "rccl" not in out:
            def exchange(raw):
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            out["rccl"] = g.RcclComm(world, rank, exchange)
        t0 = time.perf_counter()
        kr = g.l2m_all_reduce_rccl(f, out["rccl"])
        return kr, time.perf_counter() - t0

    # the histogram in both sum orders, each named: "exact" (digits that merge by addition: what N ranks all-reduce) and the reference's
    # order -- 1 on one GPU (k_l2m_seqsum every call), 2 across ranks (the observations kept, folded rank after rank at the flush)
    ref_order = 1 if dist is None else 2
    for name, mode, props, vf, so in (("l2m_counter", "counter", [("label_field", "method"), ("label_field", "code")], None, None),
                                      ("l2m_histogram", "histogram", [("label_field", "code")], "size", 0),
                                      ("l2m_histogram_reference_order", "histogram", [("label_field", "code")], "size", ref_order)):
        f = g.FilterLogToMetrics(mode, props, value_field=vf)
        if so is not None: f.set_sum_order(so)
        f.set_index_base(rank << 40)
        f.filter_dev(parsed_chunk)
        torch.cuda.synchronize()
        f.close()
        f = g.FilterLogToMetrics(mode, props, value_field=vf)      # (fresh state: the timed passes are the whole share)
        if so is not None: f.set_sum_order(so)
        f.set_index_base(rank << 40)
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(passes):
            f.filter_dev(parsed_chunk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        e = {"records_per_s_per_gpu": round(n * passes / dt, 1), "ms_per_10M_records": round(dt / passes * 1e3 * (10_000_000 / n), 3),
             "records_per_gpu": n * passes, "passes_over_resident_chunk": passes}
        if so is not None:
            e["sum_order"] = {0: "exact (fixed-point digits, rounded once at read-out)", 1: "reference (cmetrics' sequential f64 sum, every call)",
                              2: "reference across ranks (observations kept, the flush folds them rank after rank: in all_reduce_ms)"}[so]
        kr, ar_s = reduce_l2m(f)
        if kr is not None:
            e["all_reduce_ms"] = round(ar_s * 1e3, 3)
            e["ranks"] = world
            e["records_all_ranks"] = n * passes * world
            snap = f.snapshot(kr)
        else:
            snap = f.snapshot()
        e["series"] = len(snap)
        e["observations"] = int(sum(x["value"] for x in snap)) if mode == "counter" else int(sum(x["count"] for x in snap))
        out[name] = e
        f.close()
    if dist is not None:
        # the merged result against ONE rank's pass over the concatenated shards, at a size rank 0 can redo alone: every rank takes
        # the first m records of its shard, the partial aggregates are all-reduced, rank 0 regenerates all the shards' prefixes
        # (same seeds), runs them as one chunk and compares the snapshots -- label order included (global first-appearance order)
        try:
            import hashlib
            import synth
            m = min(n, 200_000)
            cut = np.zeros(1, dtype=np.uint64)
            L.flbgpu_memcpy_d2h(cut.ctypes.data, parsed_chunk.row_off + 8 * m, 8)
            sub = g.DevChunk(parsed_chunk.data, parsed_chunk.row_off, m, int(cut[0]))
            chk = {}
            for mode, props, vf in (("counter", [("label_field", "method"), ("label_field", "code")], None), ("histogram", [("label_field", "code")], "size")):
                f = g.FilterLogToMetrics(mode, props, value_field=vf)
                f.set_index_base(rank << 40)
                f.filter_dev(sub)
                kr, _ = reduce_l2m(f)
                merged = f.snapshot(kr)
                f.close()
                if rank == 0:
                    parts = [synth.apache_records(m, seed=synth.SEED + r)[0] for r in range(world)]
                    blob = b"".join(bytes(p_) for p_ in parts)
                    pz = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
                    fz = g.FilterParser("log", [pz])
                    rz, oz = fz.filter(blob)
                    f1 = g.FilterLogToMetrics(mode, props, value_field=vf)
                    f1.filter(oz)
                    single = f1.snapshot()
                    f1.close(); fz.close(); pz.close()
                    key = lambda sn: hashlib.sha256(repr([(x["labels"], x.get("value"), x.get("buckets"), x.get("count"), x.get("sum")) for x in sn]).encode()).hexdigest()[:16]
                    chk[mode] = {"merged_sha": key(merged), "single_rank_sha": key(single), "equal": key(merged) == key(single), "series": len(single)}
            if rank == 0:
                out["l2m_merge_check"] = {"records_per_rank": m, "ranks": world, **chk}
        except Exception as ex:
            out["l2m_merge_check"] = {"error": repr(ex)[:300]}
    # -- BASELINE configs[2]: NDJSON lines -> events -> two filter_grep instances (32 rules)
    try:
        out.update(measure_config2(g, torch, L, rank, world, args))
    except Exception as e:
        out["config2_ndjson_grep32"] = {"error": repr(e)[:300]}
    evc, cln = out.pop("_events_chunk", None), out.pop("_cleanup", None)
    if evc is not None:
        # -- log_to_metrics histogram of a FLOAT field (the events' "latency", labelled by "level"): the device's sum is the exact
        #    sum rounded once, the reference adds f64 values in arrival order -- the measured distance to the REAL cmetrics at this size
        try:
            out["l2m_histogram_float"] = measure_l2m_float(g, torch, L, evc, rank, world, args)
        except Exception as e:
            out["l2m_histogram_float"] = {"error": repr(e)[:300]}
    if cln is not None:
        pk_, fa_, fb_, da_, db_ = cln
        fa_.close(); fb_.close(); pk_.close()
        L.flbgpu_dev_free(da_); L.flbgpu_dev_free(db_)
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
                              # (round 6: priced on the algorithmic bytes of the WHOLE step -- msgpack in once, JSON out once; the size pass's second
                              # read of the chunk is traffic, not work.  The emit kernel alone is given beside it.)
                              "roofline": {"what": "whole step (k_fmt_size + scan + k_fmt_emit) on msgpack in + JSON out, once each", "bound": "hbm", "unit": "GB/s",
                                           "peak": HBM_PEAK_GBS, "algorithmic_bytes": in_b + out_b, "achieved": round((in_b + out_b) / dt_f / 1e9, 1),
                                           "frac": round((in_b + out_b) / dt_f / 1e9 / HBM_PEAK_GBS, 4),
                                           "k_fmt_emit_alone": {"achieved": round((in_b + out_b) / (ke_ms / 1e3) / 1e9, 1) if ke_ms else None,
                                                                "frac": round((in_b + out_b) / (ke_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if ke_ms else None}}}
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
    # -- the headline pair on MIXED shapes (the headline's chunk is one layout: 256-byte lines, one key): line lengths 80-600 B,
    #    10 % multi-key bodies (the value is looked up among other keys), 1 % legacy [ts, map] events
    try:
        import random as _rnd
        import numpy as np
        import synth as _synth
        import oracle_binding as _ob
        lens_ = MIXED_LENS
        recs_ = mixed_shape_records()
        pool_bytes = b"".join(recs_)
        tiles_ = max(1, min(n, 3_000_000) // len(recs_))
        mdata = pool_bytes * tiles_
        sizes = np.array([len(x) for x in recs_], dtype=np.uint64)
        moff = np.zeros(len(recs_) * tiles_ + 1, dtype=np.uint64)
        np.cumsum(np.tile(sizes, tiles_), out=moff[1:])
        mn = len(recs_) * tiles_
        d_md = L.flbgpu_dev_alloc(len(mdata) + 16); d_mo = L.flbgpu_dev_alloc(moff.nbytes)
        L.flbgpu_memcpy_h2d(d_md, mdata, len(mdata)); L.flbgpu_memcpy_h2d(d_mo, moff.ctypes.data, moff.nbytes)
        mch = g.DevChunk(d_md, d_mo, mn, len(mdata))
        # (round 6: nothing is selected at create.  46 % of these lines -- even length, no referer / agent -- end in a cell with two writes
        # at one position, which the three-port pair tables hand to the generic kernel; the filter notices on its first call and walks the
        # four-port tables from its second call on, trying the three-port ones again at growing distances (flbgpu_filter_paths).  Timed:
        # the first call, the second, and the steady state the filter is in after them -- tries included, they are part of the policy.)
        p3 = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
        f3 = g.FilterParser("log", [p3]); g3 = g.FilterGrep([GREP_RULE])
        ch3 = g.FilterChain([f3, g3])
        first_ms = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r3_, o3_ = ch3.filter_dev(mch)
            torch.cuda.synchronize()
            first_ms.append(round((time.perf_counter() - t0) * 1e3, 3))
        paths_after = f3.paths()
        msteps = max(steps, 40)                                  # (more than two of the policy's 16-call intervals)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(msteps):
            r3_, o3_ = ch3.filter_dev(mch)
        torch.cuda.synchronize()
        dt_x = (time.perf_counter() - t0) / msteps
        paths_end = f3.paths()
        # parity at the timed size (round 4): (a) the whole fused output against the unfused kernels (filter_parser's full output through
        # filter_grep), by hash; (b) four blocks of 1 000 input rows spread over the chunk through the oracle's two filters -- the output
        # keeps one row per input row, so a block of input rows is a block of output rows
        import hashlib as _hl
        fb = np.empty(int(o3_.bytes), dtype=np.uint8)
        L.flbgpu_memcpy_d2h(fb.ctypes.data, o3_.data, int(o3_.bytes))
        foff = np.empty(mn + 1, dtype=np.uint64)
        L.flbgpu_memcpy_d2h(foff.ctypes.data, o3_.row_off, foff.nbytes)
        rp_, op_ = f3.filter_dev(mch)
        ru_, ou_ = g3.filter_dev(op_)
        ub_ = np.empty(int(ou_.bytes), dtype=np.uint8)
        L.flbgpu_memcpy_d2h(ub_.ctypes.data, ou_.data, int(ou_.bytes))
        sha_f, sha_u = _hl.sha256(memoryview(fb)).hexdigest(), _hl.sha256(memoryview(ub_)).hexdigest()
        po_ = _ob.Parser(regex=APACHE2, time_fmt=TIME_FMT, time_key="time")
        fo_ = _ob.FilterParser("log", [po_]); go_ = _ob.Grep([GREP_RULE])
        ok_, rows_ = True, 0
        blk_ = 1000
        for start in sorted({0, mn // 3, (2 * mn) // 3, mn - blk_}):
            blob = bytes(mdata[int(moff[start]): int(moff[start + blk_])])
            r1_, w1_ = fo_.filter(blob)
            r2_, w2_ = go_.filter(w1_ if r1_ == _ob.MODIFIED else blob)
            want_ = w2_ if r2_ == _ob.MODIFIED else (w1_ if r1_ == _ob.MODIFIED else blob)
            ok_ = ok_ and bytes(fb[int(foff[start]): int(foff[start + blk_])]) == want_
            rows_ += blk_
        out["mixed_shapes"] = {"records": mn, "chunk_bytes": len(mdata), "line_lengths": lens_, "multi_key_bodies": 0.10, "legacy_events": 0.01,
                               "records_per_s_per_gpu": round(mn / dt_x, 1), "ms_per_step": round(dt_x * 1e3, 3), "chunk_GBps": round(len(mdata) / dt_x / 1e9, 1),
                               "kept": int(ch3.last_stats()[1]["out_records"]), "fused_sha256": sha_f, "fused_equals_unfused": bool(sha_f == sha_u),
                               "oracle_sample_rows": rows_, "oracle_sample_matches": bool(ok_),
                               "adaptive": {"first_calls_ms": first_ms, "steady_calls": msteps, "paths_after_three_calls": paths_after, "paths_at_the_end": paths_end,
                                            "what": "no table form selected at create: the filter's own per-call choice (flbgpu_filter_paths), tries of the set-aside build included"}}
        f3.close(); g3.close(); p3.close(); L.flbgpu_dev_free(d_md); L.flbgpu_dev_free(d_mo)
    except Exception as e:
        out["mixed_shapes"] = {"error": repr(e)[:300]}
    # -- multiline in front of the path: a file buffer of Java stack traces (and plain lines) through the built-in `java` parser
    #    (src/multiline/flb_ml_parser_java.c) on the device: rule matches, rule_to_state scan, concatenation into records
    try:
        import random as _rnd
        import ml_synth, oracle_binding as _ob
        rng_ = _rnd.Random(0x3a1)
        block = ml_synth.java_service_log(rng_, 20000)
        reps_ = max(1, min(n, 4_000_000) // 20000)
        text_ = block * reps_
        nlines = text_.count(b"\n")
        d_ml = L.flbgpu_dev_alloc(len(text_)); L.flbgpu_memcpy_h2d(d_ml, text_, len(text_))
        mp_ = g.MultilineParser(builtin="java")
        ms_ = mp_.stream()
        och, recs_, proc_ = ms_.append_dev(d_ml, len(text_), 1700000000, 5, flush=True)
        # parity on what the oracle walks in about a second: the first blocks of the same text
        k_ = min(reps_, 10)
        om = _ob.Multiline(builtin="java")
        want_ = om.append(block * k_, 1700000000, 5)[0]
        got_ = ctypes.create_string_buffer(len(want_))
        L.flbgpu_memcpy_d2h(got_, och.data, len(want_))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ms_.append_dev(d_ml, len(text_), 1700000000, 5, flush=True)
        torch.cuda.synchronize()
        dt_m = (time.perf_counter() - t0) / steps
        t0 = time.perf_counter()
        om2 = _ob.Multiline(builtin="java")
        om2.append(block * k_, 1700000000, 5)
        dt_o = time.perf_counter() - t0
        out["multiline"] = {"parser": "java (built-in, 8 regex rules)", "text": "JVM service log: lines of 80-160 B, 6 % of them open an exception with a stack trace (tests/ml_synth.py java_service_log)", "avg_line_bytes": round(len(text_) / max(nlines, 1), 1), "lines_per_s_per_gpu": round(nlines / dt_m, 1), "ms_per_step": round(dt_m * 1e3, 3),
                            "text_bytes": len(text_), "records": int(recs_), "record_bytes": int(och.bytes), "text_GBps": round(len(text_) / dt_m / 1e9, 1),
                            "product_automaton": dict(zip(("states", "classes", "live"), mp_.product())),
                            "prefix_matches_oracle": bool(got_.raw == want_), "prefix_lines": 20000 * k_,
                            "cpu_oracle_lines_per_s_1core": round(20000 * k_ / dt_o, 1)}
        ms_.close(); mp_.close(); L.flbgpu_dev_free(d_ml)
    except Exception as e:
        out["multiline"] = {"error": repr(e)[:300]}
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
        if dist is not None and shared_gpu:
            t0 = time.perf_counter()
            blobs = [None] * world
            dist.all_gather_object(blobs, st_.export())
            merged = st_.package_merged(blobs)
            e["all_reduce_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
            e["ranks"] = world
        elif dist is not None and "rccl" in out:
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
        st_.close()
        # -- flb_sp's other branch (sp_process_data): a SELECT without aggregation functions over the same resident chunk -- WHERE + projection,
        #    the projected records come back to the host (that copy is inside the time: it is what the call returns)
        try:
            SEL_SQL = "SELECT status, host AS h, latency FROM STREAM:x WHERE status >= 400;"
            ss_ = g.StreamTask(SEL_SQL)
            ret_s, out_s = ss_.do_dev(sch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                ss_.do_dev(sch)
            torch.cuda.synchronize()
            dt_q = (time.perf_counter() - t0) / steps
            ss_.profile(True)
            for _ in range(steps):
                ss_.do_dev(sch)
            prof_q = ss_.profile(False)
            ke_q = sum(v[0] / max(v[1], 1) for v in prof_q.values())
            # (round 6: the device's part and the call's are given apart -- the call returns HOST memory: the D2H copy of the projected records
            # and the Python binding's own copy of them were inside the one number before)
            es = {"records_per_s_per_gpu": round(m / (ke_q / 1e3), 1) if ke_q else None, "ms_per_step": round(ke_q, 3) if ke_q else None,
                  "what": "device time of the call's kernels (size pass + scan + emit pass), HIP events on the task's stream",
                  "call_ms_with_d2h_and_binding_copy": round(dt_q * 1e3, 3), "call_records_per_s": round(m / dt_q, 1),
                  "query": SEL_SQL, "records_out": int(ret_s),
                  "bytes_out": len(out_s), "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in prof_q.items()}}
            if ke_q:
                ab_q = sdata.nbytes * 2 + 2 * len(out_s)         # both passes read the chunk; the emit pass writes the records, the copy reads them
                es["roofline"] = {"kernel": "k_sp_select (both passes)", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": round(ab_q / (ke_q / 1e3) / 1e9, 1),
                                  "frac": round(ab_q / (ke_q / 1e3) / 1e9 / HBM_PEAK_GBS, 4)}
            if rank == 0:
                import ref_sp
                k_ = min(m, 200_000)
                sample = sdata[: int(soff[k_])].tobytes()
                got_ = ss_.do(sample)
                if ref_sp.available():
                    rr = ref_sp.RefSp(SEL_SQL)
                    t0 = time.perf_counter()
                    want_ = rr.do(sample)
                    dt_r = time.perf_counter() - t0
                    rr.close()
                    es["cpu_baseline"] = {"value": round(k_ / dt_r, 1), "unit": "records/s", "cores": 1, "kind": "reference",
                                          "sample": "%d records through the reference's own sp_process_data (oracle/_ref/ref_sp)" % k_,
                                          "identical_output": got_ == want_}
                es["head_is_prefix_of_full_output"] = out_s[:len(got_[1])] == got_[1]
            ss_.close()
            out["flb_sp_select"] = es
        except Exception as e:
            out["flb_sp_select"] = {"error": repr(e)[:300]}
        L.flbgpu_dev_free(d_sd); L.flbgpu_dev_free(d_so)
    except Exception as e:
        out["flb_sp_group_by"] = {"error": repr(e)[:300]}
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
    ucores, core_info = usable_cores()
    cpu = {"value": round(ns / cdt, 1), "unit": "records/s", "cores": 1, "kind": "port",
           "sample": "first %d records of the same seeded workload through oracle filter_parser(apache2)+filter_grep, "
                     "single thread (%d usable host cores)" % (ns, ucores), "host": core_info}
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
            secs, rin, rkept, ref_out = rf.bench_result(rf.run([rf.bench_pair_case("log", dict(regex=APACHE2, time_fmt=TIME_FMT, time_key="time"), [GREP_RULE],
                                                                                    bytes(data[: int(off[m])]), 1)], timeout=900)[0], with_output=True)
            assert rin == m, (rin, m)
            # the reference's own output bytes for these m records: verify_timed_output compares them with the device's first m rows
            import hashlib
            cpu["reference_output"] = {"records": m, "bytes": len(ref_out), "sha256": hashlib.sha256(ref_out).hexdigest()}
            del ref_out
            cpu["port"] = {"value": cpu["value"], "sample": cpu["sample"]}
            cpu.update({"value": round(rin / secs, 1), "kind": "reference", "kept": int(rkept),
                        "sample": "first %d records of the same seeded workload through the reference's own cb_filter of filter_parser(apache2) and "
                                  "filter_grep (compiled from its sources: oracle/_ref/ref_filters), single thread (%d usable host cores)" % (m, ucores)})
            ref_ok = True
    except Exception as e:
        cpu["reference_error"] = repr(e)[:200]
    try:
        # N independent processes of the same filter pair, N = the cores this lease may really use (affinity mask cut by the
        # cgroup quota -- os.cpu_count() is the machine's 256), about 8 s of work each: SURVEY 8(d)'s second baseline
        nproc = ucores
        m = min(n, 2_000_000)
        if ref_ok:
            import ref_filters as rf
            from concurrent.futures import ThreadPoolExecutor
            per = max(2000, min(200_000, m // max(nproc, 1)))
            shards = [bytes(data[int(off[i * per % max(1, m - per)]): int(off[i * per % max(1, m - per) + per])]) for i in range(nproc)]
            iters = max(1, int(8.0 * cpu["value"] / per))               # ~8 s of work per process at the single-thread rate
            def one(sh):
                return rf.bench_result(rf.run([rf.bench_pair_case("log", dict(regex=APACHE2, time_fmt=TIME_FMT, time_key="time"), [GREP_RULE], sh, iters)], timeout=300)[0])
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=nproc) as ex:
                res = list(ex.map(one, shards))
            wall = time.perf_counter() - t0
            v = sum(r[1] for r in res) * iters / wall
            cpu["nproc"] = {"processes": nproc, "cores": nproc, "value": round(v, 1), "per_core": round(v / nproc, 1), "unit": "records/s", "kind": "reference",
                            "wall_s": round(wall, 1),
                            "note": "%d independent processes (one per usable core) of the reference's filter pair, %d records x %d passes each, wall-clock aggregate "
                                    "(process start-up included)" % (nproc, per, iters)}
        else:
            v = cpu_nproc_leg((bytes(data[: int(off[m])]), np.array(off[: m + 1])), nproc)
            cpu["nproc"] = {"processes": nproc, "cores": nproc, "value": round(v, 1), "per_core": round(v / nproc, 1), "unit": "records/s", "kind": "port",
                            "note": "%d independent processes, each the oracle pair on its own shard for ~6 s, wall-clock aggregate" % nproc}
    except Exception as e:
        cpu["nproc_error"] = repr(e)[:200]
    try:
        # BASELINE.json configs[0]: in_dummy -> filter_grep (one regex) -> out_null inside the reference's OWN engine (oracle/_ref/engine,
        # built by oracle/build_engine.sh with the reference's cmake), 1 M records: the plumbing rate next to the filter-only rates above
        import subprocess
        eh = os.path.join(ROOT, "oracle", "_ref", "engine", "engine_host")
        if os.path.exists(eh):
            line = '1.2.3.4 - - [10/Oct/2000:13:55:36 -0700] "GET /a HTTP/1.1" 503 2326 "http://r" "Mozilla"'
            r = subprocess.run([eh, "configs0", "1000000", r"log ^.* 5\d\d ", line], capture_output=True, text=True, timeout=300)
            js = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if js:
                j = json.loads(js[-1])
                cpu["configs0_engine"] = {"records": int(j["records"]), "records_per_s": j["records_per_s"], "seconds": j["seconds"],
                                          "what": "BASELINE configs[0]: in_dummy -> filter_grep -> out_null in the reference's own engine (libfluent-bit.so built from its sources), 1 thread"}
    except Exception as e:
        cpu["configs0_error"] = repr(e)[:200]
    return cpu


def measure_engine_hosted(data, off, n):
    """The drop-in inside the reference's own engine (oracle/_ref/engine: the reference built with its cmake, oracle/engine/engine_host.c):
    in_lib -> filter_parser -> filter_grep -> out_lib with the built-in pair and with flb-filter_{parser,grep}_gpu.so loaded by the real
    flb_plugin_load_router -- 2 MB appends (7 000 lines a push), records/s from the first push to the last chunk out_lib hands over (the
    engine's 0.2 s flush timer is in both) -- and, without the engine's input side, one 2 MB chunk through flb_processor_run
    (the plugin's cb_filter under the unit's lock), per call.  The only numbers a user of the drop-in sees; test infrastructure, never `value`."""
    import json as _json, subprocess, tempfile
    eng = os.path.join(ROOT, "oracle", "_ref", "engine")
    host = os.path.join(eng, "engine_host")
    so = {x: os.path.join(eng, "plugins", "flb-filter_%s_gpu.so" % x) for x in ("grep", "parser")}
    if not (os.path.exists(host) and all(os.path.exists(v) for v in so.values())):
        return {"skipped": "oracle/_ref/engine not built (bash oracle/build_engine.sh where /root/reference is)"}
    pspec = "apache2|%s|%s|time|0" % (APACHE2, TIME_FMT)
    m = min(n, 1_000_000)
    out = {"lines": m, "lines_per_push": 7000}

    def last_json(r):
        ls = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return _json.loads(ls[-1]) if ls else {"error": (r.stdout[-300:] + r.stderr[-300:])}
    with tempfile.TemporaryDirectory() as d:
        blob = bytes(data[: int(off[m])])
        jl, mp = os.path.join(d, "in.json"), os.path.join(d, "in.mp")
        with open(jl, "w") as f:
            for i in range(m):
                f.write(_json.dumps([1700000000 + i, {"log": blob[int(off[i]) + 21:int(off[i + 1])].decode("latin1")}]) + "\n")
        open(mp, "wb").write(blob[: int(off[7000])])
        for name, extra, pf, gf in (("builtin", [], "parser", "grep"), ("gpu_plugins", ["-e", so["parser"], "-e", so["grep"]], "parser_gpu", "grep_gpu")):
            e = {}
            try:
                r = subprocess.run([host, "lib"] + extra + ["--parser", pspec, "--batch", "7000", jl, os.path.join(d, "out.bin"),
                                                            "--filter", pf, "key_name=log", "parser=apache2", "--filter", gf, "regex=code ^5\\d\\d$"],
                                   capture_output=True, text=True, timeout=600)
                j = last_json(r)
                e["whole_engine"] = dict(j, records_per_s=round(j["pushed"] / j["seconds"], 1)) if "seconds" in j and j.get("seconds") else j
                r = subprocess.run([host, "processor"] + extra + ["--parser", pspec, "--repeat", "50", mp, os.path.join(d, "out.mp"),
                                                                  "--unit", pf, "key_name=log", "parser=apache2", "--unit", gf, "regex=code ^5\\d\\d$"],
                                   capture_output=True, text=True, timeout=600)
                j = last_json(r)
                if "seconds" in j:
                    e["processor_2MB_chunk"] = {"ms_per_call": round(j["seconds"] / j["repeat"] * 1e3, 3), "records_per_s": round(7000 * j["repeat"] / j["seconds"], 1),
                                                "out_bytes": j["out_bytes"], "first_call_ms": round(j.get("first_call_seconds", 0.0) * 1e3, 3), "calls_timed": j["repeat"], "units": j.get("units")}
                else:
                    e["processor_2MB_chunk"] = j
            except Exception as ex:
                e["error"] = repr(ex)[:300]
            out[name] = e
    try:
        out["same_kept_bytes"] = out["builtin"]["whole_engine"]["log_bytes"] == out["gpu_plugins"]["whole_engine"]["log_bytes"] and \
            out["builtin"]["processor_2MB_chunk"]["out_bytes"] == out["gpu_plugins"]["processor_2MB_chunk"]["out_bytes"]
    except Exception:
        pass
    return out


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
        reps = 40 if nrec < 50000 else 8
        phases = []
        t0 = time.perf_counter()
        for _ in range(reps):
            ch.filter(blob)
            phases.append(g.host_phases())
        dt = (time.perf_counter() - t0) / reps
        med = {k: round(sorted(p[k] for p in phases)[len(phases) // 2], 1) for k in phases[0]}
        out[name] = {"records": nrec, "bytes": len(blob), "ms_per_call": round(dt * 1e3, 3), "records_per_s": round(nrec / dt, 1),
                     "in_GBps": round(len(blob) / dt / 1e9, 2), "phases_us_median": med}
    fp.close(); fg.close(); p.close()
    return out


def launch_ranks(args, json_fd, script=None, argv=None):
    """--gpus N without a launcher around it: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on this file
    (one rank per GPU; with fewer GPUs than ranks the ranks share devices and say so in the line) and pass rank 0's line on.
    (script / argv: another rank program under the same launcher -- tests/test_launcher.py drives the merge code on gloo.)"""
    import socket
    import subprocess
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        os.write(json_fd, (lines[-1] + "\n").encode())
    if r.returncode != 0 or not lines:
        sys.stderr.write("bench.py: the %d-rank launch ended with code %d%s\n" % (args.gpus, r.returncode, "" if lines else " and printed no line"))
        sys.exit(r.returncode or 1)


def verify_timed_output(g, L, data, off, n, fused_host, parsed_chunk, fgrep, args, cpu=None):
    """parity of the timed configuration at the timed size"""
    import hashlib
    import numpy as np
    out = {}
    # (a) fused == unfused, whole output
    kb, koff = fused_host
    r2u, o2u = fgrep.filter_dev(parsed_chunk)
    ub = np.empty(int(o2u.bytes), dtype=np.uint8)
    L.flbgpu_memcpy_d2h(ub.ctypes.data, o2u.data, int(o2u.bytes))
    h_f, h_u = hashlib.sha256(memoryview(kb)).hexdigest(), hashlib.sha256(memoryview(ub)).hexdigest()
    out["fused_sha256"] = h_f
    out["fused_equals_unfused"] = bool(h_f == h_u and r2u == g.MODIFIED)
    out["output_bytes"] = int(kb.nbytes)
    # (b) four blocks of 1 000 input rows against the oracle (the output keeps one row per input row: a dropped record is an empty row)
    if not args.no_cpu:
        import oracle_binding as ob
        po = ob.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
        fo = ob.FilterParser("log", [po]); go = ob.Grep([GREP_RULE])
        blk = min(1000, n)
        ok, rows = True, 0
        for start in sorted({0, n // 3, (2 * n) // 3, n - blk}):
            blob = bytes(data[int(off[start]): int(off[start + blk])])
            r1, w1 = fo.filter(blob)
            r2, w2 = go.filter(w1)
            want = w2 if r2 == ob.MODIFIED else w1
            got = bytes(kb[int(koff[start]): int(koff[start + blk])])
            ok = ok and got == want
            rows += blk
        out["oracle_sample_rows"] = rows
        out["oracle_sample_matches"] = bool(ok)
    # (c) VERDICT r4 weak 3: the REFERENCE's own output (its two cb_filter compiled from its sources, the run that gives cpu_baseline)
    # for the first m records of the timed chunk against the device's first m output rows, whole bytes by hash
    ro = (cpu or {}).get("reference_output")
    if ro and ro["records"] <= n:
        m = ro["records"]
        dev = memoryview(kb)[: int(koff[m])]
        out["reference_rows"] = m
        out["reference_bytes"] = int(ro["bytes"])
        out["reference_output_matches"] = bool(len(dev) == ro["bytes"] and hashlib.sha256(dev).hexdigest() == ro["sha256"])
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
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="records timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the log_to_metrics / JSON side measurements")
    ap.add_argument("--no-pmc", action="store_true", help="never start the rocprofv3 passes (roofline.traffic stays null when the committed summary is stale)")
    ap.add_argument("--pmc", action="store_true", help="measure roofline.traffic now: two more passes of the headline step under rocprofv3 (FETCH_SIZE, WRITE_SIZE)")
    ap.add_argument("--ndjson-lines", type=int, default=100_000_000, help="BASELINE configs[2]: NDJSON lines through JSON -> events -> 32-rule grep (secondary)")
    ap.add_argument("--l2m-records", type=int, default=1_000_000_000, help="BASELINE configs[3]: records through log_to_metrics over all ranks (secondary)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process becomes the launcher -- one rank per GPU through
        # torch.distributed.run on 127.0.0.1, the ranks' single JSON line (rank 0 prints it) handed on
        launch_ranks(args, json_fd)
        return

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
    ndev = max(1, torch.cuda.device_count())
    dev = local_rank % ndev
    shared_gpu = world > ndev              # more ranks than GPUs (a 1-GPU box driving the N-rank path): RCCL refuses two ranks on one
                                           # device, the collectives then run on gloo through the same merge code (fluent_bit_amd.l2m_merge,
                                           # flbgpu_sp_package_merged) -- reported as such, never as an RCCL number
    if world > 1 or os.environ.get("FLBGPU_BENCH_FORCE_DIST"):       # (forced: exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(dev)
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        torch.cuda.set_device(0)
    g = flbamd_loader.load()
    g.init(dev if world > 1 else 0)
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
    # the fused output of the last timed step, copied out before anything reuses the filters' device buffers
    fused_host = None
    if rank == 0:
        import numpy as np
        kb = np.empty(int(o2.bytes), dtype=np.uint8)
        L.flbgpu_memcpy_d2h(kb.ctypes.data, o2.data, int(o2.bytes))
        koff = np.empty(n + 1, dtype=np.uint64)
        L.flbgpu_memcpy_d2h(koff.ctypes.data, o2.row_off, koff.nbytes)
        fused_host = (kb, koff)
    # the parsed chunk itself, for the side measurements below (one unfused filter_parser run, untimed)
    r1, o1 = fparser.filter_dev(chunk)
    assert r1 == g.MODIFIED and int(o1.bytes) == parsed_bytes, (g.last_error(), int(o1.bytes), parsed_bytes)
    # BASELINE configs[1] proper -- filter_parser ALONE (the headline adds configs[0]'s grep behind it): every record parsed and
    # written back, SURVEY 8(d)'s 277 + 275 B/record
    parser_only = None
    if not args.no_secondary:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fparser.filter_dev(chunk)
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / 3
        parser_only = {"what": "filter_parser(apache2) alone, unfused kernels: k_parser_reg + scans + k_parser_emit", "records_per_s_per_gpu": round(n / dtp, 1),
                       "ms_per_step": round(dtp * 1e3, 3), "algorithmic_bytes": int(in_bytes + parsed_bytes),
                       "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "achieved": round((in_bytes + parsed_bytes) / dtp / 1e9, 1),
                                    "frac": round((in_bytes + parsed_bytes) / dtp / 8e12, 4)}}

    per_rank = None
    if dist is not None:
        # every rank's own clock next to the maximum the line is computed from (a slow rank shows up by name)
        mine = {"rank": rank, "device": dev, "seconds": round(dt, 6), "records_per_s": round(n * args.steps / dt, 1), "kept_records": kept_records}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- the timed path checked at the timed size (rank 0): (a) SHA-256 of the fused pair's output == SHA-256 of what the
    #      unfused filter_parser -> filter_grep kernels write for the same chunk; (b) 4 000 consecutive-in-blocks rows against the
    #      oracle: the output rows of four blocks of 1 000 input records (start, two inside, end) byte for byte
    verify = None
    if rank == 0:
        try:
            verify = verify_timed_output(g, L, data, off, n, fused_host, o1, fgrep, args, cpu)
        except Exception as e:
            verify = {"error": repr(e)[:300]}

    # ---- side measurements (never part of `value`): the other rows of the hot-path scope table
    secondary = None
    if not args.no_secondary:
        try:
            secondary = measure_secondary(g, torch, dist, rank, world, o1, n, args, raw_chunk=chunk, shared_gpu=shared_gpu)
            if parser_only is not None:
                secondary["parser_only"] = parser_only
            if rank == 0 and world == 1:
                secondary["host_level"] = measure_host_level(g, data, off, n)
                secondary["engine_hosted"] = measure_engine_hosted(data, off, n)
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
        live = None
        # --pmc, or by itself when the committed summary was recorded from other kernel sources than this tree's (an edit since the
        # last profile): the traffic of THIS build is measured now rather than left null (--no-pmc keeps it null)
        auto_pmc = traffic is None and not args.no_pmc and not args.pmc and world == 1
        if (args.pmc or auto_pmc) and world == 1:
            # the same step twice more under rocprofv3 (children), after this process has finished its own timing
            live, live_src = pmc_pass(n, min(args.steps, 3) if auto_pmc else args.steps, args.warmup, timeout=200 if auto_pmc else 900)
            if auto_pmc and live is not None:
                live_src += " -- run by itself: " + str(traffic_src)
            if live is None:
                traffic_src = "--pmc: " + live_src + "; " + str(traffic_src)
            else:
                hit = live.get(dom) or {}
                if hit:
                    traffic = int(hit.get("FETCH_SIZE", 0) * 1024 * 2.0 + hit.get("WRITE_SIZE", 0) * 1024 * 1.0)
                    traffic_src = live_src
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(avg_s * 1e3, 4), "launches": int(launches),
                "algorithmic_bytes_per_launch": int(alg_bytes_per_launch.get(dom, in_bytes))}
        # the whole step against the same roof: SURVEY 8(d) wire-format bytes of the fused pair = input once + what
        # filter_parser emits + what filter_grep keeps (552 + 275 x keep B/record), over the step's kernel time
        step_bytes = in_bytes + parsed_bytes + kept_bytes
        kern_ms = sum(v[0] for v in prof.values()) / args.steps
        def _tr(k):
            if live is not None:
                hs = [v for kk, v in live.items() if kk == k or (k.endswith("k_scan") and kk.startswith("k_scan_"))]
                if hs:
                    return int(sum(h.get("FETCH_SIZE", 0) * 1024 * 2.0 + h.get("WRITE_SIZE", 0) * 1024 * 1.0 for h in hs))
            return recorded_traffic(k, n)[0]
        tr = [_tr(k) for k in prof]
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
        "roofline": roof, "cpu_baseline": cpu, "kernels": kernels, "verify": verify, "secondary": secondary,
        "rccl_ranks": world if (dist is not None and not shared_gpu) else 0, "per_rank": per_rank,
        "collective_backend": None if dist is None else ("gloo (ranks share a GPU: RCCL refuses duplicate devices)" if shared_gpu else "rccl"),
    }
    if dist is not None:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()
