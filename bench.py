#!/usr/bin/env python3
"""bench.py -- records/sec through filter_parser(apache2) -> filter_grep on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident
in HBM: the chunk (10 M seeded apache-combined lines of 256 B wrapped as 277 B V2 log events,
BASELINE.json configs[1]) goes through filter_parser (conf/parsers.conf 'apache2', Key_Name log)
and the parser's output chunk goes through filter_grep (Regex code ^5\\d\\d$), chained on the
device exactly like flb_filter_do chains cb_filter calls (src/flb_filter.c:179-272).

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--records", type=int, default=10_000_000, help="records per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=3_000_000, help="records timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import flbamd_loader
    import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    g = flbamd_loader.load()
    g.init(local_rank if world > 1 else 0)
    L = g.lib()

    # ---- synthetic shard for this rank (seeded; per-GPU work is fixed => weak scaling)
    n = args.records
    t0 = time.time()
    data, off, ep = synth.apache_records(n, seed=synth.SEED + rank)
    gen_s = time.time() - t0
    in_bytes = int(data.nbytes)
    d_data = L.flbgpu_dev_alloc(in_bytes)
    d_off = L.flbgpu_dev_alloc(off.nbytes)
    assert d_data and d_off, g.last_error()
    L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, in_bytes)
    L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
    chunk = g.DevChunk(d_data, d_off, n, in_bytes)

    parser = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time", name="apache2")
    fparser = g.FilterParser("log", [parser])
    fgrep = g.FilterGrep([GREP_RULE])

    def step():
        r1, o1 = fparser.filter_dev(chunk)
        assert r1 == g.MODIFIED, g.last_error()
        r2, o2 = fgrep.filter_dev(o1)
        return o1, o2, r2

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
        o1, o2, r2 = step()
    sync_all()
    dt = time.perf_counter() - t0
    prof = dict(fparser.profile_read())
    prof.update({"grep:" + k if k == "k_scan" else k: v for k, v in fgrep.profile_read().items()})
    parsed_bytes = int(o1.bytes)
    kept_bytes = int(o2.bytes) if r2 == g.MODIFIED else parsed_bytes
    kept_records = fgrep.counts()[1]

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

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
        "k_parser_locate": in_bytes,                     # reads every chunk byte once
        "k_parser_rx": value_bytes,                      # the capture program consumes each value byte once
        "k_parser_finish": 8 * n,                        # time field + sizes
        "k_parser_emit": value_bytes + parsed_bytes,     # re-reads the values, writes the output once
        "k_grep_match": parsed_bytes,
        "k_gather": 2 * kept_bytes,
    }
    roof = None
    if dom:
        ms, launches = prof[dom]
        avg_s = ms / 1e3 / max(launches, 1)
        ach = alg_bytes_per_launch.get(dom, in_bytes) / avg_s / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                "avg_launch_ms": round(avg_s * 1e3, 4), "launches": int(launches),
                "algorithmic_bytes_per_launch": int(alg_bytes_per_launch.get(dom, in_bytes))}
    kernels = {k: {"total_ms": round(v[0], 3), "launches": int(v[1])} for k, v in prof.items()}

    cpu = None
    if not args.no_cpu and world == 1:
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

    line = {
        "metric": "log records/sec (256B apache-combined lines) through parser+grep",
        "value": round(value, 1), "unit": "records/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "filter_parser(conf/parsers.conf apache2, Key_Name log) -> filter_grep(Regex code ^5\\d\\d$) "
                               "on %d x 256B apache-combined lines per GPU (277B V2 events), chained on device, unfused" % n,
                   "records_per_gpu": n, "in_bytes": in_bytes, "parsed_bytes": parsed_bytes, "kept_records": int(kept_records),
                   "seed": synth.SEED, "parallelism": "shard%d" % world, "gen_seconds": round(gen_s, 1)},
        "roofline": roof, "cpu_baseline": cpu, "kernels": kernels,
    }
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
