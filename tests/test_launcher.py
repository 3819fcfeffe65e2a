"""bench.py --gpus N launches its own ranks: the launcher itself, driven here with world_size 2 on gloo over
the product's merge code, and the pieces of bench.py that decide who launches."""
import argparse, hashlib, json, os, subprocess, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import flbamd_loader
import l2m_model as lm
from test_l2m_merge import make_records, BOUNDS


def test_launch_ranks_two_processes_gloo_merge_equals_one_process():
    # the launcher is bench.launch_ranks; run it in a child so that its sys.exit / fd handling stay out of pytest
    code = ("import os, sys, argparse; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench; "
            "bench.launch_ranks(argparse.Namespace(gpus=2), 1, script=%r, argv=['77'])" % (ROOT, HERE, os.path.join(HERE, "rank_worker.py")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=600)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-400:], r.stderr[-800:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_agree"]
    # the same rows merged by ONE process: a world of one (no collective: the encoded rows of all records)
    g = flbamd_loader.load()
    recs = make_records(77, 6000)
    for mode in (0, 1, 2):
        obs = [(recs[i][0].encode() + b"\0", recs[i][1], i) for i in range(len(recs))]
        keys, rows = lm.encode_rows(mode, BOUNDS if mode == 2 else [], obs)
        order = np.argsort(~rows[:, 0], kind="stable")
        want = hashlib.sha256(repr(([keys[i] for i in order], rows[order].tolist())).encode()).hexdigest()
        assert d["sha"][str(mode)] == want, mode


def test_bench_decides_to_launch_only_without_a_launcher():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if args.gpus > 1 and "WORLD_SIZE" not in os.environ:' in src and "launch_ranks(args, json_fd)" in src
    assert "--nproc-per-node=%d" in src and '"--master-addr", "127.0.0.1"' in src
