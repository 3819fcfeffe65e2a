"""The N-rank RCCL path of SURVEY 8(e) as a rank program (tests/rccl_rank_worker.py) under bench.py's own launcher: with one visible
GPU it runs as a world of one (every line of the program executes, the exchange is the identity); with two or more it runs two ranks on
two devices and asserts, inside the ranks, that flbgpu_l2m_all_reduce / flbgpu_sp_timer_all_reduce give bit for bit what a single
pass over all the records gives -- so that the first multi-GPU run of the driver is a measurement, not a debugging session."""
import json, os, subprocess, sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pytestmark = pytest.mark.gpu


def run_ranks(n):
    code = ("import os, sys, argparse; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench; "
            "bench.launch_ranks(argparse.Namespace(gpus=%d), 1, script=%r, argv=[])" % (ROOT, HERE, n, os.path.join(HERE, "rccl_rank_worker.py")))
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=400)
    except subprocess.TimeoutExpired:
        # (the bare one-rank communicator of conftest.rccl_ok came up on this box -- these tests only run behind it --, so a rank program
        # that does not come back is the product's: a deadlock in flbgpu_l2m_all_reduce / flbgpu_sp_timer_all_reduce shows as a FAILURE)
        pytest.fail("the rank program did not come back within 400 s although a bare RCCL communicator came up on this box")
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-400:], r.stderr[-1500:])
    return json.loads(lines[0])


def test_rank_program_world_of_one(rccl_ok):
    d = run_ranks(1)
    assert d["rccl_ranks"] == 1 and d["ranks_agree"] and set(d["sha"]) == {"l2m_counter", "l2m_histogram", "l2m_gauge", "l2m_chain", "sp"}


def test_two_ranks_on_two_gpus_equal_the_single_pass(rccl_ok):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the two-rank RCCL run needs two devices (RCCL refuses two ranks on one)")
    d = run_ranks(2)
    assert d["rccl_ranks"] == 2 and d["ranks_agree"], d
