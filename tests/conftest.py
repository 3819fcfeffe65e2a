import os, sys, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is the checker: (re)build it if a compiler is around and it is stale/missing
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle"))
            if f.endswith((".c", ".h"))]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/lib/onigmo") and not os.path.exists(
            os.path.join(ROOT, "oracle", "_ref", "libonig_ref.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True,
                       stdout=subprocess.DEVNULL)
