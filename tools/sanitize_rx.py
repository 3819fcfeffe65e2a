"""The product's host-only regex code (rx.cpp with rx_nfa.inc / rxbt.inc, fx.cpp, rx_capi.cpp) under -fsanitize=address,undefined:
tools/sanitize_rx.sh builds /tmp/san/librx_san.so with g++ and runs this under LD_PRELOAD=libasan.  The CPU regex suites run in
process with every flbgpu_rx* / flbgpu_rxbt* entry point taken from the sanitized build (the rest of the C ABI from libflbgpu.so)."""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader
import pytest

SAN = os.environ.get("FLBGPU_RX_SAN", "/tmp/san/librx_san.so")


class Mixed:
    def __init__(self, real, san):
        object.__setattr__(self, "_real", real); object.__setattr__(self, "_san", san); object.__setattr__(self, "_hits", set())

    def __getattr__(self, name):
        if name.startswith("flbgpu_rx") and hasattr(self._san, name):
            f = getattr(self._san, name)
            if name not in self._hits:
                r = getattr(self._real, name)
                if r.argtypes is not None:
                    f.argtypes = r.argtypes
                f.restype = r.restype
                self._hits.add(name)
            return f
        return getattr(self._real, name)


def main():
    m = flbamd_loader.load()
    real = m.lib()
    mixed = Mixed(real, ctypes.CDLL(SAN))
    m._L = mixed
    files = ["test_rxbt.py", "test_rx_properties.py", "test_rx_random_patterns.py", "test_fx_tables.py", "test_stock_parsers.py", "test_grep_merge.py", "test_cabi.py"]
    rc = pytest.main(["-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"] + [os.path.join(ROOT, "tests", f) for f in files] + sys.argv[1:])
    print("sanitized entry points used:", len(mixed._hits), sorted(mixed._hits))
    return rc


if __name__ == "__main__":
    sys.exit(main())
