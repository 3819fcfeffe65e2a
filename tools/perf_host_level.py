"""PCIe-inclusive rate of the host-level call (flbgpu_filter_chain_run = flb_filter_do on a host buffer):
record indexing on the CPU + H2D + kernels + D2H.  Reported in DESIGN.md, never as bench.py's `value`."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
g = flbamd_loader.load(); g.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
APACHE2 = (r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" '
           r'(?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$')
p = g.Parser(APACHE2, time_fmt="%d/%b/%Y:%H:%M:%S %z", time_key="time")
fp = g.FilterParser("log", [p])
ch = g.FilterChain([fp, g.FilterGrep([("regex", r"code ^5\d\d$")])])
for n in [int(a) for a in sys.argv[1:]] or [7000, 100_000, 2_000_000]:
    data, off, ep = synth.apache_records(n)
    blob = bytes(data)
    ch.filter(blob); fp.filter(blob)
    reps = max(3, min(200, 2_000_000 // n))
    t0 = time.perf_counter()
    for _ in range(reps):
        r, out = ch.filter(blob)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        r, out1 = fp.filter(blob)
    dp = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    k, o2, cons = g.index_host(blob)
    ti = time.perf_counter() - t0
    print("host-level, %d records (%.1f MB): parser+grep %.3f ms = %.1f M records/s (%.2f GB/s in); parser only (%.1f MB out) %.3f ms = %.1f M records/s; "
          "record indexing alone %.3f ms" % (n, len(blob) / 1e6, dt * 1e3, n / dt / 1e6, len(blob) / dt / 1e9, len(out1) / 1e6, dp * 1e3, n / dp / 1e6, ti * 1e3))
