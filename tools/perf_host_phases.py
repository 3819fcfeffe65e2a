"""Where the wall time of one host-level call goes (flbgpu_host_phases): the [filter_parser, filter_grep] chain and filter_parser alone
on an engine-sized chunk (~2 MB) and on a 28 MB chunk, medians over the repetitions."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
import numpy as np
g = flbamd_loader.load(); g.init(0)
L = g.lib()
APACHE2 = (r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" '
           r'(?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$')
NAMES = ["index", "copy_in", "upload_wait", "chain", "download", "malloc", "total"]
p = g.Parser(APACHE2, time_fmt="%d/%b/%Y:%H:%M:%S %z", time_key="time")
fp = g.FilterParser("log", [p])
ch = g.FilterChain([fp, g.FilterGrep([("regex", r"code ^5\d\d$")])])
for n in [int(a) for a in sys.argv[1:]] or [7000, 100_000]:
    data, off, ep = synth.apache_records(n)
    blob = bytes(data)
    for name, f in (("parser+grep", ch), ("parser only", fp)):
        f.filter(blob); f.filter(blob)
        reps = 50 if n < 50000 else 10
        rows = []; walls = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r, out = f.filter(blob)
            walls.append((time.perf_counter() - t0) * 1e6)
            ph = (ctypes.c_double * 8)()
            L.flbgpu_host_phases(ph, 8)
            rows.append(list(ph)[:7])
        med = np.median(np.array(rows), axis=0)
        print("%-12s %7d records %6.1f MB in %6.2f MB out: python wall %7.1f us | " % (name, n, len(blob) / 1e6, len(out) / 1e6, float(np.median(walls)))
              + "  ".join("%s %.1f" % (k, v) for k, v in zip(NAMES, med)))
