import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
g = flbamd_loader.load(); g.init(0); L = g.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
APACHE2 = (r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" '
           r'(?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$')
data, off, ep = synth.apache_records(n)
d_data = L.flbgpu_dev_alloc(int(data.nbytes)); d_off = L.flbgpu_dev_alloc(off.nbytes)
L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, int(data.nbytes)); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
chunk = g.DevChunk(d_data, d_off, n, int(data.nbytes))
p = g.Parser(APACHE2, time_fmt="%d/%b/%Y:%H:%M:%S %z", time_key="time")
fp = g.FilterParser("log", [p])
r, o1 = fp.filter_dev(chunk)
for mode, props, vf in (("counter", [("label_field", "method"), ("label_field", "code")], None),
                        ("histogram", [("label_field", "code")], "size"),
                        ("gauge", [("label_field", "code")], "size"),
                        ("counter", [("label_field", "host")], None)):
    f = g.FilterLogToMetrics(mode, props, value_field=vf)
    f.filter_dev(o1)
    f.profile(True)
    t0 = time.perf_counter()
    for _ in range(3):
        f.filter_dev(o1)
    dt = (time.perf_counter() - t0) / 3
    print(mode, props, "%.2f ms/step %.1f M rec/s" % (dt * 1e3, n / dt / 1e6), {k: round(v[0] / v[1], 3) for k, v in f.profile_read().items()}, f.stats(), len(f.snapshot()))
    f.close()
