import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
from bench import APACHE2, TIME_FMT, GREP_RULE
n = int(sys.argv[1])
g = flbamd_loader.load(); g.init(0); L = g.lib()
data, off, ep = synth.apache_records(n)
d_data = L.flbgpu_dev_alloc(data.nbytes); d_off = L.flbgpu_dev_alloc(off.nbytes)
L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
chunk = g.DevChunk(d_data, d_off, n, data.nbytes)
p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE])
print("parser alone", flush=True)
fp.filter_dev(chunk); L.flbgpu_sync(); print(" ok", flush=True)
ch = g.FilterChain([fp, fg])
print("chain", flush=True)
ch.filter_dev(chunk); L.flbgpu_sync(); print(" ok", flush=True)
