import os, sys, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader, synth
from bench import APACHE2, TIME_FMT, GREP_RULE
n = int(sys.argv[1]); reps = int(sys.argv[2])
g = flbamd_loader.load(); g.init(0); L = g.lib()
data, off, ep = synth.apache_records(n)
d_data = L.flbgpu_dev_alloc(data.nbytes); d_off = L.flbgpu_dev_alloc(off.nbytes)
L.flbgpu_memcpy_h2d(d_data, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
chunk = g.DevChunk(d_data, d_off, n, data.nbytes)
p = g.Parser(APACHE2, time_fmt=TIME_FMT, time_key="time")
fp = g.FilterParser("log", [p]); fg = g.FilterGrep([GREP_RULE])
ch = g.FilterChain([fp, fg])
for i in range(reps):
    r, o = ch.filter_dev(chunk); L.flbgpu_sync()
    ob = np.empty(int(o.bytes), dtype=np.uint8); L.flbgpu_memcpy_d2h(ob.ctypes.data, o.data, int(o.bytes))
    st = ch.last_stats()
    print("call", i, "bytes", int(o.bytes), "kept", int(st[1]["out_records"]), hashlib.sha256(memoryview(ob)).hexdigest()[:16], flush=True)
