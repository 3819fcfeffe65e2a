#!/usr/bin/env python3
"""NDJSON lines -> log events on the device (JsonPacker.run_dev = flbgpu_json_run_dev): milliseconds per 10 M lines of
tests/ndjson_synth.py's text (200 k distinct lines tiled), and a hash of the events for comparing builds."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flbamd_loader, ndjson_synth as ns

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    g = flbamd_loader.load(); g.init(0); L = g.lib()
    nbase = 200_000
    base = ns.lines(nbase, seed=7)
    text = b"".join(base); blen = len(text)
    reps = max(1, n // nbase); n = reps * nbase
    boff = np.zeros(nbase + 1, dtype=np.uint64)
    np.cumsum(np.fromiter((len(x) for x in base), dtype=np.uint64, count=nbase), out=boff[1:])
    off = (boff[:-1][None, :] + (np.arange(reps, dtype=np.uint64) * np.uint64(blen))[:, None]).reshape(-1)
    off = np.ascontiguousarray(np.concatenate([off, np.array([reps * blen], dtype=np.uint64)]))
    d_data = L.flbgpu_dev_alloc(reps * blen + 16); d_off = L.flbgpu_dev_alloc(off.nbytes)
    for r in range(reps):
        L.flbgpu_memcpy_h2d(d_data + r * blen, text, blen)
    L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
    chunk = g.DevChunk(d_data, d_off, n, reps * blen)
    pk = g.JsonPacker()
    mode = sys.argv[2] if len(sys.argv) > 2 else ""
    pk.tile_debug(prof="prof" in mode, no_lookback="nolb" in mode)
    ev = pk.run_dev(chunk, events=True, ts=(1, 0))
    L.flbgpu_sync()
    t0 = time.perf_counter()
    for _ in range(5):
        ev = pk.run_dev(chunk, events=True, ts=(1, 0))
    L.flbgpu_sync()
    dt = (time.perf_counter() - t0) / 5
    if mode:
        ph = pk.tile_debug(prof="prof" in mode, no_lookback="nolb" in mode)
        tot = sum(ph) or 1
        print("mode %s; phase cycles (share): " % mode + ", ".join("%s %.1f%%" % (nm, 100.0 * c / tot) for nm, c in zip(
            ["stage", "A bytes", "C0 numbers", "B rows", "C tokens", "look-back", "C' headers", "D bodies"], ph)) + "; cycles per line %.0f" % (tot / n))
    host = np.empty(min(int(ev.bytes), 256 << 20), dtype=np.uint8)
    L.flbgpu_memcpy_d2h(host.ctypes.data, ev.data, host.nbytes)
    print("lines %d text %d B events %d B: %.3f ms per call = %.3f ms per 10 M lines, %.1f M lines/s; sha %s stats %s" % (
        n, reps * blen, ev.bytes, dt * 1e3, dt * 1e3 * 1e7 / n, n / dt / 1e6, hashlib.sha256(host).hexdigest()[:16], (pk.stats(), pk.tile_stats())))

if __name__ == "__main__":
    main()
