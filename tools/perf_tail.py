#!/usr/bin/env python3
"""The walk's tail skip (dev.hpp DevFx::tail_min) on lines whose tail is long: apache-combined lines of 256 bytes with a short request
path and a long user agent, and syslog lines with a long message, through filter_parser on a device-resident chunk -- k_parser_reg's
event-timed milliseconds with the skip and, FLBGPU_DEBUG_SKIP=128, with every position walked; the outputs must be the same bytes."""
import hashlib
import os
import random
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flbamd_loader

APACHE2 = r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$'
SYSLOG = r'^\<(?<pri>[0-9]+)\>(?<time>[^ ]* {1,2}[^ ]* [^ ]*) (?<ident>[a-zA-Z0-9_\/\.\-]*)(?:\[(?<pid>[0-9]+)\])?(?:[^\:]*\:)? *(?<message>.*)$'


def chunk_of(lines, n):
    """n V2 events [[ts, {}], {"log": line}] tiled from `lines` (all the same length)"""
    L = len(lines[0])
    assert all(len(x) == L for x in lines) and 32 <= L < 256
    recs = [b"\x92\x92\xd7\x00" + struct.pack(">II", 1700000000 + i, 0) + b"\x80\x81\xa3log\xd9" + bytes([L]) + x for i, x in enumerate(lines)]
    R = len(recs[0])
    tile = np.frombuffer(b"".join(recs), dtype=np.uint8)
    reps = (n + len(lines) - 1) // len(lines)
    data = np.tile(tile, reps)[: n * R]
    off = np.arange(n + 1, dtype=np.uint64) * R
    return data, off


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    g = flbamd_loader.load()
    g.init(0)
    L = g.lib()
    rng = random.Random(7)
    agents = ["Mozilla/5.0 (X11; Linux x86_64) AppleWebKit/537.36 (KHTML, like Gecko) Chrome/%d.0.%d.%d Safari/537.36 Edg/%d.0" % (rng.randrange(90, 130), rng.randrange(5000), rng.randrange(200), rng.randrange(90, 130)) for _ in range(64)]
    ap = []
    for i in range(4096):
        head = '%d.%d.%d.%d - - [10/Mar/2024:15:22:%02d +0200] "GET /p/%d HTTP/1.1" %d %d "http://example.com/%d" "' % (
            rng.randrange(256), rng.randrange(256), rng.randrange(256), rng.randrange(256), rng.randrange(60), rng.randrange(10 ** 5), rng.choice([200, 200, 404, 500]), rng.randrange(10 ** 5), rng.randrange(100))
        a = rng.choice(agents)
        line = (head + a + " " + "x" * 200)[:254] + '"'
        ap.append(line.encode())
    sy = []
    for i in range(4096):
        head = "<%d>Oct 11 22:14:%02d host%d app[%d]: " % (rng.randrange(190), rng.randrange(60), rng.randrange(50), rng.randrange(30000))
        sy.append((head + "session opened for user root by (uid=0) " + "payload=%d " % rng.randrange(10 ** 9) * 20)[:200].encode())
    for name, regex, tf, lines in (("apache2, long agent (256 B lines)", APACHE2, "%d/%b/%Y:%H:%M:%S %z", ap), ("syslog-rfc3164-local, long message (200 B lines)", SYSLOG, "%b %d %H:%M:%S", sy)):
        data, off = chunk_of(lines, n)
        d_d = L.flbgpu_dev_alloc(data.nbytes + 16); d_o = L.flbgpu_dev_alloc(off.nbytes)
        L.flbgpu_memcpy_h2d(d_d, data.ctypes.data, data.nbytes); L.flbgpu_memcpy_h2d(d_o, off.ctypes.data, off.nbytes)
        ch = g.DevChunk(d_d, d_o, n, data.nbytes)
        p = g.Parser(regex, time_fmt=tf, time_key="time")
        f = g.FilterParser("log", [p])
        f.filter_dev(ch)
        f.profile(True)
        for _ in range(5):
            r, out = f.filter_dev(ch)
        prof = f.profile_read()
        if os.environ.get("PERF_TAIL_TRACE"):
            # the kernel's own timeline (s_memtime stamps per phase): the walk's share of a wave's iteration
            path = "/tmp/flbgpu_trace_tail.bin"
            os.environ["FLBGPU_TRACE"] = "32"; os.environ["FLBGPU_TRACE_FILE"] = path
            f.filter_dev(ch); L.flbgpu_sync()
            del os.environ["FLBGPU_TRACE"]
            raw = open(path, "rb").read()
            grid, waves, iters, k = struct.unpack("<4I", raw[:16])
            t = np.frombuffer(raw[16:], dtype=np.uint64).reshape(grid, waves, iters, k).astype(np.int64)
            done = (t[..., 7] != 0) & (t[..., 0] != 0)
            d = np.diff(t, axis=-1)[done]
            print("   timeline (median cycles): ingest %d decode %d walk %d rest %d | iteration %d" % (
                np.median(d[:, 0] + d[:, 1]), np.median(d[:, 2]), np.median(d[:, 3]), np.median(d[:, 4] + d[:, 5] + d[:, 6]), np.median((t[..., 7] - t[..., 0])[done])))
        host = np.empty(min(int(out.bytes), 64 << 20), dtype=np.uint8)
        L.flbgpu_memcpy_d2h(host.ctypes.data, out.data, host.nbytes)
        ks = {k: round(v[0] / max(v[1], 1), 3) for k, v in prof.items() if v[1]}
        print("%s: n %d out %d B sha %s kernels(ms) %s" % (name, n, out.bytes, hashlib.sha256(host).hexdigest()[:16], ks), flush=True)
        f.close(); p.close(); L.flbgpu_dev_free(d_d); L.flbgpu_dev_free(d_o)


if __name__ == "__main__":
    main()
