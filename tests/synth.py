"""Seeded synthetic workloads (ctypes wrapper over fluent-bit_amd/csrc/synth.c) + tiny msgpack
helpers for hand-built records.  Harness utility."""
import ctypes, os, struct, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0xF1B17
_L = None

def _lib():
    global _L
    if _L is None:
        d = os.path.join(ROOT, "fluent-bit_amd", "csrc")
        so = os.path.join(d, "libflbsynth.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(d, "synth.c")):
            subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(d, "synth.c")], check=True)
        L = ctypes.CDLL(so)
        L.flbsynth_apache_records.restype = ctypes.c_uint64
        L.flbsynth_apache_records.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        _L = L
    return _L

def apache_records(n, line_len=256, tz_mixed=1, seed=SEED):
    """returns (bytes ndarray u8, offsets ndarray u64 [n+1], epochs ndarray u32 [n])"""
    cap = n * (line_len + 32) + 64
    out = np.empty(cap, dtype=np.uint8)
    off = np.empty(n + 1, dtype=np.uint64)
    ep = np.empty(max(n, 1), dtype=np.uint32)
    tot = _lib().flbsynth_apache_records(seed, n, line_len, tz_mixed, out.ctypes.data, cap, off.ctypes.data, ep.ctypes.data)
    assert tot > 0 or n == 0
    return out[:tot], off, ep[:n]

# ---- minimal msgpack writer for hand-built test records
def mp(o):
    if o is None: return b"\xc0"
    if o is True: return b"\xc3"
    if o is False: return b"\xc2"
    if isinstance(o, int):
        if 0 <= o < 128: return bytes([o])
        if -32 <= o < 0: return struct.pack("b", o)
        if 0 <= o < 256: return b"\xcc" + bytes([o])
        if 0 <= o < 65536: return b"\xcd" + struct.pack(">H", o)
        if 0 <= o < 2**32: return b"\xce" + struct.pack(">I", o)
        if o >= 0: return b"\xcf" + struct.pack(">Q", o)
        if o >= -128: return b"\xd0" + struct.pack("b", o)
        if o >= -32768: return b"\xd1" + struct.pack(">h", o)
        if o >= -2**31: return b"\xd2" + struct.pack(">i", o)
        return b"\xd3" + struct.pack(">q", o)
    if isinstance(o, float): return b"\xcb" + struct.pack(">d", o)
    if isinstance(o, str): o = o.encode()
    if isinstance(o, bytes):
        n = len(o)
        if n < 32: return bytes([0xa0 | n]) + o
        if n < 256: return b"\xd9" + bytes([n]) + o
        if n < 65536: return b"\xda" + struct.pack(">H", n) + o
        return b"\xdb" + struct.pack(">I", n) + o
    if isinstance(o, Raw): return o.b
    if isinstance(o, (list, tuple)):
        n = len(o)
        h = bytes([0x90 | n]) if n < 16 else (b"\xdc" + struct.pack(">H", n) if n < 65536 else b"\xdd" + struct.pack(">I", n))
        return h + b"".join(mp(x) for x in o)
    if isinstance(o, dict):
        n = len(o)
        h = bytes([0x80 | n]) if n < 16 else (b"\xde" + struct.pack(">H", n) if n < 65536 else b"\xdf" + struct.pack(">I", n))
        return h + b"".join(mp(k) + mp(v) for k, v in o.items())
    if isinstance(o, KV):
        n = len(o.items)
        h = bytes([0x80 | n]) if n < 16 else b"\xde" + struct.pack(">H", n)
        return h + b"".join(mp(k) + mp(v) for k, v in o.items)
    raise TypeError(type(o))

class Raw:
    """pre-encoded msgpack bytes"""
    def __init__(self, b): self.b = b

class KV:
    """map with explicit (possibly duplicate) key order"""
    def __init__(self, items): self.items = items

def ext_ts(sec, nsec=0):
    return Raw(b"\xd7\x00" + struct.pack(">II", sec, nsec))

def v2_record(sec, nsec, body, meta=None):
    return mp([[ext_ts(sec, nsec), meta if meta is not None else {}], body])

def legacy_record(ts, body):
    return mp([ts, body])

# ---- decoder for assertions (returns python objects; maps -> list of pairs to keep order/dups)
def unpack_all(b):
    out = []; p = 0
    while p < len(b):
        o, p = _un(b, p)
        out.append(o)
    return out

def _un(b, p):
    c = b[p]; p += 1
    if c < 0x80: return c, p
    if c >= 0xe0: return c - 256, p
    if 0xa0 <= c <= 0xbf: n = c & 31; return b[p:p+n], p + n
    if 0x90 <= c <= 0x9f: return _arr(b, p, c & 15)
    if 0x80 <= c <= 0x8f: return _map(b, p, c & 15)
    if c == 0xc0: return None, p
    if c == 0xc2: return False, p
    if c == 0xc3: return True, p
    if c == 0xca: return struct.unpack(">f", b[p:p+4])[0], p + 4
    if c == 0xcb: return struct.unpack(">d", b[p:p+8])[0], p + 8
    if c == 0xcc: return b[p], p + 1
    if c == 0xcd: return struct.unpack(">H", b[p:p+2])[0], p + 2
    if c == 0xce: return struct.unpack(">I", b[p:p+4])[0], p + 4
    if c == 0xcf: return struct.unpack(">Q", b[p:p+8])[0], p + 8
    if c == 0xd0: return struct.unpack("b", b[p:p+1])[0], p + 1
    if c == 0xd1: return struct.unpack(">h", b[p:p+2])[0], p + 2
    if c == 0xd2: return struct.unpack(">i", b[p:p+4])[0], p + 4
    if c == 0xd3: return struct.unpack(">q", b[p:p+8])[0], p + 8
    if c in (0xd4, 0xd5, 0xd6, 0xd7, 0xd8):
        n = 1 << (c - 0xd4); return ("ext", b[p], b[p+1:p+1+n]), p + 1 + n
    if c == 0xd9: n = b[p]; return b[p+1:p+1+n], p + 1 + n
    if c == 0xda: n = struct.unpack(">H", b[p:p+2])[0]; return b[p+2:p+2+n], p + 2 + n
    if c == 0xdb: n = struct.unpack(">I", b[p:p+4])[0]; return b[p+4:p+4+n], p + 4 + n
    if c == 0xc4: n = b[p]; return ("bin", b[p+1:p+1+n]), p + 1 + n
    if c == 0xdc: return _arr(b, p + 2, struct.unpack(">H", b[p:p+2])[0])
    if c == 0xdd: return _arr(b, p + 4, struct.unpack(">I", b[p:p+4])[0])
    if c == 0xde: return _map(b, p + 2, struct.unpack(">H", b[p:p+2])[0])
    if c == 0xdf: return _map(b, p + 4, struct.unpack(">I", b[p:p+4])[0])
    raise ValueError(hex(c))

def _arr(b, p, n):
    out = []
    for _ in range(n):
        o, p = _un(b, p); out.append(o)
    return out, p

def _map(b, p, n):
    out = []
    for _ in range(n):
        k, p = _un(b, p); v, p = _un(b, p); out.append((k, v))
    return ("map", out), p
