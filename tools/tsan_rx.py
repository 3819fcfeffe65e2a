"""8 threads searching the same compiled programs (what flbgpu.cpp parallel_rows does with a host rule), compiling and freeing others meanwhile,
under -fsanitize=thread: tools/sanitize_rx.sh builds /tmp/san/librx_tsan.so and runs this under LD_PRELOAD=libtsan."""
import ctypes, threading, random, sys
L = ctypes.CDLL("/tmp/san/librx_tsan.so")
L.flbgpu_rxbt_compile.restype = ctypes.c_void_p
L.flbgpu_rxbt_compile.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_char_p, ctypes.c_int]
L.flbgpu_rxbt_search.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
L.flbgpu_rxbt_free.argtypes = [ctypes.c_void_p]
L.flbgpu_rx_compile.restype = ctypes.c_void_p
L.flbgpu_rx_compile.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
L.flbgpu_rx_simulate_capture.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
pats = [rb"(?<=user=)(\w+) id=\1", rb"^(?!.*(?:health|ping)).*\d$", rb"(?>a+)b", rb"(?i)(ab)\1", rb"(a)?(?(1)b|c)", rb"a\Rb"]
err = ctypes.create_string_buffer(256)
hs = [L.flbgpu_rxbt_compile(p, len(p), 0, err, 256) for p in pats]
assert all(hs)
rp = rb"^(?<host>[^ ]*) (?<rest>.*)$"
hr = L.flbgpu_rx_compile(rp, len(rp), 0, 1, err, 256)
subj = [b"user=alice id=alice", b"GET /health 200", b"GET /x 200 5", b"aaab", b"ABab", b"ab", b"c", b"a\r\nb", b"x" * 200, b"host rest of line"]
base = {}
for i, h in enumerate(hs):
    for s in subj:
        b = (ctypes.c_int * 64)(); e = (ctypes.c_int * 64)()
        base[(i, s)] = (L.flbgpu_rxbt_search(h, s, len(s), b, e), list(b[:4]), list(e[:4]))
bad = []
def work(seed):
    rng = random.Random(seed)
    for _ in range(3000):
        i = rng.randrange(len(hs)); s = rng.choice(subj)
        b = (ctypes.c_int * 64)(); e = (ctypes.c_int * 64)()
        r = (L.flbgpu_rxbt_search(hs[i], s, len(s), b, e), list(b[:4]), list(e[:4]))
        if r != base[(i, s)]: bad.append((i, s, r))
        n = L.flbgpu_rx_simulate_capture(hr, s, len(s), b, e)
        # concurrent compiles (the NFA / table builders keep no global state)
        if rng.random() < 0.01:
            h2 = L.flbgpu_rxbt_compile(pats[i], len(pats[i]), 0, ctypes.create_string_buffer(256), 256)
            L.flbgpu_rxbt_free(h2)
ts = [threading.Thread(target=work, args=(k,)) for k in range(8)]
[t.start() for t in ts]; [t.join() for t in ts]
print("threads done, mismatches:", len(bad))
sys.exit(1 if bad else 0)
