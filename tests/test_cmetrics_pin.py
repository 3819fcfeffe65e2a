"""The arithmetic of the oracle's filter_log_to_metrics (oracle/oflb.c oflb_l2m_*: cmt_counter_inc,
cmt_gauge_set, cmt_histogram_observe restated) against the REAL cmetrics compiled from the reference
(oracle/_ref/libcmetrics_ref.so): same series in the same order, same value / bucket counts / count, and the
same f64 sum bit for bit, on random streams of labelled observations."""
import ctypes
import os
import random
import struct

import pytest

import oracle_binding as ob
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "oracle", "_ref", "libcmetrics_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libcmetrics_ref.so not built (needs /root/reference)")


def _ref():
    L = ctypes.CDLL(REF)
    L.refcmt_new.restype = ctypes.c_void_p
    L.refcmt_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    L.refcmt_update.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
    for f in ("refcmt_nseries", "refcmt_nbuckets"):
        getattr(L, f).argtypes = [ctypes.c_void_p]
    L.refcmt_bound.restype = ctypes.c_double; L.refcmt_bound.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.refcmt_label.restype = ctypes.c_char_p; L.refcmt_label.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.refcmt_value.restype = ctypes.c_double; L.refcmt_value.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.refcmt_sum.restype = ctypes.c_double; L.refcmt_sum.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.refcmt_bucket.restype = ctypes.c_uint64; L.refcmt_bucket.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.refcmt_count.restype = ctypes.c_uint64; L.refcmt_count.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.refcmt_free.argtypes = [ctypes.c_void_p]
    return L


def bits(x):
    return struct.pack("<d", x)


def run_case(L, mode, keys, bounds, obs):
    """obs: list of (label values tuple, value) -> both snapshots"""
    m = {"counter": 0, "gauge": 1, "histogram": 2}[mode]
    ka = (ctypes.c_char_p * len(keys))(*[k.encode() for k in keys])
    if bounds is None:
        h = L.refcmt_new(m, len(keys), ka, -1, None)
    else:
        ba = (ctypes.c_double * max(len(bounds), 1))(*bounds)
        h = L.refcmt_new(m, len(keys), ka, len(bounds), ba)
    for i, (labels, v) in enumerate(obs):
        la = (ctypes.c_char_p * len(labels))(*[x.encode() for x in labels])
        assert L.refcmt_update(h, 1000 + i, float(v), len(labels), la) == 0
    nb = L.refcmt_nbuckets(h)
    ref = []
    for s in range(L.refcmt_nseries(h)):
        d = dict(labels=tuple(L.refcmt_label(h, s, i) for i in range(len(keys))))
        if m == 2:
            d.update(buckets=[L.refcmt_bucket(h, s, b) for b in range(nb + 1)], count=L.refcmt_count(h, s), sum=bits(L.refcmt_sum(h, s)))
        else:
            d.update(value=bits(L.refcmt_value(h, s)))
        ref.append(d)
    ref_bounds = [L.refcmt_bound(h, i) for i in range(nb)]
    L.refcmt_free(h)
    # the same observations as log records through the oracle's filter
    props = [("label_field", k) for k in keys] + ([("bucket", repr(b)) for b in bounds] if bounds else [])
    om = ob.L2M(mode, props, value_field=None if mode == "counter" else "v")
    data = b"".join(synth.mp([[synth.ext_ts(1, i), {}], dict(zip(keys, labels), v=v)]) for i, (labels, v) in enumerate(obs))
    om.filter(data)
    _, obounds, snap = om.snapshot()
    mine = []
    for x in snap:
        d = dict(labels=x["labels"])
        if m == 2:
            d.update(buckets=x["buckets"], count=x["count"], sum=bits(x["sum"]))
        else:
            d.update(value=bits(x["value"]))
        mine.append(d)
    return ref, ref_bounds, mine, obounds


def test_oracle_l2m_arithmetic_matches_real_cmetrics():
    L = _ref()
    rng = random.Random(8)
    vals = lambda: rng.choice([0, 1, 5, 10, 0.005, 0.01, 0.25, 2.5, 7.5, 10.0, 1e-9, 1e9, 1e300, -3.5, 0.1, 0.2, 0.30000000000000004,
                               rng.random() * 20, rng.randrange(-50, 5000), float(rng.randrange(1 << 53)), 123456.789e-3])
    for it in range(60):
        mode = ["counter", "gauge", "histogram"][it % 3]
        keys = ["a", "b"][: rng.randrange(0, 3)]
        bounds = None
        if mode == "histogram" and rng.random() < 0.7:
            bounds = sorted({rng.choice([0.001, 0.5, 1, 2.5, 10, 100, 1e6, 7]) for _ in range(rng.randrange(1, 6))})
        obs = [(tuple(rng.choice(["x", "y", "zz", ""]) for _ in keys), vals()) for _ in range(rng.randrange(1, 400))]
        ref, rb, mine, ob_ = run_case(L, mode, keys, bounds, obs)
        assert rb == ob_, (mode, bounds)
        assert mine == ref, (mode, keys, bounds, obs[:5])
