"""Multi-rank path of filter_log_to_metrics on CPU (gloo, world_size 2): every rank aggregates its
shard of the records, the partial series rows are merged with torch.distributed all_reduce
(fluent_bit_amd.l2m_merge -- on the GPU box the same call runs over RCCL), and the finalized
numbers must equal the oracle's single pass over ALL the records."""
import math, os, random, struct, sys
import numpy as np
import pytest
import oracle_binding as ob
import l2m_model as lm
from synth import v2_record
import flbamd_loader

HERE = os.path.dirname(os.path.abspath(__file__))
BOUNDS = [0.5, 10.0, 100.0]


def make_records(seed, n):
    rng = random.Random(seed)
    recs = []
    for i in range(n):
        k = rng.choice(["GET", "POST", "PUT", "DELETE"]) if rng.random() < 0.97 else "rare%d" % rng.randrange(5)
        t = rng.random()
        if t < 0.6: v = float(rng.randrange(0, 5000)) / 8          # dyadic: sequential f64 sums are exact
        elif t < 0.9: v = float(rng.randrange(-100, 100))
        else: v = rng.choice([0.0, 2.0 ** 30, -(2.0 ** 30), 2.0 ** -10])
        recs.append((k, v))
    return recs


def finalize_all(g, mode, keys, rows):
    return {tuple(k.split(b"\0")[:-1]): g.finalize_row(mode, len(BOUNDS) if mode == 2 else 0, r) for k, r in zip(keys, rows)}


def test_finalize_row_single_rank():
    g = flbamd_loader.load()
    recs = make_records(5, 5000)
    data = b"".join(v2_record(1, 0, {"m": k, "v": v}) for k, v in recs)
    obs = [(k.encode() + b"\0", v, i) for i, (k, v) in enumerate(recs)]
    for mode, name in ((0, "counter"), (1, "gauge"), (2, "histogram")):
        props = [("label_field", "m")] + ([("bucket", str(b)) for b in BOUNDS] if mode == 2 else [])
        o = ob.L2M(name, props, value_field="v" if mode else None)
        o.filter(data)
        keys, rows = lm.encode_rows(mode, BOUNDS if mode == 2 else [], obs)
        got = finalize_all(g, mode, keys, rows)
        want = o.snapshot()[2]
        assert [tuple(k.split(b"\0")[:-1]) for k in keys] == [s["labels"] for s in want]       # first-appearance order
        for s in want:
            a = got[s["labels"]]
            if mode == 2:
                assert a["buckets"] == s["buckets"] and a["count"] == s["count"] and a["sum"] == s["sum"]
            else:
                assert a["value"] == s["value"]


def test_finalize_row_sum_is_exactly_rounded():
    g = flbamd_loader.load()
    rng = random.Random(8)
    for trial in range(200):
        vals = []
        for _ in range(rng.randrange(1, 60)):
            t = rng.random()
            if t < 0.4: vals.append(struct.unpack("<d", struct.pack("<Q", (rng.getrandbits(64) & 0x800FFFFFFFFFFFFF) | (rng.randrange(1, 2046) << 52)))[0] * 2.0 ** -30)
            elif t < 0.6: vals.append(struct.unpack("<d", struct.pack("<Q", rng.getrandbits(52) | (rng.getrandbits(1) << 63)))[0])
            elif t < 0.8: vals.append(rng.uniform(-1, 1))
            else:
                vals.append(rng.choice([1e300, 2.0 ** 1000, 5e-324, 1.0]))
                vals.append(-vals[-1])
        try:
            want = math.fsum(vals)
        except OverflowError:
            continue
        keys, rows = lm.encode_rows(2, [], [(b"", v, i) for i, v in enumerate(vals)])
        got = g.finalize_row(2, 0, rows[0])
        assert struct.pack("<d", got["sum"]) == struct.pack("<d", want), (vals, got["sum"], want)
        assert got["count"] == len(vals)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = flbamd_loader.load()
    recs = make_records(77, 8000)
    per = (len(recs) + world - 1) // world
    lo, hi = rank * per, min(len(recs), (rank + 1) * per)                 # contiguous shard, global indices kept
    out = {}
    for mode in (0, 1, 2):
        obs = [(recs[i][0].encode() + b"\0", recs[i][1], i) for i in range(lo, hi)]
        keys, rows = lm.encode_rows(mode, BOUNDS if mode == 2 else [], obs)
        mk, mr = g.l2m_merge(keys, rows, lm.row_words(mode, len(BOUNDS) if mode == 2 else 0), dist)
        out[mode] = (mk, mr.tolist())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_all_reduce_matches_single_pass():
    import torch.multiprocessing as tmp
    g = flbamd_loader.load()
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]                                               # identical on every rank
    recs = make_records(77, 8000)
    data = b"".join(v2_record(1, 0, {"m": k, "v": v}) for k, v in recs)
    for mode, name in ((0, "counter"), (1, "gauge"), (2, "histogram")):
        props = [("label_field", "m")] + ([("bucket", str(b)) for b in BOUNDS] if mode == 2 else [])
        o = ob.L2M(name, props, value_field="v" if mode else None)
        o.filter(data)
        want = o.snapshot()[2]
        keys, rows = res[0][mode]
        rows = np.array(rows, dtype=np.uint64)
        assert [tuple(k.split(b"\0")[:-1]) for k in keys] == [s["labels"] for s in want]
        got = finalize_all(g, mode, keys, rows)
        for s in want:
            a = got[s["labels"]]
            if mode == 2:
                assert a["buckets"] == s["buckets"] and a["count"] == s["count"] and a["sum"] == s["sum"]
            else:
                assert a["value"] == s["value"]


# ---- sum_order 2: the reference's sequential sum across ranks (round 6) ---------------------------------------------------------
def seq_records(seed, n):
    """values whose sequential f64 sum depends on the order (cancellation, magnitudes 1e-9 .. 1e16); a series that first shows up on the
    last rank, another one only in the second interval"""
    rng = random.Random(seed)
    recs = []
    for i in range(n):
        k = rng.choice(["GET", "POST", "PUT"])
        r = rng.random()
        if r < 0.03: v = rng.choice([1e16, -1e16, 9007199254740993.0, 3e15])
        elif r < 0.06: v = rng.choice([1e-9, 2.5e-7, -1e-9])
        else: v = rng.uniform(-1000, 1000) * 10.0 ** rng.randint(-6, 6)
        recs.append((k, v))
    return recs


def seq_shards(world):
    """two intervals; interval -> rank -> records (the global order is interval by interval, rank by rank)"""
    out = []
    for it in range(2):
        recs = seq_records(900 + it, 3000)
        per = (len(recs) + world - 1) // world
        shards = [recs[r * per:(r + 1) * per] for r in range(world)]
        shards[world - 1] = shards[world - 1] + [("LATE%d" % it, 0.1 * (j + 1)) for j in range(7)]       # first seen on the last rank
        if it == 1:
            shards[0] = [("LATE0", 1e-3)] + shards[0]                                                    # ... and on rank 0 an interval later
        out.append(shards)
    return out


def _seq_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = flbamd_loader.load()
    me = lm.SeqRank()
    W = lm.row_words(2, len(BOUNDS))
    res = []
    gidx = rank << 40
    for shards in seq_shards(world):
        obs = []
        for k, v in shards[rank]:
            me.observe(k.encode() + b"\0", v)
            obs.append((k.encode() + b"\0", v, gidx))
            gidx += 1
        keys, rows = lm.encode_rows(2, BOUNDS, obs)
        mk, mr = g.l2m_merge(keys, rows, W, dist)
        res.append((mk, g.l2m_chain(mk, dist, me)))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sequential_sums_across_ranks_are_the_real_cmetrics_bits(world):
    """sum_order 2: every rank keeps its interval's observations and the flush folds them rank after rank (fluent_bit_amd.l2m_chain over
    gloo, the protocol flbgpu_l2m_all_reduce runs over RCCL) -- against the REAL cmetrics (oracle/_ref/libcmetrics_ref.so) fed the
    records in that order: no tolerance, for two and three ranks, over two intervals."""
    import test_cmetrics_pin as cp
    if not os.path.exists(cp.REF):
        pytest.skip("oracle/_ref/libcmetrics_ref.so not built (needs /root/reference)")
    import torch.multiprocessing as tmp
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_seq_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(1, world):
        assert res[r] == res[0]                                           # identical on every rank
    L = cp._ref()
    flat = []
    for it, shards in enumerate(seq_shards(world)):
        for r in range(world):
            flat += [((k,), v) for k, v in shards[r]]
        # (the merged keys of an interval are the series touched IN it: the model's rows are per interval; cmetrics is cumulative)
        ref, _, _, _ = cp.run_case(L, "histogram", ["m"], BOUNDS, flat)
        want = {s["labels"][0] + b"\0": s["sum"] for s in ref}
        mk, sums = res[0][it]
        assert len(mk) >= 4
        for k, x in zip(mk, sums):
            assert struct.pack("<d", x) == want[k], (world, it, k, x, struct.unpack("<d", want[k])[0])
