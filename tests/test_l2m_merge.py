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
