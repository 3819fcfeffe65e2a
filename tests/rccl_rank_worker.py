"""A rank program for tests/test_rccl_ranks_gpu.py: one process per GPU (torch.distributed.run), an RCCL communicator made through the
C ABI (flbgpu_rccl_unique_id / flbgpu_rccl_comm_init, the id shipped over a gloo group), and the two collectives of the hot path
exactly as N ranks of a node would run them:
  * filter_log_to_metrics: rank r counts chunk r (index base = the records in front of it), flbgpu_l2m_all_reduce merges -- every
    rank must hold what ONE filter fed all the chunks in order holds, bit for bit (counter / histogram / gauge);
  * flb_sp: rank r aggregates chunk r of one window, flbgpu_sp_timer_all_reduce packages -- the records of one task fed everything.
Rank 0 prints one JSON line.  With world_size 1 the same code runs on a single GPU (the exchange is the identity)."""
import hashlib, json, os, random, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import numpy as np
import torch
import torch.distributed as dist
import flbamd_loader
import synth, sp_synth
import oracle_binding as ob

APACHE2 = (r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" '
           r'(?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$')
TF = "%d/%b/%Y:%H:%M:%S %z"
L2M = (("counter", [("label_field", "method"), ("label_field", "code")], None),
       ("histogram", [("label_field", "code"), ("bucket", "1000"), ("bucket", "100000")], "size"),
       ("gauge", [("label_field", "code")], "size"))
SQL = "SELECT host, status, COUNT(*), AVG(latency), SUM(bytes), MIN(bytes), MAX(latency) FROM STREAM:x WINDOW TUMBLING (10 SECOND) WHERE status <> 404 GROUP BY host, status;"
N_L2M, N_SP = 20000, 2000


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    assert ndev >= world, "one GPU per rank: %d ranks, %d devices" % (world, ndev)
    dist.init_process_group("gloo", rank=rank, world_size=world)        # (carries the RCCL id and the check's hashes; the data path is RCCL)
    g = flbamd_loader.load()
    g.init(local)

    def exchange(raw):
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    comm = g.RcclComm(world, rank, exchange if world > 1 else None)
    # every rank builds every chunk (seeded): its own for the sharded run, all of them for the single pass it is compared with
    po = ob.Parser(APACHE2, time_fmt=TF, time_key="time")
    parsed = [ob.FilterParser("log", [po]).filter(bytes(synth.apache_records(N_L2M, seed=synth.SEED + 100 + r)[0]))[1] for r in range(world)]
    res, ms = {}, {}
    for mode, props, vf in L2M:
        f = g.FilterLogToMetrics(mode, props, value_field=vf)
        f.set_index_base(rank * N_L2M)
        assert f.filter(parsed[rank])[0] == g.NOTOUCH
        dist.barrier()
        t0 = time.perf_counter()
        keys, rows = g.l2m_all_reduce_rccl(f, comm)
        ms["l2m_" + mode] = round((time.perf_counter() - t0) * 1e3, 3)
        one = g.FilterLogToMetrics(mode, props, value_field=vf)
        for c in parsed:
            assert one.filter(c)[0] == g.NOTOUCH
        k1, r1 = one.export()
        assert keys == k1 and np.array_equal(rows, r1), "log_to_metrics %s: the all-reduce of %d ranks differs from one filter over all the records" % (mode, world)
        assert f.snapshot((keys, rows)) == one.snapshot()
        res["l2m_" + mode] = hashlib.sha256(repr((keys, rows.tolist())).encode()).hexdigest()
        f.close(); one.close()
    # sum_order 2: the histogram sum as ONE reference process adds it up that is fed rank 0's records, then rank 1's, ... -- the chain
    # inside flbgpu_l2m_all_reduce (a broadcast of the sums per rank), two intervals; against one filter with sum_order 1 over the chunks
    # in that order, bit for bit
    mode, props, vf = L2M[1]
    f = g.FilterLogToMetrics(mode, props, value_field=vf); f.set_sum_order(2); f.set_index_base(rank << 40)
    one = g.FilterLogToMetrics(mode, props, value_field=vf); one.set_sum_order(1)
    for interval in range(2):
        cut = [int(g.index_host(c)[1][g.index_host(c)[0] // 2]) for c in parsed]                      # (a record boundary near the middle)
        part = [c[cut[q]:] if interval else c[: cut[q]] for q, c in enumerate(parsed)]
        f.filter(part[rank])
        keys, rows = g.l2m_all_reduce_rccl(f, comm)
        for c in part:
            one.filter(c)
        got, want = f.chain_sums(), one.seq_sums()
        k1, _ = one.export()
        assert keys == k1 and [x.hex() for x in map(float, got)] == [x.hex() for x in map(float, want)], \
            "log_to_metrics sum_order 2: the chain of %d ranks differs from one filter over the records in rank order (interval %d)" % (world, interval)
    res["l2m_chain"] = hashlib.sha256(repr([float(x).hex() for x in got]).encode()).hexdigest()
    f.close(); one.close()
    rng = random.Random(0x5AD)
    chunks = [sp_synth.chunk(rng, N_SP, clean=True) for _ in range(world)]
    t, one = g.StreamTask(SQL), g.StreamTask(SQL)
    t.set_index_base(rank * N_SP)
    t.do(chunks[rank])
    dist.barrier()
    t0 = time.perf_counter()
    got = t.timer_all_reduce(comm)
    ms["sp_timer"] = round((time.perf_counter() - t0) * 1e3, 3)
    for c in chunks:
        one.do(c)
    want = one.timer()
    assert got == want, "flb_sp: the all-reduce of %d ranks packages other records than one task over all the chunks" % world
    assert t.timer() == b""
    res["sp"] = hashlib.sha256(got).hexdigest()
    t.close(); one.close()
    allres = [None] * world
    dist.all_gather_object(allres, res)
    comm.close()
    dist.barrier()
    if rank == 0:
        os.write(1, (json.dumps({"rccl_ranks": world, "devices": ndev, "ranks_agree": all(r == allres[0] for r in allres), "all_reduce_ms": ms, "sha": res}) + "\n").encode())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
