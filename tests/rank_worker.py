"""A rank program for tests/test_launcher.py: started by bench.launch_ranks (torch.distributed.run, one process per rank), it
merges log_to_metrics row shards over gloo with the product's merge code (fluent_bit_amd.l2m_merge) and rank 0 prints one JSON
line -- the shape of bench.py's multi-rank run without a GPU."""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import torch.distributed as dist
import flbamd_loader
import l2m_model as lm
from test_l2m_merge import make_records, BOUNDS


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = flbamd_loader.load()
    recs = make_records(int(sys.argv[1]) if len(sys.argv) > 1 else 77, 6000)
    per = (len(recs) + world - 1) // world
    lo, hi = rank * per, min(len(recs), (rank + 1) * per)
    res = {}
    for mode in (0, 1, 2):
        obs = [(recs[i][0].encode() + b"\0", recs[i][1], i) for i in range(lo, hi)]
        keys, rows = lm.encode_rows(mode, BOUNDS if mode == 2 else [], obs)
        mk, mr = g.l2m_merge(keys, rows, lm.row_words(mode, len(BOUNDS) if mode == 2 else 0), dist)
        res[mode] = hashlib.sha256(repr((mk, mr.tolist())).encode()).hexdigest()
    allres = [None] * world
    dist.all_gather_object(allres, res)
    dist.barrier()
    if rank == 0:
        os.write(1, (json.dumps({"n_gpus": world, "ranks_agree": all(r == allres[0] for r in allres), "sha": res}) + "\n").encode())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
