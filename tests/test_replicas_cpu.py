"""N > 1 = independent replicas (no data-path collective).  The only cross-rank logic is bench.py's
replica_aggregate (MAX of the per-rank times, SUM of the per-rank tokens) -- covered here with a world_size-2
gloo group on CPU, plus the single-process path."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import bench
    elapsed = 0.5 if rank == 0 else 0.8  # the slower replica sets the job time
    dist.barrier()
    value, t = bench.replica_aggregate(elapsed, 100 * (rank + 1), world)
    q.put((rank, value, t))
    dist.destroy_process_group()


def test_two_replicas_max_time_sum_tokens():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, value, t in res:
        assert abs(t - 0.8) < 1e-9
        assert abs(value - 300 / 0.8) < 1e-6  # both ranks agree on the whole-job number


def test_single_replica():
    import bench
    v, t = bench.replica_aggregate(2.0, 50, 1)
    assert v == 25.0 and t == 2.0


def test_decode_bytes_model():
    import bench
    # fp16 B=1 S=2048 -> 14.29 GB per token-step; S -> 0: 13.21 GB
    assert abs(bench.decode_bytes_per_step(bench.LLAMA2_7B, 1, 2048) / 1e9 - 14.29) < 0.01
    assert abs(bench.decode_bytes_per_step(bench.LLAMA2_7B, 1, 0) / 1e9 - 13.21) < 0.01
    # int8 B=32 S=128 -> 8.89 GB (weights 1 B/elt, fp16 LM head and KV)
    assert abs(bench.decode_bytes_per_step(bench.LLAMA2_7B, 32, 128, 1.0) / 1e9 - 8.89) < 0.02
