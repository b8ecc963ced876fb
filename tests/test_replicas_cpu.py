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


def _run_parent(n, extra=()):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run", "--steps", "10",
                        "--warmup", "2", "--batch", "3"] + list(extra), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly one JSON line (rank 0's), whatever the replica count
    return json.loads(lines[0])


def test_bench_gpus_n_spawns_pinned_replicas():
    """`python bench.py --gpus N` with no launcher in the environment spawns N replicas itself (replicas only).
    --dry-run swaps the kernels for simulated per-rank times (rank r: 0.10 + 0.05 r s) and leaves the launcher, the
    rendezvous, the pinning and the MAX/SUM aggregation as they are."""
    for n in (2, 4):
        out = _run_parent(n)
        assert out["n_gpus"] == n and out["dry_run"] is True
        tokens = 3 * 10
        t_max = 0.10 + 0.05 * (n - 1)
        assert abs(out["value"] - n * tokens / t_max) < 1e-2            # SUM of tokens / MAX of times
        assert abs(out["ms_per_step"] - t_max / 10 * 1e3) < 1e-2
        assert out["replica_devices"] == list(range(n))                   # one device per replica: HIP_VISIBLE_DEVICES=i
        assert len(out["per_replica_tokens_per_s"]) == n
        for r, v in enumerate(out["per_replica_tokens_per_s"]):
            assert abs(v - tokens / (0.10 + 0.05 * r)) < 1e-2
        assert abs(out["scaling_efficiency"] - out["value"] / (n * out["per_replica_tokens_per_s"][0])) < 1e-3


def test_bench_single_replica_line_unchanged_by_launcher():
    out = _run_parent(1)
    assert out["n_gpus"] == 1 and out["replica_devices"] == [-1]  # N = 1: no spawn, no pinning, no process group
    assert abs(out["value"] - 30 / 0.10) < 1e-2


def test_replica_env_pins_one_device_each():
    import bench
    e = bench.replica_env(3, 8, 12345, {"CUDA_VISIBLE_DEVICES": "0,1", "PATH": "/x"})
    assert e["HIP_VISIBLE_DEVICES"] == "3" and e["RANK"] == "3" and e["LOCAL_RANK"] == "0" and e["WORLD_SIZE"] == "8"
    assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "12345" and "CUDA_VISIBLE_DEVICES" not in e
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/x"
