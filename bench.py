#!/usr/bin/env python3
"""bench.py -- decode throughput of the MI355X-native Llama-2-7B hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the configuration the metric is quoted on; fits one GPU):
Llama-2-7B, 32 layers, fp16 weights/activations/KV, batch 1, context 2048 (the K timed steps
are the last K positions of a 2048-token context, KV pre-filled), random-init weights,
synthetic token ids.  One "step" = one generated token: embedding -> 32 decoder layers ->
final RMSNorm -> LM head -> top-k -> sampling -> token id to pinned host memory, all through
the C ABI (include/llmie.h), captured once in a hipGraph and replayed.

N > 1 = N independent data-parallel replicas (one process per GPU, no collective on the data
path; torch.distributed/gloo only for the start/stop barrier and the MAX over ranks).
`python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment) spawns the N replicas
itself: the parent never imports torch or touches a GPU, it starts N fresh child processes pinned one
per device with HIP_VISIBLE_DEVICES=i (RANK/WORLD_SIZE/MASTER_* set), waits, and relays rank 0's line.

Prints ONE JSON line (rank 0).  Extra keys beside the driver contract:
  roofline      dominant kernel (gate/up GEMV + SwiGLU): algorithmic bytes per launch / mean launch
                duration measured with hipEvents on the launch stream (engine profiling pass, eager
                launches of the same step right after the timed region).
  cpu_baseline  the oracle (CPU restatement of the reference kernels, oracle/) timed on this host
                on a bounded sample (1 of 32 layers x32 + LM head), rank 0 / N=1 only.
  breakdown     per-op device time of one step from the same profiling pass.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 measured copy)

LLAMA2_7B = dict(head_num=32, kv_head_num=32, head_size=128, inter_size=11008, num_layers=32, vocab_size=32000)


def load_llmie():
    spec = importlib.util.spec_from_file_location("llmie_amd", os.path.join(ROOT, "llm-inference-engine_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["llmie_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def build_weights(torch, cfg, seed):
    """random-init Llama-2 weights of the named architecture (fp16, HF layout), resident in HBM"""
    dev = "cuda"
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    H = cfg["head_num"] * cfg["head_size"]
    QKV = (cfg["head_num"] + 2 * cfg["kv_head_num"]) * cfg["head_size"]
    I, L, V = cfg["inter_size"], cfg["num_layers"], cfg["vocab_size"]

    def w(n, k):
        # zero-mean U(-a, a) with std 1/sqrt(k): activations stay O(1) through 32 layers
        a = (3.0 / k) ** 0.5
        return ((torch.rand((n, k), generator=g, device=dev, dtype=torch.float32) * 2 - 1) * a).to(torch.float16)

    layers = []
    for _ in range(L):
        layers.append(dict(attn_norm=torch.ones(H, dtype=torch.float16, device=dev),
                           ffn_norm=torch.ones(H, dtype=torch.float16, device=dev),
                           qkv=w(QKV, H), o=w(H, H), gate_up=w(2 * I, H), down=w(H, I)))
    return dict(layers=layers, embed=w(V, H), lm_head=w(V, H),
                final_norm=torch.ones(H, dtype=torch.float16, device=dev), gen=g)


def quantize_layers(torch, llmie, layers, wfmt, group=128):
    """int8 (per-row scale) / int4 (per-group scale) / fp8-e4m3 (per-row fp32 scale) copies of the four big matrices"""
    out = []
    for lw in layers:
        q = dict(attn_norm=lw["attn_norm"], ffn_norm=lw["ffn_norm"])
        for name in ("qkv", "o", "gate_up", "down"):
            w = lw[name]
            n, k = w.shape
            if wfmt == "fp8":
                wq = torch.empty((n, k), dtype=torch.uint8, device="cuda")
                sc = torch.empty(n, dtype=torch.float32, device="cuda")
                llmie.quantize_fp8(w, wq, sc)
            elif wfmt == "int8":
                wq = torch.empty((n, k), dtype=torch.int8, device="cuda")
                sc = torch.empty(n, dtype=torch.float16, device="cuda")
                llmie.quantize_w8(w, wq, sc)
            else:
                wq = torch.empty((n, k // 2), dtype=torch.uint8, device="cuda")
                sc = torch.empty((n, k // group), dtype=torch.float16, device="cuda")
                llmie.quantize_w4(w, wq, sc, group)
            q[name] = dict(data=wq, scale=sc)
        out.append(q)
    return out


def make_decoder(torch, llmie, cfg, weights, layers, wfmt, batch, max_seq, kv_fp8=False):
    kv_shape = (cfg["num_layers"], batch, cfg["kv_head_num"], max_seq, cfg["head_size"])
    g = weights["gen"]
    if kv_fp8:  # e4m3 cache bytes, stored = e4m3(x / scale): random valid codes (|x| <= 1.75 * 2^4 / 32), no NaN patterns
        kc = torch.randint(0, 0x58, kv_shape, generator=g, device="cuda", dtype=torch.uint8)
        vc = torch.randint(0, 0x58, kv_shape, generator=g, device="cuda", dtype=torch.uint8)
        kc |= (torch.randint(0, 2, kv_shape, generator=g, device="cuda", dtype=torch.uint8) << 7)
        vc |= (torch.randint(0, 2, kv_shape, generator=g, device="cuda", dtype=torch.uint8) << 7)
    else:
        kc = (torch.randn(kv_shape, generator=g, device="cuda", dtype=torch.float32) * 0.5).to(torch.float16)
        vc = (torch.randn(kv_shape, generator=g, device="cuda", dtype=torch.float32) * 0.5).to(torch.float16)
    fmt = {"f16": llmie.W_F16, "int8": llmie.W_INT8, "int4": llmie.W_INT4, "fp8": llmie.W_FP8}[wfmt]
    ecfg = dict(cfg, max_seq_len=max_seq, max_batch=batch, rotary_dim=cfg["head_size"], rotary_base=10000.0,
                rms_eps=1e-5, dtype=llmie.F16, wfmt=fmt, int4_group=128, kv_fmt=llmie.KV_FP8 if kv_fp8 else llmie.KV_NATIVE,
                k_scale=1.0 / 32, v_scale=1.0 / 32)
    return llmie.Decoder(ecfg, layers), kc, vc


def decode_bytes_per_step(cfg, batch, ctx, wbytes=2.0, kvbytes=2):
    """weights once per step (wbytes per element; LM head stays fp16) + KV read + KV append (kvbytes/element)"""
    H = cfg["head_num"] * cfg["head_size"]
    KVH = cfg["kv_head_num"] * cfg["head_size"]
    I, L, V = cfg["inter_size"], cfg["num_layers"], cfg["vocab_size"]
    weights = L * ((H + 2 * KVH) * H + H * H + 3 * H * I) * wbytes + V * H * 2
    kv = batch * L * 2 * ctx * KVH * kvbytes + batch * L * 2 * KVH * kvbytes
    return int(weights + kv)


def cpu_baseline(cfg, ctx, budget_s=25.0):
    """Times the oracle's decode step on the host cores (bounded sample: one layer + LM head)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    rng = np.random.default_rng(0)
    nh, kvh, hs, I, V = cfg["head_num"], cfg["kv_head_num"], cfg["head_size"], cfg["inter_size"], cfg["vocab_size"]
    H, QKV = nh * hs, (nh + 2 * kvh) * hs

    def w(n, k):
        return rng.uniform(-1, 1, (n, k)).astype(np.float32) * np.float32((3.0 / k) ** 0.5)

    layer = dict(attn_norm=np.ones(H, np.float32), qkv=w(QKV, H), qkv_bias=None, o=w(H, H), o_bias=None,
                 ffn_norm=np.ones(H, np.float32), gate_up=w(2 * I, H), down=w(H, I))
    lm = w(V, H)
    kc = (rng.standard_normal((1, 1, kvh, ctx, hs)) * 0.5).astype(np.float32)
    vc = (rng.standard_normal((1, 1, kvh, ctx, hs)) * 0.5).astype(np.float32)
    ocfg = dict(head_num=nh, kv_head_num=kvh, head_size=hs, inter_size=I, num_layers=1, vocab=V, max_seq_len=ctx,
                rotary_dim=hs, rotary_base=10000.0, rms_eps=1e-5)
    x = rng.standard_normal((1, H)).astype(np.float32)
    orc.self_decoder(ocfg, [layer], x, kc, vc, ctx)  # warm-up (page in)
    t_layer, t_lm, reps = [], [], 0
    t_begin = time.perf_counter()
    while reps < 3 or (time.perf_counter() - t_begin < budget_s and reps < 12):
        t0 = time.perf_counter()
        h = orc.self_decoder(ocfg, [layer], x, kc, vc, ctx)
        t1 = time.perf_counter()
        hn, _ = orc.rmsnorm(h, np.ones(H, np.float32), 1e-5)
        logits = orc.linear(hn, lm)
        ids, vals = orc.topk(logits, 4)
        orc.sampling(ids, vals, np.zeros(1, np.int32), np.zeros(1, np.uint8), ctx, 2, V)
        t2 = time.perf_counter()
        t_layer.append(t1 - t0)
        t_lm.append(t2 - t1)
        reps += 1
    tl, tm = sorted(t_layer)[len(t_layer) // 2], sorted(t_lm)[len(t_lm) // 2]
    step_s = cfg["num_layers"] * tl + tm
    threads = int(orc.lib().orc_num_threads())
    out = dict(value=1.0 / step_s, unit="tokens/s", cores=threads, kind="port",
               sample="config C: oracle (fp32 C restatement of the reference kernels, OpenMP) decode step at ctx %d: "
                      "1 of %d layers timed (median of %d runs: %.3f s) x%d + LM head/top-k/sampling (%.3f s)"
                      % (ctx, cfg["num_layers"], reps, tl, cfg["num_layers"], tm))
    # the same layer on ONE thread (BASELINE.md section 4 asks for 1 thread and all threads)
    orc.lib().orc_set_num_threads(1)
    t0 = time.perf_counter()
    orc.self_decoder(ocfg, [layer], x, kc, vc, ctx)
    t1 = time.perf_counter()
    hn, _ = orc.rmsnorm(x, np.ones(H, np.float32), 1e-5)
    orc.topk(orc.linear(hn, lm), 4)
    t2 = time.perf_counter()
    out["one_thread"] = dict(value=1.0 / (cfg["num_layers"] * (t1 - t0) + (t2 - t1)), unit="tokens/s", cores=1,
                             sample="config C on one thread: 1 layer (%.3f s) x%d + LM head (%.3f s), single run"
                                    % (t1 - t0, cfg["num_layers"], t2 - t1))
    orc.lib().orc_set_num_threads(threads)
    del layer, lm, kc, vc

    # configs A and B: one layer, prefill of `seq` tokens then one decode step
    def small(name, nh_, hs_, I_, seq, note):
        H_, QKV_ = nh_ * hs_, 3 * nh_ * hs_
        r = np.random.default_rng(1234)
        u = lambda n, k: r.uniform(-0.05, 0.05, (n, k)).astype(np.float32)
        lw = dict(attn_norm=np.ones(H_, np.float32), qkv=u(QKV_, H_), qkv_bias=None, o=u(H_, H_), o_bias=None,
                  ffn_norm=np.ones(H_, np.float32), gate_up=u(2 * I_, H_), down=u(H_, I_))
        c = dict(head_num=nh_, kv_head_num=nh_, head_size=hs_, inter_size=I_, num_layers=1, vocab=V, max_seq_len=seq + 1,
                 rotary_dim=hs_, rotary_base=10000.0, rms_eps=1e-5)
        xs = r.standard_normal((seq, H_)).astype(np.float32)
        res = {}
        for label, nt in (("all_threads", threads), ("one_thread", 1)):
            orc.lib().orc_set_num_threads(nt)
            tp, td = [], []
            for _ in range(3 if nt > 1 else 1):
                k_ = np.zeros((1, 1, nh_, seq + 1, hs_), np.float32)
                v_ = np.zeros_like(k_)
                a0 = time.perf_counter()
                h = orc.context_decoder(c, [lw], xs, k_, v_, [seq], [0])
                a1 = time.perf_counter()
                orc.self_decoder(c, [lw], h[-1:], k_, v_, seq + 1)
                a2 = time.perf_counter()
                tp.append(a1 - a0)
                td.append(a2 - a1)
            res[label] = dict(cores=nt, prefill_tokens_per_s=round(seq / sorted(tp)[len(tp) // 2], 2),
                              decode_tokens_per_s=round(1.0 / sorted(td)[len(td) // 2], 2))
        orc.lib().orc_set_num_threads(threads)
        res["sample"] = "%s: ONE layer, prefill of %d tokens then one decode step, fp32 oracle (all threads: median of 3 runs; one thread: one run)" % (note, seq)
        return res

    out["config_A"] = small("A", 4, 32, 344, 32, "config A (hidden 128, 4 heads, I 344, seq 32)")
    out["config_B"] = small("B", nh, hs, I, 128, "config B (Llama-2-7B geometry, seq 128)")
    return out


def replica_aggregate(elapsed_s, tokens_this_rank, world):
    """Independent data-parallel replicas (no collective on the data path): the job's time is the MAX over
    ranks, its work the SUM of the per-rank tokens.  torch.distributed (gloo, CPU tensors) is control plane only.
    Returns (whole_job_tokens_per_s, max_elapsed_s)."""
    if world <= 1:
        return tokens_this_rank / elapsed_s, elapsed_s
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed_s], dtype=torch.float64)
    n = torch.tensor([float(tokens_this_rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(n.item()) / float(t.item()), float(t.item())


def replica_gather(value, world):
    """every rank's scalar, in rank order (gloo all_gather of CPU tensors; control plane only)"""
    if world <= 1:
        return [float(value)]
    import torch
    import torch.distributed as dist
    mine = torch.tensor([float(value)], dtype=torch.float64)
    out = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def replica_env(rank, world, port, base_env):
    """environment of replica `rank`: pinned to device `rank` (it sees exactly one GPU, its cuda:0)"""
    env = dict(base_env)
    env.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HIP_VISIBLE_DEVICES=str(rank), LLMIE_REPLICA_PINNED="1")
    env.pop("CUDA_VISIBLE_DEVICES", None)
    env.pop("ROCR_VISIBLE_DEVICES", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def launch_replicas(n, argv):
    """Parent of `python bench.py --gpus N` (no launcher): N fresh child processes, one per device.  Nothing here imports
    torch or initialises the GPU (a process that has must never exec/re-exec).  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=replica_env(r, n, port, os.environ),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rc = procs[0].returncode
    for p in procs[1:]:
        p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--ctx", type=int, default=2048, help="context length the timed steps end at")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--layers", type=int, default=0, help="DEBUG ONLY: fewer layers (result marked invalid)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the int8/int4/short-context side measurements")
    ap.add_argument("--only", default="", help="profiling aid: run ONE configuration and print a short line: "
                                               "decode:<f16|int8|int4|fp8>:<batch>:<ctx>[:kvfp8] or prefill:<f16|int8|int4|fp8>:<batch>:<seq>")
    ap.add_argument("--dry-run", action="store_true",
                    help="replica plumbing only (no GPU, no kernels): simulated per-rank times through the real launcher/aggregation")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the parent of N pinned replicas BEFORE torch / the GPU is touched
        raise SystemExit(launch_replicas(args.gpus, sys.argv[1:]))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if args.dry_run:
        # no kernels: rank r "takes" 0.10 + 0.05 r seconds for batch * steps tokens; everything after that is the real path
        el_r, tok_r = 0.10 + 0.05 * rank, args.batch * args.steps
        value, elapsed = replica_aggregate(el_r, tok_r, world)
        per = replica_gather(tok_r / el_r, world)
        devs = replica_gather(float(os.environ.get("HIP_VISIBLE_DEVICES", "-1") or -1), world)
        if rank == 0:
            print(json.dumps({"metric": "decode_tokens_per_s", "value": round(value, 4), "unit": "tokens/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                              "higher_is_better": True, "scaling": "weak", "dry_run": True,
                              "per_replica_tokens_per_s": [round(v, 4) for v in per],
                              "replica_devices": [int(d) for d in devs],
                              "scaling_efficiency": round(value / (world * per[0]), 4)}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    # launched by torchrun: one visible node, device = LOCAL_RANK; launched by launch_replicas: pinned, the only device is 0
    torch.cuda.set_device(0 if (world == 1 or os.environ.get("LLMIE_REPLICA_PINNED")) else local_rank)
    llmie = load_llmie()
    llmie.lib()  # fail loudly if the HIP library is missing

    cfg = dict(LLAMA2_7B)
    if args.layers:
        cfg["num_layers"] = args.layers
    B, K, W, S = args.batch, args.steps, args.warmup, args.ctx
    H, V = cfg["head_num"] * cfg["head_size"], cfg["vocab_size"]
    weights = build_weights(torch, cfg, seed=1234 + rank)
    dev = "cuda"
    TOPK, BPR = 4, 8

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    def run_decode(wfmt, layers, B, S, K, W, profile_steps, sync_ranks, kv_fp8=False):
        """K timed decode steps ending at context S (hipGraph replay); returns (elapsed_s, profile dict or None)"""
        dec, kc, vc = make_decoder(torch, llmie, cfg, weights, layers, wfmt, B, S, kv_fp8)
        ids = torch.randint(0, V, (B,), dtype=torch.int32, device=dev)
        hidden = torch.empty((B, H), dtype=torch.float16, device=dev)
        logits = torch.empty((B, V), dtype=torch.float16, device=dev)
        tmp_ids = torch.empty((B, BPR, TOPK), dtype=torch.int32, device=dev)
        tmp_vals = torch.empty((B, BPR, TOPK), dtype=torch.float16, device=dev)
        top_ids = torch.empty((B, TOPK), dtype=torch.int32, device=dev)
        top_vals = torch.empty((B, TOPK), dtype=torch.float16, device=dev)
        seq_len = torch.zeros(B, dtype=torch.int32, device=dev)
        finished = torch.zeros(B, dtype=torch.uint8, device=dev)
        host_tok = torch.empty(B, dtype=torch.int32, pin_memory=True)
        start_step = max(1, S - (K + W) + 1)
        step_dev = torch.tensor([start_step], dtype=torch.int32, device=dev)

        # One token step = 32 layers on the hidden state of the current token -> final norm + LM head -> top-k round 1 -> ONE tail
        # launch (top-k round 2, sampling, the NEXT token's input embedding into `hidden`, step counter += 1; round 3:
        # llmie_lm_head_sample_next, four launches less per step) -> token to pinned host memory.  The first token's embedding is
        # gathered once in front of the loop (it is the tail of the prompt's last step in a real run).
        llmie.input_embedding(ids, weights["embed"], hidden)

        def one_step():
            dec.forward(hidden, hidden, kc, vc, -1, step_dev=step_dev)
            dec.lm_head_sample(hidden, weights["final_norm"], weights["lm_head"], llmie.W_F16, logits, tmp_ids, tmp_vals,
                               top_ids, top_vals, seq_len, finished, ids, step=-1, end_id=-1, blocks_per_row=BPR,
                               step_dev=step_dev, embed=weights["embed"], next_hidden=hidden, advance=True)
            host_tok.copy_(ids, non_blocking=True)

        stream = torch.cuda.Stream()
        graph = None
        with torch.cuda.stream(stream):
            one_step()  # eager once (also validates every launch)
            step_dev.fill_(start_step)
        stream.synchronize()
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                one_step()
            step_dev.fill_(start_step)
        torch.cuda.synchronize()
        host_step = [start_step]

        def run(n):
            with torch.cuda.stream(stream):
                for _ in range(n):
                    if host_step[0] > S:  # context full: wrap (only when steps+warmup > ctx)
                        step_dev.fill_(1)
                        host_step[0] = 1
                    if graph is not None:
                        graph.replay()
                    else:
                        one_step()
                    host_step[0] += 1

        run(W)
        torch.cuda.synchronize()
        if sync_ranks:
            barrier()
        t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if sync_ranks:
            barrier()
        solo = None
        if sync_ranks and world > 1:
            # the same K steps on rank 0 ALONE (the other replicas idle at the barrier): the one-replica reference the
            # scaling efficiency of this job is quoted against
            if rank == 0:
                ts0 = time.perf_counter()
                run(K)
                torch.cuda.synchronize()
                solo = time.perf_counter() - ts0
            barrier()
        prof = None
        if profile_steps:
            # per-kernel timing with hipEvents on the launch stream (eager launches of the same step)
            P = profile_steps
            with torch.cuda.stream(stream):
                step_dev.fill_(max(1, S - P + 1))
                one_step()  # untimed: re-warm after the fill
                step_dev.fill_(max(1, S - P + 1))
                dec.profile_begin(P * (cfg["num_layers"] * 12 + 8))
                for _ in range(P):
                    one_step()
                prof = dec.profile_end()
        dec.close()
        del graph, kc, vc
        torch.cuda.empty_cache()
        return t1 - t0, prof, solo

    P = 4

    def run_prefill(b, s, wfmt="f16", layers=None, profile=False):
        """prefill of b sequences x s tokens (packed), all layers; returns (seconds per pass, op profile or None, flops)"""
        dec, kc, vc = make_decoder(torch, llmie, cfg, weights, layers or weights["layers"], wfmt, b, s)
        T = b * s
        ids = torch.randint(0, V, (T,), dtype=torch.int32, device=dev)
        hid = torch.empty((T, H), dtype=torch.float16, device=dev)
        lens = torch.full((b,), s, dtype=torch.int32, device=dev)
        hist = torch.zeros(b, dtype=torch.int32, device=dev)

        def once():
            llmie.input_embedding(ids, weights["embed"], hid)
            dec.prefill(hid, hid, kc, vc, lens, hist, s)

        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            once()   # eager once (also validates every launch)
        stream.synchronize()
        graph = None
        if not args.no_graph:   # fixed shape: the whole pass replays as one hipGraph, as the decode step does
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    once()
                with torch.cuda.stream(stream):
                    graph.replay()
            except RuntimeError:   # a side configuration must not take the bench line down: eager launches instead
                graph = None
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(reps):
                if graph is not None:
                    graph.replay()
                else:
                    once()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        del graph
        pprof = None
        if profile:
            dec.profile_begin(2 * (cfg["num_layers"] * 12 + 8))
            for _ in range(2):
                once()
            pprof = dec.profile_end()
        Hh, KVH, I_, L_ = H, cfg["kv_head_num"] * cfg["head_size"], cfg["inter_size"], cfg["num_layers"]
        flops = T * 2.0 * L_ * ((Hh + 2 * KVH) * Hh + Hh * Hh + 3 * Hh * I_) + b * L_ * 4.0 * Hh * s * (s + 1) / 2
        dec.close()
        del kc, vc, hid
        torch.cuda.empty_cache()
        return el, pprof, flops

    def rocprof_ref(key):
        """kernel averages of the committed rocprofv3 runs of these same configurations (profiles/r03_rocprof_roofline.json,
        written by tools/summarize_prof.py from the kernel traces / PMC passes): the trace-side counterpart of the live
        hipEvent numbers"""
        try:
            with open(os.path.join(ROOT, "profiles", "r03_rocprof_roofline.json")) as f:
                return json.load(f).get(key)
        except (OSError, ValueError):
            return None

    def roofline_block(kernel, bound, work_per_launch, us_per_launch, launches, peak, unit, ref_key):
        """achieved = algorithmic work per launch / mean launch time.  Two time bases, named apart: frac_event = hipEvent
        bracket around each launch on the launch stream, measured live in this run (includes the launch gap);
        frac_rocprof_trace = the kernel's average duration in the committed rocprofv3 kernel trace of the same configuration.
        `frac` is the live one."""
        scale = 1e9 if unit == "GB/s" else 1e12
        ach = work_per_launch / (us_per_launch * 1e-6) / scale
        blk = dict(bound=bound, kernel=kernel, achieved=round(ach, 1), peak=peak, unit=unit, frac=round(ach / peak, 4),
                   frac_event=round(ach / peak, 4), frac_rocprof_trace=None, traffic=None,
                   algorithmic_work_per_launch=work_per_launch, us_per_launch=round(us_per_launch, 2), launches_timed=launches)
        ref = rocprof_ref(ref_key)
        if ref and not args.layers:
            blk["rocprof_avg_us"] = ref.get("avg_us")
            if ref.get("avg_us"):
                blk["frac_rocprof_trace"] = round(work_per_launch / (ref["avg_us"] * 1e-6) / scale / peak, 4)
            blk["traffic"] = ref.get("hbm_bytes_per_launch")   # PMC passes of exactly this kernel name and grid, or None
            blk["traffic_source"] = ref.get("traffic_source")
            blk["rocprof_kernel"], blk["rocprof_grid"] = ref.get("kernel"), ref.get("grid")
            blk["rocprof_source"] = ref.get("source")
        return blk

    if args.only:
        kind, fmt_, b_, s_ = args.only.split(":")[:4]
        b_, s_ = int(b_), int(s_)
        ql = weights["layers"] if fmt_ == "f16" else quantize_layers(torch, llmie, weights["layers"], fmt_)
        if kind == "decode":
            kv8 = args.only.endswith(":kvfp8")
            el, pr, _ = run_decode(fmt_, ql, b_, s_, min(K, 32), min(W, 4), P, False, kv8)
            print(json.dumps({"only": args.only, "tokens_per_s": round(b_ * min(K, 32) / el, 1), "ms_per_step": round(el / min(K, 32) * 1e3, 4),
                              "ops_us_per_launch": {op: round(ms / n * 1e3, 2) for op, (ms, n) in pr.items() if n}}), flush=True)
        else:
            el, pr, fl = run_prefill(b_, s_, fmt_, ql, True)
            print(json.dumps({"only": args.only, "tokens_per_s": round(b_ * s_ / el, 1), "ms": round(el * 1e3, 3),
                              "TFLOP_per_s": round(fl / el / 1e12, 1),
                              "ops_us_per_launch": {op: round(ms / n * 1e3, 2) for op, (ms, n) in pr.items() if n}}), flush=True)
        return

    own_elapsed, prof, solo_elapsed = run_decode("f16", weights["layers"], B, S, K, W, P, True)
    value, elapsed = replica_aggregate(own_elapsed, B * K, world)
    per_replica = replica_gather(B * K / own_elapsed, world)
    ms_per_step = elapsed / K * 1e3

    breakdown = {op: dict(us_per_launch=round(ms / n * 1e3, 2), launches_per_step=n // P,
                          us_per_step=round(ms / P * 1e3, 1))
                 for op, (ms, n) in prof.items() if n}
    I = cfg["inter_size"]
    gu_bytes = 2 * I * H * 2  # fused gate_up matrix streamed once per launch (180.4 MB)
    gu_ms, gu_n = prof["gate_up_swiglu"]
    gu_us = gu_ms / gu_n * 1e3
    roofline = roofline_block("gemv_ksplit_kernel<M=%d,RPW=4,XC=2,fp16> (RMSNorm + gate/up projection + SwiGLU)" % B, "hbm", gu_bytes,
                              gu_us, gu_n, HBM_PEAK_GBS, "GB/s", "decode_f16_b1_ctx2048" if (B == 1 and S == 2048) else "")
    step_bytes = decode_bytes_per_step(cfg, B, S)
    whole = dict(algorithmic_bytes_per_step=step_bytes,
                 achieved_GBs=round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                 frac_of_hbm_peak=round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))

    out = {
        "metric": "decode_tokens_per_s", "value": round(value, 2), "unit": "tokens/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "Llama-2-7B 32-layer fp16 decode, batch %d, ctx %d (BASELINE configs[2]); "
                               "random-init weights, synthetic ids, KV pre-filled" % (B, S),
                   "global_batch": world * B, "seq_len": S, "parallelism": "replicas x%d (no collective)" % world,
                   "graph": not args.no_graph, "layers": cfg["num_layers"]},
        "metric_full": "decode tokens/s + prefill tokens/s, Llama-2-7B fp16 & int8, 1 MI355X",
        "roofline": roofline, "whole_step": whole, "breakdown": breakdown,
        "per_replica_tokens_per_s": [round(v, 2) for v in per_replica],
    }
    if world > 1 and solo_elapsed:
        # whole job / (N x one replica running alone on this node), both measured inside this run
        out["solo_replica_tokens_per_s"] = round(B * K / solo_elapsed, 2)
        out["scaling_efficiency"] = round(value / (world * (B * K / solo_elapsed)), 4)
    # ---- other configurations of the metric (rank 0, single GPU only; not the headline value) ----
    if rank == 0 and world == 1 and not args.no_extra:
        extra = {}

        def record(name, wfmt, layers, b, s, wbytes, kv_fp8=False, profile=False):
            k, w_ = min(K, 32), min(W, 4)
            el, pr, _ = run_decode(wfmt, layers, b, s, k, w_, P if profile else 0, False, kv_fp8)
            ms = el / k * 1e3
            nbytes = decode_bytes_per_step(cfg, b, s, wbytes, 1 if kv_fp8 else 2)
            extra[name] = dict(tokens_per_s=round(b * k / el, 1), ms_per_step=round(ms, 4), batch=b, ctx=s,
                               algorithmic_GB_per_step=round(nbytes / 1e9, 3),
                               frac_of_hbm_peak=round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            return pr

        def record_prefill(name, b, s, wfmt="f16", layers=None, profile=False):
            el, pr, flops = run_prefill(b, s, wfmt, layers, profile)
            extra[name] = dict(tokens_per_s=round(b * s / el, 1), ms=round(el * 1e3, 3), batch=b, seq=s,
                               TFLOP_per_s=round(flops / el / 1e12, 1),
                               frac_of_mfma_peak=round(flops / el / (5.0e15 if wfmt == "fp8" else 2.5e15), 4))
            # the HBM side of the same pass: layer weights once + K / V rows written (short prefills are weight-stream bound:
            # at 128 tokens the pass is ~128 FLOP/B, below the ridge)
            wb_ = {"f16": 2.0, "int8": 1.0, "fp8": 1.0, "int4": 0.5 + 2.0 / 128}[wfmt]
            Hh, KVH, I_, L_ = H, cfg["kv_head_num"] * cfg["head_size"], cfg["inter_size"], cfg["num_layers"]
            hb = L_ * ((Hh + 2 * KVH) * Hh + Hh * Hh + 3 * Hh * I_) * wb_ + b * s * L_ * 2 * KVH * 2
            extra[name]["algorithmic_GB_hbm"] = round(hb / 1e9, 3)
            extra[name]["frac_of_hbm_peak"] = round(hb / el / 1e9 / HBM_PEAK_GBS, 4)
            extra[name]["bound"] = "hbm" if flops / hb < 2.5e15 / (HBM_PEAK_GBS * 1e9) else "mfma"
            return pr

        if os.environ.get("LLMIE_BENCH_SMALL_BATCH_SWEEP"):  # development: small-batch decode over the weight formats
            for fmt_, wb_ in (("f16", 2.0), ("int8", 1.0), ("fp8", 1.0), ("int4", 0.5 + 2.0 / 128)):
                ql = weights["layers"] if fmt_ == "f16" else quantize_layers(torch, llmie, weights["layers"], fmt_)
                for b in (2, 3, 4, 6, 8):
                    record("decode_%s_b%d_ctx512" % (fmt_, b), fmt_, ql, b, 512, wb_)
                del ql
                torch.cuda.empty_cache()
            print(json.dumps({k: v["tokens_per_s"] for k, v in extra.items()}), flush=True)
            return
        pp = record_prefill("prefill_f16_b1_s2048", 1, 2048, profile=True)
        if pp and pp.get("gate_up_swiglu", (0, 0))[1]:
            # dominant prefill kernel: the gate/up projection GEMM with the SwiGLU epilogue (2 * T * 2I * H flops per launch)
            ms_, n_ = pp["gate_up_swiglu"]
            out["roofline_prefill"] = roofline_block(
                "gemm8p_kernel<fp16,SwiGLU>: 256 x 256 LDS-DMA MFMA tiles, eight-phase ping-pong schedule (gate/up projection + SwiGLU epilogue), T = 2048", "mfma",
                2.0 * 2048 * 2 * cfg["inter_size"] * H, ms_ / n_ * 1e3, n_, 2500.0, "TFLOP/s", "prefill_f16_b1_s2048")
            out["roofline_prefill"]["whole_pass"] = dict(extra["prefill_f16_b1_s2048"])
        record_prefill("prefill_f16_b8_s512", 8, 512)
        record_prefill("prefill_f16_b1_s128", 1, 128)   # BASELINE configs[1] shape (all 32 layers)
        q8 = quantize_layers(torch, llmie, weights["layers"], "int8")
        # the int8 half of the metric's prefill side (round 3): weight-only int8 through the eight-phase GEMM's int8 B-operand form
        p8p = record_prefill("prefill_int8_b1_s2048", 1, 2048, "int8", q8, profile=True)
        record_prefill("prefill_int8_b8_s512", 8, 512, "int8", q8)
        record_prefill("prefill_int8_b1_s128", 1, 128, "int8", q8)
        for nm in ("b1_s2048", "b8_s512", "b1_s128"):
            extra["prefill_int8_" + nm]["vs_f16"] = round(extra["prefill_int8_" + nm]["tokens_per_s"] / extra["prefill_f16_" + nm]["tokens_per_s"], 4)
        if p8p and p8p.get("gate_up_swiglu", (0, 0))[1]:
            ms_, n_ = p8p["gate_up_swiglu"]
            out["roofline_prefill_int8"] = roofline_block(
                "gemm8p_kernel<fp16 x int8,SwiGLU,WQ=8>: raw int8 weight tiles by LDS-DMA, de-quantised at fragment read (gate/up projection + SwiGLU epilogue), T = 2048",
                "mfma", 2.0 * 2048 * 2 * cfg["inter_size"] * H, ms_ / n_ * 1e3, n_, 2500.0, "TFLOP/s", "prefill_int8_b1_s2048")
            out["roofline_prefill_int8"]["whole_pass"] = dict(extra["prefill_int8_b1_s2048"])
            out["roofline_prefill_int8"]["ops_us_per_launch"] = {op: round(ms / n * 1e3, 2) for op, (ms, n) in p8p.items() if n}
        record("decode_int8_b1_ctx2048", "int8", q8, 1, 2048, 1.0)
        p8 = record("decode_int8_b32_ctx128", "int8", q8, 32, 128, 1.0, profile=True)   # BASELINE configs[3]
        if p8 and p8.get("gate_up_swiglu", (0, 0))[1]:
            # dominant kernel of the int8 batch step: the packed-weight gate/up projection (RMSNorm prologue, SwiGLU epilogue)
            ms_, n_ = p8["gate_up_swiglu"]
            out["roofline_int8"] = roofline_block(
                "pk_mfma_kernel<MT=2,int8,SwiGLU,x32> (RMSNorm + gate/up projection + SwiGLU on the tile-packed int8 image), batch 32",
                "hbm", 2 * cfg["inter_size"] * H * 1, ms_ / n_ * 1e3, n_, HBM_PEAK_GBS, "GB/s", "decode_int8_b32_ctx128")
            out["roofline_int8"]["whole_step"] = dict(extra["decode_int8_b32_ctx128"])
            out["roofline_int8"]["ops_us_per_launch"] = {op: round(ms / n * 1e3, 2) for op, (ms, n) in p8.items() if n}
        record("decode_int8_b32_ctx2048", "int8", q8, 32, 2048, 1.0)          # KV-dominated (SURVEY 8d cfg D)
        record("decode_int8_b32_ctx2048_kvfp8", "int8", q8, 32, 2048, 1.0, True)   # same with the e4m3 KV cache
        record("decode_f16_b1_ctx128", "f16", weights["layers"], 1, 128, 2.0)
        record("decode_f16_b1_ctx2048_kvfp8", "f16", weights["layers"], 1, 2048, 2.0, True)
        del q8
        torch.cuda.empty_cache()
        q4 = quantize_layers(torch, llmie, weights["layers"], "int4")
        record("decode_int4_b1_ctx2048", "int4", q4, 1, 2048, 0.5 + 2.0 / 128)
        record("decode_int4_b32_ctx128", "int4", q4, 32, 128, 0.5 + 2.0 / 128)
        record_prefill("prefill_int4_b1_s2048", 1, 2048, "int4", q4)   # fp16 image of one matrix at a time + the fp16 GEMM
        record_prefill("prefill_int4_b8_s512", 8, 512, "int4", q4)
        del q4
        torch.cuda.empty_cache()
        q8f = quantize_layers(torch, llmie, weights["layers"], "fp8")   # BASELINE configs[4]: fp8 batch sweep at ctx 512
        for b in (1, 2, 4, 8, 16, 32, 64, 128):
            record("decode_fp8_b%d_ctx512" % b, "fp8", q8f, b, 512, 1.0)
        record("decode_fp8_b128_ctx512_kvfp8", "fp8", q8f, 128, 512, 1.0, True)   # e4m3 weights + e4m3 KV cache
        record_prefill("prefill_fp8_b8_s512", 8, 512, "fp8", q8f)
        record_prefill("prefill_fp8_b1_s2048", 1, 2048, "fp8", q8f)
        del q8f
        torch.cuda.empty_cache()
        for b in (32, 128):
            record("decode_f16_b%d_ctx512" % b, "f16", weights["layers"], b, 512, 2.0)
        out["extra"] = extra
        try:   # bytes of layer-matrix storage an engine keeps resident, per format and residency flag (llmie_decoder_resident_weight_bytes)
            import ctypes as C_
            lib_ = llmie.lib()
            res = {}
            for nm, wf_ in (("f16", llmie.W_F16), ("int8", llmie.W_INT8), ("int4", llmie.W_INT4), ("fp8", llmie.W_FP8)):
                for tag, mb, fl in (("b1", 1, 0), ("b32", 32, 0), ("b32_packed_only", 32, llmie.DEC_PACKED_ONLY), ("b32_no_packed_copy", 32, llmie.DEC_NO_PACKED_COPY)):
                    if nm == "fp8" and mb == 32 and fl == llmie.DEC_PACKED_ONLY:
                        mb = 16
                    c_ = llmie.DecoderConfig(**dict(cfg, max_seq_len=2048, max_batch=mb, rotary_dim=cfg["head_size"], rotary_base=10000.0, rms_eps=1e-5,
                                                    dtype=llmie.F16, wfmt=wf_, int4_group=128, kv_fmt=0, k_scale=1.0, v_scale=1.0, flags=fl))
                    res["%s_%s" % (nm, tag)] = round(lib_.llmie_decoder_resident_weight_bytes(C_.byref(c_)) / 1e9, 3)
            out["resident_layer_weight_GB"] = res
        except Exception as e:   # a side report must never take the bench line down
            out["resident_layer_weight_GB"] = {"error": str(e)[:200]}

        # ---- the peak constants used above, with on-box measurements beside them: a device copy, and the
        #      vendor library GEMM (torch.matmul = hipBLASLt/rocBLAS) next to this library's GEMM at the same prefill shape
        def timed(fn, reps):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / reps

        peaks = dict(hbm_GBs_datasheet=HBM_PEAK_GBS, mfma_f16_TFLOPs_datasheet=2500.0, mfma_fp8_TFLOPs_datasheet=5000.0)
        try:
            src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
            dst = torch.empty_like(src)
            peaks["measured_copy_GBs"] = round(2 * src.numel() / timed(lambda: dst.copy_(src), 10) / 1e9, 1)  # read + write
            del src, dst
            # interleaved rounds in one process, median per arm (the clock the chip holds drifts over a run); the two token
            # counts are the bench's own prefill configurations (8 x 512 and 1 x 2048 tokens), the shape is the QKV projection
            def med(v):
                return sorted(v)[len(v) // 2]
            for Mg, key in ((4096, ""), (2048, "_2048tok")):
                Ng, Kg = 12288, 4096
                xa = torch.randn((Mg, Kg), device="cuda").half()
                wa = (torch.randn((Ng, Kg), device="cuda") / 64).half()
                ya = torch.empty((Mg, Ng), device="cuda", dtype=torch.float16)
                fl = 2.0 * Mg * Ng * Kg
                tv, tl = [], []
                for _ in range(3):
                    tv.append(timed(lambda: torch.matmul(xa, wa.t(), out=ya), 10))
                    tl.append(timed(lambda: llmie.linear(xa, wa, ya), 10))
                peaks["gemm_shape" + key] = [Mg, Ng, Kg]
                peaks["measured_vendor_gemm_f16_TFLOPs" + key] = round(fl / med(tv) / 1e12, 1)
                peaks["llmie_gemm_f16_TFLOPs" + key] = round(fl / med(tl) / 1e12, 1)
                if not key:   # e4m3 weights, per-token e4m3 activations: the figure includes the activation quantisation pass
                    wq = torch.empty((Ng, Kg), dtype=torch.uint8, device="cuda")
                    ws = torch.empty(Ng, dtype=torch.float32, device="cuda")
                    llmie.quantize_fp8(wa, wq, ws)
                    work = torch.empty(llmie.linear_fp8_workspace_bytes(Mg, Kg, Ng), dtype=torch.uint8, device="cuda")
                    peaks["llmie_gemm_fp8_TFLOPs"] = round(fl / med([timed(lambda: llmie.linear_fp8(xa, wq, ws, ya, work), 10) for _ in range(3)]) / 1e12, 1)
                    del wq, ws, work
            del xa, wa, ya
            torch.cuda.empty_cache()
        except Exception as e:  # a side measurement must never take the bench line down
            peaks["error"] = str(e)[:200]
        out["peaks"] = peaks
    if args.layers:
        out["config"]["INVALID_debug_layers"] = args.layers
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, S)
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
