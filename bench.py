#!/usr/bin/env python3
"""bench.py -- Flat fp32 L2 top-10 over N=10M x d=768 synthetic vectors, batch-64 queries
(BASELINE.json configs[1]); one "step" = one batch of 64 queries answered end to end.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--dim D] [--batch B] [--topk K]

N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...
(one rank per GPU, RCCL): every rank holds its own 10M-row shard (weak scaling), answers the same
query batch on it, and the per-shard partial top-K lists are merged through an all-gather
(vectorsimilarity_amd/sharded.py).  Rank 0 prints ONE JSON line.

value      = distances/s = rows(all shards) * batch * steps / wall, wall bracketed by barrier +
             device sync, max over ranks.  Vectors are resident in HBM before timing; query upload,
             kernels, candidate download, host replay and reply construction are all inside.
roofline   = dominant scan kernel: algorithmic bytes (rows * 3072 B per launch) over
             its mean HIP-event duration, against 8 TB/s.
cpu_baseline = the oracle's AVX-512 port of the reference kernel + sequential heap (oracle/vso_fast.c)
             on a bounded sample of the same synthetic rows, all host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--seed", type=int, default=47)
    ap.add_argument("--cpu-sample-rows", type=int, default=400_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mfma", type=int, default=1)
    return ap.parse_args()


def cpu_baseline(args, VecSim):
    """cpu_baseline leg -- the ONLY place bench.py touches oracle/: the reference-order AVX-512 scan +
    sequential heap (oracle port) timed on a bounded sample of the same synthetic rows, and, as the
    checker, compared bit for bit with the GPU path on exactly that sample."""
    from oracle import vso
    vso.build()
    n = min(args.cpu_sample_rows, args.rows)
    rows = vso.synth_rows_f32(args.seed, 0, n, args.dim)          # same generator, same seed: rows 0..n-1
    queries = vso.synth_rows_f32(args.seed + 1, 0, args.batch, args.dim)
    threads = max(1, min(os.cpu_count() or 1, args.batch))
    vso.flat_topk_batch_fast(vso.F32, vso.L2, rows[:20000], queries[:threads], args.topk, args.dim, threads)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        labels, scores, fast = vso.flat_topk_batch_fast(vso.F32, vso.L2, rows, queries, args.topk, args.dim, threads)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    # checker: a GPU index over the same n rows must give the same labels, order and scores
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, args.dim, VecSim.VecSimMetric_L2
    small = VecSim.BFIndex(p)
    small.add_synthetic(n, args.seed)
    small.set_option("mfma", args.mfma)
    gl, gs = small.knn_query(queries, args.topk)
    same = bool(np.array_equal(gl, labels.astype(np.int64)) and np.array_equal(gs, scores))
    return {"value": n * args.batch / best, "unit": "distances/s", "cores": threads, "kind": "port",
            "sample": "first %d of the %d synthetic rows x %d queries, top-%d, best of 3, %s kernel; "
                      "GPU result on the same sample bit-identical: %s" % (
                          n, args.rows, args.batch, args.topk, "AVX-512 intrinsics" if fast else "portable lanes", same)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # launched through torch.distributed.run (RANK set): always bring RCCL up, so a 1-rank run exercises
    # the same collective path as the 2/4/8-rank runs
    if world > 1 or "RANK" in os.environ:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    os.environ["VECSIM_GPU_DEVICE"] = str(local_rank)

    from vectorsimilarity_amd import VecSim
    from vectorsimilarity_amd.sharded import ShardedFlatIndex

    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, args.dim, VecSim.VecSimMetric_L2
    ix = ShardedFlatIndex(p, rank=rank, world=world, dist=dist)
    # weak scaling: every rank ingests `rows` synthetic vectors (its own seed => its own shard content)
    ix.add_synthetic_local(args.rows, args.seed + 1000 * rank)
    ix.local.set_option("mfma", args.mfma)

    from vectorsimilarity_amd import synth
    n_batches = args.warmup + args.steps
    qall = synth.rows_f32(args.seed + 1, 0, args.batch * n_batches, args.dim)
    qall = qall.reshape(n_batches, args.batch, args.dim)

    def sync():
        ix.device_sync()
        if dist is not None:
            dist.barrier()
        ix.device_sync()

    for w in range(args.warmup):
        ix.knn_query(qall[w], args.topk)
    ix.local.reset_stats()
    sync()
    t0 = time.perf_counter()
    last = None
    for s in range(args.steps):
        last = ix.knn_query(qall[args.warmup + s], args.topk)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    st = ix.local.stats()

    if rank == 0:
        total_rows = args.rows * world
        dists = total_rows * args.batch * args.steps
        launches = max(1, st["scan_launches"])
        avg_ms = st["scan_ms"] / launches
        bytes_per_launch = st["scan_bytes"] / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        try:  # HBM bytes per launch measured with the PMC counters (separate rocprofv3 passes, committed summary)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f).get(st["scan_kernel"])
            if t and t["workload"] == {"rows": args.rows, "dim": args.dim, "batch": args.batch}:
                traffic = t["bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "distances/sec, Flat fp32 L2 top-%d, N=%s d=%d, batch-%d" % (
                args.topk, "%dM" % (args.rows // 1_000_000) if args.rows % 1_000_000 == 0 else str(args.rows),
                args.dim, args.batch),
            "value": dists / dt,
            "unit": "distances/s",
            "qps": args.batch * args.steps / dt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "flat_fp32_l2_top%d" % args.topk, "rows_per_gpu": args.rows, "dim": args.dim,
                       "batch": args.batch, "k": args.topk, "sharding": "rows x %d, all-gather top-K merge" % world
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": st["scan_kernel"], "avg_kernel_ms": avg_ms, "launches": int(st["scan_launches"]),
                         "algorithmic_bytes_per_launch": bytes_per_launch, "other_kernels_ms_per_step":
                             st["other_ms"] / max(1, args.steps)},
            "candidates_per_query": st["candidates"] / max(1, args.steps * args.batch),
            "fallbacks": int(st["fallbacks"]),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, VecSim)
        # size-independent property at full size: replies are ascending in score
        labels, scores = last
        out["sorted"] = bool(np.all(np.diff(scores, axis=1) >= 0) and np.all(labels >= 0))
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
