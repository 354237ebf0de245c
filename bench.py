#!/usr/bin/env python3
"""bench.py -- Flat top-K over synthetic vectors resident in HBM; one "step" = one query batch answered end to end.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1|c3|c4|c5] [--rows R] [--dim D] [--batch B] [--topk K]

Configs (BASELINE.json `configs`; default c2 = the one `metric` is quoted on):
    c1  fp32 L2      100 K x 128   1 query    top-10     the reference's own CPU-runnable case
    c2  fp32 L2       10 M x 768   64 queries top-10     headline
    c3  int8 Cosine   50 M x 1024  256 queries top-100   int8 MFMA path
    c4  bf16 IP     12.5 M x 768   128 queries top-10    per GPU: 100 M rows over 8 GPUs, RCCL top-K merge
    c5  HNSW fp32 L2   1 M x 768   4096 queries top-10   M 16, efC 200, efR 128 (BASELINE: 10 M rows -- --rows 10000000; the graph is built
                                                          on the host cores first: ~3 min per million rows); --data uniform | lowrank

N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...  (one rank
per GPU).  Every rank holds its own shard of `rows` vectors (weak scaling; --scaling strong: rows / N of them, the job's
total stays `rows`), scans it for the same query batch, and the
per-shard candidate records are exchanged by ONE ncclAllGather per batch -- RCCL over xGMI, issued by the C++ host
library (csrc/vsgpu_comm.hip behind VecSimGpu_Sharded*, include/VecSim/vec_sim_gpu.h) -- and merged into the exact
single-index reply on every rank.  torch.distributed is the control plane only (gloo: rank 0's RCCL id, barriers, the
max-over-ranks of the wall time); it never touches the GPU, so the timed region is bracketed by barrier + the library's
own device drain (every C-API call returns with its HIP streams synchronised).  Rank 0 prints ONE JSON line.

value        = distances/s = rows(all shards) * batch * steps / wall (max over ranks).  Vectors are resident in HBM
               before timing; query upload, kernels, candidate download, exchange, host replay and reply construction
               are all inside.
roofline     = dominant scan kernel: algorithmic bytes (rows * storedDataSize per launch) over its mean
               HIP-event duration (measured live on the stream the kernel runs on), against 8 TB/s; for the int8
               config also the int8 MFMA rate against the 3944 TOP/s ceiling of MI355X_MICROARCH.md.
cpu_baseline = the oracle's restatement of the reference kernel + sequential heap on a bounded sample of the same
               synthetic rows: single thread (the reference's actual behaviour, brute_force.h:264-281) and all host
               cores (independent queries, as bindings.cpp:250-283 knn_parallel does); rank 0, N=1 only.
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
I8_MFMA_PEAK_TOPS = 3944  # same guide, 16x16x64 int8 ubench ceiling

# name: (type, metric, dim, rows per GPU, batch, k, dtype tag, generator, CPU sample rows)
CONFIGS = {
    "c1": ("FLOAT32", "L2", 128, 100_000, 1, 10, "f32", "rows_f32", 100_000),
    "c2": ("FLOAT32", "L2", 768, 10_000_000, 64, 10, "f32", "rows_f32", 1_000_000),    # CPU sample = N / 10
    "c3": ("INT8", "Cosine", 1024, 50_000_000, 256, 100, "i8", "rows_i8", 1_000_000),
    "c4": ("BFLOAT16", "IP", 768, 12_500_000, 128, 10, "bf16", "rows_bf16", 1_250_000),
    "c5": ("FLOAT32", "L2", 768, 1_000_000, 4096, 10, "f32", "rows_f32", 0),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--topk", type=int, default=0)
    ap.add_argument("--seed", type=int, default=47)
    ap.add_argument("--cpu-sample-rows", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--same-gpu", action="store_true", help="test aid: all ranks of a multi-rank run on GPU 0")
    ap.add_argument("--mfma", type=int, default=1)
    ap.add_argument("--readers", type=int, default=2,
                    help="threads submitting batches (the reference's readers: bindings.cpp:250-283 knn_parallel); 2 lets one "
                         "batch's upload / probe / re-rank / download / host replay overlap with the next batch's scan kernel")
    ap.add_argument("--data", default="lowrank", choices=["uniform", "lowrank"],
                    help="c5 only.  uniform: BASELINE's i.i.d. U[-1,1) rows (intrinsic dimension = d: no graph index finds neighbours "
                         "there); lowrank: 32 latent factors mixed into d dims + 5%% noise (embedding-like)")
    ap.add_argument("--ef", type=int, default=128, help="c5: efRuntime")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1.  weak (default, what the driver's plain --gpus N run measures): every rank holds --rows vectors, the "
                         "job holds N x rows (config 4 is defined that way: 100 M rows over 8 GPUs).  strong: the job holds --rows "
                         "vectors, rank r holds its share of them (BASELINE's headline reads N = 10 M at 1 / 2 / 4 / 8 GPUs)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="tuning: VecSimGpu_SetOption on the index (e.g. probe_div=64); not used by the default run")
    a = ap.parse_args()
    typ, metric, dim, rows, batch, k, tag, gen, cpu_rows = CONFIGS[a.config]
    a.type_name, a.metric_name, a.dtype, a.gen = typ, metric, tag, gen
    a.dim = a.dim or dim
    a.rows = a.rows or rows
    a.batch = a.batch or batch
    a.topk = a.topk or k
    a.cpu_sample_rows = a.cpu_sample_rows or cpu_rows
    return a


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(args, VecSim, synth):
    """cpu_baseline leg -- the ONLY place bench.py touches oracle/: the reference-order scan + sequential heap (oracle
    port) timed on a bounded sample of the same synthetic rows, and, as the checker, compared bit for bit with the GPU
    path on exactly that sample."""
    from oracle import vso
    vso.build()
    n = min(args.cpu_sample_rows, args.rows)
    vt = getattr(vso, {"FLOAT32": "F32", "INT8": "I8", "BFLOAT16": "BF16"}[args.type_name])
    gen = getattr(synth, args.gen)
    # the sample = rows 0..n-1 of the same generator and seed.  They are generated ON THE GPU into a small index (the device twin
    # of the host generator, csrc/exact_kernels.hpp k_fill_*; tests/test_gpu_flat_parity.py pins the twins on each other) and read
    # back as stored blobs -- the host generator takes minutes per million rows.  The same index is the checker's GPU side below.
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = getattr(VecSim, "VecSimType_" + args.type_name), args.dim, getattr(VecSim, "VecSimMetric_" + args.metric_name)
    small = VecSim.BFIndex(p)
    small.add_synthetic(n, args.seed)
    small.set_option("mfma", args.mfma)
    rows = small.stored_rows(0, n)
    head = gen(args.seed, 0, min(n, 64), args.dim)              # host twin on the first rows: same bytes
    assert np.array_equal(rows[:len(head), :head.view(np.uint8).reshape(len(head), -1).shape[1]], head.view(np.uint8).reshape(len(head), -1)) \
        or args.metric_name == "Cosine", "device and host generators disagree"
    qraw = gen(args.seed + 1, 0, args.batch, args.dim)
    if args.metric_name == "Cosine":                         # int8 Cosine: stored blob = elements + float norm
        def with_norm(a):
            out = np.zeros((a.shape[0], args.dim + 4), dtype=np.uint8)
            out[:, :args.dim] = a.view(np.uint8)
            for i in range(a.shape[0]):
                vso.normalize(out[i], args.dim, vt)
            return out
        assert np.array_equal(rows[:len(head)], with_norm(head)), "device and host generators disagree"
        queries, km = with_norm(qraw), vso.COSINE
    else:
        rows = rows.view({"FLOAT32": np.float32, "INT8": np.int8, "BFLOAT16": np.uint16}[args.type_name])
        queries, km = qraw, (vso.L2 if args.metric_name == "L2" else vso.IP)
    nproc = os.cpu_count() or 1
    # one query per thread at a time, as the reference's own parallel mode does (bindings.cpp:250-283 knn_parallel): a batch of B
    # queries keeps at most B threads busy, whatever the core count
    threads = max(1, min(nproc, args.batch))
    # the index follows the host's CPUID like the reference's choosers do (bf16 IP: vdpbf16ps on avx512_bf16 hosts): same tier here
    from vectorsimilarity_amd import _capi
    tier_name = _capi.load().VecSimGpu_HostTier().decode()
    tier = {"AVX512": vso.TIER_AVX512, "SCALAR": vso.TIER_SCALAR, "AVX512_BF16": vso.TIER_AVX512_BF16}[tier_name]
    vso.flat_topk_batch_fast(vt, km, rows[:2000], queries[:1], args.topk, args.dim, 1, tier)

    def best_of(nq, th, reps):
        best, res = None, None
        for _ in range(reps):
            t0 = time.perf_counter()
            res = vso.flat_topk_batch_fast(vt, km, rows, queries[:nq], args.topk, args.dim, th, tier)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, res
    nq1 = max(1, min(args.batch, 4))
    t1, _ = best_of(nq1, 1, 5)
    tall, (labels, scores, fast) = best_of(args.batch, threads, 5)
    # checker: the GPU index over the same n rows must give the same labels, order and scores
    gl, gs = small.knn_query(qraw, args.topk)
    same = bool(np.array_equal(gl, labels.astype(np.int64)) and np.array_equal(gs, scores))
    return {"value": n * args.batch / tall, "unit": "distances/s", "cores": threads, "kind": "port",
            "single_thread": {"value": n * nq1 / t1, "unit": "distances/s", "cores": 1,
                              "note": "the reference scans single-threaded (brute_force.h:264-281)"},
            "cpu": cpu_model(), "nproc": nproc,
            "tier": tier_name,
            "sample": "first %d of the %d synthetic rows x %d queries (single thread: %d), top-%d, best of 5, %s kernel (tier %s); "
                      "GPU result on the same sample bit-identical: %s" % (
                          n, args.rows, args.batch, nq1, args.topk,
                          "AVX-512 intrinsics (oracle/vso_fast.c twin of the reference kernel)" if fast else "portable reference-order lanes",
                          tier_name, same)}


def run_c5(args):
    """config 5: HNSW fp32 L2 (graph built on the host cores, queries on the GPU: k_hnsw_search).  One step = one batch of
    `batch` queries answered end to end.  value = queries/s; roofline = the gathered row bytes (distance evaluations x
    storedDataSize) over the search kernel's HIP-event time against the HBM peak (the kernel is latency-, not stream-bound);
    recall@10 against the exact Flat answer computed on the GPU; cpu_baseline = the reference's search loop
    (hnsw.h:1983-2084, restated in oracle/vso_hnsw.c) on the SAME graph, one thread and all host cores."""
    from vectorsimilarity_amd import VecSim, synth
    n, dim, nq, k = args.rows, args.dim, args.batch, args.topk
    rows = np.empty((n, dim), dtype=np.float32)
    mix = synth.rows_f32(49, 0, 32, dim)

    def gen(seed, r0, cnt):
        part = synth.rows_f32(seed, r0, cnt, dim)
        if args.data == "lowrank":
            part = (synth.rows_f32(seed, r0, cnt, 32) @ mix + 0.05 * part).astype(np.float32)
        return part
    for r0 in range(0, n, 200_000):
        rows[r0:min(n, r0 + 200_000)] = gen(args.seed, r0, min(n, r0 + 200_000) - r0)
    nb_distinct = min(args.warmup + args.steps, 4)
    qsets = [gen(args.seed + 1 + b, 0, nq) for b in range(nb_distinct)]
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2, 16, 200, args.ef
    ix = VecSim.HNSWIndex(p)
    t0 = time.perf_counter()
    ix.add_vectors(rows, np.arange(n))
    build_s = time.perf_counter() - t0
    for w in range(max(1, args.warmup)):
        ix.knn_query(qsets[w % nb_distinct], k)
    # batches from `--readers` threads, batch s on thread s % readers (the reference's parallel readers, bindings.cpp:250-283: a second
    # reader searches through a graph view on its own stream, so its staging, download and reply construction overlap with the
    # other's search kernel), as the Flat configs do
    readers = max(1, min(args.readers, args.steps))
    if readers > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(readers)
        list(pool.map(lambda t: ix.knn_query(qsets[t % nb_distinct], k), range(readers)))   # (every lane has searched once)
    ix.reset_stats()
    t0 = time.perf_counter()
    if readers == 1:
        last = None
        for s in range(args.steps):
            last = ix.knn_query(qsets[(args.warmup + s) % nb_distinct], k)
    else:
        res = list(pool.map(lambda t: [ix.knn_query(qsets[(args.warmup + s) % nb_distinct], k) for s in range(t, args.steps, readers)][-1:],
                            range(readers)))
        last = res[(args.steps - 1) % readers][0]
    dt = time.perf_counter() - t0
    st = ix.stats()
    evals = st["scan_rows"]   # rows gathered = distance evaluations, summed over the lanes
    kms = st["scan_ms"] / max(1, st["scan_launches"])
    # recall@k against the exact answer (Flat index on the same GPU)
    bp = VecSim.BFParams()
    bp.type, bp.dim, bp.metric = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2
    bf = VecSim.BFIndex(bp)
    bf.add_vectors(rows, np.arange(n))
    qlast = qsets[(args.warmup + args.steps - 1) % nb_distinct]
    nx = min(nq, 1024)
    exact = np.concatenate([bf.knn_query(qlast[i:i + 64], k)[0] for i in range(0, nx, 64)])[:nx]
    recall = sum(len(set(last[0][i]) & set(exact[i])) for i in range(nx)) / (nx * k)
    del bf
    gathered = evals / args.steps * dim * 4
    out = {
        "metric": "QPS, HNSW fp32 L2 top-%d, N=%d d=%d M=16 efC=200 efR=%d, batch-%d" % (k, n, dim, args.ef, nq),
        "value": nq * args.steps / dt, "unit": "queries/s", "qps": nq * args.steps / dt,
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "c5: hnsw_f32_l2_top%d" % k, "rows_per_gpu": n, "dim": dim, "batch": nq, "k": k, "M": 16,
                   "efConstruction": 200, "efRuntime": args.ef, "reader_threads": readers,
                   "rows_kind": "i.i.d. U[-1,1) (BASELINE's generator)" if args.data == "uniform" else
                                "32 latent factors mixed into %d dims + 5%% noise (embedding-like)" % dim,
                   "note": "BASELINE quotes N = 10 M; the default run builds %d rows (host build %.0f s)" % (n, build_s)},
        "recall_at_%d" % k: recall, "recall_queries": nx,
        "dist_evals_per_query": evals / (args.steps * nq), "host_build_s": build_s,
        "roofline": {"bound": "gather", "achieved": gathered / (kms * 1e-3) / 1e9 if kms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (gathered / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kms > 0 else 0.0, "traffic": None,
                     "kernel": st["scan_kernel"], "avg_kernel_ms": kms, "launches": int(st["scan_launches"]),
                     "algorithmic_bytes_per_launch": gathered,
                     "note": "random 3 KiB row gathers: latency-bound, the HBM peak is the ceiling of the data path, not a target"},
    }
    if not args.no_cpu_baseline:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import vso
        vso.build()
        g = ix.graph()
        nproc = os.cpu_count() or 1
        nq1, nqa = 64, min(nq, max(256, 8 * nproc))

        def one(j):
            return vso.hnsw_search(vso.F32, vso.L2, rows, g, qlast[j], k, args.ef, dim)
        same = True
        best1 = None
        for _ in range(3):
            t0 = time.perf_counter()
            res = [one(j) for j in range(nq1)]
            d1 = time.perf_counter() - t0
            best1 = d1 if best1 is None else min(best1, d1)
        for j in range(nq1):   # the checker: same graph, same loop -> same reply
            el, es, _ = res[j]
            same = same and bool(np.array_equal(last[0][j][:len(el)], el.astype(np.int64)) and np.array_equal(last[1][j][:len(es)], es))
        with ThreadPoolExecutor(nproc) as pool:
            besta = None
            for _ in range(3):
                t0 = time.perf_counter()
                list(pool.map(one, range(nqa)))
                da = time.perf_counter() - t0
                besta = da if besta is None else min(besta, da)
        out["cpu_baseline"] = {"value": nqa / besta, "unit": "queries/s", "cores": nproc, "kind": "port",
                               "single_thread": {"value": nq1 / best1, "unit": "queries/s", "cores": 1},
                               "cpu": cpu_model(), "nproc": nproc,
                               "sample": "the reference's search loop (oracle/vso_hnsw.c) on the same graph: %d queries on one thread, %d on "
                                         "%d threads (one query per thread at a time), best of 3; GPU replies on the first %d queries "
                                         "identical: %s" % (nq1, nqa, nproc, nq1, same)}
    out["sorted"] = bool(np.all(np.diff(np.where(last[0] >= 0, last[1], np.inf), axis=1) >= 0))
    print(json.dumps(out))


def main():
    args = parse()
    if args.config == "c5":
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("config 5 (HNSW) is single-GPU: replicas only, graph traversal is global (DESIGN.md 7)")
        return run_c5(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_gpu:   # test aid: every rank on GPU 0 (a one-GPU box exercising the N > 1 code path; RCCL refuses duplicate devices,
        local_rank = 0  # so the exchange then runs over the agreed gloo fallback below)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ["VECSIM_GPU_DEVICE"] = str(local_rank)
    # the product's libraries (and with them the ROCm runtime they were built against) come first
    from vectorsimilarity_amd import VecSim, synth
    from vectorsimilarity_amd.sharded import ShardedFlatIndex
    dist = None
    # launched through torch.distributed.run (RANK set): always build the communicator, so a 1-rank run exercises
    # the same RCCL exchange as the 2/4/8-rank runs
    distributed = world > 1 or "RANK" in os.environ
    if distributed:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    p = VecSim.BFParams()
    p.type, p.dim, p.metric = getattr(VecSim, "VecSimType_" + args.type_name), args.dim, getattr(VecSim, "VecSimMetric_" + args.metric_name)
    # rows this rank holds.  weak: --rows each.  strong: the job's --rows dealt evenly (the first rows % N ranks hold one more).
    my_rows = args.rows if args.scaling == "weak" else args.rows // world + (1 if rank < args.rows % world else 0)
    total_rows = args.rows * world if args.scaling == "weak" else args.rows
    if distributed:
        # RCCL exchange.  Should the communicator fail to come up on any rank (no librccl, no peer access), every rank falls back to
        # the same records over torch.distributed (gloo) -- slower, still exact -- and the line says so; never a silent mix.
        transport, why = "rccl", None
        try:
            ix = ShardedFlatIndex(p, rank=rank, world=world, dist=dist, device=local_rank)
        except RuntimeError as e:
            ix, why = None, str(e)
        flags = [None] * world
        dist.all_gather_object(flags, why)
        if any(f is not None for f in flags):
            transport = "gloo (RCCL communicator failed: %s)" % next(f for f in flags if f is not None)
            ix = ShardedFlatIndex(p, rank=rank, world=world, dist=dist, device=local_rank, transport="dist")
        args.exchange_transport = transport
        ix.add_synthetic_local(my_rows, args.seed)     # shard r holds its vectors of seed + 1000 r
        local = ix.local
    else:
        ix = local = VecSim.BFIndex(p)
        ix.add_synthetic(args.rows, args.seed)
    local.set_option("mfma", args.mfma)
    for o in args.opt:
        name, val = o.split("=", 1)
        local.set_option(name, int(val))

    n_batches = args.warmup + args.steps
    gen = getattr(synth, args.gen)
    nb_distinct = min(n_batches, 8)                       # host-side query sets, cycled
    qsets = [gen(args.seed + 1 + b, 0, args.batch, args.dim) for b in range(nb_distinct)]

    from vectorsimilarity_amd import _capi

    def sync():
        # the timed region is bracketed by a barrier and a device-wide synchronisation on both sides (the query calls are
        # synchronous, so the device is idle here anyway; hipDeviceSynchronize through the product's library covers every stream
        # of this process on its GPU, torch's included, without initialising a second HIP context through torch.cuda)
        if dist is not None:
            dist.barrier()
        _capi.load().VecSimGpu_DeviceSynchronize()
        if dist is not None:
            dist.barrier()

    # The sharded exchange is a collective, so every rank must pair the same batches: batch b carries sequence number b and is
    # answered by reader thread b % readers on every rank (the library issues the exchanges in sequence order whatever the
    # threads' relative speed), so one batch's exchange + merge runs under the next batch's scan, as on a single GPU.
    readers = max(1, min(args.readers, args.steps))
    seq = [0]   # next sequence number of this rank's sharded index (same on every rank by construction)

    def answer(b, qs):
        return ix.knn_query(qs, args.topk, seq=b) if distributed else ix.knn_query(qs, args.topk)

    def run_batches(first_seq, qsel, count):
        """`count` batches, batch j (sequence number first_seq + j) on thread j % readers; returns the last batch's reply"""
        if readers == 1:
            out = None
            for j in range(count):
                out = answer(first_seq + j, qsets[qsel(j) % nb_distinct])
            return out
        res = list(pool.map(lambda t: [answer(first_seq + j, qsets[qsel(j) % nb_distinct]) for j in range(t, count, readers)][-1:],
                            range(readers)))
        return res[(count - 1) % readers][0]
    pool = None
    if readers > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(readers)
    nwarm = max(args.warmup, 2 * readers if readers > 1 else 0)   # (the reader lanes' scratch is sized on their first batches)
    run_batches(seq[0], lambda j: j, nwarm)
    seq[0] += nwarm
    local.reset_stats()
    if distributed:
        ix.reset_stats()
    sync()
    t0 = time.perf_counter()
    last = run_batches(seq[0], lambda j: args.warmup + j, args.steps)
    seq[0] += args.steps
    sync()
    dt = time.perf_counter() - t0
    if pool is not None:
        pool.shutdown()
    st = local.stats()
    per_rank = None
    my_dt = dt
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # where a rank's time went, per batch: scan kernel (HIP events), shard scan call (wall), waiting for the exchange turn,
        # exchange, merge + replies -- gathered so that a scaling line can show where non-linearity comes from
        sst = ix.stats()
        nb = max(1, sst["batches"])
        mine = torch.tensor([st["scan_ms"] / max(1, st["scan_launches"]), sst["scan_ms"] / nb, sst["turn_wait_ms"] / nb,
                             sst["exchange_ms"] / nb, sst["merge_ms"] / nb, sst["exchange_bytes"] / nb, float(ix._lib.VecSimGpu_ShardedWorld(ix._h)),
                             my_dt / args.steps * 1e3, float(my_rows)],
                            dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        # fixed_ms_per_batch: what a batch costs this rank beyond its scan kernel (probe, threshold, re-rank, select, exchange,
        # merge, launch gaps) -- the part that does NOT shrink with rows / N under strong scaling
        per_rank = [{"rank": r, "rows": int(t[8]), "ms_per_batch": float(t[7]), "scan_kernel_ms": float(t[0]),
                     "fixed_ms_per_batch": float(t[7] - t[0]), "shard_scan_call_ms": float(t[1]), "turn_wait_ms": float(t[2]),
                     "exchange_ms": float(t[3]), "merge_ms": float(t[4]), "exchange_bytes": float(t[5]),
                     "rccl_world": int(t[6])} for r, t in enumerate(allr)]
        assert args.exchange_transport != "rccl" or all(p["rccl_world"] == world for p in per_rank), per_rank   # every communicator spans all N ranks

    if rank == 0:
        dists = total_rows * args.batch * args.steps
        launches = max(1, st["scan_launches"])
        avg_ms = st["scan_ms"] / launches
        bytes_per_launch = st["scan_bytes"] / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_source = None, None
        try:  # HBM bytes per launch from the PMC counters (separate rocprofv3 passes; committed summary, not this run)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f).get(st["scan_kernel"])
            if t and t["workload"] == {"rows": args.rows, "dim": args.dim, "batch": args.batch}:
                traffic = t["bytes_per_launch"]
                traffic_source = "profiles/pmc_traffic.json (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, separate run; not measured by this run)"
        except (OSError, ValueError, KeyError):
            pass
        rows_tag = "%dM" % (args.rows // 1_000_000) if args.rows % 1_000_000 == 0 else str(args.rows)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": st["scan_kernel"], "avg_kernel_ms": avg_ms, "launches": int(st["scan_launches"]),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                # probe + threshold kernels by their own pair of events: off by default (an event record costs the stream 3-5 us,
                # profiles/r04_event_cost.txt; the scan kernel's pair stays, it is what this object is computed from): --opt events=3
                "other_kernels_ms_per_step": (st["other_ms"] / max(1, args.steps)) if st["other_ms"] > 0 else None}
        if args.dtype == "i8" and avg_ms > 0:
            rows_per_launch = st["scan_rows"] / launches
            tops = 2.0 * rows_per_launch * args.dim * args.batch / (avg_ms * 1e-3) / 1e12
            roof["mfma"] = {"achieved": tops, "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s", "frac": tops / I8_MFMA_PEAK_TOPS,
                            "note": "balanced config: 256 queries x 1 KiB rows need 4.1 POP/s to stream at 8 TB/s"}
        out = {
            "metric": "distances/sec, Flat %s %s top-%d, N=%s d=%d, batch-%d" % (
                args.dtype if args.dtype != "f32" else "fp32", args.metric_name, args.topk, rows_tag, args.dim, args.batch),
            "value": dists / dt,
            "unit": "distances/s",
            "qps": args.batch * args.steps / dt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s: flat_%s_%s_top%d" % (args.config, args.dtype, args.metric_name.lower(), args.topk),
                       "rows_per_gpu": my_rows, "rows_total": total_rows, "dim": args.dim, "batch": args.batch, "k": args.topk,
                       "sharding": "rows x %d" % world if world > 1 else "single GPU",
                       "reader_threads": readers,
                       "exchange": (("rccl ncclAllGather" if args.exchange_transport == "rccl" else args.exchange_transport) +
                                    " of per-shard candidate records over %d rank(s), sequence-ordered, + exact host "
                                    "merge (C++ host library)" % world) if distributed else "none (plain VecSimIndex_TopKQueryBatch)"},
            "roofline": roof,
            # step time beyond the scan kernel (probe, threshold, re-rank, select, copies, host replay not hidden, launch gaps)
            "fixed_ms_per_batch": dt / args.steps * 1e3 - avg_ms,
            "per_rank_ms_per_batch": per_rank,
            "candidates_per_query": st["candidates"] / max(1, args.steps * args.batch),
            "fallbacks": int(st["fallbacks"]),
        }
        if world == 1:
            # A PREDICTION, not a measurement (no run on more than one GPU exists): strong scaling of this workload from the
            # pieces measured above -- the scan kernel shrinks with rows / G, the fixed part does not, and the exchange is priced
            # at what a 1-rank RCCL communicator cost on this box (README).  efficiency = t_1 / (G * t_G).
            fixed = dt / args.steps * 1e3 - avg_ms
            exch = 0.10
            out["predicted_strong_scaling"] = {
                "note": "prediction from 1-GPU pieces: t_G = scan_kernel / G + fixed + exchange (%.2f ms assumed); unmeasured" % exch,
                "t_ms": {str(g): avg_ms / g + fixed + (exch if g > 1 else 0.0) for g in (1, 2, 4, 8)},
                "efficiency": {str(g): (avg_ms + fixed) / (g * (avg_ms / g + fixed + exch)) for g in (2, 4, 8)}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, VecSim, synth)
        # size-independent property at full size: replies are ascending in score
        labels, scores = last
        out["sorted"] = bool(np.all(np.diff(scores, axis=1) >= 0) and np.all(labels >= 0))
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
