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
per GPU).  Every rank holds its shard, scans it for the same query batch, and the
per-shard candidate records are exchanged by ONE ncclAllGather per batch -- RCCL over xGMI, issued by the C++ host
library (csrc/vsgpu_comm.hip behind VecSimGpu_Sharded*, include/VecSim/vec_sim_gpu.h) -- and merged into the exact
single-index reply on every rank.  torch.distributed is the control plane only (gloo: rank 0's RCCL id, barriers, the
max-over-ranks of the wall time); it never touches the GPU, so the timed region is bracketed by barrier + the library's
own device drain (every C-API call returns with its HIP streams synchronised).  Rank 0 prints ONE JSON line.
Scaling: N > 1 runs TWO phases by default (--scaling both) -- strong (the job holds `rows` vectors, rows / N per rank: what
BASELINE's metric "N = 10 M at 1 / 2 / 4 / 8 GPUs" names) and weak (`rows` per rank: config 4's own definition) --; the line's
value / ms_per_step / scaling are the config's own kind (c2: strong, c4: weak), the other phase is reported under "also".
On one GPU (c2) the line also carries the MEASURED steps of rows / 2, / 4, / 8 shards: the pieces of the strong-scaling curve.

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
    ap.add_argument("--data", default=None, choices=["uniform", "lowrank", "clustered"],
                    help="row distribution.  uniform: BASELINE's i.i.d. U[-1,1) rows (the default of the Flat configs; intrinsic dimension "
                         "= d: no graph index finds neighbours there); lowrank: 32 latent factors mixed into d dims + 5%% noise "
                         "(embedding-like; the default of c5); clustered (c2 / c4): uniform rows, 1%% of them within 1e-3 of one of 64 centres, "
                         "half of the queries next to a centre -- what the filter's candidate lists look like on non-uniform data "
                         "(profiles/r06_selectivity.json).  Non-uniform Flat runs generate on the host (numpy, seeded per 200 K-row chunk) "
                         "and carry no cpu_baseline")
    ap.add_argument("--ef", type=int, default=128, help="c5: efRuntime")
    ap.add_argument("--scaling", default="both", choices=["both", "weak", "strong"],
                    help="N > 1.  strong: the job holds --rows vectors, rank r holds its share of them (BASELINE's headline reads N = 10 M "
                         "at 1 / 2 / 4 / 8 GPUs).  weak: every rank holds --rows vectors, the job holds N x rows (config 4 is defined that "
                         "way: 100 M rows over 8 GPUs).  both (default): two phases in one run -- the config's own kind first (c2: strong, "
                         "c4: weak: the line's value / scaling), the other kind under \"also\"")
    ap.add_argument("--no-shard-curve", dest="shard_curve", action="store_false",
                    help="one GPU, c2: skip measuring the rows / G shards (G = 2, 4, 8) the strong-scaling estimate is built from")
    ap.add_argument("--no-full-parity", dest="full_parity", action="store_false",
                    help="one GPU: skip comparing the timed table's last reply with the oracle's scan of all its rows (streams the "
                         "table back to the host once: ~10 s at 30 GB)")
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
    if a.data is None:
        a.data = "lowrank" if a.config == "c5" else "uniform"
    if a.config != "c5" and a.data != "uniform":
        if a.config not in ("c2", "c4") or a.gpus != 1:
            ap.error("--data lowrank / clustered: c2, c4 (one GPU) and c5 only")
        a.no_cpu_baseline = True
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
    # checker independence (round-4 advisor finding): the CPU leg reads its rows back from the GPU index it is then compared with, so
    # the rows themselves are checked against the HOST generator -- the first 64 and 31 runs of 8 spread over the whole sample
    spots = [(0, min(n, 64))] + [(r0, min(8, n - r0)) for r0 in sorted({(i * (n // 31) + 7 * i) % max(1, n - 8) for i in range(1, 32)}) if n > 72]
    eb = args.dim * {"FLOAT32": 4, "INT8": 1, "BFLOAT16": 2}[args.type_name]
    hosts = [(r0, gen(args.seed, r0, cnt, args.dim)) for r0, cnt in spots]
    for r0, h in hosts:
        hb = h.view(np.uint8).reshape(len(h), -1)
        assert np.array_equal(rows[r0:r0 + len(h), :eb], hb[:, :eb]), "device and host generators disagree at row %d" % r0
    head = hosts[0][1]
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
    tier = {"AVX512": vso.TIER_AVX512, "SCALAR": vso.TIER_SCALAR, "AVX512_BF16": vso.TIER_AVX512_BF16, "AVX512_FP16": vso.TIER_AVX512_FP16}[tier_name]
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
    row_bytes = rows.shape[1] * rows.dtype.itemsize
    return {"value": n * args.batch / tall, "unit": "distances/s", "cores": threads, "kind": "port",
            # every thread scans the whole sample for its query: the all-cores figure is bound by the host's DRAM (the sample does not
            # fit the caches), so it moves with the sample size and the box's memory, not with the kernel -- read it next to its GB/s
            "effective_GBps": n * row_bytes * args.batch / tall / 1e9,
            "single_thread": {"value": n * nq1 / t1, "unit": "distances/s", "cores": 1, "effective_GBps": n * row_bytes * nq1 / t1 / 1e9,
                              "note": "the reference scans single-threaded (brute_force.h:264-281)"},
            "cpu": cpu_model(), "nproc": nproc,
            "tier": tier_name,
            "sample": "first %d of the %d synthetic rows x %d queries (single thread: %d), top-%d, best of 5, %s kernel (tier %s); "
                      "GPU result on the same sample bit-identical: %s" % (
                          n, args.rows, args.batch, nq1, args.topk,
                          "AVX-512 intrinsics (oracle/vso_fast.c twin of the reference kernel)" if fast else "portable reference-order lanes",
                          tier_name, same)}


def full_table_parity(args, ix, n, queries, reply, which):
    """checker leg (the other place bench.py touches oracle/): the TIMED index's own reply to its last timed batch, compared for the
    queries `which` with the oracle's reference-order scan of the whole table -- the stored rows streamed back in 1 GiB pieces
    and scored on the host cores (oracle.vso.StreamTopK = brute_force.h:242-291 over a running candidate set).  Labels, order
    and scores must be bit-equal.  Returns the verdict and what it cost."""
    from oracle import vso
    from vectorsimilarity_amd import _capi
    vso.build()
    t0 = time.perf_counter()
    vt = getattr(vso, {"FLOAT32": "F32", "INT8": "I8", "BFLOAT16": "BF16"}[args.type_name])
    tier_name = _capi.load().VecSimGpu_HostTier().decode()
    tier = {"AVX512": vso.TIER_AVX512, "SCALAR": vso.TIER_SCALAR, "AVX512_BF16": vso.TIER_AVX512_BF16, "AVX512_FP16": vso.TIER_AVX512_FP16}[tier_name]
    qsel = np.ascontiguousarray(queries[which])
    if args.metric_name == "Cosine":                         # int8 Cosine: query blob = elements + float norm
        qb = np.zeros((len(which), args.dim + 4), dtype=np.uint8)
        qb[:, :args.dim] = qsel.view(np.uint8)
        for i in range(len(which)):
            vso.normalize(qb[i], args.dim, vt)
        qsel, km = qb, vso.COSINE
    else:
        km = vso.L2 if args.metric_name == "L2" else vso.IP
    threads = min(64, os.cpu_count() or 1)
    st = vso.StreamTopK(vt, km, qsel, args.topk, args.dim, threads=threads, tier=tier)
    rb = ix.stored_rows(0, 1).shape[1]
    per = max(1, (1 << 30) // rb)
    buf = np.empty(per * rb, dtype=np.uint8)
    for r0 in range(0, n, per):
        st.feed(ix.stored_rows(r0, min(per, n - r0), out=buf), r0)
    el, es = st.result()
    labels, scores = reply
    same = all(bool(np.array_equal(labels[qi], el[j]) and np.array_equal(scores[qi], es[j])) for j, qi in enumerate(which))
    return {"same": bool(same and st.rows_seen == n), "rows": int(st.rows_seen), "queries": [int(x) for x in which], "host_threads": threads,
            "tier": tier_name, "seconds": time.perf_counter() - t0,
            "how": "the timed index's reply to its last timed batch vs the oracle's reference-order scan of ALL its stored rows "
                   "(streamed back in 1 GiB pieces), labels + order + scores bit for bit"}


# ---- non-uniform rows for the Flat configs (--data lowrank | clustered): host-generated, seeded per chunk
def _nu_shared(args):
    rng = np.random.default_rng([args.seed, 7])
    return {"mix": rng.standard_normal((32, args.dim), dtype=np.float32) / np.float32(np.sqrt(32.0)),
            "centres": rng.random((64, args.dim), dtype=np.float32) * 2 - 1}


def _nu_rows(args, shared, chunk_id, cnt, queries=False):
    """float32 rows of chunk `chunk_id` (queries: the batch's own stream)"""
    rng = np.random.default_rng([args.seed, 11 if queries else 13, chunk_id])
    u = rng.random((cnt, args.dim), dtype=np.float32) * 2 - 1
    if args.data == "lowrank":
        return (rng.standard_normal((cnt, 32), dtype=np.float32) @ shared["mix"] + np.float32(0.05) * u).astype(np.float32)
    # clustered: 1 % of the rows (half of the queries) within 1e-3 of one of 64 centres
    near = (np.arange(cnt) % 2 == 0) if queries else (rng.random(cnt) < 0.01)
    c = rng.integers(0, 64, int(near.sum()))
    u[near] = shared["centres"][c] + np.float32(1e-3) * u[near]
    return u


def _nu_encode(args, rows):
    if args.type_name == "BFLOAT16":
        u = np.ascontiguousarray(rows, dtype=np.float32).view(np.uint32)
        return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)
    return rows


def fill_nonuniform(args, ix, n):
    from concurrent.futures import ThreadPoolExecutor
    shared = _nu_shared(args)
    step = 200_000
    chunks = [(i, r0, min(step, n - r0)) for i, r0 in enumerate(range(0, n, step))]
    with ThreadPoolExecutor(8) as pool:     # generation on a few host threads, ingest in order
        for (i, r0, cnt), rows in zip(chunks, pool.map(lambda c: _nu_encode(args, _nu_rows(args, shared, c[0], c[2])), chunks)):
            ix.add_vectors(rows, np.arange(r0, r0 + cnt))
    return shared


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
        "config": {"workload": "c5: hnsw_f32_l2_top%d, N=%d of BASELINE's 10000000 rows%s" % (
                       k, n, "" if n == 10_000_000 else " (REDUCED: the graph is built on the host cores first, ~36 min at 10 M; --rows 10000000 runs it)"),
                   "rows_per_gpu": n, "rows_baseline": 10_000_000, "dim": dim, "batch": nq, "k": k, "M": 16,
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


def build_index(args, p, my_rows, rank, world, local_rank, dist, distributed, VecSim, ShardedFlatIndex):
    """this rank's index over `my_rows` device-generated rows: a plain Flat index, or (launched through torch.distributed.run) its
    shard of a sharded one with the RCCL exchange -- falling back, on ALL ranks alike, to the same records over gloo should any
    rank's communicator fail to come up; the line then says so"""
    if not distributed:
        ix = VecSim.BFIndex(p)
        if args.data == "uniform":
            ix.add_synthetic(my_rows, args.seed)
        else:
            args.nu_shared = fill_nonuniform(args, ix, my_rows)
        ix.set_option("mfma", args.mfma)
        for o in args.opt:
            name, val = o.split("=", 1)
            ix.set_option(name, int(val))
        return ix, ix, None
    # Which buffers the collective runs on (csrc/vsgpu_comm.hip): `mapped` -- the records' host blocks, mapped into the device, ARE the
    # all-gather's send / receive buffers: one kernel on the exchange stream -- or `staged` -- device buffers with a copy in front
    # and behind: the canonical RCCL shape and the library's default for more than one rank, +0.2 ms per batch on a busy GPU
    # (profiles/r05a_*).  No multi-GPU box was available to the builder, so the choice is made HERE, by test: every rank tries
    # `mapped`, all-gathers rank-stamped bytes of a record's size and checks every slice (VecSimGpu_ShardedExchangeSelfTest, 20 s
    # limit); unless every rank passes, all of them build a fresh communicator in `staged` form and test again; should that fail
    # too, the same records travel over torch.distributed (gloo) -- slower, still exact -- and the line says which one ran.
    def attempt(mode):
        if mode is not None:
            os.environ["VECSIM_GPU_EXCHANGE"] = mode
        os.environ.setdefault("VECSIM_GPU_EXCHANGE_TIMEOUT_MS", "20000")
        ix, why = None, None
        try:
            ix = ShardedFlatIndex(p, rank=rank, world=world, dist=dist, device=local_rank)
            if not ix.exchange_self_test(81920):
                why = "self-test of the %s exchange failed: %s" % (ix.exchange_mode(), ix._lib.VecSimGpu_LastError().decode())
        except RuntimeError as e:
            why = str(e)
        flags = [None] * world
        dist.all_gather_object(flags, why)
        bad = [f for f in flags if f is not None]
        return (ix, None) if not bad else (None, bad[0])
    forced = os.environ.get("VECSIM_GPU_EXCHANGE")
    transport, notes = None, []
    for mode in ([forced] if forced else ["mapped", "staged"]):
        ix, why = attempt(mode)
        if ix is not None:
            transport = "rccl"
            break
        notes.append("%s: %s" % (mode, why))
    if transport is None:
        transport = "gloo (RCCL exchange failed: %s)" % "; ".join(notes)
        ix = ShardedFlatIndex(p, rank=rank, world=world, dist=dist, device=local_rank, transport="dist")
    args.exchange_mode = ix.exchange_mode() + (" (after %s)" % "; ".join(notes) if notes and transport == "rccl" else "")
    ix.add_synthetic_local(my_rows, args.seed)     # shard r holds its vectors of seed + 1000 r
    local = ix.local
    local.set_option("mfma", args.mfma)
    for o in args.opt:
        name, val = o.split("=", 1)
        local.set_option(name, int(val))
    return ix, local, transport


def timed_phase(args, ix, local, transport, my_rows, steps, qsets, rank, world, dist, distributed, readers, nwarm):
    """nwarm untimed batches, then exactly `steps` timed ones bracketed by barrier + device synchronisation on both sides; the wall
    time is the MAX over ranks.  Returns what the JSON line is made of."""
    from vectorsimilarity_amd import _capi
    nb_distinct = len(qsets)

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
    def answer(b, qs):
        return ix.knn_query(qs, args.topk, seq=b) if distributed else ix.knn_query(qs, args.topk)

    pool = None
    if readers > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(readers)

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

    run_batches(0, lambda j: j, nwarm)
    local.reset_stats()
    if distributed:
        ix.reset_stats()
    sync()
    t0 = time.perf_counter()
    last = run_batches(nwarm, lambda j: args.warmup + j, steps)
    sync()
    dt = time.perf_counter() - t0
    if pool is not None:
        pool.shutdown()
    st = local.stats()
    per_rank, my_dt = None, dt
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
                             my_dt / steps * 1e3, float(my_rows), float(local._lib.VecSimGpu_IndexDevice(local._h))],
                            dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        # fixed_ms_per_batch: what a batch costs this rank beyond its scan kernel (probe, threshold, re-rank, select, exchange,
        # merge, launch gaps) -- the part that does NOT shrink with rows / N under strong scaling
        per_rank = [{"rank": r, "rows": int(t[8]), "ms_per_batch": float(t[7]), "scan_kernel_ms": float(t[0]),
                     "fixed_ms_per_batch": float(t[7] - t[0]), "shard_scan_call_ms": float(t[1]), "turn_wait_ms": float(t[2]),
                     "exchange_ms": float(t[3]), "merge_ms": float(t[4]), "exchange_bytes": float(t[5]),
                     "rccl_world": int(t[6]), "device": int(t[9])} for r, t in enumerate(allr)]
        assert transport != "rccl" or all(q["rccl_world"] == world for q in per_rank), per_rank   # every communicator spans all N ranks
    launches = max(1, st["scan_launches"])
    return {"device": int(local._lib.VecSimGpu_IndexDevice(local._h)), "dt": dt, "steps": steps, "st": st, "per_rank": per_rank, "last": last, "avg_kernel_ms": st["scan_ms"] / launches,
            "ms_per_step": dt / steps * 1e3, "fixed_ms_per_batch": dt / steps * 1e3 - st["scan_ms"] / launches,
            "candidates_per_query": st["candidates"] / max(1, steps * args.batch)}


# which kind of scaling a config's own definition names (BASELINE.json): the headline metric reads "N = 10 M at 1 / 2 / 4 / 8 GPUs"
# -- the job's rows are fixed, strong --; config 4 is "100 M rows over 8 GPUs", 12.5 M per GPU whatever the count -- weak
HEADLINE_SCALING = {"c1": "strong", "c2": "strong", "c3": "strong", "c4": "weak"}


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
        import torch  # noqa: F401
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        # self-diagnosis before anything collective touches a GPU: every rank's device must exist (a rank that fails later would
        # leave the others inside a communicator's rendezvous, which has no timeout)
        from vectorsimilarity_amd import _capi
        ndev = _capi.load().VecSimGpu_DeviceCount()
        flags = [None] * world
        dist.all_gather_object(flags, None if local_rank < ndev else "rank %d wants GPU %d, %d visible" % (rank, local_rank, ndev))
        bad = [f for f in flags if f is not None]
        if bad:
            if rank == 0:
                print("bench.py: " + "; ".join(bad) + " -- one process per GPU; on a one-GPU box pass --same-gpu to exercise the N > 1 path", file=sys.stderr)
            dist.destroy_process_group()
            raise SystemExit(2)

    p = VecSim.BFParams()
    p.type, p.dim, p.metric = getattr(VecSim, "VecSimType_" + args.type_name), args.dim, getattr(VecSim, "VecSimMetric_" + args.metric_name)
    # Phases.  One GPU: a single phase (strong and weak coincide).  N > 1 and --scaling both (default): the config's headline kind
    # first -- its figures are the line's `value` / `ms_per_step` / `scaling` --, then the other kind, reported under "also".
    # weak: every rank holds --rows.  strong: the job's --rows dealt evenly (the first rows % N ranks hold one more).
    head = HEADLINE_SCALING[args.config] if args.scaling == "both" else args.scaling
    kinds = [head] if (world == 1 or args.scaling != "both") else [head, "weak" if head == "strong" else "strong"]
    gen = getattr(synth, args.gen)
    nb_distinct = min(args.warmup + args.steps, 8)                       # host-side query sets, cycled
    if args.data == "uniform":
        qsets = [gen(args.seed + 1 + b, 0, args.batch, args.dim) for b in range(nb_distinct)]
    else:
        qsets = [_nu_encode(args, _nu_rows(args, _nu_shared(args), b, args.batch, queries=True)) for b in range(nb_distinct)]
    readers = max(1, min(args.readers, args.steps))
    # (the reader lanes' scratch is sized on their first batches; the low-precision filters' first launches run 15-30 % slow --
    # profiles/r06_warmup_cliff.txt: the shader clock ramps over the first ~25 ms of load after any idle half second -- so their steady state needs ten warm-up batches)
    nwarm = max(args.warmup, 2 * readers if readers > 1 else 0, 10 if args.dtype in ("i8", "bf16") else 0)
    phases = []
    for kind in kinds:
        my_rows = args.rows if kind == "weak" else args.rows // world + (1 if rank < args.rows % world else 0)
        total_rows = args.rows * world if kind == "weak" else args.rows
        ix, local, transport = build_index(args, p, my_rows, rank, world, local_rank, dist, distributed, VecSim, ShardedFlatIndex)
        ph = timed_phase(args, ix, local, transport, my_rows, args.steps, qsets, rank, world, dist, distributed, readers, nwarm)
        ph.update(kind=kind, my_rows=my_rows, total_rows=total_rows, transport=transport)
        if world == 1 and not distributed and args.full_parity and not phases:
            # the table that was just timed, against the oracle (first, middle and last query of the last timed batch)
            ph["full_table_parity"] = full_table_parity(args, local, my_rows, qsets[(args.warmup + args.steps - 1) % nb_distinct], ph["last"],
                                                        sorted(set(range(0, args.batch, max(1, args.batch // 8))) | {args.batch - 1}))
        phases.append(ph)
        del ix, local                          # (the next phase's rows need the room)
    ph = phases[0]
    st, dt, my_rows, total_rows = ph["st"], ph["dt"], ph["my_rows"], ph["total_rows"]
    args.exchange_transport = ph["transport"]

    # One GPU, config 2: the strong-scaling curve's per-shard pieces, MEASURED here -- the step of a shard of rows / G vectors on
    # this GPU through the same path (same readers, same exchange when launched through torch.distributed.run) for G = 2, 4, 8.
    # What a G-GPU job adds to that is the exchange across G ranks, the one term this box cannot measure.
    shard_curve = None
    if world == 1 and args.shard_curve and args.config in ("c2",):
        shard_curve = {}
        for g in (2, 4, 8):
            rows_g = args.rows // g
            ixg, localg, trg = build_index(args, p, rows_g, rank, world, local_rank, dist, distributed, VecSim, ShardedFlatIndex)
            pg = timed_phase(args, ixg, localg, trg, rows_g, args.steps, qsets, rank, world, dist, distributed, readers, nwarm)
            shard_curve[str(g)] = {"rows": rows_g, "ms_per_step": pg["ms_per_step"], "scan_kernel_ms": pg["avg_kernel_ms"],
                                   "fixed_ms_per_batch": pg["fixed_ms_per_batch"], "candidates_per_query": pg["candidates_per_query"],
                                   "exchange_ms_one_rank": (pg["per_rank"][0]["exchange_ms"] if pg["per_rank"] else None),
                                   "turn_wait_ms": (pg["per_rank"][0]["turn_wait_ms"] if pg["per_rank"] else None),
                                   "merge_ms": (pg["per_rank"][0]["merge_ms"] if pg["per_rank"] else None)}
            del ixg, localg

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
            if t and t["workload"] == {"rows": my_rows, "dim": args.dim, "batch": args.batch}:
                traffic = t["bytes_per_launch"]
                traffic_source = "profiles/pmc_traffic.json (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, separate run; not measured by this run)"
        except (OSError, ValueError, KeyError):
            pass
        rows_tag = "%dM" % (total_rows // 1_000_000) if total_rows % 1_000_000 == 0 else str(total_rows)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": st["scan_kernel"], "avg_kernel_ms": avg_ms, "launches": int(st["scan_launches"]),
                "warmup_batches": nwarm,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                # probe + threshold kernels by their own pair of events: off by default (an event record costs the stream 3-5 us,
                # profiles/r06_event_cost.txt; the scan kernel's pair stays, it is what this object is computed from): --opt events=3
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
            "higher_is_better": True, "scaling": ph["kind"], "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s: flat_%s_%s_top%d" % (args.config, args.dtype, args.metric_name.lower(), args.topk),
                       "rows_per_gpu": my_rows, "rows_total": total_rows, "dim": args.dim, "batch": args.batch, "k": args.topk,
                       "sharding": "rows x %d" % world if world > 1 else "single GPU",
                       "reader_threads": readers,
                       "exchange_buffers": getattr(args, "exchange_mode", None),
                       # self-diagnosis of a multi-GPU run: the size every rank's RCCL communicator reports, and which device each rank drove
                       "rccl_world": ([q["rccl_world"] for q in ph["per_rank"]] if ph["per_rank"] else None),
                       "rank_devices": ([q["device"] for q in ph["per_rank"]] if ph["per_rank"] else [ph["device"]]),
                       "exchange_transport": args.exchange_transport,
                       "exchange": (("rccl ncclAllGather" if args.exchange_transport == "rccl" else args.exchange_transport) +
                                    " of per-shard candidate records over %d rank(s), sequence-ordered, + exact host "
                                    "merge (C++ host library)" % world) if distributed else "none (plain VecSimIndex_TopKQueryBatch)"},
            "roofline": roof,
            # step time beyond the scan kernel (probe, threshold, re-rank, select, copies, host replay not hidden, launch gaps)
            "fixed_ms_per_batch": dt / args.steps * 1e3 - avg_ms,
            "per_rank_ms_per_batch": ph["per_rank"],
            "candidates_per_query": st["candidates"] / max(1, args.steps * args.batch),
            "fallbacks": int(st["fallbacks"]),
            "retries": int(st["retries"]),     # queries that took a second filter pass (candidate list overflow: near-duplicate clusters)
            "rows_kind": {"uniform": "i.i.d. U[-1,1) (BASELINE's generator, device-generated)",
                          "lowrank": "32 latent factors mixed into %d dims + 5%% noise (host-generated, numpy PCG64 seeded per chunk)" % args.dim,
                          "clustered": "U[-1,1) rows, 1%% within 1e-3 of one of 64 centres; every second query next to a centre"}[args.data],
        }
        if len(phases) > 1:
            o = phases[1]
            ol = max(1, o["st"]["scan_launches"])
            out["also"] = {"scaling": o["kind"], "value": o["total_rows"] * args.batch * args.steps / o["dt"], "unit": "distances/s",
                           "qps": args.batch * args.steps / o["dt"], "ms_per_step": o["ms_per_step"], "rows_per_gpu": o["my_rows"],
                           "rows_total": o["total_rows"], "scan_kernel_ms": o["avg_kernel_ms"],
                           "hbm_frac": (o["st"]["scan_bytes"] / ol / (o["avg_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if o["avg_kernel_ms"] > 0 else 0.0,
                           "fixed_ms_per_batch": o["fixed_ms_per_batch"], "per_rank_ms_per_batch": o["per_rank"],
                           "note": "same run, second phase: the other kind of scaling (the line's value / scaling are the config's own "
                                   "definition: %s)" % ph["kind"]}
        if shard_curve is not None:
            # t_G = the MEASURED step of a rows / G shard on this GPU + the exchange across G ranks (ASSUMED: not measurable on one
            # GPU; 0.10 ms, not hidden -- conservative: with two batches in flight batch i's exchange runs under batch i+1's scan).
            # efficiency = t_1 / (G * t_G).
            exch = 0.10
            t1 = dt / args.steps * 1e3
            out["strong_scaling_from_measured_shards"] = {
                "note": "per-shard steps measured on ONE GPU (rows / G vectors, same path); only the cross-GPU exchange term (%.2f ms, "
                        "not hidden) is assumed; no run on more than one GPU" % exch,
                "shards": shard_curve, "exchange_ms_assumed": exch,
                "t_ms": dict([("1", t1)] + [(g, v["ms_per_step"] + exch) for g, v in shard_curve.items()]),
                "efficiency": {g: t1 / (int(g) * (v["ms_per_step"] + exch)) for g, v in shard_curve.items()}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, VecSim, synth)
        # size-independent property at full size: replies are ascending in score
        labels, scores = ph["last"]
        out["sorted"] = bool(np.all(np.diff(scores, axis=1) >= 0) and np.all(labels >= 0))
        if "full_table_parity" in ph:
            out["full_table_parity"] = ph["full_table_parity"]["same"]
            out["full_table_parity_detail"] = ph["full_table_parity"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
