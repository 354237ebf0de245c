#!/usr/bin/env python3
"""BASELINE.json config 5 shape, scaled to what a single-threaded host build can ingest in minutes:
HNSW fp32 L2, d=768, M=16, efC=200, efR=128, top-10.  Reports QPS, recall@10 vs the exact Flat answer
(computed on the GPU), distance evaluations per query and the achieved random-gather bandwidth.
    python tools/bench_hnsw.py [--rows 200000] [--queries 4096]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=200_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--queries", type=int, default=4096)
ap.add_argument("--M", type=int, default=16)
ap.add_argument("--efc", type=int, default=200)
ap.add_argument("--ef", type=int, default=128)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--data", default="uniform", choices=["uniform", "lowrank"],
                help="uniform: BASELINE's U[-1,1) i.i.d. rows (intrinsic dimension = d, hard for any graph index); "
                     "lowrank: 32 latent factors mixed into d dims + 5%% noise (embedding-like)")
ap.add_argument("--slots", default="", help="comma list of hnsw_slots values to time (waves per CU)")
ap.add_argument("--efs", default="", help="comma list of further efRuntime values: recall and QPS at each (same graph)")
ap.add_argument("--threads", type=int, default=0, help="host linking threads (VECSIM_HNSW_BUILD_THREADS); 1 = the sequential insert path")
ap.add_argument("--chunk", type=int, default=0, help="add the rows in chunks of this many (progress lines); 0 = one bulk call")
ap.add_argument("--readers", type=int, default=0, help="also time this many reader threads, each submitting batches of --queries (reader lanes)")
ap.add_argument("--exact-queries", type=int, default=0, help="recall on the first this-many queries only (0 = all)")
ap.add_argument("--iterator", type=int, default=0, help="also time the batch iterator's graph walk on this many queries: 10 batches of --k results each, "
                                                         "against the oracle's twin of the walk on the host (one thread)")
a = ap.parse_args()
if a.threads:
    os.environ["VECSIM_HNSW_BUILD_THREADS"] = str(a.threads)

# rows are generated in slices so the generator's temporaries stay small at millions of rows
rows = np.empty((a.rows, a.dim), dtype=np.float32)
q = synth.rows_f32(48, 0, a.queries, a.dim)
mix = synth.rows_f32(49, 0, 32, a.dim)
for r0 in range(0, a.rows, 200_000):
    r1 = min(a.rows, r0 + 200_000)
    part = synth.rows_f32(47, r0, r1 - r0, a.dim)
    if a.data == "lowrank":
        part = (synth.rows_f32(47, r0, r1 - r0, 32) @ mix + 0.05 * part).astype(np.float32)
    rows[r0:r1] = part
if a.data == "lowrank":
    q = (synth.rows_f32(48, 0, a.queries, 32) @ mix + 0.05 * q).astype(np.float32)
p = VecSim.HNSWParams()
p.type, p.dim, p.metric, p.M, p.efConstruction, p.efRuntime = VecSim.VecSimType_FLOAT32, a.dim, VecSim.VecSimMetric_L2, a.M, a.efc, a.ef
ix = VecSim.HNSWIndex(p)
t0 = time.perf_counter()
if a.chunk:
    for r0 in range(0, a.rows, a.chunk):
        r1 = min(a.rows, r0 + a.chunk)
        ix.add_vectors(rows[r0:r1], np.arange(r0, r1))
        print("built %d rows in %.0f s" % (r1, time.perf_counter() - t0), flush=True)
else:
    ix.add_vectors(rows, np.arange(a.rows))
build_s = time.perf_counter() - t0
ix.knn_query(q[:64], a.k)          # uploads rows + graph
ix.reset_stats()
best = None
for _ in range(3):
    t0 = time.perf_counter()
    labels, dists = ix.knn_query(q, a.k)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
for sl in [int(x) for x in a.slots.split(",") if x]:
    ix.set_option("hnsw_slots", sl)
    ix.knn_query(q, a.k)
    tb = None
    for _ in range(3):
        t0 = time.perf_counter()
        ix.knn_query(q, a.k)
        dt = time.perf_counter() - t0
        tb = dt if tb is None else min(tb, dt)
    print("hnsw_slots %d: %.2f ms per %d queries -> %.0f QPS" % (sl, tb * 1e3, a.queries, a.queries / tb), flush=True)
evals = ix.last_distance_evals()
st = ix.stats()
if a.readers > 1:   # reader lanes: R threads, 6 batches each, against one thread answering the same 6 R batches
    from concurrent.futures import ThreadPoolExecutor
    nb = 6
    t0 = time.perf_counter()
    for _ in range(nb * a.readers):
        ix.knn_query(q, a.k)
    one = time.perf_counter() - t0
    with ThreadPoolExecutor(a.readers) as pool:
        t0 = time.perf_counter()
        list(pool.map(lambda t: [ix.knn_query(q, a.k) for _ in range(nb)], range(a.readers)))
        many = time.perf_counter() - t0
    print("readers: 1 thread %.0f QPS, %d threads %.0f QPS (x %.2f)" % (nb * a.readers * a.queries / one, a.readers,
          nb * a.readers * a.queries / many, one / many), flush=True)
bp = VecSim.BFParams()
bp.type, bp.dim, bp.metric = VecSim.VecSimType_FLOAT32, a.dim, VecSim.VecSimMetric_L2
bf = VecSim.BFIndex(bp)
bf.add_vectors(rows, np.arange(a.rows))
nx = a.exact_queries or a.queries
exact = np.concatenate([bf.knn_query(q[i:i + 64], a.k)[0] for i in range(0, nx, 64)])[:nx]
recall = sum(len(set(labels[i]) & set(exact[i])) for i in range(nx)) / (nx * a.k)
ef_curve = []
for ef in [int(x) for x in a.efs.split(",") if x]:
    qp = VecSim.VecSimQueryParams()
    qp.hnswRuntimeParams.efRuntime = ef
    ix.knn_query(q, a.k, qp)
    tb = None
    for _ in range(3):
        t0 = time.perf_counter()
        le, _ = ix.knn_query(q, a.k, qp)
        dt = time.perf_counter() - t0
        tb = dt if tb is None else min(tb, dt)
    ef_curve.append({"ef": ef, "qps": a.queries / tb, "recall": sum(len(set(le[i]) & set(exact[i])) for i in range(nx)) / (nx * a.k),
                     "dist_evals_per_query": ix.last_distance_evals() / a.queries})
if a.iterator:
    # the batch iterator (hnsw_batch_iterator.h:96-230): host walk, GPU distances fetched VECSIM_HNSW_ITER_AHEAD expansions ahead
    from oracle import vso   # (the CPU twin of the walk: a baseline beside the number, as in bench.py)
    g = ix.graph()
    nbat = 10
    t0 = time.perf_counter()
    got = []
    for i in range(a.iterator):
        it = ix.create_batch_iterator(q[i])
        got.append([it.get_next_results(a.k, VecSim.BY_SCORE) for _ in range(nbat)])
    it_ms = (time.perf_counter() - t0) * 1e3 / (a.iterator * nbat)
    t0 = time.perf_counter()
    same = True
    for i in range(a.iterator):
        want, _ = vso.hnsw_iterate(0, 0, rows, g, q[i], a.ef, [a.k] * nbat, a.dim)
        same = same and all(np.array_equal(l[0][:len(wl)], wl.astype(np.int64)) and np.array_equal(d[0][:len(wd)], wd)
                            for (l, d), (wl, wd) in zip(got[i], want))
    cpu_ms = (time.perf_counter() - t0) * 1e3 / (a.iterator * nbat)
    # exactness of what the walk hands out: the first nbat * k labels against the exact order
    hit = sum(len(set(np.concatenate([l[0] for l, _ in got[i]]).tolist()) & set(bf.knn_query(q[i:i + 1], nbat * a.k)[0][0].tolist()))
              for i in range(min(a.iterator, 16))) / (min(a.iterator, 16) * nbat * a.k)
    print(json.dumps({"iterator": "HNSW batch iterator, %d queries x %d batches of %d, efR %d" % (a.iterator, nbat, a.k, a.ef),
                      "ms_per_batch": it_ms, "cpu_oracle_ms_per_batch": cpu_ms, "equal_to_oracle": bool(same),
                      "recall_of_first_%d" % (nbat * a.k): hit, "ahead": os.environ.get("VECSIM_HNSW_ITER_AHEAD", "8")}), flush=True)
kms = st["scan_ms"] / max(1, st["scan_launches"])
print(json.dumps({"config": "HNSW fp32 L2 N=%d d=%d M=%d efC=%d efR=%d k=%d data=%s" % (a.rows, a.dim, a.M, a.efc, a.ef, a.k, a.data),
                  "host_build_s": build_s, "queries": a.queries, "batch_ms": best * 1e3, "qps": a.queries / best,
                  "recall_at_%d" % a.k: recall, "dist_evals_per_query": evals / a.queries,
                  "search_kernel_ms": kms, "gather_GBps": evals * a.dim * 4 / (kms * 1e-3) / 1e9,
                  "build_threads": a.threads or "all cores (<= 64)", "ef_curve": ef_curve}))
