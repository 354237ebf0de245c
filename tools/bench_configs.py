#!/usr/bin/env python3
"""Kernel-level numbers for the other BASELINE.json Flat configs on one MI355X (parity cases elsewhere;
these are not bench.py lines): config 3 int8 Cosine d=1024 batch-256 top-100, config 4 (per-GPU shard)
bf16 IP d=768 batch-128 top-10, config 1 fp32 L2 100K x 128 single query.
    python tools/bench_configs.py [--scale 1.0]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--only", default="")
a = ap.parse_args()

CASES = [
    ("C1 fp32 L2 100K x128 B=1 k=10", VecSim.VecSimType_FLOAT32, VecSim.VecSimMetric_L2, 128, 100_000, 1, 10, synth.rows_f32, 512, False),
    ("C3 int8 Cosine 50M x1024 B=256 k=100", VecSim.VecSimType_INT8, VecSim.VecSimMetric_Cosine, 1024, 50_000_000, 256, 100, synth.rows_i8, 1028, True),
    ("C4 bf16 IP 12.5M x768 B=128 k=10 (one of 8 shards)", VecSim.VecSimType_BFLOAT16, VecSim.VecSimMetric_IP, 768, 12_500_000, 128, 10, synth.rows_bf16, 1536, True),
]
import faulthandler
faulthandler.enable()
for name, typ, metric, dim, n, nq, k, gen, row_bytes, scaled in CASES:
    if a.only and not name.startswith(a.only):
        continue
    if scaled:
        n = int(n * a.scale)
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = typ, dim, metric
    ix = VecSim.BFIndex(p)
    print('filling', name, n, flush=True)
    ix.add_synthetic(n, 47)
    print('filled', flush=True)
    qs = [gen(48 + i, 0, nq, dim) for i in range(3)]
    ix.knn_query(qs[0], k)
    ix.reset_stats()
    t0 = time.perf_counter()
    for s in range(a.steps):
        l, d = ix.knn_query(qs[s % 3], k)
    dt = (time.perf_counter() - t0) / a.steps
    st = ix.stats()
    kms = st["scan_ms"] / max(1, st["scan_launches"])
    bytes_ = n * row_bytes
    print(json.dumps({"config": name, "rows": n, "ms_per_batch": dt * 1e3, "distances_per_s": n * nq / dt,
                      "qps": nq / dt, "scan_kernel": st["scan_kernel"], "scan_kernel_ms": kms,
                      "launches_per_batch": st["scan_launches"] / a.steps,
                      "scan_GBps": bytes_ / (kms * 1e-3) / 1e9 if kms else None,
                      "hbm_frac": bytes_ / (kms * 1e-3) / 8e12 if kms else None,
                      "candidates_per_query": st["candidates"] / (a.steps * nq), "fallbacks": st["fallbacks"],
                      "sorted": bool(np.all(np.diff(d, axis=1) >= 0))}))
    del ix
