#!/usr/bin/env python3
"""A/B of the low-precision MFMA filter variants on the config-3 (int8) and config-4 (bf16) shapes."""
import argparse
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--scale", type=float, default=0.2)
ap.add_argument("--i8", default="0,7,9,15")
ap.add_argument("--bf16", default="0,3,8")
a = ap.parse_args()
CASES = [("i8 cos 1024 B256", VecSim.VecSimType_INT8, VecSim.VecSimMetric_Cosine, 1024, int(50_000_000 * a.scale), 256, 100, synth.rows_i8, 1028),
         ("bf16 ip 768 B128", VecSim.VecSimType_BFLOAT16, VecSim.VecSimMetric_IP, 768, 6_000_000, 128, 10, synth.rows_bf16, 1536)]
for name, typ, metric, dim, n, nq, k, gen, rb in CASES:
    if not (a.i8 if 'i8' in name else a.bf16):
        continue
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = typ, dim, metric
    ix = VecSim.BFIndex(p)
    ix.add_synthetic(n, 47)
    q = [gen(48 + i, 0, nq, dim) for i in range(2)]
    base = [ix.knn_query(x, k) for x in q]
    res = {}
    for r in range(a.rounds):
        # "v" (one workgroup per CU in the grid) or "v:w" (w workgroups per CU)
        combos = [(int(x.split(':')[0]), int(x.split(':')[1]) if ':' in x else 1)
                  for x in (a.i8 if 'i8' in name else a.bf16).split(',') if x]
        for v, w in combos:
            ix.set_option("lowp_qsplit", 1 if v == 100 else 0)
            ix.set_option("lowp_variant", 0 if v == 100 else v)
            ix.set_option("wg_per_cu", w)
            ix.reset_stats()
            for b in range(2):
                l, d = ix.knn_query(q[b], k)
                assert np.array_equal(l, base[b][0]) and np.array_equal(d, base[b][1]), (name, v, w)
            st = ix.stats()
            res.setdefault((v, w), []).append(st["scan_ms"] / st["scan_launches"])
    print(name, "rows", n)
    for (v, w), ts in sorted(res.items(), key=lambda kv: min(kv[1])):
        print("  variant %d wg/cu %d  mean %.3f ms  min %.3f ms  %.0f GB/s" % (v, w, np.mean(ts), min(ts), n * rb / min(ts) / 1e6))
    del ix
