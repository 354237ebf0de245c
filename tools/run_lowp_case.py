#!/usr/bin/env python3
"""One low-precision Flat case a few times (profiling target): --case i8|bf16, --rows N, --variant V."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="i8")
ap.add_argument("--rows", type=int, default=8_000_000)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--dbg", type=int, default=0)
a = ap.parse_args()
if a.case == "i8":
    typ, metric, dim, nq, k, gen, rb = VecSim.VecSimType_INT8, VecSim.VecSimMetric_Cosine, 1024, 256, 100, synth.rows_i8, 1028
else:
    typ, metric, dim, nq, k, gen, rb = VecSim.VecSimType_BFLOAT16, VecSim.VecSimMetric_IP, 768, 128, 10, synth.rows_bf16, 1536
p = VecSim.BFParams()
p.type, p.dim, p.metric = typ, dim, metric
ix = VecSim.BFIndex(p)
ix.add_synthetic(a.rows, 47)
ix.set_option("lowp_variant", a.variant)
q = gen(48, 0, nq, dim)
ix.knn_query(q, k)
ix.set_option("lowp_dbg", a.dbg)
ix.reset_stats()
for _ in range(a.reps):
    ix.knn_query(q, k)
st = ix.stats()
ms = st["scan_ms"] / st["scan_launches"]
print("%s rows %d variant %d: %.3f ms/launch  %.0f GB/s" % (a.case, a.rows, a.variant, ms, a.rows * rb / ms / 1e6))
