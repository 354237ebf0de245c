#!/usr/bin/env python3
"""Elimination timing of the low-precision filter kernel (option lowp_dbg): which part of a unit costs what."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=8_000_000)
a = ap.parse_args()
for case in ("i8",):
    if case == "i8":
        typ, metric, dim, nq, k, gen, rb = VecSim.VecSimType_INT8, VecSim.VecSimMetric_Cosine, 1024, 256, 100, synth.rows_i8, 1028
    else:
        typ, metric, dim, nq, k, gen, rb = VecSim.VecSimType_BFLOAT16, VecSim.VecSimMetric_IP, 768, 128, 10, synth.rows_bf16, 1536
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = typ, dim, metric
    ix = VecSim.BFIndex(p)
    ix.add_synthetic(a.rows, 47)
    q = gen(48, 0, nq, dim)
    ix.knn_query(q, k)
    for variant in (0, 1, 2):
        ix.set_option("lowp_variant", variant)
        for dbg in (0, 1, 2, 3, 5, 6, 7):
            ix.set_option("lowp_dbg", dbg)
            ix.reset_stats()
            for _ in range(3):
                try:
                    ix.knn_query(q, k)
                except Exception:
                    pass
            st = ix.stats()
            ms = st["scan_ms"] / max(1, st["scan_launches"])
            print("%s variant %d dbg %d (%s%s%s): %.3f ms  %.0f GB/s-equivalent" % (
                case, variant, dbg, "noEpi " if dbg & 1 else "", "noMFMA " if dbg & 2 else "", "noDMA" if dbg & 4 else "",
                ms, a.rows * rb / ms / 1e6), flush=True)
    ix.set_option("lowp_dbg", 0)
    del ix
