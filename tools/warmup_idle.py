#!/usr/bin/env python3
"""Is the slow start of the scan kernels an idle effect (clock / power state) or a first-touch effect (page tables, scratch growth)?
Config 4's table, one reader.  Per batch the scan kernel's own HIP-event time (stats delta):
  phase A  12 batches back to back            (fresh index: the cliff)
  phase B  sleep 0.5 s, 6 batches
  phase C  sleep 3 s, 6 batches               (if these are slow again the cause is idleness, not first touch)
  phase D  3 s of sleep with a tiny kernel every 2 ms on the same stream (keeps the device awake), 6 batches
  phase E  a NEW index of the same size built while the old one is alive, first 6 batches right after its fill (first touch without idleness)
"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vectorsimilarity_amd import VecSim, synth  # noqa: E402


def make(n, dim=768):
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_BFLOAT16, dim, VecSim.VecSimMetric_IP
    ix = VecSim.BFIndex(p)
    ix.add_synthetic(n, 47)
    return ix


def batches(ix, q, count, k=10):
    out = []
    for _ in range(count):
        ix.reset_stats()
        ix.knn_query(q, k)
        out.append(ix.stats()["scan_ms"] * 1e3)
    return " ".join("%d" % x for x in out)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
q = synth.rows_bf16(48, 0, 128, 768)
ix = make(n)
print("A fresh index      :", batches(ix, q, 12))
time.sleep(0.5)
print("B after 0.5 s idle :", batches(ix, q, 6))
time.sleep(3.0)
print("C after 3 s idle   :", batches(ix, q, 6))
small = make(4096)
t0 = time.time()
while time.time() - t0 < 3.0:
    small.knn_query(q[:1], 1)
    time.sleep(0.002)
print("D 3 s kept awake   :", batches(ix, q, 6))
ix2 = make(n)
print("E second index, right after its fill:", batches(ix2, q, 6))
print("F first index again:", batches(ix, q, 4))
