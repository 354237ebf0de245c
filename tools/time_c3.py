import os, sys, time
sys.path.insert(0, "/root/repo") if os.path.exists("/root/repo/vectorsimilarity_amd") else None
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from vectorsimilarity_amd import VecSim, synth
p = VecSim.BFParams(); p.type, p.dim, p.metric = VecSim.VecSimType_INT8, 1024, VecSim.VecSimMetric_Cosine
ix = VecSim.BFIndex(p); ix.add_synthetic(10_000_000, 47)
q = synth.rows_i8(48, 0, 256, 1024)
ix.knn_query(q, 100)
for _ in range(3):
    t0 = time.perf_counter(); ix.knn_query(q, 100); print("python wall ms", (time.perf_counter() - t0) * 1e3, flush=True)
st = ix.stats(); print(st)
