import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from vectorsimilarity_amd import VecSim
rng = np.random.default_rng(0)
n, dim = 300_000, 128
rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
p = VecSim.BFParams(); p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, dim, VecSim.VecSimMetric_L2
ix = VecSim.BFIndex(p); ix.add_vectors(rows, np.arange(n))
q = rng.uniform(-1, 1, (5, dim)).astype(np.float32)
for k in (1000, 3000, 20000, 300000):
    ix.set_option("mfma", 1); ix.reset_stats()
    t0 = time.perf_counter(); l1, d1 = ix.knn_query(q, k); t1 = time.perf_counter() - t0
    st = ix.stats()
    ix.set_option("mfma", 0)
    l2, d2 = ix.knn_query(q, k)
    d = np.sum((rows[None, :, :] - q[:, None, :]) ** 2, axis=2) if k == 1000 else None
    print("k", k, "ms %.1f" % (t1 * 1e3), st["scan_kernel"], "fallbacks", st["fallbacks"], "same as exact path:", np.array_equal(l1, l2) and np.array_equal(d1, d2),
          "sorted", bool(np.all(np.diff(d1, axis=1) >= 0)), flush=True)
