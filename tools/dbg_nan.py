import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from vectorsimilarity_amd import VecSim
from util import TYPES, METRICS, random_vectors
from oracle import vso
for typ, metric, dim in (("bf16", "IP", 72), ("f32", "IP", 72), ("bf16", "L2", 72), ("f32", "IP", 128), ("bf16", "IP", 96)):
    for bad in ((1, 3), (2000, 3), None):
        rng = np.random.default_rng(82)
        n = 3000
        rows = random_vectors(rng, n, dim, typ, vso)
        q = random_vectors(rng, 5, dim, typ, vso)
        if bad:
            rows[bad[0], bad[1]] = {"f32": np.float32(np.nan), "bf16": np.uint16(0x7FC0)}[typ]
        p = VecSim.BFParams()
        p.type, p.dim, p.metric = TYPES[typ], dim, METRICS[metric]
        ix = VecSim.BFIndex(p)
        ix.add_vectors(rows, np.arange(n))
        for k in (1, 2, 10):
            ix.reset_stats()
            l, d = ix.knn_query(q, k)
            st = ix.stats()
            print(typ, metric, dim, "bad", bad, "k", k, "labels0", l[:, 0], st.get("scan_kernel"), st.get("fallbacks"))
