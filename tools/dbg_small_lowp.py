import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
os.environ["VECSIM_GPU_TIER"] = "avx512"
import numpy as np
from oracle import vso
from util import METRICS, TYPES, random_vectors, stored_rows
from vectorsimilarity_amd import VecSim
vso.build()
def run(typ, metric, dim, n, nq, k, seed):
    rng = np.random.default_rng(seed)
    rows = random_vectors(rng, n, dim, typ, vso); q = random_vectors(rng, nq, dim, typ, vso)
    p = VecSim.BFParams(); p.type, p.dim, p.metric = TYPES[typ], dim, METRICS[metric]
    ix = VecSim.BFIndex(p); ix.add_vectors(rows, np.arange(n))
    ix.set_option("dense_pairs_sliced", 0)
    ix.reset_stats()
    l, d = ix.knn_query(q, k)
    st = ix.stats()
    bad = 0
    for j in range(nq):
        sr = stored_rows(vso, rows, typ, metric); sq = stored_rows(vso, q[j:j+1], typ, metric)[0]
        km = METRICS["IP"] if (metric == "Cosine" and typ not in ("i8","u8")) else METRICS[metric]
        el, es = vso.flat_topk(TYPES[typ], km, sr, sq, k, dim)
        if not (np.array_equal(l[j], el.astype(np.int64)) and np.array_equal(d[j], es)):
            bad += 1
            print("  MISMATCH q", j, "got", l[j][:6], d[j][:3], "want", el[:6], es[:3])
    print(typ, metric, dim, n, nq, k, "kernel", st["scan_kernel"], "cand", st["candidates"], "fallbacks", st["fallbacks"], "retries", st["retries"], "bad", bad)
run("bf16", "IP", 64, 60000, 2, 10, 60074)
for nq in (1, 2, 3, 8, 64):
    for dim in (32, 64, 100, 128, 256):
        for typ in ("bf16", "f16"):
            run(typ, "IP", dim, 60000, nq, 10, 5)
