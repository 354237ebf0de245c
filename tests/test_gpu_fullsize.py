"""GPU: the tables bench.py TIMES, checked against the oracle at FULL size (BASELINE.json configs 2, 3 and 4): slab-table
lookups across hundreds / thousands of slabs, byte offsets far beyond 4 GiB, and the last (partial) tile only exist there.
The stored rows are streamed back in <= 1 GiB pieces and scored by the oracle's reference-order kernels on the host cores
(oracle.vso.StreamTopK = brute_force.h:242-291 over a running candidate set); labels, order and scores must be bit-equal
(tolerance: none).  A few queries per config are checked -- the first, one in the middle, the last query tile, and queries
PLANTED on rows of the last slab (row N-1 must come back first)."""
import os

import numpy as np
import pytest

from util import METRICS, TYPES, stored_rows
from vectorsimilarity_amd import VecSim, synth

pytestmark = pytest.mark.gpu

PIECE_BYTES = 1 << 30


def _tier(vso):
    return {"avx512": vso.TIER_AVX512, "scalar": vso.TIER_SCALAR, "avx512_bf16": vso.TIER_AVX512_BF16,
            "avx512_fp16": vso.TIER_AVX512_FP16}[os.environ.get("VECSIM_GPU_TIER", "avx512")]


def stream_oracle(vso, ix, typ, metric, dim, n, qblobs, k, threads, label_of=None):
    """the oracle's reply for the query blobs `qblobs` over the n stored rows of `ix`, streamed back piece by piece"""
    km = METRICS["IP"] if (metric == "Cosine" and typ not in ("i8", "u8")) else METRICS[metric]
    st = vso.StreamTopK(TYPES[typ], km, qblobs, k, dim, threads=threads, tier=_tier(vso))
    probe = ix.stored_rows(0, 1)
    per = max(1, PIECE_BYTES // probe.shape[1])
    buf = np.empty(per * probe.shape[1], dtype=np.uint8)
    for r0 in range(0, n, per):
        c = min(per, n - r0)
        st.feed(ix.stored_rows(r0, c, out=buf), r0)
    assert st.rows_seen == n
    return st.result(label_of)


# (typ, metric, dim, rows, batch, k, generator): BASELINE.json configs 2, 3, 4 (4: one GPU's shard of the 100 M rows)
FULL = [("f32", "L2", 768, 10_000_000, 64, 10, "rows_f32"),
        ("i8", "Cosine", 1024, 50_000_000, 256, 100, "rows_i8"),
        ("bf16", "IP", 768, 12_500_000, 128, 10, "rows_bf16")]


@pytest.mark.parametrize("typ,metric,dim,n,nq,k,gen", FULL, ids=["c2", "c3", "c4"])
def test_full_size_table_matches_oracle(vso, typ, metric, dim, n, nq, k, gen):
    g = getattr(synth, gen)
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = TYPES[typ], dim, METRICS[metric]
    ix = VecSim.BFIndex(p)
    ix.add_synthetic(n, 47)
    assert ix.index_size() == n
    # the device generator against its host twin: the first rows, the last rows, and three runs in between
    eb = dim * {"f32": 4, "i8": 1, "bf16": 2}[typ]
    for r0 in (0, n // 3 + 11, n // 2 + 5, (5 * n) // 6 + 1, n - 8):
        dev = ix.stored_rows(r0, 8)
        host = g(47, r0, 8, dim)
        assert np.array_equal(dev[:, :eb], host.view(np.uint8).reshape(8, -1)[:, :eb]), (typ, r0)
    q = g(48, 0, nq, dim).copy()
    # planted queries: the raw elements of row N-1 (last query of the batch = last query tile), of the last slab's middle, of row 0
    plant = {nq - 1: n - 1, nq // 2: n - 1000, 1: 0}
    for qi, row in plant.items():
        q[qi] = g(47, row, 1, dim)[0]
    labels, dists = ix.knn_query(q, k)
    assert ix.stats()["scan_kernel"] != ""
    # config 2: every query of the batch; the others: a spread of them (the host scan is 7 M distances / s / core at 1 KiB rows)
    check = list(range(nq)) if typ == "f32" else sorted(set(range(0, nq, nq // 16)) | {nq - 2} | set(plant))
    qblobs = stored_rows(vso, q[check], typ, metric)
    threads = min(64, os.cpu_count() or 1)
    el, es = stream_oracle(vso, ix, typ, metric, dim, n, qblobs, k, threads)
    for j, qi in enumerate(check):
        assert np.array_equal(labels[qi], el[j]), (typ, qi, labels[qi][:5], el[j][:5])
        assert np.array_equal(dists[qi], es[j]), (typ, qi)
    for qi, row in plant.items():                       # the planted row is its own nearest neighbour
        assert labels[qi][0] == row, (typ, qi, labels[qi][:3])
    # a single query goes down a different path (no batch): same reply for the planted last row
    l1, d1 = ix.knn_query(q[nq - 1], k)
    assert np.array_equal(l1[0], labels[nq - 1]) and np.array_equal(d1[0], dists[nq - 1])
    if typ != "f32":
        return
    # mutation at full size (brute_force.h:196-224): deleting label 5 moves the LAST row (label N-1) into internal id 5 -- a D2D
    # copy from the last slab into the first; the table is one row shorter and the planted query must still find label N-1 first
    assert ix.delete_vector(5) == 1 and ix.index_size() == n - 1
    labels2, dists2 = ix.knn_query(q, k)
    assert labels2[nq - 1][0] == n - 1 and dists2[nq - 1][0] == dists[nq - 1][0]
    moved = ix.stored_rows(5, 1)
    assert np.array_equal(moved[0, :eb], g(47, n - 1, 1, dim).view(np.uint8).reshape(-1)[:eb])

    def label_of(ids):
        out = ids.copy()
        out[ids == 5] = n - 1
        return out
    sub = [0, 1, nq // 2, nq - 1]
    el2, es2 = stream_oracle(vso, ix, typ, metric, dim, n - 1, stored_rows(vso, q[sub], typ, metric), k, threads, label_of)
    for j, qi in enumerate(sub):
        assert np.array_equal(labels2[qi], el2[j]) and np.array_equal(dists2[qi], es2[j]), ("after delete", qi)
