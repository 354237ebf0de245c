"""Parity tests of the measured-slower tuning kernels (csrc/mfma_i8ks_kernels.hpp, `make -C vectorsimilarity_amd/csrc TUNING=1`).
They are not part of the shipped library, so they live outside tests/: run explicitly on a GPU box with
    make -C vectorsimilarity_amd/csrc clean all TUNING=1 && python -m pytest tools/tuning_tests -q
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import METRICS, TYPES, random_vectors, stored_rows  # noqa: E402
from test_gpu_flat_parity import kernel_metric, make_index  # noqa: E402

pytestmark = pytest.mark.gpu   # (a plain `pytest -m "not gpu"` from the repo root must not run these)


@pytest.fixture(scope="session")
def vso():
    from oracle import vso as m
    m.build()
    m.lib()
    return m


@pytest.mark.parametrize("flavour", [1, 2, 3])
@pytest.mark.parametrize("typ,metric,dim,n,nq,k", [
    ("i8", "Cosine", 1024, 30_011, 70, 100),    # BASELINE config 3 shape (scaled down); last tile partial
    ("i8", "Cosine", 1024, 9_000, 256, 10),     # a full 256-query tile
    ("i8", "L2", 1024, 20_000, 300, 10),        # two query tiles
    ("i8", "IP", 900, 25_013, 64, 10),          # zero query columns past dim
    ("u8", "L2", 1024, 30_000, 70, 10),
    ("u8", "IP", 800, 20_005, 33, 100),
])
def test_i8_ksplit_filter_bit_exact(vso, typ, metric, dim, n, nq, k, flavour):
    """K-split int8 filter (mfma_i8ks_kernels.hpp: two k-halves per dot product, partial sums exchanged through LDS,
    screening deferred by one tile) against the oracle and against the 16 x 16 filter kernel."""
    rng = np.random.default_rng(dim * 7 + n + nq)
    rows = random_vectors(rng, n, dim, typ, vso)
    q = random_vectors(rng, nq, dim, typ, vso)
    ix = make_index(typ, metric, dim)
    ix.add_vectors(rows, np.arange(n))
    ix.set_option("dense_pairs", 0)
    ix.set_option("lowp_ksplit", 0)
    l0, d0 = ix.knn_query(q, k)
    ix.set_option("lowp_ksplit", flavour)
    ix.reset_stats()
    l1, d1 = ix.knn_query(q, k)
    st = ix.stats()
    if st["scan_kernel"] != "k_i8_filter_ksplit":
        pytest.skip("K-split kernel is compiled into the tuning build only (make TUNING=1)")
    assert np.array_equal(l0, l1) and np.array_equal(d0, d1)
    srows = stored_rows(vso, rows, typ, metric)
    sq = stored_rows(vso, q, typ, metric)
    km = kernel_metric(typ, metric)
    for j in range(0, nq, 3):
        sc = vso.scan(TYPES[typ], km, srows, sq[j], dim)
        el, es = vso.topk_replay(sc, k)
        assert np.array_equal(l1[j], el.astype(np.int64)), (typ, metric, dim, j)
        assert np.array_equal(d1[j], es), (typ, metric, dim, j)


def test_i8_ksplit_many_workgroup_tiles_and_ties(vso):
    """Every workgroup walks several tiles (exchange slots and aux buffers wrap), rows repeat (exact score ties)."""
    rng = np.random.default_rng(5)
    dim, n, nq, k = 1024, 150_000, 40, 20
    base = random_vectors(rng, 5_000, dim, "i8", vso)
    rows = base[rng.integers(0, 5_000, n)]
    q = random_vectors(rng, nq, dim, "i8", vso)
    ix = make_index("i8", "Cosine", dim)
    ix.add_vectors(rows, np.arange(n))
    ix.set_option("dense_pairs", 0)
    ix.set_option("wg_per_cu", 1)
    ix.set_option("lowp_ksplit", 1)
    l1, d1 = ix.knn_query(q, k)
    if ix.stats()["scan_kernel"] != "k_i8_filter_ksplit":
        pytest.skip("K-split kernel is compiled into the tuning build only (make TUNING=1)")
    srows = stored_rows(vso, rows, "i8", "Cosine")
    sq = stored_rows(vso, q, "i8", "Cosine")
    for j in range(0, nq, 4):
        sc = vso.scan(TYPES["i8"], kernel_metric("i8", "Cosine"), srows, sq[j], dim)
        el, es = vso.topk_replay(sc, k)
        assert np.array_equal(l1[j], el.astype(np.int64)) and np.array_equal(d1[j], es), j


