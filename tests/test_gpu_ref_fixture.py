"""GPU: the HIP path in the scalar tier (VECSIM_GPU_TIER=scalar) against tests/golden/ref_scalar_random.json -- bits the
REFERENCE'S OWN compiled scalar kernels produced on seeded random inputs (tests/golden/make_ref_scalar_random.py).  Through
the C API: an index per (type, metric, dim), the case's four raw rows added, the raw query asked for all four -- so Cosine
cases also pin the product's ingest / query normalisation (blob_prep.h) against normalize_naive.h as compiled.
Tolerance: none.  Nothing here touches the oracle: fixture in, C API out."""
import json
import os
import sys

import numpy as np
import pytest

from util import METRICS, TYPES
from vectorsimilarity_amd import VecSim

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
import refgen  # noqa: E402


@pytest.fixture(scope="module")
def fixture():
    with open(os.path.join(GOLD, "ref_scalar_random.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def distances(fixture):
    return fixture["distances"]


@pytest.mark.parametrize("typ", refgen.TYPES)
def test_scalar_tier_scores_equal_the_compiled_reference(monkeypatch, distances, typ):
    monkeypatch.setenv("VECSIM_GPU_TIER", "scalar")
    cases = [e for e in distances if e["type"] == typ]
    assert len(cases) == len(refgen.METRICS) * len(refgen.DIMS)
    bad = []
    for e in cases:
        metric, dim = e["metric"], e["dim"]
        rows, q = refgen.distance_inputs(typ, metric, dim)
        p = VecSim.BFParams()
        p.type, p.dim, p.metric = TYPES[typ], dim, METRICS[metric]
        ix = VecSim.BFIndex(p)
        ix.add_vectors(rows, np.arange(refgen.ROWS))
        labels, dists = ix.knn_query(q[None, :], refgen.ROWS)
        got = [None] * refgen.ROWS
        for lab, d in zip(labels[0], dists[0]):
            got[int(lab)] = refgen.hexbits(np.float64(d) if typ == "f64" else np.float32(d))[0]
        if got != e["scores"]:
            bad.append((metric, dim, got, e["scores"]))
        # the same numbers through getDistanceFrom (brute_force_single.h:202-212; it takes the blob as it is -- no query
        # preprocessing -- so Cosine, whose queries are normalised by the top-k path, is left to the check above)
        if metric == "Cosine":
            continue
        d0 = ix.get_distance_from(0, q)
        h0 = refgen.hexbits(np.float64(d0) if typ == "f64" else np.float32(d0))[0]
        if h0 != e["scores"][0]:
            bad.append((metric, dim, "get_distance_from", h0, e["scores"][0]))
    assert not bad, "%s: %d cases differ, first %r" % (typ, len(bad), bad[:3])


@pytest.mark.parametrize("tier", ["avx512", "scalar"])
def test_uint8_rows_beyond_33025_elements_equal_the_compiled_reference(monkeypatch, fixture, tier):
    """above the 32-bit bound every reference chooser returns the scalar kernel and its 64-bit total (spaces.h:57-66,
    L2_space.cpp:474-476, IP_space.cpp:788,843) -- whatever the host's tier; 255 x 255 x 33 026 no longer fits an int32"""
    monkeypatch.setenv("VECSIM_GPU_TIER", tier)
    bad = []
    for e in fixture["wide_u8"]:
        if tier == "scalar" and e["dim"] > 40000:
            continue
        a, b = refgen.wide_u8_inputs(e["dim"], e["kind"], e["metric"])
        p = VecSim.BFParams()
        p.type, p.dim, p.metric = TYPES["u8"], e["dim"], METRICS[e["metric"]]
        ix = VecSim.BFIndex(p)
        ix.add_vectors(a, np.arange(1))
        labels, dists = ix.knn_query(b[None, :], 1)
        got = refgen.hexbits(np.float32(dists[0][0]))[0]
        if got != e["score"]:
            bad.append((e, got))
    assert not bad, bad[:3]
