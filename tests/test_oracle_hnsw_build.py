"""CPU: the oracle's HNSW INSERT path (oracle/vso_hnsw.c vso_hnsw_build) against graphs the REFERENCE itself built.

tests/golden/ref_hnsw_graphs.npz = the two serialized indexes the reference's unit tests hold
(tests/unit/data/1k-d4-L2-M8-ef_c10_FLOAT32_{single,multi_100labels}.v3, test_hnsw.cpp:1996-2052), decoded by
tests/golden/make_ref_hnsw_graphs.py.  Re-inserting their 1001 stored vectors in id order must reproduce the file:
every node's level (the level generator), the entry point, EVERY link list in the file's order, and the
unidirectional-edge sets (hnsw.h:1567-1610, 889-963, 743-797).  Tolerance: none."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_hnsw_graphs.npz")


def golden(name):
    z = np.load(GOLD)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def ref_lists(g):
    out = {}
    for node, lv, pos, nb in g["links"]:
        out.setdefault((int(node), int(lv)), []).append(int(nb))
    for i in range(int(g["n"])):
        for lv in range(int(g["levels"][i]) + 1):
            out.setdefault((i, lv), [])
    return out


def incoming_sets(lists):
    """(node, level) -> ids that point at node without being pointed back at: what the reference records as the node's
    incoming unidirectional edges (graph_data.h:17-137; checkIntegrity hnsw_serializer_impl.h:58-143 counts them this way)"""
    inc = {}
    for (i, lv), nbs in lists.items():
        for nb in nbs:
            if i not in lists[(nb, lv)]:
                inc.setdefault((nb, lv), set()).add(i)
    return inc


@pytest.mark.parametrize("name", ["single", "multi_100labels"])
@pytest.mark.parametrize("fast", [False, True])
def test_oracle_insert_path_rebuilds_the_reference_graph(vso, name, fast):
    g = golden(name)
    n, M, efc = int(g["n"]), int(g["M"]), int(g["efc"])
    assert (n, int(g["dim"]), M, int(g["M0"]), efc) == (1001, 4, 8, 16, 10)
    built = vso.hnsw_build(vso.F32, vso.L2, g["vectors"], 4, M, efc, fast=fast, labels=g["labels"])
    assert np.array_equal(built["levels"], g["levels"].astype(np.uint8))          # the level generator (seed 100)
    assert built["entry"] == int(g["entry"]) and built["max_level"] == int(g["max_level"])
    want, got = ref_lists(g), vso.graph_lists(built)
    assert want.keys() == got.keys()
    diff = [k for k in want if want[k] != got[k]]
    assert not diff, (len(diff), diff[:5], [want[k] for k in diff[:2]], [got[k] for k in diff[:2]])
    ref_inc = {}
    for node, lv, src in g["incoming"]:
        ref_inc.setdefault((int(node), int(lv)), set()).add(int(src))
    assert incoming_sets(got) == ref_inc
    assert len(g["links"]) == sum(len(v) for v in got.values()) == 7376


def test_level_generator_is_libstdcxx_minstd_rand0(vso):
    """1 / ln(M) scaling and the first draws: the golden file's own mult, and levels for another M from the same stream"""
    g = golden("single")
    assert abs(float(g["mult"]) - 1.0 / np.log(8.0)) < 1e-15
    b16 = vso.hnsw_build(vso.F32, vso.L2, g["vectors"][:50], 4, 16, 200)
    b8 = vso.hnsw_build(vso.F32, vso.L2, g["vectors"][:50], 4, 8, 10)
    assert np.all(b16["levels"] <= b8["levels"])          # same uniform draws, smaller multiplier
    assert b8["levels"][11] == 1 and b8["levels"][14] == 1 and b8["levels"][:11].sum() == 0
