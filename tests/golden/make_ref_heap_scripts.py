#!/usr/bin/env python3
"""tests/golden/ref_heap_scripts.json: random scripts of emplace / pop on the REFERENCE's own containers (oracle/_ref:
vecsim_stl::min_priority_queue, max_priority_queue, updatable_max_heap -- the state of the HNSW batch iterator and of the top-k
loops) with the size and the top after every operation.  Inputs are regenerated from the seeds; expected tops are stored.
    bash oracle/build_ref.sh && python tests/golden/make_ref_heap_scripts.py        (needs /root/reference: runs in the build container)"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def script(seed, kind, n):
    """(op, score, label): scores from a small set (ties), +-inf, and -- for the two std::priority_queue kinds -- NaN"""
    rng = np.random.default_rng(seed)
    op = (rng.random(n) < 0.35).astype(np.int32)          # 35 % pops
    pool = rng.integers(-4, 5, 12).astype(np.float64) / 2
    score = rng.choice(pool, n)
    score[rng.random(n) < 0.04] = np.inf
    score[rng.random(n) < 0.03] = -np.inf
    if kind != 2:
        score[rng.random(n) < 0.06] = np.nan
    label = rng.integers(0, 40 if kind == 2 else 1000, n).astype(np.uint64)
    return op, score, label


def run(lib, kind, op, score, label):
    n = len(op)
    sz, ts, tl = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.float64), np.zeros(n, dtype=np.uint64)
    lib(C.c_int(kind), op.ctypes.data_as(C.c_void_p), score.ctypes.data_as(C.c_void_p), label.ctypes.data_as(C.c_void_p), C.c_size_t(n),
        sz.ctypes.data_as(C.c_void_p), ts.ctypes.data_as(C.c_void_p), tl.ctypes.data_as(C.c_void_p))
    return sz, ts, tl


CASES = [(seed, kind, n) for kind in (0, 1, 2) for seed, n in ((11, 400), (12, 400), (13, 1500), (14, 60))]

if __name__ == "__main__":
    L = C.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libvsref.so"))
    out = []
    for seed, kind, n in CASES:
        op, score, label = script(seed, kind, n)
        sz, ts, tl = run(L.vsref_heap_script, kind, op, score, label)
        out.append({"seed": seed, "kind": kind, "n": n, "size": [int(x) for x in sz],
                    "top_score_bits": [int(x) for x in ts.view(np.uint64)], "top_label": [int(x) for x in tl]})
    with open(os.path.join(HERE, "ref_heap_scripts.json"), "w") as f:
        json.dump({"how": "oracle/_ref/libvsref.so:vsref_heap_script (the reference's containers, gnu++20) on make_ref_heap_scripts.script(seed, kind, n)",
                   "cases": out}, f)
    print("wrote", len(out), "scripts")
