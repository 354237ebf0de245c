#!/usr/bin/env python3
"""Exports the graphs the product's host builder makes for the reference's deterministic HNSW tests (tests/golden/
kat_hnsw.json), so that the CPU suite can pin oracle/vso_hnsw.c on those known answers without a GPU:
    python tests/golden/make_hnsw_graphs.py [out.npz]        (needs an MI355X: index construction opens a GPU context)
The file holds, per case, the stored vectors and the adjacency arrays of HNSWIndex.graph()."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from vectorsimilarity_amd import VecSim  # noqa: E402

MET = {"L2": VecSim.VecSimMetric_L2, "IP": VecSim.VecSimMetric_IP, "Cosine": VecSim.VecSimMetric_Cosine}
with open(os.path.join(HERE, "kat_hnsw.json")) as f:
    kats = json.load(f)
out = {}


def export(name, rows, labels, dim, metric, M, efc):
    p = VecSim.HNSWParams()
    p.type, p.dim, p.metric, p.M, p.efConstruction = VecSim.VecSimType_FLOAT32, dim, MET[metric], M, efc
    ix = VecSim.HNSWIndex(p)
    for v, lab in zip(rows, labels):
        ix.add_vector(v, int(lab))
    g = ix.graph()
    for k, v in g.items():
        out["%s/%s" % (name, k)] = np.asarray(v)
    stored = np.stack([ix.get_vector(int(lab))[0] for lab in labels]).astype(np.float32)   # as stored (Cosine: normalised)
    out["%s/stored" % name] = stored


for c in kats["topk"]:
    export(c["name"], np.array(c["vectors"], dtype=np.float32), c["labels"], c["dim"], c["metric"], c["M"], c["efConstruction"])
r = kats["range"]
export("rangeQuery", np.repeat(np.arange(r["n"], dtype=np.float32)[:, None], r["dim"], axis=1), list(range(r["n"])), r["dim"],
       r["metric"], r["M"], r["efConstruction"])
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "kat_hnsw_graphs.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")
