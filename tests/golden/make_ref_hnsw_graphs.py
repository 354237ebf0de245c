#!/usr/bin/env python3
"""Golden HNSW graphs BUILT BY THE REFERENCE ITSELF: the two serialized indexes its own unit tests hold
(/root/reference/tests/unit/data/1k-d4-L2-M8-ef_c10_FLOAT32_{single,multi_100labels}.v3, loaded by
tests/unit/test_hnsw.cpp:1996-2052 HNSWSerializationV3) decoded into plain arrays -> tests/golden/ref_hnsw_graphs.npz.

These are DATA files (bytes a reference build wrote with saveIndex), not source.  The layout is read the way
hnsw_serializer_impl.h:145-243 restores it (V3: hnsw_factory.cpp:173-205 header, data_blocks_container.cpp:76-112 vector
blocks with their lengths, then per element toplevel + per level {numLinks u16, links u32[], incoming count u32, ids u32[]}).
1001 fp32 vectors of dim 4, L2, M = 8, efConstruction = 10, block size 2; labels j (single) / j % 100 (multi).
What the fixture pins: the reference's INSERT path (hnsw.h:1567-1610, 889-963, 743-797, 1857-1946) -- re-inserting the
stored vectors in id order must reproduce every link list, level, the entry point and the incoming-edge sets.

    python tests/golden/make_ref_hnsw_graphs.py        (needs /root/reference; the .npz is committed)
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/tests/unit/data"


class Rd:
    def __init__(self, b):
        self.b, self.o = b, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v


def decode(path):
    r = Rd(open(path, "rb").read())
    version = r.take("i")
    algo = r.take("i")
    dim, vtype, metric, block, multi, cap = r.take("Q"), r.take("i"), r.take("i"), r.take("Q"), r.take("?"), r.take("Q")
    M, M0, efc, ef, eps, mult = r.take("Q"), r.take("Q"), r.take("Q"), r.take("Q"), r.take("d"), r.take("d")
    n, n_del, max_level, entry = r.take("Q"), r.take("Q"), r.take("Q"), r.take("I")
    assert version == 3 and algo == 1 and vtype == 0 and metric == 0, (version, algo, vtype, metric)
    labels = np.zeros(n, dtype=np.uint64)
    flags = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        labels[i], flags[i] = r.take("Q"), r.take("B")
    nblocks = r.take("I")
    vecs = []
    for _ in range(nblocks):
        bl = r.take("I")
        for _ in range(bl):
            vecs.append(np.frombuffer(r.b, dtype=np.float32, count=dim, offset=r.o).copy())
            r.o += 4 * dim
    vecs = np.stack(vecs)
    assert vecs.shape == (n, dim)
    levels = np.zeros(n, dtype=np.int32)
    links, incoming = [], []          # flat records (node, level, position, neighbour) / (node, level, id)
    node = 0
    while node < n:
        bl = r.take("I")
        for _ in range(bl):
            top = r.take("Q")
            levels[node] = top
            for lv in range(top + 1):
                cnt = r.take("H")
                for p in range(cnt):
                    links.append((node, lv, p, r.take("I")))
                inc = r.take("I")
                for _ in range(inc):
                    incoming.append((node, lv, r.take("I")))
            node += 1
    assert r.o == len(r.b), (r.o, len(r.b))
    return dict(dim=dim, block=block, multi=int(multi), M=M, M0=M0, efc=efc, ef=ef, eps=eps, mult=mult, n=n, n_deleted=n_del,
                max_level=max_level, entry=entry, labels=labels, flags=flags, vectors=vecs, levels=levels,
                links=np.array(links, dtype=np.int64).reshape(-1, 4), incoming=np.array(incoming, dtype=np.int64).reshape(-1, 3))


def main():
    out = {}
    for name in ("single", "multi_100labels"):
        d = decode(os.path.join(SRC, "1k-d4-L2-M8-ef_c10_FLOAT32_%s.v3" % name))
        for k, v in d.items():
            out["%s/%s" % (name, k)] = np.asarray(v)
        print(name, {k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in d.items()})
    np.savez_compressed(os.path.join(HERE, "ref_hnsw_graphs.npz"), **out)


if __name__ == "__main__":
    main()
