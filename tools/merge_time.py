"""Host merge of a sharded batch (csrc/host/sharded_index.cpp merge_topk) timed at the exchange sizes of the BASELINE configs: what every
rank does per batch with the records of G shards -- k + ties candidates per query and shard.  Run as a subprocess per thread count
(VECSIM_GPU_MERGE_THREADS is read once).  No GPU involved."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def one():
    from vectorsimilarity_amd import _capi
    lib = _capi.load()
    f = lib.VecSimGpu_MergeTopK
    f.restype = C.c_int
    f.argtypes = [C.c_size_t] * 3 + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(1)
    # (name, queries, shards, cap, candidates per query and shard, k)
    for name, nq, parts, cap, cnt, k in (("c2 x8", 64, 8, 26, 10, 10), ("c4 x8", 128, 8, 26, 10, 10), ("c3 x8", 256, 8, 206, 100, 100),
                                          ("c3 x2", 256, 2, 206, 100, 100), ("c3 x1", 256, 1, 206, 100, 100)):
        gids = rng.permutation(parts * nq * cap).astype(np.uint64).reshape(parts, nq, cap)
        labels = gids.copy()
        scores = rng.random((parts, nq, cap))
        counts = np.full((parts, nq), cnt, np.uint32)
        ol = np.empty((nq, k), np.int64)
        os_ = np.empty((nq, k))
        t = []
        for _ in range(25):
            t0 = time.perf_counter()
            rc = f(nq, parts, cap, gids.ctypes.data, labels.ctypes.data, scores.ctypes.data, counts.ctypes.data, k, ol.ctypes.data, os_.ctypes.data)
            t.append(time.perf_counter() - t0)
        assert rc == 0
        print("  %-6s %3d queries x %d shards x %3d candidates, k %3d: min %.3f ms, median %.3f ms  (record %d bytes per rank)"
              % (name, nq, parts, cnt, k, min(t) * 1e3, float(np.median(t)) * 1e3, 32 + nq * (1 + 3 * cap) * 8), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        print("host cores:", os.cpu_count())
        for th in (1, 2, 4, 8, 16):
            print("VECSIM_GPU_MERGE_THREADS=%d" % th, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, VECSIM_GPU_MERGE_THREADS=str(th)), check=True)
