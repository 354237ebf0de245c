"""vectorsimilarity_amd -- MI355X-native distance-kernel hot path of VectorSimilarity.

Layout (only what the hot path needs):
  csrc/            HIP kernels + C-ABI shim (include/vsgpu.h) and the C++ host index behind the
                   VecSim C API (include/VecSim/*.h)
  _capi.py         ctypes view of libvecsim_amd.so
  VecSim.py        Python surface with the reference module's names (BFParams, BFIndex, knn_query ...)
  sharded.py       multi-GPU Flat index: row shards + all-gather of per-shard partial top-K

Importing the package does not load the native libraries; the first index does, and raises when
they are missing.  There is no CPU implementation of any distance.
"""
__all__ = ["VecSim", "_capi", "sharded"]
