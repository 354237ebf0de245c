"""Multi-GPU Flat index: a binding over the C ABI's VecSimGpu_Sharded* entry points (include/VecSim/
vec_sim_gpu.h; host code csrc/host/sharded_index.cpp, exchange csrc/vsgpu_comm.hip).

Partition, candidate records, the RCCL all-gather over xGMI and the exact global merge all live in the C++
host library; Python only carries arrays across and -- when the ranks are `torch.distributed` processes --
hands rank 0's 128-byte RCCL id to the others (`broadcast_object_list`, the one thing the library leaves to
its caller).  torch never touches the data path.

    ShardedFlatIndex(params, rank, world, dist)           one process per GPU, RCCL exchange
    ShardedFlatIndex(params, shards=G, devices=[...])      one process drives G shards (may share a GPU)
    ShardedFlatIndex(params, rank, world, dist, transport="dist")   exchange through `dist` itself (gloo in CPU tests)
    ShardedFlatIndex(params, rank, world, transport=(allgather, broadcast))   the caller's own callables (_capi.ALLGATHER_FN shapes)
"""
import ctypes as C
import os

import numpy as np

from . import VecSim, _capi

OVERFLOW = 0xFFFFFFFF


def block_owner(gid, block, world):
    return (gid // block) % world


def local_to_gid(local_id, rank, block, world):
    return ((local_id // block) * world + rank) * block + local_id % block


def gid_to_local(gid, block, world):
    b = gid // block
    return (b // world) * block + gid % block


def _dist_transport(dist):
    """(allgather, broadcast) callbacks that move the library's byte records with torch.distributed"""
    import torch

    def allgather(_user, send, nbytes, recv):
        try:
            world = dist.get_world_size()
            src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
            out = torch.empty((world * nbytes,), dtype=torch.uint8)
            dist.all_gather_into_tensor(out, torch.from_numpy(src.copy()))
            C.memmove(recv, out.numpy().ctypes.data, world * nbytes)
            return 0
        except Exception:  # a Python exception must not unwind through the C caller
            import traceback
            traceback.print_exc()
            return -1

    def broadcast(_user, buf, nbytes, root):
        try:
            arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(nbytes,))
            t = torch.from_numpy(arr.copy())
            dist.broadcast(t, src=root)
            C.memmove(buf, t.numpy().ctypes.data, nbytes)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return -1

    return _capi.ALLGATHER_FN(allgather), _capi.BROADCAST_FN(broadcast)


class ShardedFlatIndex:
    def __init__(self, params, rank=0, world=1, dist=None, shards=None, devices=None, device=None, transport="rccl",
                 external=None):
        self._lib = lib = _capi.load()
        p = _capi.VecSimParams()
        p.algo = _capi.VecSimAlgo_BF
        p.algoParams.bfParams = params
        self.block = params.blockSize or 1024
        self.dim, self.type, self.metric = params.dim, params.type, params.metric
        self._keep = []
        if shards is not None:
            devs = list(devices) if devices is not None else [0] * shards
            arr = (C.c_int * shards)(*devs)
            self._h = lib.VecSimGpu_ShardedNewLocal(C.byref(p), shards, arr)
            self.rank, self.world = -1, shards
        elif external is not None:
            add_fn, cand_fn = external
            ag, _bc = _dist_transport(dist)
            a, c = _capi.SHARD_ADD_FN(add_fn), _capi.SHARD_CAND_FN(cand_fn)
            self._keep += [ag, a, c]
            self._h = lib.VecSimGpu_ShardedNewExternal(C.byref(p), rank, world, a, c, ag, None)
            self.rank, self.world = rank, world
        else:
            if device is None:
                device = int(os.environ.get("VECSIM_GPU_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            if isinstance(transport, tuple):   # the caller's own (allgather, broadcast) callables: MPI, a test's failing transport
                ag, bc = _capi.ALLGATHER_FN(transport[0]), _capi.BROADCAST_FN(transport[1])
                self._keep += [ag, bc]
                self._h = lib.VecSimGpu_ShardedNewWithTransport(C.byref(p), rank, world, device, ag, bc, None)
            elif transport == "dist":
                ag, bc = _dist_transport(dist)
                self._keep += [ag, bc]
                self._h = lib.VecSimGpu_ShardedNewWithTransport(C.byref(p), rank, world, device, ag, bc, None)
            else:
                # Pre-flight, agreed by all ranks BEFORE anyone enters ncclCommInitRank (which blocks until every rank has joined
                # and has no timeout): a rank whose device is missing must fail everybody at once, not leave the others waiting
                if world > 1:
                    ndev = lib.VecSimGpu_DeviceCount()
                    mine = None if 0 <= device < ndev else "rank %d: device %d out of range (%d visible)" % (rank, device, ndev)
                    flags = [None] * world
                    dist.all_gather_object(flags, mine)
                    bad = [f for f in flags if f is not None]
                    if bad:
                        raise RuntimeError("sharded index: " + "; ".join(bad))
                uid = [None]
                if rank == 0:
                    buf = (C.c_char * 128)()
                    if lib.VecSimGpu_ShardedGetUniqueId(buf) != 0:
                        raise RuntimeError("RCCL id: %s" % lib.VecSimGpu_LastError().decode())
                    uid[0] = bytes(buf)
                if world > 1:
                    dist.broadcast_object_list(uid, src=0)
                self._h = lib.VecSimGpu_ShardedNew(C.byref(p), rank, world, device, uid[0])
            self.rank, self.world = rank, world
        if not self._h:
            err = lib.VecSimGpu_LastError()
            raise RuntimeError("sharded index: %s" % (err.decode() if err else "bad parameters"))
        self._qbytes = lib.VecSimParams_GetQueryBlobSize(self.type, self.dim, self.metric)

    # ---- shards held by this process ----
    def local_index(self, shard=None):
        """the shard's own Flat index (stats, options) as a non-owning VecSim.BFIndex"""
        s = self.rank if shard is None else shard
        h = self._lib.VecSimGpu_ShardedLocalIndex(self._h, max(s, 0))
        if not h:
            return None
        ix = VecSim.VecSimIndex.__new__(VecSim.BFIndex)
        VecSim.VecSimIndex.__init__(ix, None, borrowed_handle=h)
        ix._parent = self
        return ix

    @property
    def local(self):
        return self.local_index()

    # ---- ingest (SPMD: call on every rank with the same arguments) ----
    def _blob(self, a):
        want = VecSim._NP[self.type]
        a = np.asarray(a)
        if a.dtype != want:
            a = a.view(np.uint16) if (a.dtype.itemsize == 2 and want == np.uint16) else a.astype(want)
        return np.ascontiguousarray(a)

    def add_vector(self, vector, label):
        v = self._blob(vector)
        return self._lib.VecSimGpu_ShardedAddVector(self._h, v.ctypes.data_as(C.c_void_p), int(label))

    def add_vectors(self, vectors, labels):
        v = self._blob(vectors)
        lab = np.ascontiguousarray(labels, dtype=np.uint64)
        n = self._lib.VecSimGpu_ShardedAddVectorsBulk(self._h, v.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p),
                                                      lab.size)
        if n < 0:
            raise RuntimeError("sharded bulk add failed: %s" % self._lib.VecSimGpu_LastError().decode())
        return n

    def add_synthetic_local(self, rows_per_shard, seed):
        """weak-scaling fill: every shard s appends rows_per_shard device-generated rows (seed + 1000 s); the
        equivalent single index is the concatenation of the shards, label = gid.  Append-only afterwards."""
        n = self._lib.VecSimGpu_ShardedAddSyntheticLocal(self._h, rows_per_shard, seed)
        if n < 0:
            raise RuntimeError("synthetic fill failed: %s" % self._lib.VecSimGpu_LastError().decode())
        return n

    def delete_vector(self, label):
        return self._lib.VecSimGpu_ShardedDeleteVector(self._h, int(label))

    def index_size(self):
        return self._lib.VecSimGpu_ShardedIndexSize(self._h)

    def device_sync(self):
        pass  # every C-API call returns with its streams drained

    # ---- query ----
    def knn_query(self, queries, k, order=VecSim.BY_SCORE, seq=None, query_param=None, with_codes=False):
        """seq: the batch's position in the stream of batches every process answers (several reader threads per process:
        VecSimGpu_ShardedTopKQueryBatchArraysSeq); None = one reader, call order is the order"""
        q = self._blob(queries)
        q = q.reshape(-1, q.shape[-1])
        stride = q.strides[0]
        if stride < self._qbytes:  # int8/uint8 Cosine: room for the appended norm
            buf = np.zeros((q.shape[0], self._qbytes), dtype=np.uint8)
            buf[:, :stride] = q.view(np.uint8).reshape(q.shape[0], stride)
            q, stride = buf, self._qbytes
        nq = q.shape[0]
        labels = np.empty((nq, k), dtype=np.int64)
        dists = np.empty((nq, k), dtype=np.float64)
        codes = np.zeros(nq, dtype=np.int32)
        qp = C.byref(query_param) if query_param is not None else None
        rc = self._lib.VecSimGpu_ShardedTopKQueryBatchArraysSeq(self._h, q.ctypes.data_as(C.c_void_p), nq, stride, k, qp,
                                                               order, labels.ctypes.data_as(C.c_void_p),
                                                               dists.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p),
                                                               0xFFFFFFFFFFFFFFFF if seq is None else int(seq))
        if rc != 0:
            raise RuntimeError("sharded top-k failed: %s" % self._lib.VecSimGpu_LastError().decode())
        return (labels, dists, codes) if with_codes else (labels, dists)

    def stats(self):
        """wall ms per phase since reset_stats(): shard scans, waiting for the exchange turn, exchange, merge + replies"""
        out = (C.c_double * 6)()
        self._lib.VecSimGpu_ShardedGetStats(self._h, out)
        return {"scan_ms": out[0], "turn_wait_ms": out[1], "exchange_ms": out[2], "merge_ms": out[3], "batches": int(out[4]),
                "exchange_bytes": out[5]}

    def reset_stats(self):
        self._lib.VecSimGpu_ShardedResetStats(self._h)

    def abort(self):
        """give up on the other processes (VecSimGpu_ShardedAbort): exchanges in flight fail, later queries are refused"""
        self._lib.VecSimGpu_ShardedAbort(self._h)

    def exchange_self_test(self, nbytes=81920):
        """SPMD: True on every process iff the exchange moves rank-stamped bytes correctly between all of them"""
        return self._lib.VecSimGpu_ShardedExchangeSelfTest(self._h, nbytes) == 0

    def exchange_mode(self):
        return self._lib.VecSimGpu_ShardedExchangeMode(self._h).decode()

    def reset_seq(self):
        """start a new stream of numbered batches at 0 (every process, nothing in flight)"""
        self._lib.VecSimGpu_ShardedResetSeq(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.VecSimGpu_ShardedFree(self._h)
            self._h = None


def merge_topk(counts, gids, labels, scores, k):
    """counts [G,nq], gids/labels/scores [G,nq,cap] -> (labels int64 [nq,k], scores f64 [nq,k])"""
    lib = _capi.load()
    parts, nq, cap = gids.shape
    out_l = np.empty((nq, k), dtype=np.int64)
    out_s = np.empty((nq, k), dtype=np.float64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.VecSimGpu_MergeTopK(nq, parts, cap, p(gids), p(labels), p(scores), p(counts), k, p(out_l), p(out_s))
    if rc != 0:
        raise RuntimeError("merge saw an overflow marker")
    return out_l, out_s
