"""Multi-GPU Flat index: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests), vector blocks dealt round-robin to the ranks, one exchange step
per query batch.

Partition: vector number i of the equivalent single index (its internal id, `gid`)
lives in block b = i // blockSize; block b belongs to rank b % G; its local id is
(b // G) * blockSize + i % blockSize.  Every rank sees every add call (SPMD) and keeps only its own
blocks, so ingest needs no communication.

Query: every rank computes, on its own GPU, every local row with score <= T_local (the k-th
smallest local score) -- VecSimIndex_TopKCandidatesBatch -- packs them into a fixed-size int64
record [nq, 1 + 3*cap] and ONE all_gather_into_tensor moves the G records to every rank (tens of
KB: latency-bound, no bandwidth tuning needed).  The merge replays the reference's sequential heap
over the union in gid order (VecSimGpu_MergeTopK), which reproduces the single-index reply exactly,
ties included.  No other collective touches the data path.
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


class ShardedFlatIndex:
    def __init__(self, params, rank=0, world=1, dist=None, local_index=None, gather_device=None):
        self.rank, self.world, self.dist = rank, world, dist
        self.block = params.blockSize or 1024
        self.dim, self.type, self.metric = params.dim, params.type, params.metric
        self.local = local_index if local_index is not None else VecSim.BFIndex(params)
        self.n_global = 0            # vectors added so far across all ranks (same on every rank)
        self._label_gid = {}         # label -> gid of vectors this rank owns (for locate/delete)
        self._synthetic = None       # (rows_per_rank,) when filled by add_synthetic_local
        self._gather_device = gather_device
        self._lib = None

    # ---- ingest (SPMD: call on every rank with the same arguments) ----
    def add_vector(self, vector, label):
        gid = self.n_global
        self.n_global += 1
        if block_owner(gid, self.block, self.world) == self.rank:
            self._label_gid[int(label)] = gid
            return self.local.add_vector(vector, label)
        return 1

    def add_vectors(self, vectors, labels):
        vectors = np.asarray(vectors)
        labels = np.asarray(labels)
        gids = np.arange(self.n_global, self.n_global + len(labels))
        mine = ((gids // self.block) % self.world) == self.rank
        self.n_global += len(labels)
        if mine.any():
            self.local.add_vectors(np.ascontiguousarray(vectors[mine]), labels[mine])
            for lab, g in zip(labels[mine], gids[mine]):
                self._label_gid[int(lab)] = int(g)
        return len(labels)

    def add_synthetic_local(self, rows_per_rank, seed):
        """weak-scaling fill: THIS rank appends rows_per_rank device-generated rows.  The equivalent
        single index is the concatenation of the shards in rank order: gid = rank*rows_per_rank + local
        id, label = gid."""
        assert self.n_global == 0
        self.local.add_synthetic(rows_per_rank, seed)
        self._synthetic = rows_per_rank
        self.n_global = rows_per_rank * self.world

    def locate(self, label):
        """(owner rank, local row) of a label; synthetic fills use label == gid"""
        if self._synthetic is not None:
            return label // self._synthetic, label % self._synthetic
        gid = self._label_gid[label]
        return block_owner(gid, self.block, self.world), gid_to_local(gid, self.block, self.world)

    def index_size(self):
        return self.n_global

    def device_sync(self):
        pass  # every C-API call returns with its stream drained

    # ---- query ----
    def _candidates(self, queries, k, cap):
        """local candidate record: int64 [nq, 1 + 3*cap] = count | gids | labels | score bits"""
        q = np.ascontiguousarray(queries)
        nq = q.shape[0]
        ids = np.zeros((nq, cap), dtype=np.uint32)
        labels = np.zeros((nq, cap), dtype=np.uint64)
        scores = np.zeros((nq, cap), dtype=np.float64)
        counts = np.zeros(nq, dtype=np.uint32)
        self.local.topk_candidates(q, k, cap, ids, labels, scores, counts)
        rec = np.zeros((nq, 1 + 3 * cap), dtype=np.int64)
        rec[:, 0] = counts
        lid = ids.astype(np.int64)
        if self._synthetic is not None:
            gids = self.rank * self._synthetic + lid    # contiguous shards
            rec[:, 1 + cap:1 + 2 * cap] = gids          # label := gid (globally unique)
        else:
            gids = ((lid // self.block) * self.world + self.rank) * self.block + lid % self.block
            rec[:, 1 + cap:1 + 2 * cap] = labels.view(np.int64)
        rec[:, 1:1 + cap] = gids
        rec[:, 1 + 2 * cap:] = scores.view(np.int64)
        return rec

    def _all_gather(self, rec):
        import torch
        dev = self._gather_device
        if dev is None:
            dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        t = torch.from_numpy(rec).to(dev)
        # concatenated form (world*nq, W): accepted by both RCCL and gloo
        out = torch.empty((self.world * rec.shape[0], rec.shape[1]), dtype=torch.int64, device=dev)
        self.dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().reshape((self.world,) + tuple(rec.shape))

    def knn_query(self, queries, k):
        queries = np.ascontiguousarray(queries)
        if queries.ndim == 1:
            queries = queries[None, :]
        if self.dist is None or (self.world == 1 and not os.environ.get("VECSIM_FORCE_GATHER")):
            return self.local.knn_query(queries, k)
        nq = queries.shape[0]
        cap = max(2 * k, k + 16)
        rec = self._candidates(queries, k, cap)
        allrec = self._all_gather(rec)                      # [G, nq, 1+3cap]
        counts = np.ascontiguousarray(allrec[:, :, 0]).astype(np.uint32)
        gids = np.ascontiguousarray(allrec[:, :, 1:1 + cap]).view(np.uint64)
        labels = np.ascontiguousarray(allrec[:, :, 1 + cap:1 + 2 * cap]).view(np.uint64)
        scores = np.ascontiguousarray(allrec[:, :, 1 + 2 * cap:]).view(np.float64)
        if (counts == OVERFLOW).any():
            # more than `cap` rows tie at some shard's k-th score: redo with room for every tie
            return self._knn_query_wide(queries, k)
        return merge_topk(counts, gids, labels, scores, k)

    def _knn_query_wide(self, queries, k):
        cap = 4 * max(2 * k, k + 16)
        while True:
            rec = self._candidates(queries, k, cap)
            allrec = self._all_gather(rec)
            counts = np.ascontiguousarray(allrec[:, :, 0]).astype(np.uint32)
            if not (counts == OVERFLOW).any():
                gids = np.ascontiguousarray(allrec[:, :, 1:1 + cap]).view(np.uint64)
                labels = np.ascontiguousarray(allrec[:, :, 1 + cap:1 + 2 * cap]).view(np.uint64)
                scores = np.ascontiguousarray(allrec[:, :, 1 + 2 * cap:]).view(np.float64)
                return merge_topk(counts, gids, labels, scores, k)
            cap *= 8


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
