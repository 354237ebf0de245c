"""CPU, world_size 2 over gloo: the multi-GPU Flat path of the C++ host library (csrc/host/sharded_index.cpp:
SPMD ingest bookkeeping, block partition, candidate records, exchange, exact global merge) with

  * the exchange done by torch.distributed/gloo through the library's transport callbacks, and
  * this rank's storage + scan provided through the library's external-shard callbacks by an oracle-backed
    stand-in (tests only -- the product's shard is the GPU Flat index; there is no GPU in this tier).

Checks that the merged reply equals the single-index reference reply, ties and overwrites included."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShard:
    """External shard: rows live in numpy, distances come from the CPU oracle"""

    def __init__(self, dim):
        from oracle import vso
        self.vso, self.dim = vso, dim
        self.rows, self.labels = [], []

    def add(self, _user, blob, label):
        v = np.ctypeslib.as_array(C.cast(blob, C.POINTER(C.c_float)), shape=(self.dim,)).copy()
        if label in self.labels:   # overwrite in place
            self.rows[self.labels.index(label)] = v
            return 0
        self.rows.append(v)
        self.labels.append(int(label))
        return 1

    def candidates(self, _user, queries, nq, stride, k, cap, ids, labels, scores, counts):
        ids = np.ctypeslib.as_array(C.cast(ids, C.POINTER(C.c_uint32)), shape=(nq, cap))
        labels = np.ctypeslib.as_array(C.cast(labels, C.POINTER(C.c_uint64)), shape=(nq, cap))
        scores = np.ctypeslib.as_array(C.cast(scores, C.POINTER(C.c_double)), shape=(nq, cap))
        counts = np.ctypeslib.as_array(C.cast(counts, C.POINTER(C.c_uint32)), shape=(nq,))
        rows = np.stack(self.rows) if self.rows else np.zeros((0, self.dim), np.float32)
        for qi in range(nq):
            q = np.ctypeslib.as_array(C.cast(queries + qi * stride, C.POINTER(C.c_float)), shape=(self.dim,))
            if len(rows) == 0:
                counts[qi] = 0
                continue
            s = self.vso.scan(0, 0, rows, q.copy(), self.dim)
            kk = min(k, len(s))
            T = np.partition(s, kk - 1)[kk - 1]
            keep = np.nonzero(s <= T)[0]
            if len(keep) > cap:
                counts[qi] = 0xFFFFFFFF
                continue
            counts[qi] = len(keep)
            ids[qi, :len(keep)] = keep
            labels[qi, :len(keep)] = np.array(self.labels, dtype=np.uint64)[keep]
            scores[qi, :len(keep)] = s[keep]
        return 0


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vso
    from vectorsimilarity_amd import VecSim
    from vectorsimilarity_amd.sharded import ShardedFlatIndex

    rng = np.random.default_rng(5)           # same data on every rank (SPMD ingest)
    dim, n, nq, k, block = 16, 1000, 6, 10, 32
    base = rng.integers(-2, 3, (30, dim)).astype(np.float32)
    rows = base[rng.integers(0, 30, n)]       # duplicates => ties across shards
    labels = rng.permutation(n) + 100
    queries = base[:nq].copy()
    p = VecSim.BFParams()
    p.type, p.dim, p.metric, p.blockSize = 0, dim, 0, block
    shard = OracleShard(dim)
    ix = ShardedFlatIndex(p, rank=rank, world=world, dist=dist, external=(shard.add, shard.candidates))
    # the SPMD transport check bench.py chooses its exchange buffers by: rank-stamped bytes through the all-gather, every slice checked
    assert ix.exchange_self_test(4096) and ix.exchange_mode() == "transport"
    ix.add_vectors(rows[:700], labels[:700])
    for i in range(700, n):
        assert ix.add_vector(rows[i], labels[i]) == 1
    # overwrites: the label keeps its place in the equivalent single index, the count does not move
    for i in (3, 250, 777):
        rows[i] = base[(i * 7) % 30]
        assert ix.add_vector(rows[i], labels[i]) == 0
    assert ix.index_size() == n
    ok = True
    for kk in (k, 40):   # k = 40: hundreds of rows tie at the k-th score => the wide (overflow) retry
        got_l, got_s = ix.knn_query(queries, kk)
        for qi in range(nq):
            el, es = vso.flat_topk(0, 0, rows, queries[qi], kk, dim, labels.astype(np.uint64))
            ok &= np.array_equal(got_l[qi], el.astype(np.int64)) and np.array_equal(got_s[qi], es)
    # two reader threads per process: batch b is answered by thread b % 2 under sequence number b -- the exchanges pair up
    # across the processes whatever the threads' relative speed (the external shard's scan holds the GIL, so the threads of
    # the two processes really do run at unrelated paces)
    from concurrent.futures import ThreadPoolExecutor
    qsets = [base[(3 * b) % 24:(3 * b) % 24 + nq].copy() for b in range(8)]

    def reader(t):
        return [(b, ix.knn_query(qsets[b], k, seq=b)) for b in range(t, 8, 2)]
    ix.reset_stats()
    with ThreadPoolExecutor(2) as pool:
        for part in pool.map(reader, range(2)):
            for b, (got_l, got_s) in part:
                for qi in range(nq):
                    el, es = vso.flat_topk(0, 0, rows, qsets[b][qi], k, dim, labels.astype(np.uint64))
                    ok &= np.array_equal(got_l[qi], el.astype(np.int64)) and np.array_equal(got_s[qi], es)
    st = ix.stats()
    ok &= st["batches"] == 8 and st["exchange_bytes"] > 0
    # a sequence number that has already been answered is an error on every process (it used to wait for ever and strand the
    # peers in their collective); reset_seq starts a new stream at 0
    try:
        ix.knn_query(qsets[0], k, seq=3)
        ok = False
    except RuntimeError:
        pass
    ix.reset_seq()
    got_l, got_s = ix.knn_query(qsets[1], k, seq=0)
    for qi in range(nq):
        el, es = vso.flat_topk(0, 0, rows, qsets[1][qi], k, dim, labels.astype(np.uint64))
        ok &= np.array_equal(got_l[qi], el.astype(np.int64)) and np.array_equal(got_s[qi], es)
    # the timeout callback fires on ONE process only: its verdict travels in the exchange, both processes return TimedOut
    # replies (a process that left before the collective would hang the other)
    cb = VecSim.set_timeout_callback(lambda ctx: 1 if (ctx == 7 and rank == 1) else 0)
    qp = VecSim.VecSimQueryParams()
    qp.timeoutCtx = 7
    _, _, codes = ix.knn_query(queries, k, query_param=qp, with_codes=True)
    ok &= bool(np.all(codes == VecSim._capi.VecSim_QueryReply_TimedOut))
    qp.timeoutCtx = 8
    got_l, _, codes = ix.knn_query(queries, k, query_param=qp, with_codes=True)
    ok &= bool(np.all(codes == VecSim._capi.VecSim_QueryReply_OK)) and bool(np.all(got_l >= 0))
    VecSim.set_timeout_callback(None)
    del cb
    # an external (append-only) shard cannot move rows: delete is refused on every process alike, nothing changes
    ok &= ix.delete_vector(int(labels[5])) == -1 and ix.index_size() == n
    got_l, got_s = ix.knn_query(queries, k)
    for qi in range(nq):
        el, es = vso.flat_topk(0, 0, rows, queries[qi], k, dim, labels.astype(np.uint64))
        ok &= np.array_equal(got_l[qi], el.astype(np.int64)) and np.array_equal(got_s[qi], es)
    owned = len(shard.rows)
    # the RCCL form's pre-flight: no GPU here, so every rank's device is out of range -- ALL ranks must fail, at once and with the
    # reason, before anyone enters the communicator's rendezvous (one rank failing alone used to leave the others blocked in it)
    import time
    t0 = time.time()
    try:
        ShardedFlatIndex(p, rank=rank, world=world, dist=dist, device=rank)
        ok = False
    except RuntimeError as e:
        ok &= "out of range" in str(e) and time.time() - t0 < 20
    dist.barrier()
    dist.destroy_process_group()
    out.put((rank, bool(ok), owned))


def test_sharded_flat_world2_matches_single_index():
    from oracle import vso
    vso.build()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sum(o for _, _, o in res) == 1000 and all(o > 400 for _, _, o in res)   # balanced, nothing lost


def test_block_partition_is_a_bijection():
    from vectorsimilarity_amd.sharded import block_owner, gid_to_local, local_to_gid
    block, world = 32, 3
    seen = set()
    for gid in range(5000):
        r = block_owner(gid, block, world)
        loc = gid_to_local(gid, block, world)
        assert local_to_gid(loc, r, block, world) == gid
        seen.add((r, loc))
    assert len(seen) == 5000


def test_wide_merge_equals_the_sequential_heap_loop():
    """a wide and deep batch (merged on several host threads, the no-tie shortcut for most queries) against the plain loop of
    brute_force.h:264-281 over the union in gid order, restated here: queries with distinct scores, queries full of ties at the k-th
    score, queries with fewer candidates than k"""
    from vectorsimilarity_amd.sharded import merge_topk
    rng = np.random.default_rng(77)
    nq, parts, cap, k = 224, 8, 48, 32
    gids = np.zeros((parts, nq, cap), dtype=np.uint64)
    labels = np.zeros((parts, nq, cap), dtype=np.uint64)
    scores = np.zeros((parts, nq, cap), dtype=np.float64)
    counts = np.zeros((parts, nq), dtype=np.uint32)
    for q in range(nq):
        kind = q % 4
        perm = rng.permutation(parts * cap)
        for p in range(parts):
            n = int(rng.integers(0, 4)) if kind == 3 else int(rng.integers(36, cap + 1))
            counts[p, q] = n
            g = np.sort(perm[p * cap:p * cap + n])
            gids[p, q, :n] = g
            labels[p, q, :n] = (g * 7 + 3) % 1009 + 1000 * g
            sc = rng.random(n)
            if kind == 1:
                sc = np.floor(sc * 6) / 6          # heavy ties, some across the k-th score
            elif kind == 2:
                sc = np.floor(sc * 300) / 300      # a few ties
            scores[p, q, :n] = sc
    assert int(counts.sum()) + nq * k * 8 >= 100000   # (the work estimate past which the merge runs on threads, sharded_index.cpp)
    l, s = merge_topk(counts, gids, labels, scores, k)
    for q in range(nq):
        cands = sorted((int(gids[p, q, i]), int(labels[p, q, i]), float(scores[p, q, i])) for p in range(parts) for i in range(counts[p, q]))
        heap = []   # (score, label), the largest evicted
        for _, lab, sc in cands:
            if len(heap) < k or sc < max(heap)[0]:
                heap.append((sc, lab))
                if len(heap) > k:
                    heap.remove(max(heap))
        heap.sort()
        got = [(float(s[q, i]), int(l[q, i])) for i in range(len(heap))]
        assert got == heap, q
        assert all(int(x) == -1 for x in l[q, len(heap):])


def test_merge_replays_ties_like_the_reference():
    """VecSimGpu_MergeTopK on hand-made partials == sequential heap over the union in gid order"""
    from vectorsimilarity_amd.sharded import merge_topk
    # the SURVEY tie probe split over two shards: scan order (label,dist) = (5,16),(9,16),(1,4),(2,16),(0,16)
    gids = np.zeros((2, 1, 4), dtype=np.uint64)
    labels = np.zeros((2, 1, 4), dtype=np.uint64)
    scores = np.zeros((2, 1, 4), dtype=np.float64)
    counts = np.array([[3], [2]], dtype=np.uint32)
    gids[0, 0, :3], labels[0, 0, :3], scores[0, 0, :3] = [0, 2, 4], [5, 1, 0], [16, 4, 16]
    gids[1, 0, :2], labels[1, 0, :2], scores[1, 0, :2] = [1, 3], [9, 2], [16, 16]
    for k, exp in ((2, [1, 5]), (3, [1, 5, 9])):
        l, s = merge_topk(counts, gids, labels, scores, k)
        assert list(l[0]) == exp
