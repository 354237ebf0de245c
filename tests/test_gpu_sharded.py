"""GPU: the shard leg of the multi-GPU Flat index on real hardware (BASELINE config 4's defining feature).

One MI355X plays G shards (VecSimGpu_ShardedNewLocal with every shard on device 0): rows are dealt block-round-robin
into G real GPU Flat indexes, every shard runs VecSimIndex_TopKCandidatesBatch's scan, the records are merged by the
host library, and the reply must equal BOTH the single-index GPU reply and the oracle -- labels, order and scores,
0 ulp, ties included.  A world-1 RCCL communicator (ncclCommInitRank + ncclAllGather through librccl inside
libvsgpu.so) and a torchrun-launched 1-rank bench prove the exchange path and the torch + libvsgpu.so combination
in one process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from util import METRICS, TYPES, random_vectors, stored_rows
from vectorsimilarity_amd import VecSim
from vectorsimilarity_amd.sharded import ShardedFlatIndex

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def params(typ, metric, dim, block):
    p = VecSim.BFParams()
    p.type, p.dim, p.metric, p.blockSize = TYPES[typ], dim, METRICS[metric], block
    return p


def kernel_metric(typ, metric):
    return METRICS["IP"] if (metric == "Cosine" and typ not in ("i8", "u8")) else METRICS[metric]


def oracle_topk(vso, typ, metric, rows, q, k, labels):
    st = stored_rows(vso, rows, typ, metric)
    qq = stored_rows(vso, q[None, :], typ, metric)[0]
    return vso.flat_topk(TYPES[typ], kernel_metric(typ, metric), st, qq, k, rows.shape[1], labels)


def check_equal(vso, sharded, single, typ, metric, rows, labels, queries, k, oracle_queries=None):
    gl, gs = sharded.knn_query(queries, k)
    sl, ss = single.knn_query(queries, k)
    assert np.array_equal(gl, sl) and np.array_equal(gs, ss)
    for qi in (range(len(queries)) if oracle_queries is None else oracle_queries):
        el, es = oracle_topk(vso, typ, metric, rows, queries[qi], k, labels.astype(np.uint64))
        assert np.array_equal(gl[qi], el.astype(np.int64)) and np.array_equal(gs[qi], es), (typ, metric, qi)


@pytest.mark.parametrize("G", [2, 3, 8])
def test_one_gpu_as_G_shards_fp32_ties(vso, G):
    """fp32 L2 on duplicated small-integer rows: equal scores straddle shard boundaries, so the merge's gid order
    decides the reply"""
    rng = np.random.default_rng(100 + G)
    dim, n, nq, k, block = 32, 6000, 9, 10, 64
    base = rng.integers(-2, 3, (40, dim)).astype(np.float32)
    rows = base[rng.integers(0, 40, n)]
    labels = rng.permutation(n) + 7
    queries = base[:nq].copy()
    sx = ShardedFlatIndex(params("f32", "L2", dim, block), shards=G)
    one = VecSim.BFIndex(params("f32", "L2", dim, block))
    sx.add_vectors(rows[:4000], labels[:4000])
    one.add_vectors(rows[:4000], labels[:4000])
    for i in range(4000, n):
        assert sx.add_vector(rows[i], labels[i]) == 1
        one.add_vector(rows[i], labels[i])
    assert sx.index_size() == n
    sizes = [sx.local_index(s).index_size() for s in range(G)]
    assert sum(sizes) == n and max(sizes) - min(sizes) <= block
    check_equal(vso, sx, one, "f32", "L2", rows, labels, queries, k)
    # k = 200: far more than cap = 2k rows tie at a shard's k-th score => the wide retry
    check_equal(vso, sx, one, "f32", "L2", rows, labels, queries[:3], 200)


def test_overwrite_and_delete_keep_single_index_order(vso):
    """overwrite keeps the row's place; delete moves the GLOBAL last row into the hole (brute_force.h:196-224), across
    shards, so ties keep resolving like the single index"""
    rng = np.random.default_rng(5)
    dim, n, G, block = 16, 1500, 3, 32
    base = rng.integers(-2, 3, (25, dim)).astype(np.float32)
    rows = base[rng.integers(0, 25, n)].copy()
    labels = np.arange(n) + 1000
    sx = ShardedFlatIndex(params("f32", "L2", dim, block), shards=G)
    one = VecSim.BFIndex(params("f32", "L2", dim, block))
    sx.add_vectors(rows, labels)
    one.add_vectors(rows, labels)
    for i in (5, 700, 1499):
        v = base[(i * 3) % 25]
        assert sx.add_vector(v, labels[i]) == 0 and one.add_vector(v, labels[i]) == 0
    assert sx.index_size() == n
    for lab in (1000 + 3, 1000 + 1499, 1000 + 64, 1000 + 777, 1000 + 1498, 1000 + 0):
        assert sx.delete_vector(lab) == 1 and one.delete_vector(lab) == 1
    assert sx.delete_vector(1000 + 3) == 0
    assert sx.index_size() == one.index_size() == n - 6
    # more rows after the deletes land where the single index puts them
    extra = base[rng.integers(0, 25, 100)]
    for j in range(100):
        assert sx.add_vector(extra[j], 50_000 + j) == 1
        one.add_vector(extra[j], 50_000 + j)
    q = base[:8].copy()
    for k in (1, 10, 60):
        gl, gs = sx.knn_query(q, k)
        sl, ss = one.knn_query(q, k)
        assert np.array_equal(gl, sl) and np.array_equal(gs, ss), k


def test_config4_shape_bf16_ip_shards(vso):
    """BASELINE config 4's shape per query batch: bf16 IP, d = 768, 128 queries, top-10 -- on the low-precision MFMA
    filter path of every shard (dense_pairs = 0 forces it at this test size)"""
    rng = np.random.default_rng(44)
    dim, n, nq, k, G = 768, 24_000, 128, 10, 8
    f = rng.standard_normal((n, dim)).astype(np.float32)
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    rows = vso.f32_to_bf16(f)
    qf = rng.standard_normal((nq, dim)).astype(np.float32)
    qf /= np.linalg.norm(qf, axis=1, keepdims=True)
    queries = vso.f32_to_bf16(qf)
    labels = np.arange(n) + 1
    sx = ShardedFlatIndex(params("bf16", "IP", dim, 1024), shards=G)
    one = VecSim.BFIndex(params("bf16", "IP", dim, 1024))
    sx.add_vectors(rows, labels)
    one.add_vectors(rows, labels)
    for s in range(G):
        sx.local_index(s).set_option("dense_pairs", 0)
        sx.local_index(s).reset_stats()
    one.set_option("dense_pairs", 0)
    check_equal(vso, sx, one, "bf16", "IP", rows, labels, queries, k, oracle_queries=range(0, nq, 9))
    for s in range(G):
        st = sx.local_index(s).stats()
        assert "lowp" in st["scan_kernel"] and st["fallbacks"] == 0, st


def test_config3_shape_int8_cosine_top100_across_eight_shards(vso):
    """BASELINE config 3's shape per query batch: int8 Cosine, d = 1024, 256 queries, top-100 -- every shard on the 32 x 32 x 32 filter,
    and the exchange at ITS size: 1.27 MB of records per shard, 800 candidates per query for the merge, which reads them where the
    collective wrote them and runs on several host threads (host/sharded_index.cpp merge_topk_strided).  int8 scores tie in droves: the
    merge's global-id order decides."""
    rng = np.random.default_rng(303)
    dim, n, nq, k, G = 1024, 40_000, 256, 100, 8
    rows = random_vectors(rng, n, dim, "i8", vso)
    queries = random_vectors(rng, nq, dim, "i8", vso)
    labels = rng.permutation(n) + 5
    sx = ShardedFlatIndex(params("i8", "Cosine", dim, 512), shards=G)
    one = VecSim.BFIndex(params("i8", "Cosine", dim, 512))
    sx.add_vectors(rows, labels)
    one.add_vectors(rows, labels)
    for s in range(G):
        sx.local_index(s).set_option("dense_pairs", 0)
        sx.local_index(s).reset_stats()
    one.set_option("dense_pairs", 0)
    check_equal(vso, sx, one, "i8", "Cosine", rows, labels, queries, k, oracle_queries=range(0, nq, 37))
    for s in range(G):
        st = sx.local_index(s).stats()
        assert "x32" in st["scan_kernel"] and st["fallbacks"] == 0, st
    # a second batch with fewer candidates than k in some shards' lists and a short last tile of queries
    check_equal(vso, sx, one, "i8", "Cosine", rows, labels, queries[:131], k, oracle_queries=(0, 130))


@pytest.mark.parametrize("typ,metric,dim", [("i8", "Cosine", 128), ("f32", "Cosine", 100), ("u8", "L2", 64), ("f16", "IP", 96)])
def test_other_types_shard_identically(vso, typ, metric, dim):
    rng = np.random.default_rng(dim)
    n, nq, k, G = 9000, 20, 25, 3
    rows = random_vectors(rng, n, dim, typ, vso)
    queries = random_vectors(rng, nq, dim, typ, vso)
    labels = rng.permutation(n)
    sx = ShardedFlatIndex(params(typ, metric, dim, 256), shards=G)
    one = VecSim.BFIndex(params(typ, metric, dim, 256))
    sx.add_vectors(rows, labels)
    one.add_vectors(rows, labels)
    check_equal(vso, sx, one, typ, metric, rows, labels, queries, k, oracle_queries=range(0, nq, 4))


def test_rccl_world1_exchange_in_process(vso):
    """one rank, real communicator: ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclBroadcast run through
    librccl inside libvsgpu.so -- the RCCL + vsgpu combination in one process, no torch involved"""
    rng = np.random.default_rng(8)
    dim, n, nq, k = 64, 20_000, 16, 10
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    queries = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    labels = np.arange(n)
    sx = ShardedFlatIndex(params("f32", "L2", dim, 1024), rank=0, world=1, device=0)   # transport = rccl
    one = VecSim.BFIndex(params("f32", "L2", dim, 1024))
    sx.add_vectors(rows, labels)
    one.add_vectors(rows, labels)
    check_equal(vso, sx, one, "f32", "L2", rows, labels, queries, k, oracle_queries=range(0, nq, 5))
    assert sx.delete_vector(17) == 1 and one.delete_vector(17) == 1
    gl, gs = sx.knn_query(queries, k)
    sl, ss = one.knn_query(queries, k)
    assert np.array_equal(gl, sl) and np.array_equal(gs, ss)


@pytest.mark.parametrize("mode", ["staged", "mapped"])
def test_rccl_exchange_modes_on_a_communicator_of_one(vso, monkeypatch, mode):
    """round-4 review: a communicator of MORE than one rank defaults to `staged` (device send / receive buffers + two copies,
    csrc/vsgpu_comm.hip), and no test had ever executed that branch.  Both forms, forced through $VECSIM_GPU_EXCHANGE on a
    communicator of one: the sharded index (all-gather per batch, two readers) equals the single index and the oracle."""
    monkeypatch.setenv("VECSIM_GPU_EXCHANGE", mode)
    rng = np.random.default_rng(80)
    dim, n, nq, k = 48, 30_000, 12, 10
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    queries = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    labels = np.arange(n) + 5
    sx = ShardedFlatIndex(params("f32", "L2", dim, 1024), rank=0, world=1, device=0)
    assert sx.exchange_mode() == "rccl-" + mode
    one = VecSim.BFIndex(params("f32", "L2", dim, 1024))
    sx.add_vectors(rows, labels)
    one.add_vectors(rows, labels)
    check_equal(vso, sx, one, "f32", "L2", rows, labels, queries, k, oracle_queries=range(0, nq, 4))
    assert sx.delete_vector(5 + 17) == 1 and one.delete_vector(5 + 17) == 1
    from concurrent.futures import ThreadPoolExecutor
    want = one.knn_query(queries, k)
    sx.reset_seq()
    with ThreadPoolExecutor(2) as pool:
        for gl, gs in pool.map(lambda b: sx.knn_query(queries, k, seq=b), range(6)):
            assert np.array_equal(gl, want[0]) and np.array_equal(gs, want[1])
    st = sx.stats()
    assert st["exchange_ms"] > 0 and st["exchange_bytes"] > 0, st


@pytest.mark.parametrize("mode", ["staged", "mapped"])
def test_comm_allgather_broadcast_abort_through_the_c_abi(monkeypatch, mode):
    """include/vsgpu.h comm group on its own (no index): ncclAllGather and ncclBroadcast in both buffer forms on a communicator of
    one -- a single-rank delete never reaches the broadcast (hole and last row share the owner), so it is exercised here --, then
    the failure path: the test hook fails the chosen collective the way an RCCL error would, the communicator is aborted
    (ncclCommAbort) and refuses further calls instead of leaving peers inside a collective."""
    import ctypes as C
    from vectorsimilarity_amd import _capi
    monkeypatch.setenv("VECSIM_GPU_EXCHANGE", mode)
    monkeypatch.setenv("VECSIM_GPU_EXCHANGE_FAIL_AT", "4")
    G = C.CDLL(_capi.GPU_LIB_PATH)
    G.vsgpu_ctx_create.restype = C.c_void_p
    G.vsgpu_ctx_create.argtypes = [C.c_int]
    G.vsgpu_ctx_destroy.argtypes = [C.c_void_p]
    G.vsgpu_comm_create.restype = C.c_void_p
    G.vsgpu_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    for f in ("vsgpu_comm_destroy", "vsgpu_comm_staged", "vsgpu_comm_abort"):
        getattr(G, f).argtypes = [C.c_void_p]
    G.vsgpu_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    G.vsgpu_comm_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    G.vsgpu_last_error.restype = C.c_char_p
    ctx = G.vsgpu_ctx_create(0)
    assert ctx
    uid = (C.c_char * 128)()
    assert G.vsgpu_comm_unique_id(uid) == 0
    comm = G.vsgpu_comm_create(ctx, 0, 1, uid)
    assert comm, G.vsgpu_last_error()
    assert G.vsgpu_comm_staged(comm) == (1 if mode == "staged" else 0)
    rng = np.random.default_rng(3)
    for nbytes in (8, 81_920, 1_300_000):          # an agreement word, config 4's record, config 3's (k = 100) record
        send = rng.integers(0, 256, nbytes, dtype=np.uint8)
        recv = np.zeros(nbytes, dtype=np.uint8)
        assert G.vsgpu_comm_allgather(comm, send.ctypes.data, nbytes, recv.ctypes.data) == 0, G.vsgpu_last_error()
        assert np.array_equal(send, recv)
    buf = rng.integers(0, 256, 3072 + 8, dtype=np.uint8)   # a moved row with its status word
    keep = buf.copy()
    assert G.vsgpu_comm_broadcast(comm, buf.ctypes.data, buf.size, 0) == 0, G.vsgpu_last_error()
    assert np.array_equal(buf, keep)
    assert G.vsgpu_comm_broadcast(comm, buf.ctypes.data, buf.size, 1) != 0      # no such root: refused before any collective
    # collective number 4 (0-based) fails: aborted, then dead
    recv = np.zeros(8, dtype=np.uint8)
    assert G.vsgpu_comm_allgather(comm, keep.ctypes.data, 8, recv.ctypes.data) != 0
    assert b"aborted" in G.vsgpu_last_error()
    assert G.vsgpu_comm_allgather(comm, keep.ctypes.data, 8, recv.ctypes.data) != 0
    assert b"aborted by an earlier failure" in G.vsgpu_last_error()
    G.vsgpu_comm_destroy(comm)
    G.vsgpu_ctx_destroy(ctx)


def test_query_after_abort_is_refused_not_hung(vso):
    """VecSimGpu_ShardedAbort: the next exchange returns an error at once"""
    rng = np.random.default_rng(9)
    dim, n = 32, 5000
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    sx = ShardedFlatIndex(params("f32", "L2", dim, 1024), rank=0, world=1, device=0)
    sx.add_vectors(rows, np.arange(n))
    sx.knn_query(rows[:4], 5)
    sx.abort()
    with pytest.raises(RuntimeError, match="abort"):
        sx.knn_query(rows[:4], 5)


def test_multi_value_delete_that_fails_half_way_can_be_retried(vso, monkeypatch):
    """round-4 advisor finding (sharded_index.cpp deleteVector, multi-value): a label's vectors go one swap-delete at a time; when
    one of them failed, the ids already removed stayed in the label's list although they then named OTHER rows (swapped into the
    holes), and a retried delete removed unrelated rows.  The third row move reports a failed read on its owner (test hook: the
    failure that leaves every shard untouched): the delete returns -1, the retry removes exactly the rest, and the three-shard
    index keeps answering like a single index that deleted the label."""
    monkeypatch.setenv("VECSIM_GPU_TEST_FAIL_REMOVE_AT", "2")
    rng = np.random.default_rng(12)
    dim, n, block, G = 16, 600, 8, 3
    rows = rng.integers(-3, 4, (n, dim)).astype(np.float32)
    labels = rng.integers(0, 40, n)                        # ~15 vectors per label
    pm = params("f32", "L2", dim, block)
    pm.multi = True
    sx = ShardedFlatIndex(pm, shards=G)
    monkeypatch.delenv("VECSIM_GPU_TEST_FAIL_REMOVE_AT")
    one = VecSim.BFIndex(pm)
    sx.add_vectors(rows, labels)
    one.add_vectors(rows, labels)
    lab = int(labels[3])
    want = int(np.sum(labels == lab))
    assert want >= 6
    assert sx.delete_vector(lab) == -1
    gone = n - sx.index_size()
    assert 2 <= gone < want                                # (2, or more when a removed row was the last one: no move, no hook)
    assert sx.delete_vector(lab) == want - gone            # the retry takes up where the failed call stopped
    assert one.delete_vector(lab) == want
    assert sx.index_size() == one.index_size() == n - want
    q = rows[:9].copy()
    for k in (10, 50):
        gl, gs = sx.knn_query(q, k)
        sl, ss = one.knn_query(q, k)
        assert np.array_equal(gl, sl) and np.array_equal(gs, ss), k
    other = int(labels[5]) if int(labels[5]) != lab else int(labels[6])
    assert sx.delete_vector(other) == one.delete_vector(other) > 0
    gl, gs = sx.knn_query(q, 10)
    sl, ss = one.knn_query(q, 10)
    assert np.array_equal(gl, sl) and np.array_equal(gs, ss)


@pytest.mark.parametrize("typ,metric,dim,k", [("f32", "L2", 64, 10), ("f32", "Cosine", 32, 5), ("bf16", "IP", 72, 10), ("i8", "Cosine", 64, 8)])
def test_nan_score_rows_and_queries_across_shards(vso, typ, metric, dim, k):
    """brute_force.h:272 lets a NaN-score row into the heap only while it fills (internal ids below k): rows that can score NaN
    among the first k gids -- on different shards --, NaN / Inf / zero queries, and a tail NaN row: the sharded reply equals
    the single index's (itself pinned on the oracle's every-row replay in tests/test_gpu_flat_parity.py), NaNs included"""
    rng = np.random.default_rng(dim + k)
    n, G, block = 3000, 3, 2          # tiny blocks: gids 0 .. k-1 spread over all three shards
    rows = random_vectors(rng, n, dim, typ, vso)
    q = random_vectors(rng, 6, dim, typ, vso)
    if metric == "Cosine":
        rows[1] = 0
        rows[4] = 0
        rows[n - 1] = 0
        q[3] = 0
    else:
        nan = {"f32": np.float32(np.nan), "bf16": np.uint16(0x7FC0)}[typ]
        inf = {"f32": np.float32(np.inf), "bf16": np.uint16(0x7F80)}[typ]
        rows[1, 3] = nan
        rows[4, 0] = nan
        rows[n - 1, 2] = nan
        q[3, 1] = nan
        q[4, 0] = inf
    labels = np.arange(n) * 3 + 1
    sx = ShardedFlatIndex(params(typ, metric, dim, block), shards=G)
    one = VecSim.BFIndex(params(typ, metric, dim, block))
    sx.add_vectors(rows, labels)
    one.add_vectors(rows, labels)
    for kk in (k, 1, 40):
        gl, gs = sx.knn_query(q, kk)
        sl, ss = one.knn_query(q, kk)
        assert np.array_equal(gl, sl) and np.array_equal(gs, ss, equal_nan=True), (typ, metric, kk)
    # NaN rows only beyond gid k (delete the head ones: the tail row moves into gid 1 first, then goes too): finite queries take
    # the ordinary candidate path again, NaN queries still the every-row one
    for lab in (labels[1], labels[4], labels[n - 1]):
        assert sx.delete_vector(int(lab)) == 1 and one.delete_vector(int(lab)) == 1
    gl, gs = sx.knn_query(q, k)
    sl, ss = one.knn_query(q, k)
    assert np.array_equal(gl, sl) and np.array_equal(gs, ss, equal_nan=True)


@pytest.mark.parametrize("typ,metric,dim,G", [("f32", "L2", 32, 3), ("bf16", "IP", 64, 2), ("i8", "Cosine", 48, 4)])
def test_multi_value_index_across_shards(vso, typ, metric, dim, G):
    """BFParams.multi over shards: a label's vectors land on different shards, every shard returns the rows that cover k local
    labels, the union goes through the reference's label-keyed heap in gid order (brute_force_multi.h:108-277) -- the reply
    equals the single multi-value index's (pinned on the oracle in tests/test_gpu_flat_parity.py), duplicated rows (ties) included"""
    rng = np.random.default_rng(dim + G)
    n, n_labels, nq, block = 4000, 700, 7, 16
    base = random_vectors(rng, 900, dim, typ, vso)
    rows = base[rng.integers(0, 900, n)]           # repeated vectors: equal scores under different labels
    labels = rng.integers(0, n_labels, n)
    q = random_vectors(rng, nq, dim, typ, vso)
    pm = params(typ, metric, dim, block)
    pm.multi = True
    sx = ShardedFlatIndex(pm, shards=G)
    one = VecSim.BFIndex(pm)
    sx.add_vectors(rows, labels)
    one.add_vectors(rows, labels)
    assert sx.index_size() == n
    for k in (10, 1, 300, 900):                    # 900 > the label count: every label comes back
        gl, gs = sx.knn_query(q, k)
        sl, ss = one.knn_query(q, k)
        assert np.array_equal(gl, sl) and np.array_equal(gs, ss), (typ, metric, k)
    # label-wise deletes (brute_force_multi.h:133-150): every vector of the label goes, one swap-delete of the equivalent single
    # index at a time -- rows cross shards on the way -- and the two indexes keep giving the same replies, ties included
    for lab in [int(labels[0]), int(labels[17]), int(labels[n - 1]), 10 ** 9]:
        want = int(np.sum(labels == lab))
        assert sx.delete_vector(lab) == one.delete_vector(lab) == want
        assert sx.index_size() == one.index_size()
        labels = labels.copy()
        labels[labels == lab] = -1
        for k in (10, 300):
            gl, gs = sx.knn_query(q, k)
            sl, ss = one.knn_query(q, k)
            assert np.array_equal(gl, sl) and np.array_equal(gs, ss), (typ, metric, lab, k)
    # and new vectors after the deletes land where the single index puts them
    extra = random_vectors(rng, 50, dim, typ, vso)
    sx.add_vectors(extra, np.arange(50) % 7)
    one.add_vectors(extra, np.arange(50) % 7)
    gl, gs = sx.knn_query(q, 10)
    sl, ss = one.knn_query(q, 10)
    assert np.array_equal(gl, sl) and np.array_equal(gs, ss)


def test_multi_value_shard_with_hundreds_of_identical_rows_under_distinct_labels(vso):
    """round-3 advisor finding: more than max(k, 64) rows of ONE shard tie at the k-th row score (one vector stored under many
    labels).  The shard's candidate call used to report overflow whatever room the caller offered, so the sharded query failed;
    it now grows its own room.  Reply == the single multi-value index's."""
    rng = np.random.default_rng(8)
    dim, G, block = 24, 2, 16
    base = rng.integers(-3, 4, (40, dim)).astype(np.float32)
    rows = np.concatenate([np.repeat(base[:1], 400, axis=0), base[rng.integers(1, 40, 1200)]])   # 400 copies of one vector
    labels = np.concatenate([np.arange(400) + 1000, rng.integers(0, 300, 1200)])
    perm = rng.permutation(len(rows))
    rows, labels = rows[perm], labels[perm]
    pm = params("f32", "L2", dim, block)
    pm.multi = True
    sx = ShardedFlatIndex(pm, shards=G)
    one = VecSim.BFIndex(pm)
    sx.add_vectors(rows, labels)
    one.add_vectors(rows, labels)
    q = np.concatenate([base[:1], base[5:8]])          # the first query IS the duplicated vector: 400 rows at score 0
    for k in (5, 10, 150):
        gl, gs = sx.knn_query(q, k)
        sl, ss = one.knn_query(q, k)
        assert np.array_equal(gl, sl) and np.array_equal(gs, ss), k


def test_concurrent_readers_with_sequence_numbers(vso):
    """two reader threads on one sharded index (VecSimGpu_ShardedTopKQueryBatchArraysSeq): scans overlap on the shards' reader
    lanes, exchanges go in sequence order; every batch's reply equals the one-reader reply.  Through a real 1-rank RCCL
    communicator, so the ordered exchange path runs."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(21)
    dim, n, nq, k = 96, 60_000, 24, 10
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    sx = ShardedFlatIndex(params("f32", "L2", dim, 1024), rank=0, world=1, device=0)
    sx.add_vectors(rows, np.arange(n))
    sx.local.set_option("dense_pairs", 0)
    qsets = [rng.uniform(-1, 1, (nq, dim)).astype(np.float32) for _ in range(12)]
    want = [sx.knn_query(qs, k) for qs in qsets]
    sx.reset_stats()

    def reader(t):
        return [(b, sx.knn_query(qsets[b], k, seq=b)) for b in range(t, 12, 2)]
    with ThreadPoolExecutor(2) as pool:
        for part in pool.map(reader, range(2)):
            for b, (gl, gs) in part:
                assert np.array_equal(gl, want[b][0]) and np.array_equal(gs, want[b][1]), b
    st = sx.stats()
    assert st["batches"] == 12 and st["exchange_ms"] > 0 and st["scan_ms"] > 0, st


def test_synthetic_weak_scaling_fill_matches_concatenation(vso):
    """bench.py's fill: shard s holds rows generated from seed + 1000 s; the equivalent single index is their
    concatenation with label = gid"""
    dim, per, G, nq, k = 128, 5000, 4, 8, 10
    sx = ShardedFlatIndex(params("f32", "L2", dim, 1024), shards=G)
    sx.add_synthetic_local(per, 47)
    assert sx.index_size() == per * G and sx.add_vector(np.zeros(dim, np.float32), 1) == -1   # append-only afterwards
    rows = np.concatenate([vso.synth_rows_f32(47 + 1000 * s, 0, per, dim) for s in range(G)])
    queries = vso.synth_rows_f32(48, 0, nq, dim)
    gl, gs = sx.knn_query(queries, k)
    for qi in range(nq):
        el, es = vso.flat_topk(0, 0, rows, queries[qi], k, dim)
        assert np.array_equal(gl[qi], el.astype(np.int64)) and np.array_equal(gs[qi], es)


def test_torchrun_one_rank_bench_uses_rccl():
    """the driver's launch line with one rank: torch.distributed rendezvous + RCCL communicator inside libvsgpu.so +
    the scan, in one process; the JSON line must come back and say the replies were sorted"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--rows", "200000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["sorted"] and out["n_gpus"] == 1 and out["config"]["exchange"].startswith("rccl"), out


class _ThreadBus:
    """an all-gather between `world` rank THREADS of one process: what MPI / RCCL do between processes, for the in-process test
    below (the callbacks have _capi.ALLGATHER_FN's shape: user, send, nbytes, recv)"""

    def __init__(self, world, timeout):
        import threading
        self.world, self.slots = world, [b""] * world
        self.barrier = threading.Barrier(world, timeout=timeout)
        self.fail_rank, self.fail_at, self.calls = -1, -1, [0] * world

    def callbacks(self, rank):
        import ctypes as C
        import threading

        def allgather(_user, send, nbytes, recv):
            try:
                n = self.calls[rank]
                self.calls[rank] += 1
                if rank == self.fail_rank and n >= self.fail_at:
                    return -1                                   # this rank's transport fails: it never shows up
                self.slots[rank] = C.string_at(send, nbytes)
                self.barrier.wait()
                out = b"".join(self.slots)
                self.barrier.wait()
                C.memmove(recv, out, len(out))
                return 0
            except threading.BrokenBarrierError:
                return -1                                       # a peer never arrived within the time limit
            except Exception:
                import traceback
                traceback.print_exc()
                return -1

        def broadcast(_user, buf, nbytes, root):
            return -1                                           # (append-only index: no delete, no broadcast)
        return allgather, broadcast


def test_eight_ranks_in_one_process_two_readers_each_then_one_rank_fails(vso):
    """Round-5 review item 8.  The DISTRIBUTED code path (VecSimGpu_ShardedNewWithTransport: one ShardedIndex per rank, turn-ordered
    exchanges, merge of 8 records on every rank) at G = 8 and config 4's per-shard shape / 10 (1.25 M x 768 bf16 IP rows per rank,
    128 queries, top-10), all eight ranks as threads of this process on one GPU, two reader threads per rank submitting numbered
    batches.  (RCCL itself refuses a second rank on the same device, and the local 8-shard index has no exchange at all, so the
    records travel through a thread barrier with the transport callbacks' exact shape; the RCCL staged / mapped buffers are
    exercised on a communicator of one above.)  Every rank's every reply must equal the local 8-shard index's -- which the
    oracle check pins for three queries -- and when one rank's transport fails, every rank comes back with an error within the
    time limit instead of hanging."""
    from concurrent.futures import ThreadPoolExecutor
    from vectorsimilarity_amd import synth
    G, per, dim, nq, k, nb = 8, 1_250_000, 768, 128, 10, 8
    p = params("bf16", "IP", dim, 1024)
    bus = _ThreadBus(G, timeout=20.0)
    ranks = [ShardedFlatIndex(p, rank=r, world=G, device=0, transport=bus.callbacks(r)) for r in range(G)]
    local = ShardedFlatIndex(p, shards=G)
    with ThreadPoolExecutor(G) as pool:
        list(pool.map(lambda ix: ix.add_synthetic_local(per, 47), ranks))
    local.add_synthetic_local(per, 47)
    assert all(ix.index_size() == G * per for ix in ranks) and local.index_size() == G * per
    qsets = [synth.rows_bf16(48 + b, 0, nq, dim) for b in range(nb)]
    want = [local.knn_query(qs, k) for qs in qsets]
    # oracle: the equivalent single index is the concatenation of the shards (gid = shard * per + local id, label = gid)
    check = [0, 77, 127]
    st = vso.StreamTopK(TYPES["bf16"], METRICS["IP"], qsets[0][check], k, dim, threads=min(64, os.cpu_count() or 1))
    buf = np.empty(per * dim * 2, dtype=np.uint8)
    for s in range(G):
        st.feed(local.local_index(s).stored_rows(0, per, out=buf), s * per)
    el, es = st.result()
    for j, qi in enumerate(check):
        assert np.array_equal(want[0][0][qi], el[j]) and np.array_equal(want[0][1][qi], es[j]), qi

    def rank_job(r):
        ix = ranks[r]

        def reader(t):
            return [(b, ix.knn_query(qsets[b], k, seq=b)) for b in range(t, nb, 2)]
        with ThreadPoolExecutor(2) as rp:
            return [x for part in rp.map(reader, range(2)) for x in part]
    with ThreadPoolExecutor(G) as pool:
        for r, replies in enumerate(pool.map(rank_job, range(G))):
            assert len(replies) == nb
            for b, (gl, gs) in replies:
                assert np.array_equal(gl, want[b][0]) and np.array_equal(gs, want[b][1]), (r, b)
    for ix in ranks:
        stt = ix.stats()
        assert stt["batches"] == nb and stt["exchange_bytes"] > 0
    # rank 5's transport dies at its next exchange: nobody hangs, everybody reports
    bus.fail_rank, bus.fail_at = 5, bus.calls[5]
    for ix in ranks:
        ix.reset_seq()

    def failing(r):
        try:
            ranks[r].knn_query(qsets[0], k, seq=0)
            return None
        except RuntimeError as e:
            return str(e)
    import time
    t0 = time.perf_counter()
    with ThreadPoolExecutor(G) as pool:
        errs = list(pool.map(failing, range(G)))
    assert all(e is not None for e in errs), errs
    assert time.perf_counter() - t0 < 60
