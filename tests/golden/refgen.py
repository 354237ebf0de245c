"""Deterministic inputs of tests/golden/ref_scalar_random.json.

The fixture stores SEEDS, not inputs; this module turns a seed into the input bytes, identically in the script that
made the fixture (make_ref_scalar_random.py, which ran the reference's own compiled code on them) and in the tests
that replay it (tests/test_ref_fixture.py on the oracle, tests/test_gpu_ref_fixture.py on the HIP path).  The
generator is splitmix64 over a counter, written out here so that no library version can change the stream.
"""
import hashlib
import math

import numpy as np

TYPES = ("f32", "f64", "bf16", "f16", "i8", "u8")
METRICS = ("L2", "IP", "Cosine")
DIMS = tuple(range(1, 41)) + (127, 768, 1024)
ROWS = 4          # rows per (type, metric, dim) case; one query


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def u64(seed, count):
    """`count` 64-bit words of stream `seed`"""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        return _mix(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + idx * np.uint64(0x9E3779B97F4A7C15))


def case_seed(*parts):
    h = hashlib.sha256(("|".join(str(p) for p in parts)).encode()).digest()
    return int.from_bytes(h[:8], "little") >> 1


def uniform_f64(seed, count):
    """U[-1, 1) doubles with 53 random mantissa bits"""
    return (u64(seed, count) >> np.uint64(11)).astype(np.float64) * (2.0 ** -52) - 1.0


def f32_to_bf16_rne(f):
    """round-to-nearest-even bf16 bits of finite floats (IEEE operation; the reference's own conversion is pinned
    separately by the `conversions` section of the fixture)"""
    u = np.asarray(f, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def vectors(seed, count, dim, typ):
    """`count` vectors of `dim` elements in the stored element type (bf16 / fp16 as uint16 bit patterns)"""
    n = count * dim
    if typ == "f64":
        return uniform_f64(seed, n).reshape(count, dim)
    if typ == "f32":
        return uniform_f64(seed, n).astype(np.float32).reshape(count, dim)
    if typ == "bf16":
        return f32_to_bf16_rne(uniform_f64(seed, n).astype(np.float32)).reshape(count, dim)
    if typ == "f16":
        return uniform_f64(seed, n).astype(np.float32).astype(np.float16).view(np.uint16).reshape(count, dim)
    b = (u64(seed, n) >> np.uint64(56)).astype(np.uint8)
    return (b.view(np.int8) if typ == "i8" else b).reshape(count, dim)


def wide_floats(seed, count):
    """fp32 bit patterns over the whole exponent range, with specials, ties-to-even cases and subnormals mixed in"""
    bits = (u64(seed, count) >> np.uint64(32)).astype(np.uint32)
    special = np.array([0x00000000, 0x80000000, 0x7F800000, 0xFF800000, 0x7FC00000, 0x7F800001, 0x00000001, 0x007FFFFF,
                        0x00800000, 0x33000000, 0x33800000, 0x38000000, 0x387FC000, 0x387FE000, 0x38800000, 0x477FE000,
                        0x477FF000, 0x47800000, 0x3F808000, 0x3F818000, 0x3F80FFFF, 0x3F807FFF, 0x7F7FFFFF, 0x3F801000,
                        0x3F803000, 0x3F802000, 0x38801000, 0x33000001, 0x32FFFFFF, 0xC77FF000], dtype=np.uint32)
    bits[:len(special)] = special
    # a block of exact halfway cases for fp16 (13 dropped bits = 0x1000) and bf16 (16 dropped bits = 0x8000)
    k = len(special)
    m = (count - k) // 4
    bits[k:k + m] = (bits[k:k + m] & np.uint32(0xFFFFE000)) | np.uint32(0x1000)
    bits[k + m:k + 2 * m] = (bits[k + m:k + 2 * m] & np.uint32(0xFFFF0000)) | np.uint32(0x8000)
    return bits.view(np.float32)


def sq8_storage(seed, count, dim, metric, centred_ip_slot=False):
    """`count` SQ8 storage blobs (types/sq8.h:19-62): dim codes + {min, delta, sum[, sum_squares]} as FP32.  The
    metadata are random plausible values -- the kernels are pure functions of the blob bytes and never check
    that the metadata describe the codes."""
    nmeta = 4 if metric == "L2" else 3
    codes = (u64(seed, count * dim) >> np.uint64(56)).astype(np.uint8).reshape(count, dim)
    r = uniform_f64(seed + 1, count * 4).reshape(count, 4)
    meta = np.empty((count, nmeta), dtype=np.float32)
    meta[:, 0] = (r[:, 0] * 0.9).astype(np.float32)                      # min
    meta[:, 1] = ((r[:, 1] + 1.5) / 255.0).astype(np.float32)            # delta
    meta[:, 2] = (r[:, 2] * dim * 0.5).astype(np.float32)                # sum
    if nmeta == 4:
        meta[:, 3] = ((r[:, 3] + 1.0) * dim * 0.33).astype(np.float32)   # sum of squares
    out = np.empty((count, dim + 4 * nmeta), dtype=np.uint8)
    out[:, :dim] = codes
    out[:, dim:] = meta.view(np.uint8).reshape(count, 4 * nmeta)
    return out


def sq8_query_f32(seed, dim, metric):
    """FP32 query blob: dim values + y_sum [+ y_sum_squares] (sums accumulated in float as a plain loop would)"""
    y = uniform_f64(seed, dim).astype(np.float32)
    w = y.astype(np.float64)
    extra = [np.float32(math.fsum(w))]                 # fsum: exactly rounded, no dependence on a library's blocking
    if metric == "L2":
        extra.append(np.float32(math.fsum(w * w)))
    return np.concatenate([y, np.array(extra, dtype=np.float32)])


def sq8_query_f16(seed, dim, metric):
    """FP16 query blob: dim fp16 values, then FP32 metadata (unaligned when dim is odd) -- as bytes"""
    y = uniform_f64(seed, dim).astype(np.float32).astype(np.float16)
    w = y.astype(np.float64)
    extra = [np.float32(math.fsum(w))]
    if metric == "L2":
        extra.append(np.float32(math.fsum(w * w)))
    return np.concatenate([y.view(np.uint8), np.array(extra, dtype=np.float32).view(np.uint8)])


def score_stream(seed, n, levels, nan_every=0):
    """`n` fp32-representable scores drawn from `levels` distinct values (heavy ties), optional NaNs"""
    v = (u64(seed, n) % np.uint64(levels)).astype(np.float64) * 0.125
    if nan_every:
        v[(u64(seed + 7, n) % np.uint64(nan_every)) == 0] = np.nan
    return v


def labels_unique(seed, n):
    """a permutation of n distinct labels (spread out, so label order != id order)"""
    key = u64(seed + 3, n)
    return (np.argsort(key, kind="stable").astype(np.uint64) * np.uint64(7) + np.uint64(11))


def labels_multi(seed, n, distinct):
    return (u64(seed + 5, n) % np.uint64(distinct)).astype(np.uint64)


def hexbits(x):
    """IEEE bits of a float32 / float64 numpy scalar or array as hex strings"""
    a = np.atleast_1d(x)
    if a.dtype == np.float32:
        return ["%08x" % v for v in a.view(np.uint32)]
    return ["%016x" % v for v in a.astype(np.float64).view(np.uint64)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---------------------------------------------------------------------------------------------------------------
# The fixture as a function of a BACKEND.  make_ref_scalar_random.py evaluates it on the reference's compiled code
# (oracle/vsref.py) and commits the result; the tests evaluate it on the oracle's scalar tier and compare the
# dictionaries.  A backend offers:
#   distance(type_id, kernel_metric_id, a, b, dim) -> float          normalize(blob, dim, type_id) in place
#   f32_to_bf16 / f32_to_f16 / bf16_to_f32 / f16_to_f32 (arrays)     sq8_distance(kind, metric_id, storage, query, dim)
#   topk(scores, k, labels, multi, wide) -> (labels, scores)
TYPE_ID = {t: i for i, t in enumerate(TYPES)}
METRIC_ID = {m: i for i, m in enumerate(METRICS)}
NP_OF = {"f32": np.float32, "f64": np.float64, "bf16": np.uint16, "f16": np.uint16, "i8": np.int8, "u8": np.uint8}


def kernel_metric(typ, metric):
    """fp Cosine runs the IP kernel on normalised blobs; int8 / uint8 have a Cosine kernel (spaces.cpp:24-148)"""
    return "IP" if (metric == "Cosine" and typ not in ("i8", "u8")) else metric


def distance_inputs(typ, metric, dim):
    """raw (un-normalised) rows [ROWS, dim] and one query [dim] of the case"""
    seed = case_seed("dist", typ, metric, dim)
    v = vectors(seed, ROWS + 1, dim, typ)
    return v[:ROWS], v[ROWS]


def stored_form(backend, vecs, typ, metric):
    """what the index stores / what a query becomes: Cosine -> normalised (fp) or norm appended (int8 / uint8)"""
    vecs = np.ascontiguousarray(vecs)
    if metric != "Cosine":
        return vecs
    n, dim = vecs.shape
    if typ in ("i8", "u8"):
        out = np.zeros((n, dim + 4), dtype=np.uint8)
        out[:, :dim] = vecs.view(np.uint8)
    else:
        out = vecs.copy()
    for i in range(n):
        backend.normalize(out[i], dim, TYPE_ID[typ])
    return out


def _score_hex(typ, values):
    a = np.asarray(values, dtype=np.float64)
    return hexbits(a if typ == "f64" else a.astype(np.float32))


def compute_distances(backend, dims=DIMS):
    out = []
    for typ in TYPES:
        for metric in METRICS:
            for dim in dims:
                rows, q = distance_inputs(typ, metric, dim)
                st = stored_form(backend, rows, typ, metric)
                qq = stored_form(backend, q[None, :], typ, metric)[0]
                km = METRIC_ID[kernel_metric(typ, metric)]
                sc = [backend.distance(TYPE_ID[typ], km, st[i], qq, dim) for i in range(ROWS)]
                e = {"type": typ, "metric": metric, "dim": dim, "scores": _score_hex(typ, sc)}
                if metric == "Cosine":
                    e["stored_sha"] = sha(np.concatenate([st.reshape(-1).view(np.uint8), qq.reshape(-1).view(np.uint8)]))
                out.append(e)
    return out


CONV_COUNT = 8192


def compute_conversions(backend):
    f = wide_floats(case_seed("conv"), CONV_COUNT)
    allh = np.arange(65536, dtype=np.uint16)
    b = backend.f32_to_bf16(f)
    h = backend.f32_to_f16(f)
    return {"count": CONV_COUNT,
            "f32_to_bf16_head": ["%04x" % v for v in b[:64]], "f32_to_bf16_sha": sha(b),
            "f32_to_f16_head": ["%04x" % v for v in h[:64]], "f32_to_f16_sha": sha(h),
            "bf16_to_f32_all_sha": sha(backend.bf16_to_f32(allh).view(np.uint32)),
            "f16_to_f32_all_sha": sha(backend.f16_to_f32(allh).view(np.uint32))}


NORM_DIMS = (1, 2, 3, 7, 16, 33, 128, 768)


def compute_normalize(backend):
    out = []
    for typ in TYPES:
        for dim in NORM_DIMS:
            v = vectors(case_seed("norm", typ, dim), 2, dim, typ)
            if typ in ("f32", "f64"):
                v = v * NP_OF[typ](3.75)        # norms away from 1
            st = stored_form(backend, v, typ, "Cosine")
            e = {"type": typ, "dim": dim, "sha": sha(st.view(np.uint8))}
            if dim <= 7:
                e["bytes"] = st.view(np.uint8).tobytes().hex()
            out.append(e)
    return out


SQ8_DIMS = (1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 768, 1024)


def compute_sq8(backend, dims=SQ8_DIMS):
    out = []
    for kind in ("fp32", "fp16", "sq8"):
        for metric in METRICS:
            for dim in dims:
                seed = case_seed("sq8", kind, metric, dim)
                # symmetric SQ8-SQ8 blobs carry {min, delta, sum} for IP / Cosine and + sum_squares for L2 as well
                st = sq8_storage(seed, ROWS + 1, dim, metric)
                if kind == "fp32":
                    q = sq8_query_f32(seed + 11, dim, metric)
                elif kind == "fp16":
                    q = sq8_query_f16(seed + 11, dim, metric)
                else:
                    q = st[ROWS]
                sc = [backend.sq8_distance(kind, METRIC_ID[metric], st[i], q, dim) for i in range(ROWS)]
                out.append({"kind": kind, "metric": metric, "dim": dim, "scores": _score_hex("f32", sc)})
    return out


TOPK_CASES = [  # (n, k, levels, multi, wide, nan_every, distinct labels for multi)
    (1, 1, 1, 0, 0, 0, 0), (5, 10, 2, 0, 0, 0, 0), (40, 3, 2, 0, 0, 0, 0), (200, 10, 5, 0, 0, 0, 0), (200, 10, 5, 0, 1, 0, 0),
    (1000, 1, 3, 0, 0, 0, 0), (1000, 50, 7, 0, 0, 0, 0), (2000, 100, 1000, 0, 1, 0, 0), (2000, 100, 11, 0, 0, 0, 0),
    (500, 20, 4, 0, 0, 9, 0), (500, 20, 4, 0, 1, 3, 0), (64, 64, 2, 0, 0, 5, 0), (300, 7, 1, 0, 0, 0, 0),
    (40, 3, 2, 1, 0, 0, 9), (200, 10, 5, 1, 0, 0, 30), (1000, 50, 7, 1, 0, 0, 120), (2000, 100, 1000, 1, 1, 0, 400),
    (2000, 10, 3, 1, 0, 0, 25), (300, 7, 1, 1, 0, 0, 50), (500, 600, 6, 1, 0, 0, 40),
]


def topk_inputs(case):
    n, k, levels, multi, wide, nan_every, distinct = case
    seed = case_seed("topk", *case)
    scores = score_stream(seed, n, levels, nan_every)
    labels = labels_multi(seed, n, distinct) if multi else labels_unique(seed, n)
    return scores, labels


def compute_topk(backend):
    out = []
    for case in TOPK_CASES:
        n, k, levels, multi, wide, nan_every, distinct = case
        scores, labels = topk_inputs(case)
        ol, osc = backend.topk(scores, k, labels, bool(multi), bool(wide))
        out.append({"case": list(case), "labels": [int(x) for x in ol], "scores": hexbits(np.asarray(osc, dtype=np.float64))})
    return out


# uint8 rows beyond 33 025 elements: the choosers return the scalar kernel there (spaces.h:57-66, L2_space.cpp:474-476,
# IP_space.cpp:788,843), whose 64-bit total is exact where a 32-bit one overflows.  "max": every element 255 against 255
# (IP, Cosine) or against 0 (L2) -- the worst case the bound is derived from; "rand": seeded bytes.
WIDE_U8 = [(33025, "max"), (33026, "max"), (33026, "rand"), (40000, "max"), (50001, "rand")]


def wide_u8_inputs(dim, kind, metric):
    if kind == "max":
        a = np.full((1, dim), 255, dtype=np.uint8)
        b = np.full(dim, 0 if metric == "L2" else 255, dtype=np.uint8)
        return a, b
    v = vectors(case_seed("wide_u8", dim, metric), 2, dim, "u8")
    return v[:1], v[1]


def compute_wide_u8(backend):
    out = []
    for dim, kind in WIDE_U8:
        for metric in METRICS:
            a, b = wide_u8_inputs(dim, kind, metric)
            st = stored_form(backend, a, "u8", metric)
            qq = stored_form(backend, b[None, :], "u8", metric)[0]
            sc = backend.distance(TYPE_ID["u8"], METRIC_ID[metric], st[0], qq, dim)
            out.append({"dim": dim, "kind": kind, "metric": metric, "score": _score_hex("f32", [sc])[0]})
    return out


def compute_all(backend):
    return {"distances": compute_distances(backend), "conversions": compute_conversions(backend),
            "normalize": compute_normalize(backend), "sq8": compute_sq8(backend), "topk": compute_topk(backend),
            "wide_u8": compute_wide_u8(backend)}
