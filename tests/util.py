"""shared helpers for the test-suite (data encoders, oracle adapters)"""
import numpy as np

TYPES = {"f32": 0, "f64": 1, "bf16": 2, "f16": 3, "i8": 4, "u8": 5}
METRICS = {"L2": 0, "IP": 1, "Cosine": 2}
TIERS = {"avx512": 0, "scalar": 1, "avx512_bf16": 2, "avx512_fp16": 3}


def encode(vso, values, typ):
    """python floats/ints -> raw numpy array of the stored element type (bf16/fp16 as uint16,
    converted with the reference's own rounding restated in the oracle)"""
    if typ == "f32":
        return np.asarray(values, dtype=np.float32)
    if typ == "f64":
        return np.asarray(values, dtype=np.float64)
    if typ == "bf16":
        return vso.f32_to_bf16(np.asarray(values, dtype=np.float32))
    if typ == "f16":
        return vso.f32_to_f16(np.asarray(values, dtype=np.float32))
    if typ == "i8":
        return np.asarray(values, dtype=np.int8)
    if typ == "u8":
        return np.asarray(values, dtype=np.uint8)
    raise ValueError(typ)


def random_vectors(rng, n, dim, typ, vso):
    if typ in ("f32", "f64"):
        return rng.uniform(-1, 1, (n, dim)).astype(np.float32 if typ == "f32" else np.float64)
    if typ in ("bf16", "f16"):
        f = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
        if typ == "bf16":
            u = f.view(np.uint32)
            return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
        return f.astype(np.float16).view(np.uint16)
    if typ == "i8":
        return rng.integers(-128, 128, (n, dim), dtype=np.int8)
    return rng.integers(0, 256, (n, dim), dtype=np.uint8)


def stored_rows(vso, rows, typ, metric):
    """what the index stores for `rows`: Cosine => normalised (fp) / norm appended (int)"""
    t, m = TYPES[typ], METRICS[metric]
    n, dim = rows.shape
    if metric != "Cosine":
        return np.ascontiguousarray(rows)
    if typ in ("i8", "u8"):
        out = np.zeros((n, dim + 4), dtype=np.uint8)
        out[:, :dim] = rows.view(np.uint8)
        for i in range(n):
            vso.normalize(out[i], dim, t)
        return out
    out = np.ascontiguousarray(rows).copy()
    for i in range(n):
        vso.normalize(out[i], dim, t)
    return out
