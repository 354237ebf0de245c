"""Host twin of the device synthetic-row generator (csrc/exact_kernels.hpp:k_fill_uniform_f32).

value(seed, idx) is a pure function (splitmix64-style finaliser of seed + idx*golden, top 24 bits
mapped to U[-1,1) exactly), so benchmarks can make query batches on the host and tests can
re-create any row of a device-filled index without copying it back.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def hash32(seed, idx):
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + idx * _GOLD
        x ^= x >> np.uint64(30)
        x *= _M1
        x ^= x >> np.uint64(27)
        x *= _M2
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(32)).astype(np.uint32)


def rows_f32(seed, first_row, nrows, dim):
    idx = np.arange(first_row * dim, (first_row + nrows) * dim, dtype=np.uint64)
    u = (hash32(seed, idx) >> np.uint32(8)).astype(np.float32)
    return (u * np.float32(1.0 / 8388608.0) - np.float32(1.0)).reshape(nrows, dim)


def rows_bf16(seed, first_row, nrows, dim):
    """uint16 bf16 bits: the fp32 synthetic values rounded to nearest even"""
    u = rows_f32(seed, first_row, nrows, dim).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def rows_f16(seed, first_row, nrows, dim):
    return rows_f32(seed, first_row, nrows, dim).astype(np.float16).view(np.uint16)


def rows_i8(seed, first_row, nrows, dim):
    idx = np.arange(first_row * dim, (first_row + nrows) * dim, dtype=np.uint64)
    return (hash32(seed, idx) >> np.uint32(24)).astype(np.uint8).view(np.int8).reshape(nrows, dim)
