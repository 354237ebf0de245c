"""ctypes loader for oracle/_ref/libvsref.so -- the REFERENCE's own scalar translation units compiled by
oracle/build_ref.sh (plain g++ on the files under /root/reference, no stand-ins) behind oracle/ref_driver.cpp.

TEST INFRASTRUCTURE ONLY, like everything under oracle/.  Used by tests/golden/make_ref_scalar_random.py (to make
the committed fixture) and by tests/test_ref_fixture.py (live comparison wherever the library exists).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_ref", "libvsref.so")

_lib = None


def build():
    """runs the recipe; a no-op where /root/reference is absent (the GPU box)"""
    subprocess.run(["sh", os.path.join(_HERE, "build_ref.sh")], check=True)
    return _LIB


def available():
    return os.path.exists(_LIB)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_LIB)
        vp, sz, dbl, i = C.c_void_p, C.c_size_t, C.c_double, C.c_int
        L.vsref_distance.restype = dbl
        L.vsref_distance.argtypes = [i, i, sz, vp, vp]
        L.vsref_normalize.restype = None
        L.vsref_normalize.argtypes = [vp, sz, i]
        for name in ("vsref_f32_to_bf16_n", "vsref_f32_to_f16_n", "vsref_bf16_to_f32_n", "vsref_f16_to_f32_n"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [vp, sz, vp]
        for name in ("vsref_sq8_fp32_distance", "vsref_sq8_fp16_distance", "vsref_sq8_sq8_distance"):
            getattr(L, name).restype = dbl
            getattr(L, name).argtypes = [i, sz, vp, vp]
        L.vsref_topk.restype = sz
        L.vsref_topk.argtypes = [vp, vp, sz, sz, i, i, vp, vp]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def distance(vtype, metric, a, b, dim):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return lib().vsref_distance(vtype, metric, dim, _ptr(a), _ptr(b))


def normalize(blob, dim, vtype):
    assert blob.flags["C_CONTIGUOUS"]
    lib().vsref_normalize(_ptr(blob), dim, vtype)
    return blob


def _conv(name, a, src, dst):
    a = np.ascontiguousarray(a, dtype=src)
    out = np.empty(a.shape, dtype=dst)
    getattr(lib(), name)(_ptr(a), a.size, _ptr(out))
    return out


def f32_to_bf16(a):
    return _conv("vsref_f32_to_bf16_n", a, np.float32, np.uint16)


def f32_to_f16(a):
    return _conv("vsref_f32_to_f16_n", a, np.float32, np.uint16)


def bf16_to_f32(a):
    return _conv("vsref_bf16_to_f32_n", a, np.uint16, np.float32)


def f16_to_f32(a):
    return _conv("vsref_f16_to_f32_n", a, np.uint16, np.float32)


def sq8_distance(kind, metric, storage, query, dim):
    """kind: 'fp32' | 'fp16' (asymmetric: storage blob vs query blob) | 'sq8' (two storage blobs)"""
    storage = np.ascontiguousarray(storage)
    query = np.ascontiguousarray(query)
    return getattr(lib(), "vsref_sq8_%s_distance" % kind)(metric, dim, _ptr(storage), _ptr(query))


def topk(scores, k, labels=None, multi=False, wide=False):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint64)
    cap = max(1, min(k, scores.size))
    ol = np.empty(cap, dtype=np.uint64)
    osc = np.empty(cap, dtype=np.float64)
    c = lib().vsref_topk(_ptr(scores), None if lab is None else _ptr(lab), scores.size, k, int(multi), int(wide),
                         _ptr(ol), _ptr(osc))
    return ol[:c].copy(), osc[:c].copy()
