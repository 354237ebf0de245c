"""CPU: the drop-in boundary.  (1) our public structs have the reference's layout (fixture from the
reference header), (2) both shared libraries load and export every symbol the headers declare,
(3) without a GPU the product refuses to build an index -- no CPU fallback."""
import ctypes as C
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLD)


def test_struct_layout_matches_reference_fixture():
    from make_abi_layout import probe
    ours = json.loads(probe(os.path.join(ROOT, "include", "VecSim", "vec_sim_common.h")))
    with open(os.path.join(GOLD, "abi_layout.json")) as f:
        ref = json.load(f)
    assert ours == ref


def test_ctypes_mirror_matches_header():
    from vectorsimilarity_amd import _capi
    with open(os.path.join(GOLD, "abi_layout.json")) as f:
        ref = json.load(f)
    assert C.sizeof(_capi.BFParams) == ref["sizeof(BFParams)"]
    assert C.sizeof(_capi.HNSWParams) == ref["sizeof(HNSWParams)"]
    assert C.sizeof(_capi.SVSParams) == ref["sizeof(SVSParams)"]
    assert C.sizeof(_capi.TieredIndexParams) == ref["sizeof(TieredIndexParams)"]
    assert C.sizeof(_capi.AlgoParams) == ref["sizeof(AlgoParams)"]
    assert C.sizeof(_capi.VecSimParams) == ref["sizeof(VecSimParams)"]
    assert _capi.VecSimParams.logCtx.offset == ref["offsetof(VecSimParams,logCtx)"]
    assert C.sizeof(_capi.VecSimQueryParams) == ref["sizeof(VecSimQueryParams)"]
    assert _capi.VecSimQueryParams.timeoutCtx.offset == ref["offsetof(VecSimQueryParams,timeoutCtx)"]
    assert C.sizeof(_capi.VecSimIndexBasicInfo) == ref["sizeof(VecSimIndexBasicInfo)"]


def _declared(header):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b((?:VecSim|vsgpu)[A-Za-z_0-9]*)\s*\(", txt)) - {"VecSim_OK"}


def test_libraries_export_every_declared_symbol():
    from vectorsimilarity_amd import _capi
    L = _capi.load()
    G = C.CDLL(_capi.GPU_LIB_PATH)
    inc = os.path.join(ROOT, "include")
    declared = set()
    for h in ("VecSim/vec_sim.h", "VecSim/query_results.h", "VecSim/info_iterator.h", "VecSim/vec_sim_debug.h", "VecSim/vec_sim_gpu.h"):
        declared |= _declared(os.path.join(inc, h))
    assert declared == set(_capi.EXPORTS), declared ^ set(_capi.EXPORTS)
    for name in sorted(declared):
        assert hasattr(L, name), name
    gdecl = _declared(os.path.join(inc, "vsgpu.h"))
    assert gdecl == set(_capi.GPU_EXPORTS), gdecl ^ set(_capi.GPU_EXPORTS)
    for name in sorted(gdecl):
        assert hasattr(G, name), name


def test_no_gpu_means_no_index():
    from vectorsimilarity_amd import VecSim, _capi
    if _capi.load().VecSimGpu_DeviceCount() > 0:
        pytest.skip("a GPU is visible")
    p = VecSim.BFParams()
    p.type, p.dim, p.metric = VecSim.VecSimType_FLOAT32, 16, VecSim.VecSimMetric_L2
    with pytest.raises(RuntimeError, match="no HIP device"):
        VecSim.BFIndex(p)


def test_product_never_references_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "vectorsimilarity_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or fn == "Makefile":
                txt = open(os.path.join(base, fn)).read()
                if re.search(r"(import|from)\s+oracle|libvso|vso_[a-z]+\(|#include\s+\"vso", txt):
                    bad.append(fn)
    assert not bad, bad


def test_public_headers_compile_as_plain_c(tmp_path):
    """the drop-in boundary is a C ABI: every public header must be usable from a C translation unit"""
    import subprocess
    src = tmp_path / "use_headers.c"
    src.write_text('#include "VecSim/vec_sim.h"\n#include "VecSim/query_results.h"\n#include "VecSim/info_iterator.h"\n'
                   '#include "VecSim/vec_sim_gpu.h"\n#include "vsgpu.h"\n'
                   "int main(void) { VecSimParams p; VecSim_InfoField f; (void)p; (void)f; return (int)sizeof(VecSimQueryParams) == 0; }\n")
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_stats_struct_layout_is_pinned(tmp_path):
    """round-4 advisor finding: VecSimGpuStats / vsgpu_stats changed field order between rounds 3 and 4.  The layout is append-only
    from here: every offset and the size are pinned, in the C headers and in the ctypes mirror."""
    import subprocess
    from vectorsimilarity_amd import _capi
    fields = ["scan_ms", "scan_launches", "scan_rows", "scan_bytes", "other_ms", "candidates", "fallbacks", "scan_kernel", "retries"]
    want = {"scan_ms": 0, "scan_launches": 8, "scan_rows": 16, "scan_bytes": 24, "other_ms": 32, "candidates": 40, "fallbacks": 48,
            "scan_kernel": 56, "retries": 120, "sizeof": 128}
    src = tmp_path / "stats_layout.c"
    body = "".join('    printf("%%s %%s %%zu\\n", "%s", "%s", offsetof(%s, %s));\n' % (t, f, t, f)
                   for t in ("VecSimGpuStats", "vsgpu_stats") for f in fields)
    body += "".join('    printf("%%s sizeof %%zu\\n", "%s", sizeof(%s));\n' % (t, t) for t in ("VecSimGpuStats", "vsgpu_stats"))
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "VecSim/vec_sim_gpu.h"\n#include "vsgpu.h"\nint main(void) {\n' + body +
                   "    return 0;\n}\n")
    exe = tmp_path / "stats_layout"
    r = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {}
    for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines():
        t, f, off = line.split()
        got.setdefault(t, {})[f] = int(off)
    assert got["VecSimGpuStats"] == want and got["vsgpu_stats"] == want, got
    for f in fields:
        assert getattr(_capi.VecSimGpuStats, f).offset == want[f], f
    assert C.sizeof(_capi.VecSimGpuStats) == want["sizeof"]
