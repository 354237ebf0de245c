"""GPU: a bounded slice of tools/fuzz_parity.py -- random (type, metric, dim, rows, batch, k) on the filter paths against the exact
path of the same index (0 ulp); the full soak (7 minutes, 1185 shapes, no mismatch) is a tools/ run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,readers", [(3, 0), (11, 2)])
def test_random_shapes_filter_path_equals_exact_path(seed, readers):
    """(readers 2: every shape is also queried by two threads at once -- reader lanes, scan chain; the soaks with 2 and 3 readers:
    798 + 288 shapes, no mismatch)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--seconds", "12", "--seed", str(seed),
                        "--readers", str(readers)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches 0" in r.stdout, r.stdout[-2000:]


def test_random_shapes_sharded_index_equals_single_index():
    """tools/fuzz_sharded.py: G shards on one GPU against the single index -- partition, candidate rule, gid-order merge, ties across
    shards, deletes (the soak: 1845 shapes in 3 minutes, no mismatch)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_sharded.py"), "--seconds", "10", "--seed", "9"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches 0" in r.stdout, r.stdout[-2000:]


def test_random_hnsw_indexes_batch_iterator_equals_the_oracle_walk():
    """tools/fuzz_hnsw_iter.py: the HNSW batch iterator's graph walk (host heaps, GPU distances) against oracle/vso_hnsw.c's twin over
    random indexes -- metrics, M, ef, single / multi-value, deleted labels, tied distances, random batch sizes (the soak: 18 015
    iterations, 223 044 batches in 4 minutes, no mismatch)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_hnsw_iter.py"), "--seconds", "12", "--seed", "5"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches 0" in r.stdout, r.stdout[-2000:]


def test_random_wide_row_shapes_filter_path_equals_exact_path():
    """tools/fuzz_parity.py --wide: only the wide-row kernels' shapes (16-bit rows of 2049 .. 8192 elements, 8-bit rows of 4097 .. 16384, fp32
    3073 .. 8192; batches on both sides of the 16 / 32 / 64 queries a workgroup holds -- four column blocks at kernel width 96), two reader
    threads (the soaks: 642 + 492 shapes, no mismatch)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--wide", "--seconds", "15", "--seed", "21", "--readers", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches 0" in r.stdout, r.stdout[-2000:]
