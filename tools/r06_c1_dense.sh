#!/bin/bash
# round 6: single queries on small tables -- the sliced dense path (dense_sliced_bytes) against the six-kernel filter path
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c 'import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("%8.4f ms/step  kernel %.4f (%s)  sorted %s" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["kernel"], d["sorted"]))'; }
( time timeout 600 python -m pytest tests/test_gpu_flat_parity.py -q -m gpu -x -k "sliced_dense" ) 2>&1 | tail -5
for rows in 50000 100000 200000 400000 800000; do for dp in 0 100000000000; do for r in 1 2; do
  echo "== rows $rows dense_sliced_bytes $dp readers $r"; python bench.py --config c1 --rows $rows --steps 400 --warmup 20 --readers $r --no-cpu-baseline --no-full-parity --opt dense_sliced_bytes=$dp 2>&1 | line
done; done; done
for b in 2 4; do for dp in 0 100000000000; do
  echo "== rows 100000 batch $b dense_sliced_bytes $dp readers 2"; python bench.py --config c1 --batch $b --steps 400 --warmup 20 --no-cpu-baseline --no-full-parity --opt dense_sliced_bytes=$dp 2>&1 | line
done; done
