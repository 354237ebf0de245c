#!/bin/bash
# round 6: reader lanes at the 1.25 M-row shard of config 2 (what a rank of an 8-GPU strong-scaling job runs) and at config 2
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c 'import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("%8.4f ms/step  kernel %.4f  fixed %.4f  cand/q %.0f" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["fixed_ms_per_batch"], d["candidates_per_query"]))'; }
run() { echo "== $*"; timeout 200 python bench.py --no-cpu-baseline --no-shard-curve --no-full-parity "$@" 2>&1 | line; }
for r in 1 2 3 4; do run --config c2 --rows 1250000 --steps 120 --warmup 12 --readers $r; done
for r in 2 3 4; do run --config c2 --steps 40 --warmup 8 --readers $r; done
for r in 2 3; do run --config c4 --steps 30 --warmup 12 --readers $r; done
