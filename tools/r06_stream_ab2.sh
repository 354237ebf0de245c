#!/bin/bash
# round 6: streaming threshold, lock-free list: 0 = full probe, 1 = streaming, 4 = full probe + streaming kernel without insertions / re-reads (its epilogue's cost alone)
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c 'import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("%8.4f ms/step  kernel %.4f  fixed %.4f  cand/q %.0f fallbacks %d retries %d parity %s" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["fixed_ms_per_batch"], d["candidates_per_query"], d["fallbacks"], d["retries"], d.get("full_table_parity")))'; }
( time timeout 300 python -m pytest tests/test_gpu_flat_parity.py -q -m gpu -x -k "streaming_threshold" ) 2>&1 | tail -6
run() { echo "== $*"; timeout 150 python bench.py --config c2 --no-cpu-baseline --no-shard-curve "$@" 2>&1 | line; }
run --steps 40 --warmup 5 --opt stream_tau=0
run --rows 1250000 --steps 100 --warmup 10 --opt stream_tau=0
for cfg in "8" "4" "16"; do set -- $cfg
  o="--opt stream_tau=1 --opt stream_refresh=$1"
  run --steps 40 --warmup 5 $o
  run --rows 1250000 --steps 100 --warmup 10 $o
done
run --steps 40 --warmup 5 --opt stream_tau=0
for d in lowrank clustered; do for st in 0 1; do run --data $d --steps 20 --warmup 5 --opt stream_tau=$st; done; done
