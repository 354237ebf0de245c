#!/bin/bash
# round 6: wave-private survivor queues in k_mfma_filter (was: one queue per workgroup behind a returning LDS atomic)
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c 'import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("%8.4f ms/step  kernel %.4f  fixed %.4f  cand/q %.0f fallbacks %d retries %d parity %s" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["fixed_ms_per_batch"], d["candidates_per_query"], d["fallbacks"], d["retries"], d.get("full_table_parity")))'; }
run() { echo "== $*"; timeout 200 python bench.py --no-cpu-baseline --no-shard-curve "$@" 2>&1 | line; }
( timeout 600 python -m pytest tests/test_gpu_flat_parity.py -q -m gpu -x -k "mfma_filter_path or streaming or fp64 or overflow or ties" 2>&1 | tail -3 )
for rep in 1 2; do
run --config c2 --steps 40 --warmup 5
run --config c2 --rows 1250000 --steps 100 --warmup 10
run --config c2 --rows 2500000 --steps 60 --warmup 10
done
run --config c2 --steps 40 --warmup 5 --opt probe_div=128
run --config c2 --data clustered --steps 20 --warmup 5
