#!/bin/bash
# round 6: the streaming threshold (option stream_tau) against the full probe, same box
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c 'import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("%8.4f ms/step  kernel %.4f  fixed %.4f  cand/q %.0f fallbacks %d retries %d parity %s" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["fixed_ms_per_batch"], d["candidates_per_query"], d["fallbacks"], d["retries"], d.get("full_table_parity")))'; }
( time timeout 300 python -m pytest tests/test_gpu_flat_parity.py -q -m gpu -x -k "streaming_threshold" ) 2>&1 | tail -6
for rep in 1 2; do for st in 0 1; do
  echo "== c2 stream_tau=$st";        timeout 150 python bench.py --config c2 --steps 40 --warmup 5 --no-cpu-baseline --no-shard-curve --opt stream_tau=$st 2>&1 | line
  echo "== shard8 stream_tau=$st";    timeout 150 python bench.py --config c2 --rows 1250000 --steps 100 --warmup 10 --no-cpu-baseline --no-shard-curve --opt stream_tau=$st 2>&1 | line
  echo "== shard4 stream_tau=$st";    timeout 150 python bench.py --config c2 --rows 2500000 --steps 60 --warmup 10 --no-cpu-baseline --no-shard-curve --opt stream_tau=$st 2>&1 | line
done; done
for d in lowrank clustered; do for st in 0 1; do
  echo "== c2 $d stream_tau=$st"; timeout 150 python bench.py --config c2 --data $d --steps 20 --warmup 5 --no-shard-curve --opt stream_tau=$st 2>&1 | line
done; done
echo "== c2 one reader"; for st in 0 1; do timeout 150 python bench.py --config c2 --steps 30 --warmup 5 --readers 1 --no-cpu-baseline --no-shard-curve --no-full-parity --opt stream_tau=$st 2>&1 | line; done
