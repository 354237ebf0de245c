#!/bin/bash
# round 6: A/B of the per-batch stream operations (upload by kernel, select into the pinned block) on config 1, a 1.25 M-row shard and config 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06c
mkdir -p $OUT
cd $R
( time timeout 600 python -m pytest tests/test_gpu_flow_scenarios.py -q -m gpu -x ) > $OUT/flow.log 2>&1
tail -3 $OUT/flow.log
line() { python -c 'import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("%8.4f ms/step  kernel %.4f  fixed %.4f  sorted %s" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["fixed_ms_per_batch"], d["sorted"]))'; }
for rep in 1 2; do
for o in "upload_kernel=0 sel_mapped=0" "upload_kernel=1 sel_mapped=0" "upload_kernel=0 sel_mapped=1" "upload_kernel=1 sel_mapped=1"; do
  opts=""; for x in $o; do opts="$opts --opt $x"; done
  echo "== c1 [$o]";      python bench.py --config c1 --steps 400 --warmup 20 --no-cpu-baseline --no-full-parity $opts 2>&1 | line
  echo "== c1 r1 [$o]";   python bench.py --config c1 --steps 400 --warmup 20 --readers 1 --no-cpu-baseline --no-full-parity $opts 2>&1 | line
  echo "== shard8 [$o]";  python bench.py --config c2 --rows 1250000 --steps 100 --warmup 10 --no-cpu-baseline --no-shard-curve --no-full-parity $opts 2>&1 | line
  echo "== c2 [$o]";      python bench.py --config c2 --steps 40 --warmup 5 --no-cpu-baseline --no-shard-curve --no-full-parity $opts 2>&1 | line
done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
