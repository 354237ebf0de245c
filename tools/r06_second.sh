#!/bin/bash
# round 6, second GPU call: flow scenarios, the strengthened full-size test, event-flag cost, timelines of config 1 / a 1.25 M-row shard / config 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06b
mkdir -p $OUT
cd $R
( time timeout 1200 python -m pytest tests/test_gpu_flow_scenarios.py -q -m gpu ) > $OUT/flow.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu ) > $OUT/fullsize.log 2>&1
( cd tools/stream && hipcc --offload-arch=gfx950 -O2 -o /tmp/event_cost event_cost.hip && /tmp/event_cost ) > $OUT/event_cost.txt 2>&1
cd /tmp && export TMPDIR=/tmp
tl() {  # name, bench args
  local name=$1; shift
  rm -rf $OUT/t_$name
  rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/t_$name -o r1 -- python $R/bench.py "$@" --no-cpu-baseline --no-shard-curve --no-full-parity > $OUT/log_$name.txt 2>&1
  echo "== $name: $(tail -1 $OUT/log_$name.txt | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["fixed_ms_per_batch"])')"
  python $R/tools/timeline.py $(find $OUT/t_$name -name "*.db" | head -1) 44
  rm -rf $OUT/t_$name
}
( tl c1_r1 --config c1 --steps 40 --warmup 5 --readers 1
  tl c1_r2 --config c1 --steps 40 --warmup 5 --readers 2
  tl shard8_r1 --config c2 --rows 1250000 --steps 30 --warmup 5 --readers 1
  tl shard8_r2 --config c2 --rows 1250000 --steps 30 --warmup 5 --readers 2
  tl c2_r2 --config c2 --steps 20 --warmup 5 --readers 2 ) > $OUT/timelines.txt 2>&1
tail -15 $OUT/flow.log; tail -4 $OUT/fullsize.log; cat $OUT/event_cost.txt
