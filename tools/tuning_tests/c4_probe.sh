#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r04_c4_probe.txt
: > $OUT
for d in 0 24 48 64 96 128; do
  python $R/bench.py --no-cpu-baseline --config c4 --steps 30 --opt probe_div=$d 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('c4 probe_div=$d  step %.3f ms  kernel %.3f ms  other %.3f  cand/q %.0f' % (d['ms_per_step'], r['avg_kernel_ms'], r['other_kernels_ms_per_step'], d['candidates_per_query']))" >> $OUT
done
cat $OUT
