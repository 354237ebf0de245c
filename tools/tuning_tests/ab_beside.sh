#!/bin/bash
# A/B of the "beside" mode (ctx option aux_cus: CUs reserved for the small kernels so that they run beside other batches' scans;
# 0 = round 3's two-gate chain) under bench conditions.  Output: gpurun_out/ab_beside.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/ab_beside.txt
: > $OUT
line() {  # config, extra args...
  cfg=$1; shift
  python $R/bench.py --config $cfg --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('%-4s %-34s step %.3f ms  kernel %.3f ms (%.3f)  fixed %.3f  other %.3f  cand/q %.0f' % ('$cfg', '$*', d['ms_per_step'], r['avg_kernel_ms'], r['frac'], d['fixed_ms_per_batch'], r['other_kernels_ms_per_step'], d['candidates_per_query']))" >> $OUT
}
for a in 0 32 16 64; do line c2 --opt aux_cus=$a; done
line c2 --opt aux_cus=32 --readers 1
line c2 --opt aux_cus=0 --readers 1
line c2 --opt aux_cus=32 --readers 3
line c2 --opt aux_cus=32
for a in 0 8 16 32; do line c4 --steps 30 --opt aux_cus=$a; done
line c4 --steps 30 --opt aux_cus=8 --readers 1
for a in 0 8 16; do line c3 --steps 12 --warmup 3 --opt aux_cus=$a; done
line c1 --steps 200 --opt aux_cus=32
line c1 --steps 200 --opt aux_cus=0
cat $OUT
