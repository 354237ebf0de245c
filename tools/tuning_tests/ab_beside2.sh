#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/ab_beside2.txt
: > $OUT
line() {
  cfg=$1; shift
  python $R/bench.py --config $cfg --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('%-4s %-58s step %.3f ms  kernel %.3f ms (%.3f)  fixed %.3f  other %.3f' % ('$cfg', '$*', d['ms_per_step'], r['avg_kernel_ms'], r['frac'], d['fixed_ms_per_batch'], r['other_kernels_ms_per_step']))" >> $OUT
}
line c2 --opt aux_cus=0
line c2 --opt aux_cus=32 --opt wg_per_cu=3
line c2 --opt aux_cus=32 --opt wg_per_cu=4
line c2 --opt aux_cus=32 --opt scan_all_cus=1
line c2 --opt aux_cus=8 --opt scan_all_cus=1
line c2 --opt aux_cus=16 --opt scan_all_cus=1
line c2 --opt aux_cus=64 --opt scan_all_cus=1
line c2 --opt aux_cus=32 --opt wg_per_cu=3 --readers 1
line c2 --opt aux_cus=0 --opt wg_per_cu=3
line c4 --steps 30 --opt aux_cus=0
line c4 --steps 30 --opt aux_cus=32 --opt scan_all_cus=1
line c4 --steps 30 --opt aux_cus=8 --opt scan_all_cus=1
line c3 --steps 12 --warmup 3 --opt aux_cus=8 --opt scan_all_cus=1
line c3 --steps 12 --warmup 3 --opt aux_cus=32 --opt scan_all_cus=1
cat $OUT
