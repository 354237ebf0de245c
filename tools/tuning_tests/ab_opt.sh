#!/bin/bash
# A/B of a 0/1 context option under bench conditions on one box: tools/tuning_tests/ab_opt.sh NAME cfg...  -> stdout
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
name=$1; shift
for rep in 1 2; do
  for cfg in "$@"; do
    for v in 0 1; do
      steps=40; [ $cfg = c1 ] && steps=400; [ $cfg = c3 ] && steps=20
      python bench.py --config $cfg --steps $steps --warmup 5 --no-cpu-baseline --opt $name=$v 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $name=$v rep $rep: %.4f ms per batch, scan kernel %.4f ms' % (d['ms_per_step'], d['roofline']['avg_kernel_ms']))"
    done
  done
done
