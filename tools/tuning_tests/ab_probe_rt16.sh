#!/bin/bash
# A/B of the fp32 probe's tile shape (option probe_rt16) under bench conditions, same box.  -> gpurun_out/ab_probe_rt16.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
  for v in 0 1; do
    for cfg in c2 c1; do
      python bench.py --config $cfg --steps $([ $cfg = c1 ] && echo 200 || echo 40) --warmup 5 --no-cpu-baseline --opt probe_rt16=$v 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg probe_rt16=$v rep $rep: %.4f ms per batch, scan kernel %.4f ms, candidates/query %.0f' % (d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['candidates_per_query']))"
    done
  done
done
