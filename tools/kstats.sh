#!/bin/bash
# rocprofv3 kernel stats of bench.py for the given configs: tools/kstats.sh c4 c3 ...  -> gpurun_out/kstats_<cfg>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  O=$R/gpurun_out/ks_$c
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O/trace -o r1 -- python $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
  python $R/profiles/summarize.py $O $O/summary > /dev/null 2>&1
  head -12 $O/summary_kernel_stats.txt 2>/dev/null || ls $O $O/trace | head
done
