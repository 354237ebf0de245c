#!/bin/bash
# same-box A/B of two builds of libvsgpu.so (vectorsimilarity_amd/ab/libvsgpu_prev.so against libvsgpu_new.so): the bench line of
# the given configs, alternating.  usage: ab_lib.sh "c4 --steps 30" "c3 --steps 12 --warmup 3" ...   Output: gpurun_out/ab_lib.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/ab_lib.txt
: > $OUT
for round in 1 2; do
  for args in "$@"; do
    for v in prev new; do
      cp $R/vectorsimilarity_amd/ab/libvsgpu_$v.so $R/vectorsimilarity_amd/libvsgpu.so
      python $R/bench.py --no-cpu-baseline --config $args 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('%-5s %-32s step %.3f ms  kernel %.3f ms (%.3f)  fixed %.3f  cand/q %.0f' % ('$v', '$args', d['ms_per_step'], r['avg_kernel_ms'], r['frac'], d['fixed_ms_per_batch'], d['candidates_per_query']))" >> $OUT
    done
  done
done
cp $R/vectorsimilarity_amd/ab/libvsgpu_new.so $R/vectorsimilarity_amd/libvsgpu.so
cat $OUT
