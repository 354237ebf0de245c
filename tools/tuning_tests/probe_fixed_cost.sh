#!/bin/bash
# fixed cost of a probe launch: the probe kernel's duration (rocprofv3) against the number of probe tiles, ONE reader thread (nothing runs beside it)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c4; do
  for cap in 4096 8192 16384 32768; do
    O=$R/gpurun_out/pfc_${cfg}_$cap; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --stats -d $O/trace -o r1 -- python $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --readers 1 --opt probe_cap=$cap > $O/log.txt 2>&1
    python $R/profiles/summarize.py $O $O/summary > /dev/null 2>&1
    echo "== $cfg probe_cap=$cap: $(grep '^{' $O/log.txt | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4f ms per batch, candidates/query %.0f" % (d["ms_per_step"], d["candidates_per_query"]))')"
    grep -E "k_mfma_filter|k_probe_threshold" $O/summary_kernel_stats.txt | cut -c1-60,79-130
  done
done
