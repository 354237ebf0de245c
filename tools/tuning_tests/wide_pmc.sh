#!/bin/bash
# HBM traffic of the wide-row filter against its algorithmic bytes: do the query tiles of a row tile share the rows through L2?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for spec in "f32 4096 64" "f32 8192 64" "f32 4096 16" "bf16 4096 64"; do
  set -- $spec
  O=$R/gpurun_out/wide_pmc_$1_$2_$3; rm -rf $O; mkdir -p $O
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o r1 -- python $R/tools/bench_dims.py --type $1 --batch $3 $2 > $O/log.txt 2>&1
  echo "== $spec: $(grep dim $O/log.txt | tail -1)"
  python $R/tools/pmc_dump.py $O wide | grep -v "MODE\|, 0, 0" | head -4
  python $R/tools/pmc_dump.py $O wide | head -6 | cut -c1-160
done
