#!/bin/bash
# where a wide-row batch spends its time besides the filter kernel: rocprofv3 kernel statistics of tools/bench_dims.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c/wide_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "bf16 IP 128 3072" "bf16 IP 64 3072" "f32 L2 64 8192"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats -d $O/t_$1_$3_$4 -o r1 -- python $R/tools/bench_dims.py --type $1 --metric $2 --batch $3 $4 > $O/log_$1_$3_$4.txt 2>&1
  tail -1 $O/log_$1_$3_$4.txt
  f=$(find $O/t_$1_$3_$4 -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("   %-70s calls %5s total %10.1f us avg %9.1f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3))
PY
done
