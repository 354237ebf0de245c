#!/bin/bash
# how much of a probe kernel's time is fixed (launch, fragment load, ring fill) and how much scales with its tiles: rocprofv3 kernel
# stats of config 4 / config 2 at several probe divisors.  Output: gpurun_out/probe_fixed.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/probe_fixed.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in c4 c2; do
  for d in 0 200 2000; do
    rm -rf /tmp/pf
    rocprofv3 --kernel-trace --stats -d /tmp/pf -o r1 -- python $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --opt probe_div=$d > /dev/null 2>&1
    python - "$cfg probe_div=$d" >> $OUT <<'PY'
import sqlite3, glob, sys
db = sqlite3.connect(glob.glob('/tmp/pf/**/r1_results.db', recursive=True)[0])
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if 'k_mfma_filter' in name or 'exact_pairs' in name or 'select' in name or 'threshold' in name:
        print("%-20s %-70s calls %4d avg %9.1f us" % (sys.argv[1], name[:70], calls, avg))
PY
  done
done
cat $OUT
