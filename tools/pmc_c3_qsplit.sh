#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/qsplit; rm -rf $O; mkdir -p $O
python $R/bench.py --config c3 --rows 10000000 --steps 10 --warmup 2 --no-cpu-baseline --opt lowp_qsplit=1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qsplit step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['avg_kernel_ms'],3))"
python $R/bench.py --config c3 --rows 10000000 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['avg_kernel_ms'],3))"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o r1 -- python $R/bench.py --config c3 --rows 10000000 --steps 3 --warmup 1 --no-cpu-baseline --opt lowp_qsplit=1 > $O/log 2>&1
python - <<P
import sqlite3
d=sqlite3.connect("$O/pmc_fetch/r1_results.db")
for k,c,n,v,dur in d.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration)/1000.0 from counters_collection group by kernel_name, counter_name order by avg(value) desc limit 3"):
    print(k[:70], c, n, round(v/1e6,3), "GB(x2 for gfx950)", round(dur,1), "us")
P
