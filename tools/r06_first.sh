#!/bin/bash
# round 6, first GPU call: the full-size parity test, the default bench line with full_table_parity, c3 / c4 lines, then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06a
mkdir -p $OUT
cd $R
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu ) > $OUT/fullsize.log 2>&1
timeout 600 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 600 python bench.py --config c3 --steps 20 --warmup 10 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
timeout 600 python bench.py --config c4 --steps 20 --warmup 10 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $OUT/gpu_suite.log 2>&1
tail -5 $OUT/fullsize.log; tail -3 $OUT/gpu_suite.log
for c in c2 c3 c4; do python - <<PY
import json
d=json.loads(open("$OUT/bench_$c.json").read().strip().splitlines()[-1])
print("$c", d["ms_per_step"], d["roofline"]["frac"], d.get("full_table_parity"), d.get("full_table_parity_detail",{}).get("seconds"))
PY
done
