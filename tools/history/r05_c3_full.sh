#!/bin/bash
# full-size config 3 (50 M x 1024 int8 Cosine, 256 queries, top-100) with the given lowp_x32 values (TUNING build for the non-shipped ones)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_c3
mkdir -p $O
cd $R
for v in "$@"; do
  if [ "$v" = "default" ]; then python bench.py --config c3 --steps 12 --no-cpu-baseline > $O/bench_c3_$v.json 2> $O/bench_c3_$v.err
  else python bench.py --config c3 --steps 12 --no-cpu-baseline --opt lowp_x32=$v > $O/bench_c3_$v.json 2> $O/bench_c3_$v.err; fi
  python - <<PY
import json
d=json.loads(open("$O/bench_c3_$v.json").read().strip().splitlines()[-1])
print("$v", "ms/step %.3f kernel %.3f frac %.3f mfma %.3f cand/q %.0f fallbacks %d sorted %s %s"%(d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["roofline"]["mfma"]["frac"], d["candidates_per_query"], d["fallbacks"], d["sorted"], d["roofline"]["kernel"]))
PY
done
