#!/bin/bash
# round 4, one GPU call: full -m gpu suite, same-box A/B of the previous and the current libvsgpu.so on c1 .. c4, the probe size
# for config 3 (k = 100), HNSW with two reader threads.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File" | tail -12 > gpurun_out/r04_gputest_b.log
bash tools/tuning_tests/ab_lib.sh "c4 --steps 30" "c2" "c1 --steps 300" "c3 --steps 12 --warmup 3" > /dev/null
: > gpurun_out/r04_c3_probe.txt
for d in 64 48 32 24 16; do
  python bench.py --no-cpu-baseline --config c3 --steps 12 --warmup 3 --opt probe_div=$d 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('c3 probe_div=$d  step %.3f ms  kernel %.3f ms  other %.3f  cand/q %.0f' % (d['ms_per_step'], r['avg_kernel_ms'], r['other_kernels_ms_per_step'], d['candidates_per_query']))" >> gpurun_out/r04_c3_probe.txt
done
python tools/bench_hnsw.py --rows 200000 --data lowrank --queries 4096 --readers 2 > gpurun_out/r04_hnsw_readers.txt 2>&1
VSGPU_TIMING=1 python bench.py --config c1 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r04_c1_timing.txt > /dev/null
tail -5 gpurun_out/r04_gputest_b.log; cat gpurun_out/ab_lib.txt gpurun_out/r04_c3_probe.txt; grep readers gpurun_out/r04_hnsw_readers.txt; tail -4 gpurun_out/r04_c1_timing.txt
