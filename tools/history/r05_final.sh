#!/bin/bash
# round 5, closing measurements: config 3 with the per-table kernel choice, the default line's rocprofv3 stats + PMC traffic without the shard curve
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05b
mkdir -p $O $O/cfg_c3
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --config c3 --steps 20 --warmup 10 > $O/bench_c3.json 2> $O/bench_c3.err
rocprofv3 --kernel-trace --stats -d $O/cfg_c3/trace -o r1 -- python $R/bench.py --config c3 --steps 10 --warmup 10 --no-cpu-baseline > $O/cfg_c3/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/cfg_c3/pmc_fetch -o r1 -- python $R/bench.py --config c3 --steps 3 --warmup 2 --no-cpu-baseline > $O/cfg_c3/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/cfg_c3/pmc_write -o r1 -- python $R/bench.py --config c3 --steps 3 --warmup 2 --no-cpu-baseline > $O/cfg_c3/pmc_write.log 2>&1
python $R/profiles/summarize.py $O/cfg_c3 $O/cfg_c3/summary || true
rocprofv3 --kernel-trace --stats -d $O/trace -o r1 -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-shard-curve > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shard-curve > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shard-curve > $O/pmc_write.log 2>&1
python $R/profiles/summarize.py $O $O/summary || true
python $R/profiles/make_pmc_traffic.py $O > $O/pmc_traffic.json || true
tail -1 $O/bench_c3.json | cut -c1-600; head -6 $O/cfg_c3/summary_kernel_stats.txt; head -5 $O/summary_kernel_stats.txt
