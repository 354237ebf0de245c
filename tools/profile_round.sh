#!/bin/bash
# Round-end evidence: the default bench line, rocprofv3 kernel stats of the same command, the two PMC passes for
# HBM traffic (separate runs, --kernel-trace only), and -- with a second argument "all" -- the bench lines and kernel
# stats of the other BASELINE configs.  Outputs under gpurun_out/prof_$1; profiles/summarize.py turns them into the
# text files kept in profiles/.
TAG=${1:-r1}
ALL=${2:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_line.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r1 -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-shard-curve --no-full-parity > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shard-curve --no-full-parity > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shard-curve --no-full-parity > $OUT/pmc_write.log 2>&1
python $R/profiles/summarize.py $OUT $OUT/summary || true
if [ "$ALL" = "all" ]; then
  for c in c3 c4; do
    timeout 600 python $R/bench.py --config $c --steps 20 --warmup 10 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  done
  # config 1 is one query per step (40 us): 400 steps, so that starting the reader threads does not weigh on the timed region
  timeout 600 python $R/bench.py --config c1 --steps 400 --warmup 20 > $OUT/bench_c1.json 2> $OUT/bench_c1.err
  # config 5: the graph is built on the host cores first (~3 min per million rows); both row distributions
  timeout 900 python $R/bench.py --config c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
  timeout 600 python $R/bench.py --config c5 --rows 200000 --data uniform --steps 5 --warmup 1 > $OUT/bench_c5_uniform.json 2> $OUT/bench_c5_uniform.err
  for c in c3 c4; do
    mkdir -p $OUT/cfg_$c
    rocprofv3 --kernel-trace --stats -d $OUT/cfg_$c/trace -o r1 -- python $R/bench.py --config $c --steps 10 --warmup 10 --no-cpu-baseline --no-full-parity > $OUT/cfg_$c/trace.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg_$c/pmc_fetch -o r1 -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-full-parity > $OUT/cfg_$c/pmc_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg_$c/pmc_write -o r1 -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-full-parity > $OUT/cfg_$c/pmc_write.log 2>&1
    python $R/profiles/summarize.py $OUT/cfg_$c $OUT/cfg_$c/summary || true
  done
fi
python $R/profiles/make_pmc_traffic.py $OUT > $OUT/pmc_traffic.json || true
tail -1 $OUT/bench_line.json | cut -c1-1200
