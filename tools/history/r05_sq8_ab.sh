#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
python tools/bench_sq8.py --rows 10000000 --batches 64,128 --metric L2 2>&1 | grep -v "^ingest" | tee $O/sq8_bench.txt
python tools/bench_sq8.py --rows 10000000 --batches 128 --metric IP 2>&1 | grep -v "^ingest" | tee -a $O/sq8_bench.txt
bash tools/r05_sq8_pmc.sh
