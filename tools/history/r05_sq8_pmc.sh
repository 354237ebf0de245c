#!/bin/bash
# HBM traffic of the SQ8 filter (is the 4 x 256 B request shape over-fetching?): rocprofv3 PMC FETCH_SIZE on tools/bench_sq8.py, 4 M rows
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq8
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_sq8 -o r1 -- python $R/tools/bench_sq8.py --rows 4000000 --batches 128 --steps 4 > $O/sq8_pmc.log 2>&1
python $R/tools/pmc_dump.py /tmp/pmc_sq8 "k_mfma_filter_lowp" 2>&1 | tee $O/sq8_pmc.txt
tail -4 $O/sq8_pmc.log
