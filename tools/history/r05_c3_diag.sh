#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_c3
mkdir -p $O
cd $R
DOPTS=$(echo $1 | sed 's/\([0-9]*\)/lowp_x32=\1/g')
timeout 900 python tools/time_lowp_kernels.py --rows ${2:-8388608} --opts $DOPTS 2>&1 | grep -v "x32 phases\|wave [0-7]"
