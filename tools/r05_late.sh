#!/bin/bash
# late round 5: wide-row A/B (ring depth, stages per unit), then the wide-row parity tests on the in-tree library
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
VERSIONS="${VERSIONS:-base ns6u2 ns8u2 base}" bash tools/r05_wide_ns.sh > /dev/null 2>&1
cat gpurun_out/r05c/wide_ns2.txt
timeout 1500 python -m pytest tests/test_gpu_flat_parity.py -x -q -m gpu -k "wide or lowp_mfma_filter_path or mfma_filter" 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/r05c/wide_tests.txt
