#!/bin/bash
# late round 5: wide-row parity tests on the in-tree library first, then the A/B against ab/libvsgpu_base.so
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
timeout 1500 python -m pytest tests/test_gpu_flat_parity.py -x -q -m gpu -k "wide or lowp_mfma_filter_path or mfma_filter" 2>&1 | grep -E "passed|failed|error|assert" | head -20 | tee gpurun_out/r05c/wide_tests.txt
VERSIONS="${VERSIONS:-base cur base cur}" bash tools/r05_wide_ns.sh > /dev/null 2>&1
cat gpurun_out/r05c/wide_ns2.txt
