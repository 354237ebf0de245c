#!/bin/bash
# late round 5: A/B of wide-row kernel builds (ab/libvsgpu_<name>.so against the in-tree library = "cur"), then the wide-row parity
# tests on $TESTLIB (default: the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
SPECS="${SPECS:-}" VERSIONS="${VERSIONS:-cur}" bash tools/r05_wide_ns.sh > /dev/null 2>&1
cat gpurun_out/r05c/wide_ns2.txt
if [ -n "$TESTLIB" ]; then cp vectorsimilarity_amd/ab/libvsgpu_$TESTLIB.so vectorsimilarity_amd/libvsgpu.so; fi
timeout 1500 python -m pytest tests/test_gpu_flat_parity.py -x -q -m gpu -k "wide or lowp_mfma_filter_path or mfma_filter" 2>&1 | grep -E "passed|failed|error|assert" | head -20 | tee gpurun_out/r05c/wide_tests.txt
timeout 400 python tools/fuzz_parity.py --wide --seconds 120 --seed 61 --readers 2 2>&1 | tail -2 | tee -a gpurun_out/r05c/wide_tests.txt
