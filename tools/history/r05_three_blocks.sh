#!/bin/bash
# three column blocks per workgroup at kernel width 128 k-steps (default where 48-query tiles mean fewer passes) against two (option wide_blocks=2)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
timeout 1500 python -m pytest tests/test_gpu_flat_parity.py -x -q -m gpu -k "wide or lowp_mfma_filter_path or mfma_filter" 2>&1 | grep -E "passed|failed|error|assert" | head -20 | tee gpurun_out/r05c/wide_tests.txt
{
for spec in "bf16 IP 128 4096" "bf16 IP 96 4096" "bf16 IP 40 4096" "bf16 IP 64 4096" "i8 L2 128 8192" "i8 L2 40 8192" "f32 L2 128 4096" "f32 L2 96 4096" "u8 Cosine 128 8192"; do
  set -- $spec
  for o in 0 2; do
    echo "wide_blocks=$o $(python tools/bench_dims.py --type $1 --metric $2 --batch $3 --opt wide_blocks=$o $4 2>&1 | tail -1)"
  done
done
} | tee gpurun_out/r05c/three_blocks.txt
timeout 400 python tools/fuzz_parity.py --wide --seconds 150 --seed 71 --readers 2 2>&1 | tail -2 | tee -a gpurun_out/r05c/wide_tests.txt
