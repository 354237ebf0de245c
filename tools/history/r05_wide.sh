#!/bin/bash
# round 5: wide rows after spreading the row requests over a stage's k-steps (compare profiles/r04_wide_blocks4.txt, r04_wide_dims.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05b
cd $R
{
for spec in "bf16 IP 3072" "bf16 L2 2560" "f16 L2 3072" "i8 L2 6144" "i8 Cosine 5000" "bf16 IP 4096" "bf16 IP 6144" "f32 L2 4096" "f32 L2 8192" "i8 L2 16384"; do
  set -- $spec
  for b in 64 128; do
    python tools/bench_dims.py --type $1 --metric $2 --batch $b $3 2>&1 | tail -1
  done
done
python tools/bench_dims.py --type bf16 --metric IP --batch 16 3072 2>&1 | tail -1
} | tee $R/gpurun_out/r05b/wide_spread.txt
timeout 900 python -m pytest tests/test_gpu_flat_parity.py -m gpu -x -q -k "lowp_mfma_filter_path or wide" 2>&1 | tail -2
