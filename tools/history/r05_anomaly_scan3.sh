#!/bin/bash
# third outlier scan: dims that are not multiples of the kernel widths, batch 64 and 128
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
{
for tm in "f32 L2" "bf16 IP" "f16 Cosine" "i8 Cosine" "i8 L2" "u8 IP" "u8 Cosine"; do
  set -- $tm
  for b in 64 128; do
    python tools/bench_dims.py --type $1 --metric $2 --batch $b 100 300 520 800 1000 1100 1600 2100 2500 3100 3300 5000 2>&1 | grep "ms per batch"
  done
done
} | tee gpurun_out/r05c/anomaly_scan3.txt
