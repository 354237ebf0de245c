#!/bin/bash
# filter rate over kinds, dims and (non-multiple) batch sizes: looking for outliers like the uint8 Cosine padding flood
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
{
for tm in "f32 L2" "f32 Cosine" "bf16 IP" "f16 L2" "i8 L2" "i8 Cosine" "u8 Cosine" "u8 IP"; do
  set -- $tm
  for b in 40 64 100 128 200 256; do
    python tools/bench_dims.py --type $1 --metric $2 --batch $b 256 768 1024 1536 2048 3000 2>&1 | grep "ms per batch"
  done
done
} | tee gpurun_out/r05c/anomaly_scan.txt
