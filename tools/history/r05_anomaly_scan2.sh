#!/bin/bash
# second outlier scan: small batches (1 .. 32 queries) over kinds and dims; ms per batch is what matters here
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
{
for tm in "f32 L2" "bf16 IP" "i8 Cosine" "u8 Cosine" "f16 L2"; do
  set -- $tm
  for b in 1 2 4 8 16 32; do
    python tools/bench_dims.py --type $1 --metric $2 --batch $b 128 768 1024 2048 4096 2>&1 | grep "ms per batch"
  done
done
} | tee gpurun_out/r05c/anomaly_scan2.txt
