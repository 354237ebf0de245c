#!/bin/bash
# what bounds k_mfma_filter_wide at 64 queries per workgroup: the workgroup's own rate or the memory system?  bf16 3072:
# batch 64 on half / a quarter of the CUs, batch 128 with the two query tiles one after the other (grid 256 x 2) instead of side by side (128 x 2)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
{
for gx in 0 128 64 32; do echo "batch 64 wide_gx=$gx $(python tools/bench_dims.py --type bf16 --metric IP --batch 64 --opt wide_gx=$gx 3072 2>&1 | tail -1)"; done
for gx in 0 256 64 32; do echo "batch 128 wide_gx=$gx $(python tools/bench_dims.py --type bf16 --metric IP --batch 128 --opt wide_gx=$gx 3072 2>&1 | tail -1)"; done
for gx in 0 128 32; do echo "batch 64 f32 8192 wide_gx=$gx $(python tools/bench_dims.py --type f32 --metric L2 --batch 64 --opt wide_gx=$gx 8192 2>&1 | tail -1)"; done
} | tee gpurun_out/r05c/wide_gx.txt
