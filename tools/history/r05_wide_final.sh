#!/bin/bash
# the wide-row filter's rates on the round's final code (tools/bench_dims.py, ~6 GB tables, top-10, scan kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
{
for spec in "f32 L2 64 4096 6144 8192" "f32 L2 16 4096 8192" "bf16 IP 64 2560 3072 4096 6144 8192" "bf16 IP 128 3072 4096" "bf16 IP 16 4096" "f16 IP 64 3072" \
            "i8 L2 64 6144 8192 12288 16384" "i8 L2 128 6144" "i8 Cosine 128 5000" "i8 L2 16 8192" "u8 Cosine 64 6144 8192 16384" "u8 L2 64 8192"; do
  set -- $spec
  t=$1; m=$2; b=$3; shift 3
  python tools/bench_dims.py --type $t --metric $m --batch $b "$@" 2>&1 | grep "ms per batch"
done
} | tee gpurun_out/r05c/wide_final.txt
