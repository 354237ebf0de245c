#!/bin/bash
# A/B on one box: wide-row filter with WIDE_NS_ALONE = 5 / 6 ring slots (ab/libvsgpu_ns5.so, _ns6.so) against the shipped 3 (ab/libvsgpu_base.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
cp vectorsimilarity_amd/libvsgpu.so vectorsimilarity_amd/ab/libvsgpu_cur.so
{
for v in base ns5 ns6 base ns6; do
  cp vectorsimilarity_amd/ab/libvsgpu_$v.so vectorsimilarity_amd/libvsgpu.so
  for spec in "bf16 IP 3072 64" "bf16 IP 3072 128" "i8 L2 6144 128" "f32 L2 8192 64"; do
    set -- $spec
    echo "$v $(python tools/bench_dims.py --type $1 --metric $2 --batch $4 $3 2>&1 | tail -1)"
  done
done
} | tee $R/gpurun_out/r05c/wide_ns.txt
cp vectorsimilarity_amd/ab/libvsgpu_cur.so vectorsimilarity_amd/libvsgpu.so
