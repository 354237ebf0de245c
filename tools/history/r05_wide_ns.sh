#!/bin/bash
# A/B on one box: wide-row filter (k_mfma_filter_wide) with WIDE_NS_ALONE = 4 / 5 ring slots (ab/libvsgpu_ns4.so, _ns5.so: the shipped
# sources compiled with -DWIDE_NS_ALONE=n) against ab/libvsgpu_base.so; the in-tree library is put back at the end.
# First run (3 / 5 / 6 slots): profiles/r05_wide_ring_depth.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
cp vectorsimilarity_amd/libvsgpu.so vectorsimilarity_amd/ab/libvsgpu_cur.so
{
for v in ${VERSIONS:-base ns4 ns5 base}; do
  cp vectorsimilarity_amd/ab/libvsgpu_$v.so vectorsimilarity_amd/libvsgpu.so
  IFS=';' read -ra LIST <<< "${SPECS:-bf16 IP 3072 64;bf16 IP 3072 128;bf16 IP 4096 64;bf16 IP 6144 64;bf16 IP 8192 64;i8 L2 6144 128;i8 L2 8192 64;i8 L2 16384 64;u8 Cosine 8192 64;f32 L2 4096 64;f32 L2 6144 64;f32 L2 8192 64}"
  for spec in "${LIST[@]}"; do
    set -- $spec
    echo "$v $(python tools/bench_dims.py --type $1 --metric $2 --batch $4 $3 2>&1 | tail -1)"
  done
done
} | tee $R/gpurun_out/r05c/wide_ns2.txt
cp vectorsimilarity_amd/ab/libvsgpu_cur.so vectorsimilarity_amd/libvsgpu.so
