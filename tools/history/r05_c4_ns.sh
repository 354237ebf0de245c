#!/bin/bash
# A/B on one box: config 4's filter (k_mfma_filter_lowp, 8 waves x 16 queries) with LOWP_T_NS = 4 ring slots (ab/libvsgpu_lns4.so)
# against the shipped 3 (ab/libvsgpu_base.so); also the SQ8 filter's batch-128 line, which already has four
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
cp vectorsimilarity_amd/libvsgpu.so vectorsimilarity_amd/ab/libvsgpu_cur.so
{
for v in base lns4 base lns4; do
  cp vectorsimilarity_amd/ab/libvsgpu_$v.so vectorsimilarity_amd/libvsgpu.so
  python bench.py --config c4 --steps 20 --warmup 10 --no-cpu-baseline --no-shard-curve 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v c4 ms/step %.3f kernel %.3f frac %.3f' % (d['ms_per_step'], r['avg_kernel_ms'], r['frac']))"
  echo "$v $(python tools/bench_dims.py --type bf16 --metric IP --batch 128 512 1024 2>&1 | tail -2 | tr '\n' '|')"
done
} | tee gpurun_out/r05c/c4_ns.txt
cp vectorsimilarity_amd/ab/libvsgpu_cur.so vectorsimilarity_amd/libvsgpu.so
