#!/bin/bash
# widest rows: eight waves per workgroup (option wide_w8 = 1) against four, batch 64 and 16, ~6 GB tables
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for v in 0 1; do
  for spec in "f32 L2 8192" "bf16 IP 8192" "f16 L2 8192" "i8 L2 16384" "u8 IP 16384"; do
    set -- $spec
    for b in 64 16; do
      echo "wide_w8=$v $(python tools/bench_dims.py --type $1 --metric $2 --batch $b --opt wide_w8=$v $3 2>&1 | tail -1)"
    done
  done
done
