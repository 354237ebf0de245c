#!/bin/bash
# width 3072 (16-bit rows) / 6144 (8-bit rows): 64 queries per workgroup in one pass (wide_blocks = 4) against two 32-query tiles (0)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for v in 0 4; do
  for spec in "bf16 IP 3072" "bf16 L2 2560" "f16 L2 3072" "i8 L2 6144" "u8 IP 6144" "i8 Cosine 5000"; do
    set -- $spec
    for b in 64 128; do
      echo "wide_blocks=$v $(python tools/bench_dims.py --type $1 --metric $2 --batch $b --opt wide_blocks=$v $3 2>&1 | tail -1)"
    done
  done
done
