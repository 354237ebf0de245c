#!/bin/bash
# wide rows: scan kernel rate across types / widths / batch sizes (tools/bench_dims.py).  Output: gpurun_out/wide_sweep.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/wide_sweep.txt
: > $OUT
python $R/tools/bench_dims.py --type f32 --batch 64 4096 6144 8192 2>&1 | grep "dim" >> $OUT
python $R/tools/bench_dims.py --type f32 --batch 16 4096 8192 2>&1 | grep "dim" >> $OUT
python $R/tools/bench_dims.py --type bf16 --batch 64 3072 4096 6144 8192 2>&1 | grep "dim" >> $OUT
python $R/tools/bench_dims.py --type bf16 --batch 16 4096 2>&1 | grep "dim" >> $OUT
python $R/tools/bench_dims.py --type i8 --batch 64 4097 8192 12288 16384 2>&1 | grep "dim" >> $OUT
python $R/tools/bench_dims.py --type i8 --batch 16 8192 2>&1 | grep "dim" >> $OUT
cat $OUT
