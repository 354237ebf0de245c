#!/bin/bash
# timeline of a single query on 100 K x 128 (config 1): where the 77 us go.  -> gpurun_out/c1_trace.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/c1_trace; rm -rf $O; mkdir -p $O
for readers in 1 2; do
  rocprofv3 --kernel-trace --memory-copy-trace -d $O/t$readers -o r1 -- python $R/bench.py --config c1 --steps 40 --warmup 5 --no-cpu-baseline --readers $readers > $O/log$readers.txt 2>&1
  echo "== readers $readers: $(tail -1 $O/log$readers.txt | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  python $R/tools/timeline.py $(find $O/t$readers -name "*.db" | head -1) 36
done > $R/gpurun_out/c1_trace.txt 2>&1
tail -80 $R/gpurun_out/c1_trace.txt
