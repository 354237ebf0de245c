#!/bin/bash
# bench.py step time against the probe size (rows / probe_div sampled for the thresholds), at the bench's own conditions
for cfg in c2 c4; do
  for div in 32 48 64 80 96 128 192; do
    python bench.py --config $cfg --steps 30 --warmup 4 --no-cpu-baseline --opt probe_div=$div 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg div $div step %.3f ms kernel %.3f other %.3f cand/q %.0f' % (d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['other_kernels_ms_per_step'], d['candidates_per_query']))"
  done
done
