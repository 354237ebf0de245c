#!/bin/bash
# after the wide-row kernel rewrite: the whole GPU suite, the wide-row fuzz (also with concurrent readers), the stress tool
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|assert" | head -20 | tee gpurun_out/r05c/all_tests2.txt
timeout 600 python tools/fuzz_parity.py --wide --seconds 240 --seed 51 2>&1 | tail -3 | tee gpurun_out/r05c/fuzz_wide.txt
timeout 600 python tools/fuzz_parity.py --wide --seconds 180 --seed 52 --readers 3 2>&1 | tail -3 | tee -a gpurun_out/r05c/fuzz_wide.txt
timeout 600 python tools/fuzz_parity.py --seconds 120 --seed 53 2>&1 | tail -2 | tee -a gpurun_out/r05c/fuzz_wide.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r05c/all_tests2.txt
python bench.py --no-shard-curve 2>/dev/null | tail -1 | cut -c1-400 | tee gpurun_out/r05c/bench_default_line.txt
