#!/bin/bash
# late round 5: wide-row ring depth A/B, host merge timing at the configs' exchange sizes, config 3 through the 1-rank RCCL path,
# the GPU tests touched since the last full run
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
bash tools/r05_wide_ns.sh > /dev/null 2>&1
cat gpurun_out/r05c/wide_ns.txt
python tools/merge_time.py > gpurun_out/r05c/merge_time.txt 2>&1
cat gpurun_out/r05c/merge_time.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --config c3 --steps 10 --warmup 10 --no-cpu-baseline --no-shard-curve > gpurun_out/r05c/bench_c3_rccl1.json 2> gpurun_out/r05c/bench_c3_rccl1.err
tail -1 gpurun_out/r05c/bench_c3_rccl1.json | cut -c1-1500
timeout 1200 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r05c/hnsw_sharded_tests.txt
timeout 1200 python -m pytest tests/test_gpu_flat_parity.py -x -q -m gpu -k "other_tiers or default_tier" 2>&1 | tail -5 | tee gpurun_out/r05c/tier_tests.txt
