#!/bin/bash
# the N > 1 path of bench.py on the one GPU there is: two ranks sharing it (RCCL refuses duplicate devices -> the agreed gloo fallback),
# and one rank through RCCL (config 4), after the merge rewrite
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05c
cd $R
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --same-gpu --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/r05c/bench_c2_two_ranks.json 2> gpurun_out/r05c/bench_c2_two_ranks.err
tail -1 gpurun_out/r05c/bench_c2_two_ranks.json | cut -c1-1200
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config c4 --steps 20 --warmup 10 --no-cpu-baseline --no-shard-curve > gpurun_out/r05c/bench_c4_rccl1.json 2> gpurun_out/r05c/bench_c4_rccl1.err
tail -1 gpurun_out/r05c/bench_c4_rccl1.json | cut -c1-900
tail -3 gpurun_out/r05c/bench_c2_two_ranks.err
