#!/bin/bash
# round 5, first GPU call: new tests, full suite, the default bench line, the 1-rank RCCL bench lines (shard curve), two ranks on one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q > $O/sharded_tests.txt 2>&1; echo "sharded rc=$?" >> $O/rc.txt
timeout 600 python -m pytest tests/test_gpu_flat_parity.py -m gpu -x -q -k "default_tier" > $O/default_tier_tests.txt 2>&1; echo "default_tier rc=$?" >> $O/rc.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/all_tests.txt 2>&1; echo "all rc=$?" >> $O/rc.txt
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?" >> $O/rc.txt
TR="python -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
timeout 600 $TR --nproc-per-node 1 --master-port 29611 bench.py --gpus 1 --no-cpu-baseline > $O/bench_c2_rccl1.json 2> $O/bench_c2_rccl1.err; echo "rccl1 rc=$?" >> $O/rc.txt
timeout 600 $TR --nproc-per-node 1 --master-port 29612 bench.py --gpus 1 --config c4 --no-cpu-baseline > $O/bench_c4_rccl1.json 2> $O/bench_c4_rccl1.err; echo "c4 rccl1 rc=$?" >> $O/rc.txt
VECSIM_GPU_EXCHANGE=staged timeout 600 $TR --nproc-per-node 1 --master-port 29613 bench.py --gpus 1 --no-cpu-baseline --no-shard-curve > $O/bench_c2_rccl1_staged.json 2> $O/bench_c2_rccl1_staged.err; echo "staged rc=$?" >> $O/rc.txt
timeout 900 $TR --nproc-per-node 2 --master-port 29614 bench.py --gpus 2 --same-gpu --steps 20 --no-cpu-baseline > $O/bench_c2_two_ranks_one_gpu.json 2> $O/bench_c2_two_ranks_one_gpu.err; echo "2rank rc=$?" >> $O/rc.txt
cat $O/rc.txt
tail -3 $O/all_tests.txt
