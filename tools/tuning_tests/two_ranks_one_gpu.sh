#!/bin/bash
# bench.py's N > 1 path on a one-GPU box: two ranks on GPU 0 (RCCL refuses duplicate devices -> the agreed gloo fallback carries the
# exchange), weak and strong scaling, then the 1-rank RCCL run beside the plain index.  -> gpurun_out/two_ranks_one_gpu.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus $1 --steps 10 --warmup 2 --no-cpu-baseline "${@:3}" 2>&1 | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('n_gpus', d['n_gpus'], d['scaling'], 'rows/gpu', d['config']['rows_per_gpu'], 'total', d['config']['rows_total'], '| %.3f ms per batch, value %.3g %s' % (d['ms_per_step'], d['value'], d['unit']))
print('   exchange:', d['config']['exchange'][:110])
for p in (d.get('per_rank_ms_per_batch') or []): print('   ', json.dumps(p)[:230])
"; }
echo "== 2 ranks on one GPU, weak (5 M rows each)"; run 2 29521 --same-gpu --rows 5000000
echo "== 2 ranks on one GPU, strong (10 M rows dealt)"; run 2 29522 --same-gpu --scaling strong
echo "== 1 rank, RCCL communicator of one"; run 1 29523
echo "== plain index"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms per batch' % d['ms_per_step'])"
