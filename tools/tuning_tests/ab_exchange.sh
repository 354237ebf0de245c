#!/bin/bash
# the cost of the exchange on a communicator of one: plain index against `torch.distributed.run --nproc-per-node 1` (RCCL all-gather
# per batch, sequence-ordered, two reader threads), same box, alternating.  Output: gpurun_out/ab_exchange.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/ab_exchange.txt
: > $OUT
show() { grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
pr = d.get('per_rank_ms_per_batch')
print('%-22s step %.3f ms  kernel %.3f ms  fixed %.3f %s' % ('$1', d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['fixed_ms_per_batch'], ('turn_wait %.3f exchange %.3f merge %.3f' % (pr[0]['turn_wait_ms'], pr[0]['exchange_ms'], pr[0]['merge_ms'])) if pr else ''))" >> $OUT; }
for cfg in c2 c4; do
  extra=""; [ $cfg = c4 ] && extra="--steps 30"
  for i in 1 2; do
    python $R/bench.py --config $cfg $extra --no-cpu-baseline 2>/dev/null | show "$cfg plain"
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) $R/bench.py --gpus 1 --config $cfg $extra --no-cpu-baseline "$@" 2>/dev/null | show "$cfg rccl x1 $*"
  done
done
cat $OUT
