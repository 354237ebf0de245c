#!/bin/bash
# round 6: (1) filter selectivity on non-uniform rows (configs 2 and 4), (2) PMC rows of the REAL config-3 / config-4 scan kernels
# (effective clock = GRBM_GUI_ACTIVE / wall, MFMA-busy, waits), separate rocprofv3 passes with --kernel-trace only
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06e
mkdir -p $OUT
cd $R
for c in c2 c4; do
  for d in uniform lowrank clustered; do
    timeout 900 python bench.py --config $c --data $d --steps 20 --warmup 10 --no-cpu-baseline --no-shard-curve > $OUT/sel_${c}_$d.json 2> $OUT/sel_${c}_$d.err
    tail -1 $OUT/sel_${c}_$d.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("'$c' '$d'", "%.3f ms/step kernel %.3f cand/q %.0f fallbacks %d retries %d parity %s" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["candidates_per_query"], d["fallbacks"], d["retries"], d.get("full_table_parity")))'
  done
done
cd /tmp && export TMPDIR=/tmp
for c in c3 c4; do
  for SET in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
    rm -rf /tmp/pmc_$c
    timeout 300 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_$c -- python $R/bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-full-parity > /tmp/pmc_run.log 2>&1
    echo "== $c : $SET"; python $R/tools/pmc_dump.py /tmp/pmc_$c "filter" | grep -v PROBE | cut -c1-200; tail -1 /tmp/pmc_run.log | cut -c1-160
  done
done > $OUT/pmc_c3_c4.txt 2>&1
cat $OUT/pmc_c3_c4.txt | tail -70
