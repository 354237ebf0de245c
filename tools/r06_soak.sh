#!/bin/bash
# round 6 closing soaks (bounded) on the final code: the filter paths against the exact path over random shapes (the per-batch chain
# changed: upload kernel, select into the pinned block), under concurrent readers, the sharded index, the HNSW iterator walk,
# and the new reference-order HNSW insert path against the oracle
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06s
mkdir -p $O
cd $R
{
echo "## fuzz_parity 240 s"; timeout 400 python tools/fuzz_parity.py --seconds 240 --seed 701 2>&1 | tail -3
echo "## fuzz_parity --readers 2, 240 s"; timeout 400 python tools/fuzz_parity.py --seconds 240 --seed 702 --readers 2 2>&1 | tail -3
echo "## fuzz_parity --wide --readers 3, 180 s"; timeout 400 python tools/fuzz_parity.py --seconds 180 --seed 703 --wide --readers 3 2>&1 | tail -3
echo "## fuzz_parity --stream --readers 2, 240 s"; timeout 500 python tools/fuzz_parity.py --seconds 240 --seed 707 --stream --readers 2 2>&1 | tail -3
echo "## fuzz_sharded 180 s"; timeout 400 python tools/fuzz_sharded.py --seconds 180 --seed 704 2>&1 | tail -3
echo "## fuzz_hnsw_iter 180 s"; timeout 400 python tools/fuzz_hnsw_iter.py --seconds 180 --seed 705 2>&1 | tail -3
echo "## fuzz_hnsw_build 300 s"; timeout 500 python tools/fuzz_hnsw_build.py --seconds 300 --seed 706 2>&1 | tail -6
echo "## stress_flat"; for kd in f32 bf16 i8; do timeout 300 python tools/stress_flat.py 300 $kd 2>&1 | tail -2; done
} | tee $O/soak.txt
