#!/bin/bash
# round 5 closing soaks (bounded), on the final code: filter paths against the exact path over random shapes (incl. wide rows, alone and
# under concurrent readers), the HNSW batch iterator's walk against the oracle twin (with deletes: compactions are deferred while a
# walker lives, run between them), the sharded index (its merge reads the exchange records in place and may run on several threads)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
{
echo "## fuzz_parity 300 s"; timeout 500 python tools/fuzz_parity.py --seconds 300 --seed 601 2>&1 | tail -3
echo "## fuzz_parity --wide 300 s"; timeout 500 python tools/fuzz_parity.py --seconds 300 --seed 602 --wide 2>&1 | tail -3
echo "## fuzz_parity --wide --readers 3, 240 s"; timeout 400 python tools/fuzz_parity.py --seconds 240 --seed 603 --wide --readers 3 2>&1 | tail -3
echo "## fuzz_parity --readers 2, 180 s"; timeout 400 python tools/fuzz_parity.py --seconds 180 --seed 604 --readers 2 2>&1 | tail -3
echo "## fuzz_hnsw_iter 240 s"; timeout 400 python tools/fuzz_hnsw_iter.py --seconds 240 --seed 605 2>&1 | tail -3
echo "## fuzz_sharded 240 s"; timeout 400 python tools/fuzz_sharded.py --seconds 240 --seed 606 2>&1 | tail -3
} | tee $O/soak.txt
