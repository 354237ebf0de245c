#!/bin/bash
# round 5 closing soaks (bounded): filter paths against the exact path over random shapes (incl. wide rows), the HNSW batch iterator's
# walk against the oracle twin (with deletes: compactions are deferred while a walker lives, run between them), the sharded index
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
{
echo "## fuzz_parity 240 s"; timeout 400 python tools/fuzz_parity.py --seconds 240 --seed 501 2>&1 | tail -3
echo "## fuzz_parity --wide 180 s"; timeout 400 python tools/fuzz_parity.py --seconds 180 --seed 502 --wide 2>&1 | tail -3
echo "## fuzz_parity --readers 2, 120 s"; timeout 300 python tools/fuzz_parity.py --seconds 120 --seed 503 --readers 2 2>&1 | tail -3
echo "## fuzz_hnsw_iter 240 s"; timeout 400 python tools/fuzz_hnsw_iter.py --seconds 240 --seed 504 2>&1 | tail -3
echo "## fuzz_sharded 180 s"; timeout 400 python tools/fuzz_sharded.py --seconds 180 --seed 505 2>&1 | tail -3
} | tee $O/soak.txt
