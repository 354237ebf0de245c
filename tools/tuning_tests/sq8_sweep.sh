set -x
timeout 900 python tools/bench_sq8.py --metric L2 --batches 128 --steps 10 --sweep lowp_variant=0,1,2,4,5,6,7,8,9 2>&1 | grep "mfma 1"
timeout 300 python tools/bench_sq8.py --metric L2 --batches 128 --steps 4 --rows 5000000 --opt lowp_dbg=8 2>&1 | grep -i "phases\|mfma 1"
timeout 300 python tools/bench_sq8.py --metric L2 --batches 128 --steps 4 --rows 5000000 --opt lowp_dbg=1 2>&1 | grep -i "mfma 1"
timeout 300 python tools/bench_sq8.py --metric L2 --batches 128 --steps 4 --rows 5000000 --opt lowp_dbg=2 2>&1 | grep -i "mfma 1"
timeout 300 python tools/bench_sq8.py --metric L2 --batches 128 --steps 4 --rows 5000000 --opt lowp_dbg=3 2>&1 | grep -i "mfma 1"
timeout 300 python tools/bench_sq8.py --metric L2 --batches 128 --steps 4 --rows 5000000 --opt lowp_dbg=4 2>&1 | grep -i "mfma 1"
timeout 300 python tools/bench_sq8.py --metric L2 --batches 128 --steps 4 --rows 5000000 --opt wg_per_cu=2 --sweep lowp_variant=0,3 2>&1 | grep -i "mfma 1"
