# SQ8 filter variants (make TUNING=1 build): tools/tuning_tests/sq8_sweep.sh
timeout 900 python tools/bench_sq8.py --metric L2 --batches 128 --steps 10 --sweep lowp_variant=0,3,10,1,2 2>&1 | grep "mfma 1"
timeout 600 python tools/bench_sq8.py --metric IP --batches 64,128 --steps 10 2>&1 | grep "mfma 1"
timeout 600 python -m pytest tests/test_gpu_sq8.py tests/test_gpu_sq8_centred.py -m gpu -x -q 2>&1 | tail -3
