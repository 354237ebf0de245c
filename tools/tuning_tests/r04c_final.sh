cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r04c_gputest.log
bash tools/profile_round.sh r04c all > gpurun_out/r04c_profile.log 2>&1
timeout 300 python tools/fuzz_parity.py --seconds 150 --seed 51 2>&1 | tail -2 > gpurun_out/r04c_fuzz.txt
cat gpurun_out/r04c_gputest.log gpurun_out/r04c_fuzz.txt; tail -3 gpurun_out/r04c_profile.log | cut -c1-600
