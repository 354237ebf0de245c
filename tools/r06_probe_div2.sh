cd ${GRAFT_REPO_ROOT:-.}
line() { python -c 'import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("%8.4f ms/step  kernel %.4f  fixed %.4f  cand/q %.0f fallbacks %d retries %d" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["fixed_ms_per_batch"], d["candidates_per_query"], d["fallbacks"], d["retries"]))'; }
run() { echo "== $*"; timeout 200 python bench.py --no-cpu-baseline --no-shard-curve --no-full-parity "$@" 2>&1 | line; }
for rep in 1 2; do for pd in 0 64 80 96; do run --config c2 --steps 40 --warmup 5 --opt probe_div=$pd; done; done
for rep in 1 2; do for pd in 0 12 24 32; do run --config c2 --rows 1250000 --steps 100 --warmup 10 --opt probe_div=$pd; done; done
for pd in 0 48 64; do run --config c2 --data clustered --steps 20 --warmup 5 --opt probe_div=$pd; done
