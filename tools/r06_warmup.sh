#!/bin/bash
# round 6: the first launches of the low-precision filters run 15-30 % slow -- in which circumstances?  Per-launch durations of the scan
# kernel from rocprofv3's kernel trace, in launch order, no warm-up batches, one reader.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
durs() {  # name, env..., -- bench args
  local name=$1; shift
  rm -rf /tmp/wu_$name
  env "$@" rocprofv3 --kernel-trace -d /tmp/wu_$name -o r1 -- python $R/bench.py --steps 24 --warmup 0 --readers 1 --no-cpu-baseline --no-full-parity --no-shard-curve $BENCH_ARGS > /tmp/wu_$name.log 2>&1
  echo "== $name ($BENCH_ARGS $*)"
  python - /tmp/wu_$name <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
big = [(s, e, n) for s, e, n in rows if (e - s) > 1_000_000 and "fill" not in n and "aux" not in n and "norm" not in n]
print("scan launches (us):", " ".join("%d" % ((e - s) / 1e3) for s, e, n in big))
print("gap before each (ms):", " ".join("%.1f" % ((big[i][0] - big[i - 1][1]) / 1e6) for i in range(1, len(big))))
PY
}
BENCH_ARGS="--config c4" durs c4_plain A=1
BENCH_ARGS="--config c4" durs c4_onelane VECSIM_GPU_READER_LANES=1
BENCH_ARGS="--config c4 --opt events=0" durs c4_noevents A=1
BENCH_ARGS="--config c2" durs c2_plain A=1
BENCH_ARGS="--config c3" durs c3_plain A=1
BENCH_ARGS="--config c4 --rows 3000000" durs c4_small A=1
