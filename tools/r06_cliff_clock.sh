#!/bin/bash
# effective shader clock of every scan launch after an idle second: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the launch's duration
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cc
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/cc -o r1 -- python $R/bench.py --config ${1:-c4} --steps 14 --warmup 0 --readers 1 --no-cpu-baseline --no-full-parity > /tmp/cc.log 2>&1
python - <<'PY'
import glob, sqlite3
db = sqlite3.connect(glob.glob("/tmp/cc/**/*_results.db", recursive=True)[0])
rows = list(db.execute("select start, end, kernel_name, counter_name, value from counters_collection order by start"))
print("launch  dur_us  GRBM_GUI_ACTIVE  effective_GHz")
i = 0
for s, e, k, c, v in rows:
    if (e - s) > 1_000_000 and "fill" not in k and "aux" not in k and "norm" not in k:
        print("%4d  %7.0f  %14.0f  %.3f   %s" % (i, (e - s) / 1e3, v, v / 8 / (e - s), k[:40]))
        i += 1
PY
