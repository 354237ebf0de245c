#!/usr/bin/env python3
"""Turns rocprofv3 rocpd (sqlite) outputs under gpurun_out/ into the small text summaries kept in
profiles/.   python profiles/summarize.py gpurun_out/prof_r1 profiles/r01"""
import sqlite3
import sys


def main(src, dst):
    out = []
    db = sqlite3.connect(f"{src}/trace/r1_results.db")
    out.append("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline")
    out.append("%-78s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        out.append("%-78s %8d %14.1f %12.2f %7.2f" % (name[:78], calls, total, avg, pct))
    # the dominant kernel launch by launch: rocprofv3's mean includes the warm-up launches (first passes over a fresh table: cold
    # TLBs), bench.py's HIP-event mean only the timed ones -- round 3's config-4 lines differed by 6 % for exactly that reason
    try:
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
        rows = list(db.execute("select s.kernel_name, d.end - d.start from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
        tot = {}
        for n, d in rows:
            tot[n] = tot.get(n, 0) + d
        top = max(tot, key=tot.get)
        durs = [d / 1e3 for n, d in rows if n == top]
        out.append("# dominant kernel, every launch in order (us): " + " ".join("%.0f" % d for d in durs))
        half = durs[len(durs) // 2:]
        out.append("# mean of all %d launches %.1f us; of the last %d (steady state) %.1f us" % (len(durs), sum(durs) / len(durs), len(half), sum(half) / len(half)))
    except Exception as e:   # older rocprofv3 schemas
        out.append("# (no per-launch table: %s)" % e)
    open(dst + "_kernel_stats.txt", "w").write("\n".join(out) + "\n")
    import os
    if not os.path.exists(f"{src}/pmc_fetch/r1_results.db"):   # a kernel-stats-only directory (other configs)
        print("\n".join(out))
        return
    out = ["# rocprofv3 --pmc <counter> --kernel-trace -- python bench.py --steps 5 --warmup 1 (separate passes)",
           "# FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md §HBM)",
           "%-50s %-12s %8s %16s %16s" % ("kernel", "counter", "launches", "mean_value_KB", "mean_dur_us")]
    for sub in ("pmc_fetch", "pmc_write"):
        d = sqlite3.connect(f"{src}/{sub}/r1_results.db")
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration)/1000.0 from counters_collection "
             "group by kernel_name, counter_name order by avg(value) desc")
        for k, cname, n, v, dur in d.execute(q):
            out.append("%-50s %-12s %8d %16.1f %16.2f" % (k[:50], cname, n, v, dur))
    open(dst + "_pmc.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
