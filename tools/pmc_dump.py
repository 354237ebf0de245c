#!/usr/bin/env python3
"""Per-kernel means of every counter in a rocprofv3 --pmc rocpd database: pmc_dump.py <dir> [kernel-substring]"""
import glob
import sqlite3
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else ""
for dbf in sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)):
    d = sqlite3.connect(dbf)
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration)/1000.0 from counters_collection "
         "group by kernel_name, counter_name order by kernel_name, counter_name")
    for k, c, n, v, dur in d.execute(q):
        if pat in k:
            print("%-75s %-28s n=%-4d mean=%-18.1f dur_us=%.1f" % (k[:75], c, n, v, dur))
