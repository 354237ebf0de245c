#!/usr/bin/env python3
"""GPU-side timeline of the last steps of a traced run: kernels and copies in start order with the gaps between them.
  rocprofv3 --kernel-trace --memory-copy-trace -d D -o r1 -- python bench.py --config c1 ...;  python tools/timeline.py D/r1_results.db [ops]"""
import sqlite3
import sys


def main(path, last=40):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    ops = [(s, e, n.split("(")[0][-60:], q) for s, e, n, q in
           db.execute("select d.start, d.end, s.kernel_name, d.queue_id from %s d join %s s on d.kernel_id = s.id" % (kd, ks))]
    mc = [t for t in tabs if t.startswith("rocpd_memory_copy")]
    if mc:
        cols = [r[1] for r in db.execute("pragma table_info(%s)" % mc[0])]
        size = "size" if "size" in cols else "0"
        ops += [(s, e, "COPY %s B" % b, -1) for s, e, b in db.execute("select start, end, %s from %s" % (size, mc[0]))]
    ops.sort()
    ops = ops[-last:]
    t0 = ops[0][0]
    prev_end = t0
    print("%10s %9s %9s  q  op" % ("start_us", "dur_us", "gap_us"))
    for s, e, n, q in ops:
        print("%10.1f %9.1f %9.1f %2s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, n))
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
