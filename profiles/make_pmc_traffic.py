#!/usr/bin/env python3
"""HBM bytes per launch of the dominant scan kernel of every bench config, from the rocprofv3 PMC passes tools/profile_round.sh
collected (separate --pmc FETCH_SIZE / WRITE_SIZE runs with --kernel-trace only): prints the JSON bench.py reads as
profiles/pmc_traffic.json.      python profiles/make_pmc_traffic.py gpurun_out/prof_<tag> > profiles/pmc_traffic.json
FETCH_SIZE is doubled (gfx950 reports half the bytes of a wide coalesced stream, MI355X_MICROARCH.md HBM section)."""
import json
import os
import sqlite3
import sys

# bench scan-kernel name -> (sub-directory of the profile run, substring of the kernel's name, workload bench.py checks)
CONFIGS = {
    "k_mfma_filter": ("", "k_mfma_filter<", {"rows": 10_000_000, "dim": 768, "batch": 64}),
    "k_i8_filter_x32": ("cfg_c3", "k_i8_filter_x32<", {"rows": 50_000_000, "dim": 1024, "batch": 256}),
    "k_i8_filter_x32l": ("cfg_c3", "k_i8_filter_x32l<", {"rows": 50_000_000, "dim": 1024, "batch": 256}),
    "k_mfma_filter_lowp(h16)": ("cfg_c4", "k_mfma_filter_lowp<", {"rows": 12_500_000, "dim": 768, "batch": 128}),
}


def per_kernel(db, pat, counter):
    if not os.path.exists(db):
        return {}
    d = sqlite3.connect(db)
    rows = list(d.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
                          "group by kernel_name", (counter,)))
    return {r[0]: r for r in rows if pat in r[0]}


def main(src):
    out = {}
    for name, (sub, pat, workload) in CONFIGS.items():
        base = os.path.join(src, sub) if sub else src
        fetch = per_kernel(os.path.join(base, "pmc_fetch", "r1_results.db"), pat, "FETCH_SIZE")
        if not fetch:
            continue
        # the filter instance is the one that FETCHES the most (the probe instance of the same template reads 1/48 of the
        # rows but writes more); its WRITE_SIZE is looked up under the same full kernel name, not as a maximum
        f = max(fetch.values(), key=lambda r: r[2])
        w = per_kernel(os.path.join(base, "pmc_write", "r1_results.db"), pat, "WRITE_SIZE").get(f[0])
        fetch_kb, write_kb = f[2], (w[2] if w else 0.0)
        out[name] = {"workload": workload, "fetch_size_kb_mean": round(fetch_kb, 1), "write_size_kb_mean": round(write_kb, 1),
                     "bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024), "launches": f[1],
                     "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace (tools/profile_round.sh); "
                            "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of a wide coalesced stream); "
                            "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024"}
    print(json.dumps(out, indent=2))


if __name__ == "__main__":
    main(sys.argv[1])
