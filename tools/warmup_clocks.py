#!/usr/bin/env python3
"""Companion of warmup_idle.py: sample the SMU's clock levels (sysfs pp_dpm_*) at ~1 kHz while config 4's batches run after an idle
second, and print, per batch, the scan kernel's time next to the clocks seen during it."""
import glob
import sys
import threading
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vectorsimilarity_amd import VecSim, synth  # noqa: E402

dev = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
base = dev[0].rsplit("/", 1)[0] if dev else None
names = ["pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"]
if base:
    for n in names:
        try:
            print("==", n, open(base + "/" + n).read().replace("\n", " | "))
        except OSError as e:
            print("==", n, "unreadable:", e)


def current(n):
    try:
        for line in open(base + "/" + n):
            if "*" in line:
                return line.split(":")[1].replace("*", "").strip()
    except OSError:
        pass
    return "?"


samples, stop = [], False


def sampler():
    while not stop:
        samples.append((time.perf_counter(), current("pp_dpm_sclk"), current("pp_dpm_mclk"), current("pp_dpm_fclk")))
        time.sleep(0.0005)


p = VecSim.BFParams()
p.type, p.dim, p.metric = VecSim.VecSimType_BFLOAT16, 768, VecSim.VecSimMetric_IP
ix = VecSim.BFIndex(p)
ix.add_synthetic(12_500_000, 47)
q = synth.rows_bf16(48, 0, 128, 768)
for _ in range(12):
    ix.knn_query(q, 10)
time.sleep(1.0)
th = threading.Thread(target=sampler)
th.start()
time.sleep(0.05)
marks = []
for b in range(10):
    ix.reset_stats()
    t0 = time.perf_counter()
    ix.knn_query(q, 10)
    marks.append((t0, time.perf_counter(), ix.stats()["scan_ms"]))
stop = True
th.join()
print("idle before the batches:", sorted(set(s[1:] for s in samples if s[0] < marks[0][0])))
for b, (t0, t1, ms) in enumerate(marks):
    seen = sorted(set(s[1:] for s in samples if t0 <= s[0] <= t1))
    print("batch %d: scan %.0f us   (sclk, mclk, fclk) seen: %s" % (b, ms * 1e3, seen))
