#!/usr/bin/env python3
"""Counters of the scale sweep (bench.py --scale-sweep --launch-mode 4 --scales S under rocprofv3 --pmc): per size the HBM traffic of the sweep-carrying kernel
(k_iter: the whole iteration in one launch; k_sweep: the sweep launch of the two-launch structure) per live launch, next to the sweep's algorithmic bytes, and the
fp64 matrix-core rate over the launch.  usage: scale_summary.py DIR   (DIR holds scale_<S>_{fetch,write,mfma}.csv and scale_<S>.json)"""
import collections, csv, json, os, sys

def per_launch(path, names):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if any(n in r["Kernel_Name"] for n in names):
            acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {}
    for (k, c), v in acc.items():
        live = [x for x in v if x[0] > 0.25 * max(y[0] for y in v)]
        out[(k, c)] = (sum(x[0] for x in live) / len(live), sum(x[1] for x in live) / len(live) / 1e3, len(live))
    return out

d = sys.argv[1]
print("%5s %10s %12s %-22s %12s %12s %10s %10s %12s" % ("scale", "landmarks", "alg MB/sweep", "kernel", "FETCH MB(x2)", "WRITE MB", "traffic/alg", "launch us", "MFMA TFLOP/s"))
for sc in (1, 4, 16, 64):
    jp = os.path.join(d, "scale_%d.json" % sc)
    if not os.path.exists(jp): continue
    row = json.load(open(jp))["scale_sweep"][0]
    names = ("k_iter<", "k_sweep<")
    f = per_launch(os.path.join(d, "scale_%d_fetch.csv" % sc), names); w = per_launch(os.path.join(d, "scale_%d_write.csv" % sc), names)
    m = per_launch(os.path.join(d, "scale_%d_mfma.csv" % sc), names) if os.path.exists(os.path.join(d, "scale_%d_mfma.csv" % sc)) else {}
    for (k, c), (val, us, n) in sorted(f.items()):
        fe = 2.0 * 1024.0 * val; wr = 1024.0 * w.get((k, "WRITE_SIZE"), (0, 0, 0))[0]
        mo = m.get((k, "SQ_INSTS_VALU_MFMA_MOPS_F64"))
        tf = (512.0 * mo[0] / (mo[1] * 1e-6) / 1e12) if mo else float("nan")
        alg = row["algorithmic_bytes_per_sweep"]
        print("%5d %10d %12.2f %-22s %12.2f %12.2f %10.2f %10.1f %12.3f" % (sc, row["landmarks"], alg / 1e6, k[:22], fe / 1e6, wr / 1e6, (fe + wr) / alg, us, tf))
    print("      sweep phase %.1f us = %.0f GB/s = %.1f %% of the 8 TB/s roof; %d launches per iteration; %.0f it/s" % (row.get("sweep_phase_us", float("nan")), row.get("sweep_phase_GBps", float("nan")), 100 * row.get("sweep_phase_frac_of_hbm_peak", float("nan")), row["launches_per_iteration"], row["iterations_per_s"]))
