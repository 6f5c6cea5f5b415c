#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV and the FETCH_SIZE / WRITE_SIZE --pmc CSVs of a bench.py run.

`done`-flag no-op launches (the solver enqueues iterations in chunks of 6; once the device-side `done` flag is
set the remaining launches of a chunk return immediately, ~3-5 us) are separated from LIVE launches, because
bench.py's HIP-event figure (`roofline.avg_launch_us`) times live launches only.

usage: summarize.py kernel_trace.csv [pmc_fetch.csv pmc_write.csv [pmc_mfma.csv]]
(the MFMA file holds SQ_INSTS_VALU_MFMA_MOPS_F64 / SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES of one --pmc pass)
"""
import collections
import csv
import sys


def main():
    tr = sys.argv[1]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(tr)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("%-44s %6s %10s %6s %10s %10s" % ("kernel", "calls", "avg_us", "live", "live_avg", "live_max"))
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        own = any(f in k for f in ("k_sweep", "k_reduce", "k_step<", "k_iter<", "k_solve<"))      # (k_solve: round 6, the whole solve as one resident launch)
        own_dummy = 0      # (k_sweep is a template since round 3: "void k_sweep<true>(...)"; k_iter: round 5, the one-launch iteration)
        thr = (8.0 if ("k_sweep" in k or "k_step<" in k or "k_iter<" in k or "k_solve<" in k) else 5.0) if own else 0.0
        live = [x for x in v if x > thr]
        print("%-44s %6d %10.2f %6d %10.2f %10.2f" % (k[:44], len(v), sum(v) / len(v), len(live), sum(live) / max(1, len(live)), max(v)))
        if "k_iter<" in k and live:
            # a solve = N full iterations + ONE launch that sweeps and judges the last candidate and ends the solve (no dense solve behind the judge): two populations
            full = [x for x in live if x > 0.75 * max(live)]; judge = [x for x in live if x <= 0.75 * max(live)]
            print("    of which full iterations %d, avg %.2f us; judge-only (the last launch of a solve) %d, avg %.2f us" % (len(full), sum(full) / max(1, len(full)), len(judge), sum(judge) / max(1, len(judge))))
    for f in sys.argv[2:4]:
        acc = collections.defaultdict(list)
        name = None
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"])); name = r["Counter_Name"]
        for k, v in acc.items():
            if any(f in k for f in ("k_sweep", "k_reduce", "k_step<", "k_iter<", "k_solve<")):
                live = [x for x in v if x > 0.25 * max(v)]
                print("%s %-40s live launches %4d  mean %.1f KB" % (name, k[:40], len(live), sum(live) / max(1, len(live))))
    if len(sys.argv) > 4:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(sys.argv[4])):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in acc.items():
            if ("k_sweep" in k or "k_iter<" in k or "k_solve<" in k) and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                # the Schur contraction of the landmark block: every visual workgroup of the sweep (one per compute unit) on the fp64 matrix cores
                d = [x[1] for x in v["SQ_VALU_MFMA_BUSY_CYCLES"]]
                live = [i for i, x in enumerate(d) if x > 0.5 * max(d)]
                mops = sum(v["SQ_INSTS_VALU_MFMA_MOPS_F64"][i][0] for i in live) / len(live)
                busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"][i][0] for i in live) / len(live)
                us = sum(d[i] for i in live) / len(live) / 1e3
                print("MFMA %-36s live launches %4d  %.0f MOPS x 512 fp64 flop = %.2f MFLOP per launch (= %.0f v_mfma_f64_16x16x4), MFMA busy %.0f cycles summed over the SIMDs, launch %.1f us"
                      % (k[:36], len(live), mops, mops * 512 / 1e6, mops / 4, busy, us))
                print("     -> %.3f TFLOP/s over the launch = %.2f %% of the 78.6 TFLOP/s fp64-matrix peak of the chip" % (mops * 512 / us / 1e6, 100 * (mops * 512 / us / 1e6) / 78.6))
                continue
            if not k.startswith("void k_step"):
                continue
            d = [x[1] for x in v["SQ_VALU_MFMA_BUSY_CYCLES"]]
            live = [i for i, x in enumerate(d) if x > 0.5 * max(d)]
            mops = sum(v["SQ_INSTS_VALU_MFMA_MOPS_F64"][i][0] for i in live) / len(live)
            busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"][i][0] for i in live) / len(live)
            us = sum(d[i] for i in live) / len(live) / 1e3
            print("MFMA %-36s live launches %4d  %.0f x 512 fp64 flop = %.2f MFLOP per launch, MFMA busy %.0f cycles (= %.0f v_mfma_f64_16x16x4 x 64), launch %.1f us"
                  % (k[:36], len(live), mops, mops * 512 / 1e6, busy, busy / 64, us))
            print("     -> %.1f GFLOP/s on the one CU that runs the step = %.3f %% of the 78.6 TFLOP/s fp64-matrix peak of the chip; MFMA pipes busy %.1f %% of the launch on that CU (4 SIMDs)"
                  % (mops * 512 / us / 1e3, 100 * (mops * 512 / us / 1e3) / 78600.0, 100 * busy / 4 / (us * 2400)))


if __name__ == "__main__":
    main()
