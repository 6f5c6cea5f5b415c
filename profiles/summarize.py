#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV and the FETCH_SIZE / WRITE_SIZE --pmc CSVs of a bench.py run.

`done`-flag no-op launches (the solver enqueues iterations in chunks of 6; once the device-side `done` flag is
set the remaining launches of a chunk return immediately, ~3-5 us) are separated from LIVE launches, because
bench.py's HIP-event figure (`roofline.avg_launch_us`) times live launches only.

usage: summarize.py kernel_trace.csv [pmc_fetch.csv pmc_write.csv]
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
        thr = 8.0 if k.startswith(("k_sweep", "k_reduce", "void k_step")) else 0.0
        live = [x for x in v if x > thr]
        print("%-44s %6d %10.2f %6d %10.2f %10.2f" % (k[:44], len(v), sum(v) / len(v), len(live), sum(live) / max(1, len(live)), max(v)))
    for f in sys.argv[2:]:
        acc = collections.defaultdict(list)
        name = None
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"])); name = r["Counter_Name"]
        for k, v in acc.items():
            if k.startswith(("k_sweep", "k_reduce", "void k_step")):
                live = [x for x in v if x > 0.25 * max(v)]
                print("%s %-40s live launches %4d  mean %.1f KB" % (name, k[:40], len(live), sum(live) / max(1, len(live))))


if __name__ == "__main__":
    main()
