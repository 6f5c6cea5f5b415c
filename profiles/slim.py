#!/usr/bin/env python3
"""Keep only the rows of the solver's own kernels in a rocprofv3 CSV (kernel trace or counter collection): the bench.py command also
runs the tracker-driven leg, the marginalisation that builds the prior, torch's fill / copy kernels ... which multiply the file size; bookkeeping columns (ids of agent / queue / thread ...) are dropped as well.
usage: slim.py in.csv out.csv [name-fragment ...]   (default fragments: k_sweep k_reduce k_step k_finish k_marg)"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
frags = sys.argv[3:] or ["k_sweep", "k_reduce", "k_step", "k_finish", "k_marg"]
with open(src) as f, open(dst, "w", newline="") as g:
    r = csv.reader(f); w = csv.writer(g, quoting=csv.QUOTE_ALL)
    head = next(r)
    drop = {"Correlation_Id", "Agent_Id", "Queue_Id", "Stream_Id", "Process_Id", "Thread_Id", "Kernel_Id", "Kind", "Workgroup_Size_Y", "Workgroup_Size_Z", "Grid_Size_Y", "Grid_Size_Z", "Accum_VGPR_Count"}
    keep = [i for i, h in enumerate(head) if h not in drop]
    w.writerow([head[i] for i in keep])
    ki = head.index("Kernel_Name")
    for row in r:
        if any(fr in row[ki] for fr in frags):
            w.writerow([row[i] for i in keep])
