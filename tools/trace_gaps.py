"""gaps between consecutive kernels of a rocprofv3 --kernel-trace csv, grouped by (previous kernel -> next kernel): where an iteration's time goes that no kernel accounts for"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-28:]) for r in rows))
g = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    g[(n0, n1)].append((s1 - e0) / 1e3)
print("%-30s %-30s %7s %9s %9s %9s" % ("previous", "next", "count", "median_us", "mean_us", "max_us"))
for (a, b), v in sorted(g.items(), key=lambda kv: -len(kv[1]))[:14]:
    v.sort()
    print("%-30s %-30s %7d %9.2f %9.2f %9.2f" % (a, b, len(v), v[len(v) // 2], sum(v) / len(v), v[-1]))
