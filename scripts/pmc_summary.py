"""Summarise rocprofv3 outputs written by scripts/profile.sh (per-kernel averages)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
ks = os.path.join(out, "kernel_stats.csv")
if os.path.exists(ks):
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for row in csv.DictReader(open(ks)):
        print("  {Name:60.60s} calls={Calls:>5s} avg_ns={AverageNs:>12s} total_ns={TotalDurationNs:>14s} pct={Percentage}".format(**row))
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:48]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
    print(f"== {os.path.basename(d)} (per-dispatch averages) ==")
    for k, cs in agg.items():
        n = max(1, len(calls[k]))
        if not any(t in k for t in ("gather", "reorder", "evaluate", "traverse", "plan_kernel")):
            continue
        print("  ", k, f"dispatches={n}")
        for c, v in sorted(cs.items()):
            print(f"      {c:28s} {v / n:18.1f}")
