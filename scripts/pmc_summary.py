"""Summarise rocprofv3 outputs written by scripts/profile.sh (per-kernel averages)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for name, title in (("kernel_stats.csv", "kernel stats, default pipeline (build, traversal and evaluation streams) (rocprofv3 --kernel-trace --stats)"),
                    ("kernel_stats_single_stream.csv", "kernel stats, GVPM_PIPELINE=0 (single stream: isolated kernel durations)")):
    ks = os.path.join(out, name)
    if os.path.exists(ks):
        print(f"== {title} ==")
        for row in list(csv.DictReader(open(ks)))[:16]:
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
        if not any(t in k for t in ("gather", "reorder", "evaluate", "traverse", "plan_kernel", "reconnect", "chain_", "vpm_")):
            continue
        print("  ", k, f"dispatches={n}")
        for c, v in sorted(cs.items()):
            print(f"      {c:28s} {v / n:18.1f}")

# HBM traffic per launch of the G-BRE kernels as MI355X_MICROARCH.md prescribes: FETCH_SIZE (KB) counts a
# 128-byte request as 64 bytes on gfx950 -> doubled; WRITE_SIZE (KB) as reported.
import json
traffic = {}
def per_kernel(d, counter):
    res = {}
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        tot, calls = defaultdict(float), defaultdict(set)
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                k = row["Kernel_Name"]
                tot[k] += float(row["Counter_Value"])
                calls[k].add(row["Dispatch_Id"])
        for k in tot:
            res[k] = tot[k] / max(1, len(calls[k]))
    return res
fetch, write = per_kernel("pmc3", "FETCH_SIZE"), per_kernel("pmc4", "WRITE_SIZE")
for k in fetch:
    short = k.split("(")[0].split("::")[-1].split("<")[0]
    traffic[short] = {"fetch_kb_raw": fetch[k], "write_kb_raw": write.get(k, 0.0),
                      "hbm_bytes_per_launch": (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0}
# what the measurement was taken on (bench.py reports `traffic` only for the same workload and kernel sources)
try:
    line = json.loads([l for l in open(os.path.join(out, "bench_trace.log")).read().splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    traffic["_meta"] = {"technique": cfg["technique"], "scene": cfg["scene"], "frame": cfg["frame"],
                        "photons": cfg.get("photons_per_iter", cfg.get("records_per_iter")), "scale": float(cfg["workload"].rsplit(" ", 1)[-1]),
                        "n_gpus": line["n_gpus"], "csrc_sha": cfg["csrc_sha"]}
except (OSError, ValueError, KeyError, IndexError) as e:
    print("no bench line to take the workload from:", e)
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1, sort_keys=True)
print("== HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, bytes) ==")
for k, v in sorted(((k, v) for k, v in traffic.items() if k != "_meta"), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:10]:
    print(f"   {k:40s} {v['hbm_bytes_per_launch'] / 1e6:12.2f} MB")
