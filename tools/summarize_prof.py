"""Summarise rocprofv3 CSV output of tools/profile.sh into a short text report."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


print(f"# rocprofv3 summary for {root}")
for f in find("*kernel_stats.csv"):
    if "/trace/" not in f:
        continue
    print(f"\n## kernel stats ({os.path.relpath(f, root)})")
    for row in csv.DictReader(open(f)):
        print("  {Name:70.70s} calls={Calls:>4s} total_ns={TotalDurationNs:>12s} avg_ns={AverageNs:>12s} pct={Percentage}".format(**row))

for f in find("*kernel_trace.csv"):
    if "/trace/" not in f:
        continue
    d = defaultdict(list)
    meta = {}
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        d[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[name] = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    print("\n## kernel trace durations")
    for name, v in d.items():
        v2 = sorted(v)
        print(f"  {name[:70]:70s} n={len(v):3d} avg_us={sum(v)/len(v)/1e3:10.2f} min_us={v2[0]/1e3:10.2f} max_us={v2[-1]/1e3:10.2f} {meta[name]}")

print("\n## PMC counters (per dispatch averages for pfb kernels)")
for f in find("*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "pfb" not in name:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, cs in acc.items():
        for c, v in cs.items():
            print(f"  {name[:48]:48s} {c:24s} avg={sum(v)/len(v):18.1f} n={len(v)}")
