#!/bin/bash
# PMC passes over tools/pdw_bench.py (config 5's matrix, channelized extraction only): HBM traffic and issue-side counters of
# the PDW kernels.  Every counter set in its own rocprofv3 run, with --kernel-trace only.
set -u
root=$(pwd); out=$root/gpurun_out/prof_pdw; mkdir -p $out
export TMPDIR=/tmp PDW_BENCH_SKIP_RAW=1; cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$i -- python3 $root/tools/pdw_bench.py 28 > $out/pmc_$i.log 2>&1
  echo "pmc pass $i: exit $?"
done
cd $root
python3 - "$out" <<'PY' | tee $out/summary.txt
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "pdw_" in n:
            k = n.split("pdw_")[1].split("(")[0].split("<")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("PDW kernels on config 5's matrix (4.29 GB, F = 4194304 x 128), per-launch averages; FETCH_SIZE doubled (gfx950), KB = 1024 B")
for k, c in acc.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}
    rd, wr = 2.0 * a.get("FETCH_SIZE", 0) * 1024, a.get("WRITE_SIZE", 0) * 1024
    line = f"  pdw_{k:24s} HBM read {rd / 1e9:7.3f} GB  write {wr / 1e9:7.3f} GB"
    if "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"] > 0:
        line += "   WAIT_INST_ANY/WAVE_CYCLES {:.2f}  WAIT_ANY/WAVE_CYCLES {:.2f}  VALU busy {:.2f}  LDS conflicts/active {:.2f}".format(
            a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"],
            a["SQ_ACTIVE_INST_VALU"] / (4.0 * max(a["SQ_BUSY_CYCLES"], 1)), a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1))
    print(line)
PY
find $out -mindepth 1 -maxdepth 1 ! -name summary.txt -exec rm -rf {} +
