#!/bin/bash
# Issue-side PMC passes over tools/ab.py cases: tools/pmc_ab.sh <tag> <ab.py args...>
# (each counter set in its own rocprofv3 run with --kernel-trace only; python3 directly behind `--`)
set -u
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/pmc_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $root/tools/ab.py "$@" > $out/p$i.log 2>&1
  echo "pass $i exit $?"
done
cd $root
python3 - $out <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "pfb_" not in name or "init_tables" in name or "update_history" in name or "copy" in name:
            continue
        key = (name[:60] + "|" + name[-60:], row.get("Grid_Size", ""), row.get("VGPR_Count", ""), row.get("LDS_Block_Size", ""))
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key, cs in acc.items():
    print("==", key)
    g = {c: sum(v) / len(v) for c, v in cs.items()}
    for c in sorted(g):
        print(f"   {c:32s} {g[c]:18.1f}  (n={len(cs[c])})")
    wc = g.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in g: print(f"   {c}/WAVE_CYCLES = {g[c] / wc:.3f}")
    if g.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   LDS conflict / active = {g.get('SQ_LDS_BANK_CONFLICT', 0) / g['SQ_LDS_IDX_ACTIVE']:.3f}")
    if g.get("SQ_BUSY_CYCLES") and g.get("SQ_ACTIVE_INST_VALU"):
        print(f"   ACTIVE_INST_VALU / (4 BUSY_CYCLES) = {g['SQ_ACTIVE_INST_VALU'] / (4 * g['SQ_BUSY_CYCLES']):.3f}")
PY
