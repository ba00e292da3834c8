#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the default kernel of each bench workload (run on the GPU box from the repo root)
root=$(pwd); export TMPDIR=/tmp; cd /tmp
for w in cfg2 cfg3 cfg4 cfg5 ref56 ref560; do
  rm -rf $root/gpurun_out/prof_lds_$w
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $root/gpurun_out/prof_lds_$w -- python3 $root/bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $w --log2-samples 28 $* > /dev/null 2>&1
  python3 - $root/gpurun_out/prof_lds_$w $w <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "pfb_" in k and "init_tables" not in k and "update_history" not in k and "stream_copy" not in k:
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(sys.argv[2], k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
