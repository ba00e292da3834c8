#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the default kernel of every fused shape (run on the GPU box from the repo root)
root=$(pwd); export TMPDIR=/tmp; cd /tmp
for w in ${W:-64,12,64,int16,12 64,12,64,int8,8 256,8,256,int8,8 1024,16,1024,int16,12 128,12,64,int16,12 56,12,56,int16,12 560,12,560,int16,12 32,12,32,int16,12 16,12,16,int16,12 8,12,8,int16,12 8,12,8,cf32,0 10,12,10,int16,12 20,12,20,int16,12 40,12,40,int16,12}; do
  d=$root/gpurun_out/prof_lds; rm -rf $d
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $d -- python3 $root/tools/ab.py --log2-samples 27 --rounds 1 --steps 2 --workload $w $* "default:" > /dev/null 2>&1
  python3 - $d $w <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "pfb_" in k and "init_tables" not in k and "update_history" not in k and "stream_copy" not in k:
        acc[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    c = {n: sum(v) / len(v) for n, v in d.items()}
    print(f"{sys.argv[2]:24s} {k:50s} conflict cycles {c['SQ_LDS_BANK_CONFLICT']:12.0f} = {100 * c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):5.1f} % of LDS-active {c['SQ_LDS_IDX_ACTIVE']:12.0f}")
PY
done
