#!/bin/bash
out=gpurun_out/${1:-mall}; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $root/$out/prof -- python3 $root/tools/pdw_mall_probe.py > $root/$out/wall.txt 2>&1
cd $root
grep "F=2" $out/wall.txt
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'pdw_bracket_kernel' in r['Kernel_Name']]
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
# 13 calls per size (3 warm-up + 10), three sizes in order
for i, name in enumerate(("2^19 (134 MB)", "2^20 (268 MB)", "2^23 (2.1 GB)")):
    seg = d[13 * i + 3: 13 * (i + 1)]
    if seg:
        nbytes = (1 << (19, 20, 23)[i]) * 32 * 8
        avg = sum(seg) / len(seg)
        print(f"bracket pass, F={name}: {avg / 1e3:.1f} us = {nbytes / avg:.2f} GB/ms... {nbytes / avg / 1e3:.2f} TB/s")
PY
rm -rf $out/prof
