#!/bin/bash
out=gpurun_out/${1:-sm}; mkdir -p $out
root=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -- python3 $root/tools/pdw_small_m.py 8 16 32 56 64 > $root/$out/wall.txt 2>&1
cd $root
grep "M=" $out/wall.txt
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pdw_' in r['Name']:
        print("   %-60.60s calls=%3s avg_us=%9.1f total_us=%10.1f" % (r['Name'].replace('(anonymous namespace)::','').split('(')[0], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs'])/1e3))
PY
rm -rf $out/prof
