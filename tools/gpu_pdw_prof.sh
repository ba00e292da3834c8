#!/bin/bash
out=gpurun_out/${1:-r2p}; mkdir -p $out; export PDW_OUT=$out
root=$(pwd); export TMPDIR=/tmp; cd /tmp
python3 $root/tools/pdw_bench.py 28 > $root/$out/pdw_bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -- python3 $root/tools/pdw_bench.py 28 > $root/$out/pdw_bench_prof.txt 2>&1
cd $root
cat $out/pdw_bench.txt | grep -v amdgpu
for f in $(find $out/prof -name "*kernel_stats.csv"); do cp $f $out/pdw_kernel_stats.csv; done
rm -rf $out/prof
python3 - <<'PY'
import csv,sys
import os; rows=list(csv.DictReader(open(os.environ.get('PDW_OUT','gpurun_out/r2p')+'/pdw_kernel_stats.csv')))
for r in rows:
    print("%-70.70s calls=%4s avg_us=%9.1f total_us=%10.1f" % (r['Name'], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
