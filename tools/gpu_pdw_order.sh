#!/bin/bash
# PDW stage A/B over an environment knob: gpu_pdw_order.sh OUT VAR v1 v2 ...  (prints the edge-stage kernels too)
out=gpurun_out/${1:-r2o}; mkdir -p $out; var=$2; shift 2
root=$(pwd); export TMPDIR=/tmp; cd /tmp
for o in "$@"; do
  export $var=$o
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof$o -- python3 $root/tools/pdw_bench.py 28 > $root/$out/bench$o.txt 2>&1
  f=$(find $root/$out/prof$o -name "*kernel_stats.csv" | head -1)
  echo "$var=$o: $(grep 'PDW extraction: F' $root/$out/bench$o.txt | tail -1)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'pdw_' in n and 'Raw' not in n and 'raw' not in n:
        print("   %-40.40s calls=%3s avg_us=%8.1f" % (n.split('::')[-1].split('(')[0], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -rf $root/$out/prof$o
done
