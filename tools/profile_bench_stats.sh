#!/bin/bash
# the default bench command under the kernel trace: the bench line and the per-kernel stats of the SAME run
set -u
root=$(pwd); out=$root/gpurun_out/prof_bench; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/run -- python3 $root/bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench default under rocprofv3: exit $?"
for f in $(find $out/run -name "*kernel_stats.csv"); do cp $f $out/bench_default_kernel_stats.csv; done
rm -rf $out/run
tail -c 600 $out/bench_default.json
