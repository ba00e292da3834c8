#!/bin/bash
# The round's profile set (run on the GPU box: `bash tools/profile_round.sh r03`; every counter set in its own rocprofv3
# run, never together with a trace domain other than --kernel-trace).  Output under gpurun_out/prof_<round>/keep,
# summarised by tools/summarize_pmc.py; copy what is judged into profiles/<round>_*.
set -u
rnd=${1:-r03}
root=$(pwd); out=$root/gpurun_out/prof_$rnd; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
# 1. the default bench command under the kernel trace: the bench line and the per-kernel stats of the SAME run
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_default -- python3 $root/bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench default under rocprofv3: exit $?"
# 2. HBM traffic of the headline kernel (cfg2) and of every other frame-major shape: FETCH_SIZE / WRITE_SIZE passes,
#    then the issue-side counters the round's analysis uses (cfg4_v3 = the schedule-13 variant of M = 1024)
for wl in cfg2 cfg1 cfg3 cfg4 cfg4_v3 cfg5 ref56 ref560; do
  l2=30; case $wl in cfg1|cfg5|ref56|ref560) l2=28;; esac
  extra=""; name=$wl; case $wl in cfg4_v3) name=cfg4; extra="--variant 3";; esac
  B="python3 $root/bench.py --workload $name $extra --log2-samples $l2 --no-cpu-baseline --no-other-workloads --sustained-s 0 --steps 5 --warmup 2"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_${wl}_$i -- $B > $out/pmc_${wl}_$i.log 2>&1
    echo "pmc $wl pass $i: exit $?"
  done
done
cd $root
python3 tools/summarize_pmc.py $out > $out/summary.txt 2>&1; tail -70 $out/summary.txt
# keep what is judged (stats + the summary), drop the bulky per-dispatch traces: gpurun_out/ only travels back under 64 MiB
mkdir -p $out/keep
for f in $(find $out/bench_default -name "*kernel_stats.csv"); do cp $f $out/keep/bench_default_kernel_stats.csv; done
cp $out/summary.txt $out/pmc_traffic.json $out/bench_default.json $out/keep/ 2>/dev/null
find $out -mindepth 1 -maxdepth 1 ! -name keep -exec rm -rf {} +
