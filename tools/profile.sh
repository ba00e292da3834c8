#!/bin/bash
# rocprofv3 passes for the bench command (run on the GPU box from the repo root).
#   tools/profile.sh <tag> [bench args...]
# Pass 1: kernel trace + stats.  Passes 2..: PMC counters, each in its own run (never mixed with
# tracing other than --kernel-trace).  Outputs under gpurun_out/prof_<tag>/.
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
B="python3 $root/bench.py --no-cpu-baseline --steps 5 --warmup 2 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- $B > $out/pmc$i.log 2>&1
done
cd $root
python3 tools/summarize_prof.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
