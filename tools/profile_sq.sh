#!/bin/bash
# issue-side PMC passes: tools/profile_sq.sh <tag> [bench args...]
set -u
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/prof_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
B="python3 $root/bench.py --no-cpu-baseline --steps 5 --warmup 2 $*"
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU2" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES" \
           "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- $B > $out/pmc$i.log 2>&1
done
cd $root
python3 tools/summarize_prof.py $out | grep -E "pfb_(tile|fast|strided)" | sed -E 's/void pfb::pfb_[a-z]+_kernel<pfb::FastCfg<[^ ]*( [0-9a-z, ]*)? / /' | cut -c1-120
