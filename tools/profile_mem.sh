#!/bin/bash
# memory-side PMC passes only: tools/profile_mem.sh <tag> [bench args...]
set -u
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/prof_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
B="python3 $root/bench.py --no-cpu-baseline --steps 5 --warmup 2 $*"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- $B > $out/pmc$i.log 2>&1
done
cd $root
python3 tools/summarize_prof.py $out | grep -E "pfb_(tile|fast|strided)" | sed -E 's/void pfb::pfb_[a-z]+_kernel<pfb::FastCfg<[^ ]*( [0-9a-z, ]*)? / /' | cut -c1-120
