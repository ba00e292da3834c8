#!/bin/bash
# FETCH/WRITE/TCC-only PMC passes: tools/profile_fetch.sh <tag> [bench args...]
set -u
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/prof_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
B="python3 $root/bench.py --no-cpu-baseline --steps 5 --warmup 2 $*"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- $B > $out/pmc$i.log 2>&1
done
cd $root
echo "== $tag: $*"
python3 tools/summarize_prof.py $out | grep -E "pfb_(tile|fast|strided)" | sed -E 's/void pfb::pfb_[a-z]+_kernel<pfb::FastCfg<[^ ]*( [0-9a-z, ]*)? / /' | cut -c1-100
