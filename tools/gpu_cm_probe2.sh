#!/bin/bash
# channel-major: power-of-two frame counts (column stride = 2^k bytes) against odd ones, every shape's default route
out=gpurun_out/${1:-r2e}; mkdir -p $out
{
for spec in "1024,16,1024,int16,16 30 3072" "560,12,560,int16,12 28 1680" "64,12,64,int16,12 30 192" "256,8,256,int8,8 30 768" "128,12,64,int16,12 28 192"; do
  set -- $spec
  echo "== $1  F = 2^k"
  timeout -k 10 300 python tools/ab.py --channel-major --workload $1 --log2-samples $2 --rounds 3 --steps 3 "default:" "slabs:sched=9" "tiles16:sched=10,tw=16" 2>&1 | grep -v "amdgpu.ids"
  echo "== $1  F = 2^k + 3"
  timeout -k 10 300 python tools/ab.py --channel-major --workload $1 --samples $(( (1 << $2) + $3 )) --rounds 3 --steps 3 "default:" "slabs:sched=9" "tiles16:sched=10,tw=16" 2>&1 | grep -v "amdgpu.ids"
done
} > $out/pow2_vs_odd.txt 2>&1
cat $out/pow2_vs_odd.txt
