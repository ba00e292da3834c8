#!/bin/bash
out=gpurun_out/${1:-r2l}; mkdir -p $out
{
echo "== cfg5"; timeout -k 10 300 python tools/ab.py --workload 128,12,64,int16,12 --log2-samples 28 --rounds 4 --steps 4 "default:" "wide:exp=128" "s11_64:sched=11,fpb=64" "s11_64_wide:sched=11,fpb=64,exp=128" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== cfg3"; timeout -k 10 300 python tools/ab.py --workload 256,8,256,int8,8 --rounds 4 --steps 4 "default:" "wide:exp=128" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== cfg2"; timeout -k 10 300 python tools/ab.py --workload 64,12,64,int16,12 --rounds 4 --steps 4 "default:" "wide:exp=128" "s0:sched=0" "s0_wide:sched=0,exp=128" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== cfg4"; timeout -k 10 300 python tools/ab.py --workload 1024,16,1024,int16,16 --rounds 4 --steps 4 "default:" "wide:exp=128" 2>&1 | grep -v "amdgpu.ids\|in ptr"
} > $out/wide_probe.txt 2>&1; cat $out/wide_probe.txt
