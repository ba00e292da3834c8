#!/bin/bash
# do run lengths that are NOT powers of two (CUs no longer at the same offset modulo the channel interleave) help?
out=gpurun_out/${1:-r2f}; mkdir -p $out
{
echo "== cfg4 frame-major"
timeout -k 10 300 python tools/ab.py --workload 1024,16,1024,int16,16 --rounds 3 --steps 3 "fpb512:" "fpb480:fpb=480" "fpb544:fpb=544" "fpb520:fpb=520" "fpb504:fpb=504" "fpb264:fpb=264" "fpb1032:fpb=1032" "fpb520nr:fpb=520,remap=0" 2>&1 | grep -v amdgpu.ids
echo "== cfg5 frame-major"
timeout -k 10 300 python tools/ab.py --workload 128,12,64,int16,12 --log2-samples 28 --rounds 3 --steps 3 "fpb512:" "fpb480:fpb=480" "fpb544:fpb=544" "fpb520:fpb=520" "fpb264:fpb=264" "fpb136:fpb=136" "fpb1032:fpb=1032" 2>&1 | grep -v amdgpu.ids
echo "== cfg3 frame-major"
timeout -k 10 300 python tools/ab.py --workload 256,8,256,int8,8 --rounds 3 --steps 3 "fpb32:" "fpb36:fpb=36" "fpb28:fpb=28" "fpb68:fpb=68" "fpb132:fpb=132" "fpb260:fpb=260" "fpb516:fpb=516" 2>&1 | grep -v amdgpu.ids
echo "== ref56 frame-major"
timeout -k 10 300 python tools/ab.py --workload 56,12,56,int16,12 --log2-samples 28 --rounds 3 --steps 3 "fpb512:" "fpb528:fpb=528" "fpb496:fpb=496" 2>&1 | grep -v amdgpu.ids
echo "== ref560 frame-major"
timeout -k 10 300 python tools/ab.py --workload 560,12,560,int16,12 --log2-samples 28 --rounds 3 --steps 3 "fpb512:" "fpb520:fpb=520" "fpb504:fpb=504" 2>&1 | grep -v amdgpu.ids
echo "== cfg4 channel-major fused"
timeout -k 10 300 python tools/ab.py --channel-major --workload 1024,16,1024,int16,16 --rounds 3 --steps 3 "t32:sched=10" "t32_480:sched=10,fpb=480" "t32_544:sched=10,fpb=544" "slabs:sched=9" 2>&1 | grep -v amdgpu.ids
} > $out/fpb_sweep.txt 2>&1
cat $out/fpb_sweep.txt
