#!/bin/bash
out=gpurun_out/${1:-r2k}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "software_pipelined or every_schedule" > $out/pytest.log 2>&1; tail -4 $out/pytest.log
{
echo "== cfg3"; timeout -k 10 300 python tools/ab.py --workload 256,8,256,int8,8 --rounds 4 --steps 4 "default:" "s11_fpb32:sched=11,fpb=32" "s11_fpb64:sched=11,fpb=64" "s11_fpb128:sched=11,fpb=128" "s11_fpb512:sched=11,fpb=512" "s0_fpb512:sched=0,fpb=512" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== cfg5"; timeout -k 10 300 python tools/ab.py --workload 128,12,64,int16,12 --log2-samples 28 --rounds 4 --steps 4 "default:" "s11_fpb512:sched=11,fpb=512" "s11_fpb128:sched=11,fpb=128" "s11_fpb64:sched=11,fpb=64" "s11_fpb32:sched=11,fpb=32" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== ref56"; timeout -k 10 300 python tools/ab.py --workload 56,12,56,int16,12 --log2-samples 28 --rounds 4 --steps 4 "default:" "s0:sched=0" "s11_fpb512:sched=11,fpb=512" "s11_fpb128:sched=11,fpb=128" "s11_fpb32:sched=11,fpb=32" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== cfg2"; timeout -k 10 300 python tools/ab.py --workload 64,12,64,int16,12 --rounds 4 --steps 4 "default:" "s0:sched=0" "s11_fpb512:sched=11,fpb=512" "s11_fpb64:sched=11,fpb=64" "s11_fpb32:sched=11,fpb=32" 2>&1 | grep -v "amdgpu.ids\|in ptr"
} > $out/overlap.txt 2>&1; cat $out/overlap.txt
