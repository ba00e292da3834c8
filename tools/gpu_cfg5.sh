#!/bin/bash
out=gpurun_out/${1:-r2n}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "software_pipelined or every_schedule or switches or golden or channel_major_routes" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
{
echo "== cfg5"; timeout -k 10 300 python tools/ab.py --workload 128,12,64,int16,12 --log2-samples 28 --rounds 5 --steps 4 "default:" "s11_32:sched=11,fpb=32" "s11_128:sched=11,fpb=128" "s0_512:sched=0,fpb=512" "s0_136:sched=0,fpb=136" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== cfg5 channel-major"; timeout -k 10 300 python tools/ab.py --channel-major --workload 128,12,64,int16,12 --log2-samples 28 --rounds 4 --steps 4 "default:" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== cfg3"; timeout -k 10 300 python tools/ab.py --workload 256,8,256,int8,8 --rounds 4 --steps 4 "default:" "s0_32:sched=0,fpb=32" 2>&1 | grep -v "amdgpu.ids\|in ptr"
} > $out/cfg5.txt 2>&1; cat $out/cfg5.txt
