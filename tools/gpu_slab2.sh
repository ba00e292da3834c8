#!/bin/bash
out=gpurun_out/${1:-r2j}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "channel_major" > $out/pytest_cm.log 2>&1; tail -3 $out/pytest_cm.log
{
timeout -k 10 300 python tools/ab.py --channel-major --workload 1024,16,1024,int16,16 --rounds 3 --steps 3 "default:" "slab64k:slab=65536" "slab32k:slab=32768" "slab256k:slab=262144" "fused:sched=10" 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/ab.py --channel-major --log2-samples 28 --workload 560,12,560,int16,12 --rounds 3 --steps 3 "default:" "slab64k:slab=65536" "slab32k:slab=32768" "slab256k:slab=262144" "fused:sched=10" 2>&1 | grep -v amdgpu.ids
} > $out/slab2.txt 2>&1; cat $out/slab2.txt
