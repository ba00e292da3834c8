#!/bin/bash
out=gpurun_out/${1:-r2g}; mkdir -p $out
{
echo "== cfg4 channel-major: slab transposer tile frames (exp = ft << 20), slab length"
timeout -k 10 300 python tools/ab.py --channel-major --workload 1024,16,1024,int16,16 --rounds 3 --steps 3 "ft64:sched=9,exp=$((64<<20))" "ft128:sched=9,exp=$((128<<20))" "ft256:sched=9,exp=$((256<<20))" "ft256_slab32k:sched=9,exp=$((256<<20)),slab=32768" "ft256_slab512k:sched=9,exp=$((256<<20)),slab=524288" "ft256_slab8k:sched=9,exp=$((256<<20)),slab=8192" 2>&1 | grep -v amdgpu.ids
echo "== M=560"
timeout -k 10 300 python tools/ab.py --channel-major --log2-samples 28 --workload 560,12,560,int16,12 --rounds 3 --steps 3 "ft64:sched=9,exp=$((64<<20))" "ft128:sched=9,exp=$((128<<20))" "ft256:sched=9,exp=$((256<<20))" 2>&1 | grep -v amdgpu.ids
} > $out/slab_sweep.txt 2>&1
cat $out/slab_sweep.txt
