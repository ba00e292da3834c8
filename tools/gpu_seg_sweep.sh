#!/bin/bash
for W in 24,12,24 200,12,200 100,12,100 30,12,30; do
  echo "== $W"; timeout -k 10 200 python tools/ab.py --workload $W,int16,12 --log2-samples 28 --rounds 3 --steps 4 "default:" "f64:fpb=64" "f128:fpb=128" "f256:fpb=256" "f512:fpb=512" "f1024:fpb=1024" "f256_rr:fpb=256,remap=0" "f512_rr:fpb=512,remap=0" "s11_128:sched=11,fpb=128" "s7_256:sched=7,fpb=256" 2>&1 | grep -v "amdgpu.ids\|in ptr\|^case\|yardstick"
done
