#!/bin/bash
# channel-major vs frame-major on every fused shape (run on the GPU box from the repo root; CM=1 for channel-major);
# sample counts are not powers of two: with F = 2^k every channel's column starts on the same HBM channel/bank
for w in "64,12,64,int16,12 268000000" "64,12,64,int8,8 268000000" "256,8,256,int8,8 1073000000" "128,12,64,int16,12 268000000" \
         "56,12,56,int16,12 268000000" "32,12,32,int16,12 268000000" "560,12,560,int16,12 268000000" "1024,16,1024,int16,12 1073000000" \
         "8,12,8,int16,12 268000000" "16,12,16,int16,12 268000000" "20,12,20,int16,12 268000000" "40,12,40,int16,12 268000000"; do
  set -- $w
  python tools/ab.py --samples $2 --workload $1 ${CM:+--channel-major} "default:" | tail -1
done
