#!/bin/bash
# the three-pass mixed-radix shapes: tuned default against the 2-frame-chunk team plan (variant 1)
for M in 200 250 280 320 400 500 512; do
  echo "== M=$M"; timeout -k 10 200 python tools/ab.py --workload $M,12,$M,int16,12 --log2-samples 28 --rounds 4 --steps 4 "default:" "2f:var=1" "2f_512:var=1,fpb=512" "2f_128:var=1,fpb=128" 2>&1 | grep -v "amdgpu.ids\|in ptr\|^case\|yardstick"
done
