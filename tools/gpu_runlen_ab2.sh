#!/bin/bash
# team kernels, default (balanced) run length against fixed 512 on stream lengths that are not whole rounds
for n in 1073741824 700000000 1000000000 536870912 300000000 100000000 20000000 5242880; do
  echo "== cfg4 n=$n"; timeout -k 10 120 python tools/ab.py --workload 1024,16,1024,int16,16 --samples $n --rounds 3 --steps 4 "auto:" "fpb512:fpb=512" 2>&1 | grep -v "amdgpu.ids\|in ptr\|^case\|yardstick"
done
for n in 268435456 200000000 100000000 10000000; do
  echo "== ref560 n=$n"; timeout -k 10 120 python tools/ab.py --workload 560,12,560,int16,12 --samples $n --rounds 3 --steps 4 "auto:" "fpb512:fpb=512" 2>&1 | grep -v "amdgpu.ids\|in ptr\|^case\|yardstick"
done
echo "== ref560 int8: lockstep (default) vs teams (variant 1)"; timeout -k 10 120 python tools/ab.py --workload 560,12,560,int8,8 --log2-samples 28 --rounds 3 --steps 4 "default:" "teams:var=1" 2>&1 | grep -v "amdgpu.ids\|in ptr"
