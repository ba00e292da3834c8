#!/bin/bash
python tools/chk_variant.py 560 12 2>&1 | grep -v amdgpu | tail -3
echo "== ref560"; timeout -k 10 300 python tools/ab.py --workload 560,12,560,int16,12 --log2-samples 28 --rounds 6 --steps 4 "teams4:" "teams2x2:var=2" "t2_256:var=2,fpb=256" "t2_512:var=2,fpb=512" "t2_1024:var=2,fpb=1024" "t2_128:var=2,fpb=128" "t2_rr:var=2,remap=0" 2>&1 | grep -v "amdgpu.ids\|in ptr"
