#!/bin/bash
# grouped XCD remap (PFB_OPT_XCD_REMAP = G: each XCD takes G consecutive runs at a time) against round-robin and the full
# remap, on the short-run kernels
echo "== cfg5"; timeout -k 10 300 python tools/ab.py --workload 128,12,64,int16,12 --log2-samples 28 --rounds 5 --steps 4 "default:" "rr:remap=0" "full:remap=1" "g2:remap=2" "g4:remap=4" "g8:remap=8" "g16:remap=16" "g32:remap=32" "fpb64_g8:fpb=64,remap=8" "fpb64_g4:fpb=64,remap=4" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== cfg3"; timeout -k 10 300 python tools/ab.py --workload 256,8,256,int8,8 --rounds 5 --steps 4 "default:" "rr:remap=0" "full:remap=1" "g2:remap=2" "g4:remap=4" "g8:remap=8" "g16:remap=16" "g32:remap=32" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== cfg4 teams run length"; timeout -k 10 300 python tools/ab.py --workload 1024,16,1024,int16,16 --rounds 4 --steps 4 "default:" "fpb256:fpb=256" "fpb384:fpb=384" "fpb768:fpb=768" "fpb1024:fpb=1024" "rr:remap=0" "g2:remap=2" "g4:remap=4" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== ref560"; timeout -k 10 300 python tools/ab.py --workload 560,12,560,int16,12 --log2-samples 28 --rounds 4 --steps 4 "default:" "fpb256:fpb=256" "fpb128:fpb=128" "fpb1024:fpb=1024" "rr:remap=0" "g2:remap=2" "g4:remap=4" 2>&1 | grep -v "amdgpu.ids\|in ptr"
