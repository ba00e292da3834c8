#!/bin/bash
# team kernels: run length = the stream divided evenly over the CUs (one or a few runs per CU) against the fixed defaults
echo "== cfg4 2^30"; timeout -k 10 300 python tools/ab.py --workload 1024,16,1024,int16,16 --rounds 4 --steps 4 "default:" "fpb1024:fpb=1024" "fpb2048:fpb=2048" "fpb4096:fpb=4096" "fpb4096_rr:fpb=4096,remap=0" "fpb2048_rr:fpb=2048,remap=0" "fpb1368:fpb=1368" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== cfg4 2^29+"; timeout -k 10 300 python tools/ab.py --workload 1024,16,1024,int16,16 --samples 700000000 --rounds 4 --steps 4 "default:" "fpb1024:fpb=1024" "fpb2672:fpb=2672" "fpb1336:fpb=1336" "fpb672:fpb=672" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== ref560 2^28"; timeout -k 10 300 python tools/ab.py --workload 560,12,560,int16,12 --log2-samples 28 --rounds 4 --steps 4 "default:" "fpb1024:fpb=1024" "fpb1880:fpb=1880" "fpb944:fpb=944" "fpb632:fpb=632" "fpb472:fpb=472" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== ref560 int8"; timeout -k 10 300 python tools/ab.py --workload 560,12,560,int8,8 --log2-samples 28 --rounds 3 --steps 4 "default:" "fpb1880:fpb=1880" "fpb944:fpb=944" 2>&1 | grep -v "amdgpu.ids\|in ptr"
