#!/bin/bash
echo "== cfg4 variants"; timeout -k 10 300 python tools/ab.py --workload 1024,16,1024,int16,16 --rounds 4 --steps 4 "teams:" "v1_16w:var=1" "v1_16w_128:var=1,fpb=128" "v1_16w_64:var=1,fpb=64" "v2_8w:var=2" "v2_8w_128:var=2,fpb=128" "v2_8w_64:var=2,fpb=64" "v3_duo:var=3" 2>&1 | grep -v "amdgpu.ids\|in ptr"
