#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "1024 or 560 or variant or one_hot or run_length" > gpurun_out/teams_pytest.txt 2>&1; tail -2 gpurun_out/teams_pytest.txt
echo "== cfg4"; timeout -k 10 300 python tools/ab.py --workload 1024,16,1024,int16,16 --rounds 6 --steps 4 "default:" "fpb512:fpb=512" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== ref560"; timeout -k 10 300 python tools/ab.py --workload 560,12,560,int16,12 --log2-samples 28 --rounds 6 --steps 4 "default:" "fpb512:fpb=512" 2>&1 | grep -v "amdgpu.ids\|in ptr"
echo "== ref560 int8"; timeout -k 10 300 python tools/ab.py --workload 560,12,560,int8,8 --log2-samples 28 --rounds 4 --steps 4 "default:" 2>&1 | grep -v "amdgpu.ids\|in ptr"
