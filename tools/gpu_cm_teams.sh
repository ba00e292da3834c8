#!/bin/bash
# channel-major routes of the team plans: parity test + interleaved A/B timing (run on the GPU box)
set -o pipefail
out=gpurun_out/${1:-r2c}
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "channel_major" > $out/pytest_cm.log 2>&1; tail -4 $out/pytest_cm.log
timeout -k 10 200 python tools/ab.py --channel-major --workload 1024,16,1024,int16,16 --rounds 4 --steps 4 "default:" "tiles32:sched=10" "tiles16:sched=10,tw=16" "slabs:sched=9" "t32_fpb256:sched=10,fpb=256" "t32_fpb1024:sched=10,fpb=1024" > $out/ab_cfg4_cm.txt 2>&1; cat $out/ab_cfg4_cm.txt
timeout -k 10 200 python tools/ab.py --channel-major --log2-samples 28 --workload 560,12,560,int16,12 --rounds 4 --steps 4 "default:" "tiles32:sched=10" "tiles16:sched=10,tw=16" "slabs:sched=9" > $out/ab_560_cm.txt 2>&1; cat $out/ab_560_cm.txt
