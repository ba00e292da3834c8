#!/bin/bash
# channel-major schedules side by side (run on the GPU box from the repo root)
W=${W:-64,12,64,int16,12}; N=${N:-268000000}
python tools/ab.py --samples $N --workload $W --channel-major "default:" "s0:sched=0" "s2w8:sched=2,tw=8" "s8_8x1:sched=8,tw=8,fpb=8" "s8_2x2:sched=8,tw=2,fpb=16" "s8_4x2:sched=8,tw=4,fpb=16" "s8_2x4:sched=8,tw=2,fpb=32" 2>&1 | grep -v "amdgpu.ids\|^#"
