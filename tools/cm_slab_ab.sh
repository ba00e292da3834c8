#!/bin/bash
# channel-major: fused instantiation vs frame-major slabs + transpose (run on the GPU box from the repo root)
W=${W:-1024,16,1024,int16,12}; N=${N:-1073000000}
python tools/ab.py --samples $N --workload $W --channel-major "fused_s0:sched=0,var=${VAR:-0}" "slab32k:sched=9,slab=32768" "slab64k:sched=9,slab=65536" "slab128k:sched=9,slab=131072" "slab256k:sched=9,slab=262144" "slab512k:sched=9,slab=524288" 2>&1 | grep -v "amdgpu.ids\|^#"
