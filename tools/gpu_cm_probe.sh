#!/bin/bash
# where does the fused channel-major team kernel spend its time?  experiment bits switch stages off (timing only)
out=gpurun_out/${1:-r2d}; mkdir -p $out
timeout -k 10 300 python tools/ab.py --channel-major --workload 1024,16,1024,int16,16 --rounds 3 --steps 3 "tiles32:sched=10" "noload:sched=10,exp=16" "nostore:sched=10,exp=32" "noflush:sched=10,exp=48" "nolastpass:sched=10,exp=64" "nothing:sched=10,exp=112" "slabs:sched=9" > $out/probe.txt 2>&1; cat $out/probe.txt
root=$(pwd); export TMPDIR=/tmp; cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/$out/pmc$i -- python3 $root/tools/ab.py --channel-major --workload 1024,16,1024,int16,16 --rounds 1 --steps 2 "tiles32:sched=10" > $root/$out/pmc$i.log 2>&1
done
cd $root
python3 tools/summarize_prof.py $out 2>/dev/null | grep -E "teams_cm|transpose" | cut -c1-200 | sed -E 's/pfb::FastCfg<[^>]*>//' > $out/pmc_summary.txt; cat $out/pmc_summary.txt
