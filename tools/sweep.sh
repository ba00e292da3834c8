#!/bin/bash
# usage: tools/sweep.sh "<bench args A>" "<bench args B>" ...   (each run prints one summary line)
mkdir -p gpurun_out
for a in "$@"; do
  python bench.py --no-cpu-baseline --steps 10 $a 2>/dev/null > gpurun_out/sweep_tmp.json
  python -c "import sys,json; d=json.load(open('gpurun_out/sweep_tmp.json')); print('%-50s MS/s %9.0f  GB/s %7.1f  kernel_ms %.4f  step_ms %.4f' % (sys.argv[1], d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms'], d['ms_per_step']))" "$a"
done
