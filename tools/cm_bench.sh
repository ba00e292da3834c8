#!/bin/bash
# channel-major vs frame-major on every fused shape (run on the GPU box from the repo root)
for w in "64,12,64,int16,12 30" "256,8,256,int8,8 30" "128,12,64,int16,12 28" "56,12,56,int16,12 28" "560,12,560,int16,12 28" "1024,16,1024,int16,12 30"; do
  set -- $w
  python tools/ab.py --log2-samples $2 --workload $1 ${CM:+--channel-major} "default:" | tail -1
done
