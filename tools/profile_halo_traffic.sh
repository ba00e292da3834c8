#!/bin/bash
# cfg5 / cfg3: HBM-side read traffic (FETCH_SIZE, doubled: gfx950) and kernel time of the short-run defaults against runs
# grouped per XCD (halo rows served by that XCD's L2) and against long runs (halo read once per 512 frames).
set -u
root=$(pwd); out=$root/gpurun_out/halo_traffic; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
run() {  # name workload log2 extra...
  name=$1; wl=$2; l2=$3; shift 3
  B="python3 $root/bench.py --workload $wl --log2-samples $l2 --no-cpu-baseline --no-other-workloads --sustained-s 0 --steps 10 --warmup 5 $*"
  $B > $out/$name.json 2> $out/$name.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_$name -- $B > $out/pmc_$name.log 2>&1
  echo "$name: exit $?"
}
run cfg5_default cfg5 28
run cfg5_grouped16 cfg5 28 --xcd-remap 16
run cfg5_one_xcd cfg5 28 --xcd-remap 1
run cfg5_runs512 cfg5 28 --frames-per-block 512
run cfg3_default cfg3 30
run cfg3_grouped32 cfg3 30 --xcd-remap 32
run cfg3_one_xcd cfg3 30 --xcd-remap 1
run cfg3_runs512 cfg3 30 --frames-per-block 512
cd $root
python3 - "$out" <<'PY' | tee $out/summary.txt
import csv, glob, json, os, sys
root = sys.argv[1]
ALG_IN = {"cfg5": (1 << 28) * 4, "cfg3": (1 << 30) * 2}
print("input bytes fetched from the fabric per launch (FETCH_SIZE x 2 x 1024: the guide's gfx950 correction) and kernel time (un-profiled run, HIP events)")
for name in ("cfg5_default", "cfg5_grouped16", "cfg5_one_xcd", "cfg5_runs512", "cfg3_default", "cfg3_grouped32", "cfg3_one_xcd", "cfg3_runs512"):
    vals = []
    for f in glob.glob(os.path.join(root, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if any(k in row["Kernel_Name"] for k in ("pfb_overlap_kernel", "pfb_fast_kernel")) and row["Counter_Name"] == "FETCH_SIZE":
                vals.append(float(row["Counter_Value"]))
    try:
        line = json.loads(open(os.path.join(root, name + ".json")).read().strip().splitlines()[-1])
        ms, frac = line["roofline"]["kernel_ms"], line["roofline"]["frac"]
    except Exception:
        ms, frac = float("nan"), float("nan")
    rd = 2048.0 * sum(vals) / max(len(vals), 1)
    alg = ALG_IN[name[:4]]
    print(f"  {name:16s} read {rd / 1e9:6.3f} GB = {rd / alg:5.3f} x the input   kernel {ms:7.4f} ms   frac {frac:.3f}")
PY
find $out -mindepth 1 -maxdepth 1 ! -name summary.txt -exec rm -rf {} +
