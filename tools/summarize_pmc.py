"""Summarise gpurun_out/prof_rNN (tools/profile_round.sh): kernel stats of the default bench run, and per shape the HBM
traffic (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md 'HBM', WRITE_SIZE as is) and the issue-side
counters of its channelizer kernel.  Writes <dir>/pmc_traffic.json for the headline kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
ALG = {"cfg2": (1 << 30) * 12, "cfg1": (1 << 28) * 16, "cfg3": (1 << 30) * 10, "cfg4": (1 << 30) * 12, "cfg4_v3": (1 << 30) * 12,
       "cfg5": (1 << 28) * 20, "ref56": (1 << 28) * 12, "ref560": (1 << 28) * 12}


def channelizer_kernel(name):
    return "pfb" in name and any(k in name for k in ("paired", "fast_kernel", "teams", "pairs_sliding", "seg_kernel", "tile", "overlap", "twin"))


print("## default bench run under rocprofv3 --kernel-trace --stats")
for f in sorted(glob.glob(os.path.join(root, "bench_default", "**", "*kernel_stats.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if "pfb" in row["Name"] or "transpose" in row["Name"]:
            print("  {:90.90s} calls={:>5s} avg_ns={:>11s} min_ns={:>10s} max_ns={:>10s} pct={}".format(
                row["Name"], row["Calls"], row["AverageNs"], row.get("MinNs", ""), row.get("MaxNs", ""), row["Percentage"]))
try:
    line = json.loads(open(os.path.join(root, "bench_default.json")).read().strip().splitlines()[-1])
    r = line["roofline"]
    print(f"  bench line of the same run: kernel_ms={r['kernel_ms']} frac={r['frac']} sustained_kernel_ms={r.get('sustained_kernel_ms')} "
          f"sustained_frac={r.get('sustained_frac')} yardstick={r.get('measured_stream_copy_gbs')} GB/s")
except Exception as e:  # noqa: BLE001
    print("  (no bench line)", e)

print("\n## per shape: channelizer kernel counters, per-launch averages")
traffic = {}
for wl in ALG:
    acc = defaultdict(list)
    kname = None
    for f in sorted(glob.glob(os.path.join(root, f"pmc_{wl}_[0-9]", "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if channelizer_kernel(row["Kernel_Name"]):
                kname = row["Kernel_Name"]
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    if not acc:
        continue
    avg = {k: sum(v) / len(v) for k, v in acc.items()}
    rd = 2.0 * avg.get("FETCH_SIZE", float("nan")) * 1024.0   # KB -> bytes, x2: gfx950 reports half of streamed reads
    wr = avg.get("WRITE_SIZE", float("nan")) * 1024.0
    tot = rd + wr
    print(f"  {wl}: {kname[:70] if kname else ''}")
    print(f"     HBM read {rd / 1e9:8.3f} GB  write {wr / 1e9:8.3f} GB  total {tot / 1e9:8.3f} GB = {tot / ALG[wl]:.3f} x algorithmic ({ALG[wl] / 1e9:.3f} GB)")
    if "SQ_LDS_IDX_ACTIVE" in avg:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_ACTIVE_INST_VALU counts wave instructions (4 issue cycles
        # each on one of 1024 SIMDs): VALU issue share = 4 INST / (1024 BUSY / 32) = INST / (8 BUSY)
        print("     LDS bank-conflict cycles / LDS-active cycles = {:.3f};  WAIT_INST_ANY / WAVE_CYCLES = {:.3f};  WAIT_ANY / WAVE_CYCLES = {:.3f};  "
              "VALU issue share = ACTIVE_INST_VALU / (8 x BUSY_CYCLES) = {:.3f}".format(
                  avg["SQ_LDS_BANK_CONFLICT"] / max(avg["SQ_LDS_IDX_ACTIVE"], 1), avg["SQ_WAIT_INST_ANY"] / avg["SQ_WAVE_CYCLES"],
                  avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"], avg["SQ_ACTIVE_INST_VALU"] / (8.0 * avg["SQ_BUSY_CYCLES"])))
    traffic[wl] = dict(kernel=kname, FETCH_SIZE_KB=avg.get("FETCH_SIZE"), WRITE_SIZE_KB=avg.get("WRITE_SIZE"), hbm_read_bytes_per_launch=rd,
                       hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=tot, algorithmic_bytes_per_launch=ALG[wl],
                       traffic_over_algorithmic=tot / ALG[wl], counters={k: v for k, v in avg.items() if k.startswith("SQ_")})
if "cfg2" in traffic:
    head = dict(what="HBM traffic of one launch of the headline kernel (cfg2, 2^30 samples) and of the other frame-major shapes: rocprofv3 PMC, "
                     "FETCH_SIZE and WRITE_SIZE in separate --pmc passes (tools/profile_round.sh)",
                corrections="gfx950: FETCH_SIZE reports 1/2 of streamed read bytes (MI355X_MICROARCH.md 'HBM'): doubled; WRITE_SIZE as is; KB = 1024 B",
                **{k: v for k, v in traffic["cfg2"].items() if k != "counters"}, shapes=traffic)
    json.dump(head, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
    print("\nwrote pmc_traffic.json")
