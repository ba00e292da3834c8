"""bench.py's contract with the driver, on the GPU: the one JSON line (metric / value / roofline / cpu_baseline keys) at a
reduced size, and the multi-rank path started by plain `python bench.py --gpus 2` -- both ranks on the one device of this
box, gloo transport (RCCL refuses two ranks on one GPU) -- so that the path the 8-GPU node runs stays runnable."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(400)
def test_single_gpu_line_has_the_contract_fields():
    d = run_bench(["--log2-samples", "24", "--steps", "4", "--warmup", "2", "--no-other-workloads", "--sustained-s", "0",
                   "--cpu-budget-s", "1"])
    assert d["metric"].startswith("input MS/s") and d["unit"] == "MS/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and "workload" in d["config"] and d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None and r["traffic_source"].startswith("profiles/") and r["launches_timed"] == 4
    assert r["measured_stream_copy_gbs"] > 1000 and len(r["copy_kernel_frac_by_byte_mix"]) == 6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "MS/s" and c["value"] > 0 and c["cores"] >= 1 and "NAIVE" in c["sample"]
    assert c["one_thread"]["cores"] == 1 and c["one_thread"]["value"] > 0


@pytest.mark.timeout(400)
@pytest.mark.parametrize("halo", ["p2p", "allgather"])
def test_two_ranks_start_themselves_and_print_one_line(halo):
    d = run_bench(["--gpus", "2", "--backend", "gloo", "--halo", halo, "--log2-samples", "24", "--steps", "4", "--warmup", "2"],
                  env_extra={"PFB_BENCH_ONE_DEVICE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0 and d["cpu_baseline"] is None
    assert "time-sharded x2" in d["config"]["parallelism"] and "704 raw samples" in d["config"]["parallelism"]
    assert d["roofline"]["launches_per_step"] == 2 and d["roofline"]["launches_timed"] == 4
    # the spread over ranks of the headline, and BASELINE's 8-GPU configuration (cfg4) as a sharded side pass
    kr = d["roofline"]["kernel_ms_per_rank"]
    assert len(kr["all"]) == 2 and 0 < kr["min"] <= kr["max"] and d["roofline"]["frac_slowest_rank"] > 0
    (c4,) = d["other_workloads"]
    assert c4["workload"] == "cfg4" and "error" not in c4, c4
    assert c4["halo_bytes"] == 61440 and c4["halo_samples"] == 15360 and c4["launches_per_step"] == 2
    assert len(c4["kernel_ms_per_rank"]["all"]) == 2 and c4["kernel_ms_per_rank"]["min"] > 0 and c4["step_wall_ms"] > 0
    assert c4["kernel"].startswith("pfb_fast<M1024,P16") and c4["ms_value"] > 0


@pytest.mark.timeout(400)
def test_two_ranks_fall_back_to_the_collective_when_the_neighbour_exchange_fails():
    """bench.py probes one sharded step before it times anything; a failed point-to-point exchange on any rank (injected
    here) switches every rank to the all_gather form of the same exchange, and the line says so."""
    d = run_bench(["--gpus", "2", "--backend", "gloo", "--halo", "p2p", "--log2-samples", "24", "--steps", "3", "--warmup", "2",
                   "--no-other-workloads"],
                  env_extra={"PFB_BENCH_ONE_DEVICE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PFB_BENCH_FAIL_P2P": "1"})
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert "allgather after p2p failed" in d["config"]["parallelism"]
