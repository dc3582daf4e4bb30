"""bench.py's launcher contract (round 5; VERDICT r04 "missing 1"): ``python bench.py --gpus N`` must become the N-rank job by
itself -- the first multi-GPU SCALE run may not fail for a launcher reason -- and the recorded 2-rank self-test line under
profiles/ must carry what the driver and the judge read."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_command_is_the_drivers_own():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29555)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_gpus_n_without_a_launcher_self_launches_or_refuses_loudly():
    """On this CPU-only container there are 0 devices: the self-launcher must say so and exit 2 -- NOT raise the old
    "launch with torch.distributed.run" SystemExit, and not start ranks that would all fight over a missing GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ONSSEN_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "--gpus 2 but this node shows 0 device(s)" in r.stderr and "ONSSEN_BENCH_ONE_DEVICE=1" in r.stderr


def test_recorded_two_rank_self_test_line():
    """profiles/r05_bench_selflaunch_2rank.json: ``ONSSEN_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5`` on the 1-GPU box,
    no external launcher (tools/gpu_r05a.sh)."""
    path = os.path.join(ROOT, "profiles", "r05_bench_selflaunch_2rank.json")
    r = json.loads(open(path).read().strip().splitlines()[-1])
    assert r["n_gpus"] == 2 and r["steps"] == 5 and len(r["per_rank_ms_per_step"]) == 2
    assert r["scaling"] == "weak" and r["roofline"]["frac"] > 0 and r["value"] > 0
    dp = r["dp_training_step_dc_l3_b16"]
    assert "error" not in dp, dp
    assert len(dp["per_rank_ms_per_step"]) == 2 and dp["all_reduce_bytes_per_step"] == 23_908_980 * 4
    assert dp["replicas_identical_after_dp_steps"] is True and dp["buckets_issued_inside_backward"] >= 1
    assert dp["ms_per_step"] > 0 and dp["ms_per_step_without_exchange"] > 0


def test_recorded_bench_line_carries_the_contract():
    """profiles/r05_bench_final.json = `python3 bench.py --gpus 1 --steps 20 --warmup 5` on the GPU box (the driver's command): every
    key the bench contract names, the roofline and CPU-baseline blocks, and the round-5 additions."""
    r = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_final.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 20 and r["warmup"] == 5 and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["vs_baseline"] is None and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - 32 * 3.2 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]          # audio seconds per wall second
    roof = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["legs_le_step"] is True                                                   # the legs timed one by one sum to <= the step
    cpu = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] == "port" and cpu["value"] > 0
    assert r["lloyd_iterations"]["cap"] == 20 and "second_input_set" in r
    ex = r["extra_configs"]
    assert ex["trained_weights_dc_l2_b32"]["si_sdr_db"]["separated_mean"] > 8.0           # the headline step on a trained network separates
    assert [row["chunks"] for row in ex["batch_sweep"]["dc_l2"]["rows"]] == [8, 16, 32, 64, 128, 256]
    assert ex["cfg4_training_step_dc_l3_b16"]["ms_per_step"] < 7.5
