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


def test_recorded_eight_rank_self_test_line():
    """profiles/r06_bench_selflaunch_8rank.json: ``ONSSEN_BENCH_ONE_DEVICE=1 python bench.py --gpus 8 --steps 3
    --warmup 1`` on the 1-GPU box (tools/gpu_r06e.sh) -- the command line the driver's first N = 8 SCALE run uses, with every rank on
    cuda:0 over gloo and launch-per-step kernels.  What it proves is the HARNESS at world 8: the self-launcher, the rendezvous, the
    barrier + max-over-ranks timing, the whole-job value, the data-parallel leg with its 7 buckets, and the per-rank health block the
    line carries at N > 1 (round 6).  Its timings say nothing about 8 GPUs."""
    path = os.path.join(ROOT, "profiles", "r06_bench_selflaunch_8rank.json")
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                     # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert len(r["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in r["per_rank_ms_per_step"])
    assert abs(r["ms_per_step"] - max(r["per_rank_ms_per_step"])) <= 1e-6 * r["ms_per_step"]      # MAX over ranks
    audio_per_step = 8 * r["config"]["chunks_per_gpu"] * r["config"]["frames_per_chunk"] * 64 / 8000.0
    assert abs(r["value"] - audio_per_step / (r["ms_per_step"] * 1e-3)) <= 1e-6 * r["value"]        # whole job: all eight ranks' chunks
    assert r["config"]["parallelism"].startswith("utterance-sharded x8") and r["vs_baseline"] is None
    health = r["per_rank_health"]
    assert len(health) == 8
    for h in health:
        assert set(h) == {"xcd_placement_independent_protocol_used", "persistent_launch_aborts", "calls_rerun_after_abort",
                          "persistent_stack_launches", "launch_per_step_stack_launches", "NCCL_MAX_NCHANNELS", "device_index"}
        assert h["persistent_launch_aborts"] == 0 and h["calls_rerun_after_abort"] == 0
    assert "gloo" in r["collective_layer"]["backend"] and r["collective_layer"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    dp = r["dp_training_step_dc_l3_b16"]
    assert "error" not in dp, dp
    assert len(dp["per_rank_ms_per_step"]) == 8 and dp["all_reduce_bytes_per_step"] == 23_908_980 * 4
    assert len(dp["bucket_bytes"]) == 7 and sum(dp["bucket_bytes"]) == dp["all_reduce_bytes_per_step"]
    assert dp["replicas_identical_after_dp_steps"] is True and dp["buckets_issued_inside_backward"] >= 1
    assert len(dp["last_loss_per_rank"]) == 8 and len(set(dp["last_loss_per_rank"])) == 8       # every rank trained on its own batches


def test_recorded_bench_line_carries_the_contract():
    """profiles/r06c_bench_final.json = `python3 bench.py --gpus 1 --steps 20 --warmup 5` on the GPU box (the driver's command): every
    key the bench contract names, the roofline and CPU-baseline blocks (round 6: BASELINE.md section 3's protocol), the round-5 additions and
    (round 6b) the two-batch pipeline the headline step runs through, with the one-batch-at-a-time step beside it."""
    r = json.loads(open(os.path.join(ROOT, "profiles", "r06c_bench_final.json")).read().strip().splitlines()[-1])
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
    # the pipelined step: one persistent launch holds two layers' recurrences (twice the FLOPs of one), every step is a whole batch
    pl = r["config"]["pipeline"]
    assert pl["depth"] == 2 and abs(pl["batch_latency_ms"] - 2 * r["ms_per_step"]) < 1e-9
    assert "pair launch" in roof["kernel"] and abs(roof["algorithmic_flop_per_launch"] - 2 * 2.0 * 2 * 32 * 4 * 600 * 600 * 400) < 1.0
    assert abs(roof["achieved"] - roof["algorithmic_flop_per_launch"] / (roof["us_per_launch"] * 1e-6) / 1e12) < 1e-6 * roof["achieved"]
    assert set(roof["legs_ms"]) == {"stft_logmag", "threshold_target_map", "input_proj_l0_with_split", "pair_recurrence_l1_prev_l0_this",
                                    "input_proj_l1", "fc_dc_l2norm_active_rows_only", "init_lloyd_masks", "mask_istft"}
    seq = roof["one_batch_at_a_time_form"]
    assert seq["kernel"] == "lstm_xcd_kernel" and 0 < seq["frac"] < roof["frac"]
    one = r["one_batch_at_a_time_step"]
    assert one["ms_per_step"] > r["ms_per_step"] and abs(one["x_real_time"] - 32 * 3.2 / (one["ms_per_step"] * 1e-3)) < 1e-6 * one["x_real_time"]
    cpu = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] == "port" and cpu["value"] > 0
    # physical cores stated, the threads actually used, median of 10 at B = 1 and B = 32,
    # the network at the best thread count of a scan up to all physical cores, both KMeans n_init settings
    assert cpu["cores_physical"] >= cpu["threads_used"] == cpu["cores"] >= 1 and cpu["host_threads"] >= cpu["cores_physical"]
    assert cpu["kmeans_n_init"] == 10 and "n_init=10" in cpu["sample"]
    assert set(cpu["whole_path"]) == {"B1", "B1_kmeans_auto", "B32", "B32_kmeans_auto"}
    assert all(v["passes"] >= 3 and v["ms"] > 0 for v in cpu["whole_path"].values()) and 3 <= cpu["whole_path"]["B32"]["passes"] <= 10      # (time-capped: a slow box gets 9)
    assert abs(cpu["value"] - cpu["whole_path"]["B32"]["x_real_time"]) < 1e-9
    assert cpu["whole_path"]["B32_kmeans_auto"]["ms"] < cpu["whole_path"]["B32"]["ms"]
    net = cpu["network_only"]
    assert 3 <= net["B1"]["passes"] <= 10 and 3 <= net["B32"]["passes"] <= 10
    assert str(cpu["cores_physical"]) in net["B32"]["threads_tried"] and str(net["B32"]["threads"]) in net["B32"]["threads_tried"]
    assert r["lloyd_iterations"]["cap"] == 20 and "second_input_set" in r
    ex = r["extra_configs"]
    assert ex["trained_weights_dc_l2_b32"]["si_sdr_db"]["separated_mean"] > 8.0           # the headline step on a trained network separates
    tp = ex["trained_weights_dc_l2_b32"]["pipelined"]                                      # ... and so does the pipelined one, faster
    assert tp["si_sdr_db_separated_mean"] > 8.0 and tp["ms_per_step"] < ex["trained_weights_dc_l2_b32"]["ms_per_step"]
    assert [row["chunks"] for row in ex["batch_sweep"]["dc_l2"]["rows"]] == [8, 16, 32, 64, 128, 256]
    assert ex["cfg4_training_step_dc_l3_b16"]["ms_per_step"] < 7.5
    # round 6c: the pair launch's PMC traffic in the line (per launch, like `achieved`), and the ragged pipeline beside the plain ragged call
    assert 0.9 < roof["traffic"] / 639078400 < 1.2 and "PMC" in roof["traffic_source"]
    rg = ex["b16_ragged_utterances"]
    for order in ("as_they_come", "bucketed_by_length"):
        assert rg["pipelined"][order]["bit_identical_to_separate_dc"] is True
        assert rg["plain_call_on_2K_rows"][order]["x_real_time"] > 0
    assert rg["pipelined"]["as_they_come"]["x_real_time"] > rg["x_real_time"]
    assert rg["pipelined"]["bucketed_by_length"]["x_real_time"] > rg["bucketed_by_length"]["x_real_time"]
