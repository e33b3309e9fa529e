"""bench.py's LAST stdout line is the contract's JSON object: under 2000 characters (the driver's record keeps a 2000-character
tail), every judged number a scalar inside `roofline` / `cpu_baseline`; the bulky objects ride on an earlier `SECONDARY ` line.
Run with -m gpu; short runs, CPU baseline leg skipped here except where stated -- the default invocation includes it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split(stdout):
    """(contract object, secondary object | None): the LAST line, and the `SECONDARY ` line before it."""
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    main = [ln for ln in lines if ln.startswith("{")]
    assert len(main) == 1 and lines[-1] == main[0], stdout[-3000:]
    assert len(main[0]) <= 2000, len(main[0])
    sec = [ln[len("SECONDARY "):] for ln in lines if ln.startswith("SECONDARY ")]
    assert len(sec) <= 1
    return json.loads(main[0]), (json.loads(sec[0]) if sec else None)


def _run(extra, steps=("--steps", "4", "--warmup", "2"), cpu=False):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *steps] + ([] if cpu else ["--no-cpu-baseline"]) + extra,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    return _split(p.stdout)


def _check_roofline(r, peak):
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["peak"] == peak and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1


def test_bench_json_contract_default():
    j, sec = _run([], steps=("--steps", "8", "--warmup", "2"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "preheat_ms", "cold_first_block"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 8 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["unit"] == "samples/s" and j["scaling"] == "strong" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert j["metric"] == "MC-forward samples/sec, BayesianAlexNet CIFAR-10 bs=512 num_ens=10" and j["dtype"] == "f32"
    assert "workload" in j["config"] and "model" not in j["config"] and "4 steps per launch" in j["config"]["launch"]
    assert abs(j["value"] - 512 * 10 / (j["ms_per_step"] * 1e-3)) <= 1e-3 * j["value"]
    assert j["preheat_ms"] >= 300 and j["cold_first_block"]["value"] > 0
    r = j["roofline"]
    _check_roofline(r, 157.3)
    # every judged number is a first-level scalar of `roofline` (what the driver's record keeps), the reparam pass also nested
    for k in ("per_launch_us", "slabs_per_launch", "sustained_frac", "stats_median", "stats_p10", "stats_p90", "one_step_in_flight_ms",
              "reparam_frac", "reparam_avg_us", "reparam_draws", "reparam_back_to_back_frac", "training_step_ms", "training_step_frac",
              "dropin_loop_value", "timed_by", "value_above_p90", "odd_batch_510_value", "hooked_loop_value"):
        assert k in r and not isinstance(r[k], (dict, list)), k
    for k in ("one_step_per_launch_ms", "reparam_10draw_frac", "reparam_hbm_resident_frac"):        # (round 6: on the SECONDARY line)
        assert ("roofline." + k) in sec["main_line_extras"], k
    assert r["slabs_per_launch"] == 40 and len(r["per_launch_us"].split("/")) == 6
    assert r["stats_p10"] <= r["stats_median"] <= r["stats_p90"]
    # reparam_frac is the launch shape the timed region runs: 4 steps x 10 draws per launch (review r04: "make the judged line say
    # what the timed region does"); the 10-draw launch and the HBM-resident probe are named scalars beside it
    nbytes = (8 + 4 * 40) * 2175946                     # (8 + 4 E) bytes per parameter element, E = 40 draws per launch
    assert r["reparam_draws"] == 40 and sec["roofline_reparam_steps_per_launch"]["bytes_per_launch"] == nbytes
    assert abs(r["reparam_frac"] - nbytes / (r["reparam_avg_us"] * 1e-6) / 8e12) < 2e-3
    reg = sec["roofline_reparam_in_region"]             # the launch in the region's context: [reparam, conv1] pairs minus conv1 alone
    assert "error" not in reg and reg["draws_per_launch"] == 40 and reg["frac"] == r["reparam_frac"]
    assert abs(reg["avg_us"] - (reg["pair_us"] - reg["gemm_alone_us"])) < 0.02
    assert 0.3 < r["reparam_back_to_back_frac"] < 1.0 and r["reparam_back_to_back_frac"] == sec["roofline_reparam_steps_per_launch"]["frac"]
    mx = sec["main_line_extras"]
    assert 0 < mx["roofline.reparam_hbm_resident_frac"] < 1 and 0 < mx["roofline.reparam_10draw_frac"] < 1.2
    ts = sec["training_step"]["roofline"]                # executed FLOPs of forward + wgrad + dgrad over the step's wall time
    assert ts["bound"] == "mfma" and ts["peak"] == 157.3 and 0.05 < ts["frac"] < 1 and r["training_step_frac"] == ts["frac"]
    f32 = sec["fusion_ab_fp32"]                          # ... and on the fp32 chain: every pool either fused or priced
    assert "error" not in f32 and f32["bbb_pool1"]["shipped"] == "fused" and f32["bbb_pool2"]["fused_us"] > 0 and f32["lrt_pool1"]["fused_us"] > 0
    fa = sec["split_bf16"]["fusion_ab"]                  # N3 closed with numbers: pool1 fused and shipped, pool2 / pool3 priced
    assert "error" not in fa and fa["pool1"]["fused_us"] < fa["pool1"]["conv_us"] + fa["pool1"]["pool_us"] and fa["pool2"]["pool_us"] > 0
    assert r["value_above_p90"] == (j["value"] > r["stats_p90"]) and "hipGraph" in r["timed_by"]
    # the secondary line: the objects of rounds 1-3, unabridged
    for k in ("roofline_detail", "roofline_reparam", "stats", "one_step_in_flight", "one_step_per_launch", "dropin_loop", "configs",
              "training_step", "split_bf16", "slow_paths"):
        assert k in sec, k
    sp = sec["slow_paths"]
    assert "error" not in sp and sp["odd_batch_510"]["value"] > sp["odd_batch_510_reference_layout"]["value"] > 0
    assert sp["forward_hook_dropin_loop"]["value"] > 0
    rr = sec["roofline_reparam"]
    assert rr["bound"] == "hbm" and rr["peak"] == 8000.0 and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-3
    assert "error" not in sec["dropin_loop"] and sec["dropin_loop"]["value"] > 0
    assert "error" not in sec["training_step"] and sec["training_step"]["path"] == "chwn-autograd"
    dflt = sec["training_step"]["reference_default_config"]             # config_bayesian.py defaults: lrt, bs 256, num_ens 1
    assert "error" not in dflt and dflt["path"] == "chwn-autograd" and 0 < dflt["hipgraph_ms_per_step"] <= dflt["eager_ms_per_step"] * 1.5
    # every other BASELINE configuration is measured, each with its own roofline
    assert set(sec["configs"]) == {"configs[1]", "configs[2]", "configs[3]", "configs[4]"}
    for name, c in sec["configs"].items():
        assert "error" not in c, (name, c)
        assert c["value"] > 0 and c["roofline"] is not None
        _check_roofline(c["roofline"], 2500.0 if c["dtype"] == "bf16" else 157.3)
    assert sec["configs"]["configs[1]"]["dtype"] == "bf16" and sec["configs"]["configs[4]"]["rows_out"] == 512 * 49


def test_bench_driver_invocation_line_is_short_and_complete():
    """The driver's own command (--steps 20 --warmup 5, CPU baseline included): the last line fits the record's 2000-character
    tail with `cpu_baseline` inside, and the pre-heated 20-step figure is within 8 % of the block median of the same run."""
    j, sec = _run([], steps=("--steps", "20", "--warmup", "5"), cpu=True)
    cb = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    if cb["kind"] == "reference":
        # the same upstream modules on this box's GPU (PyTorch-ROCm's own kernels): a second reported baseline
        assert cb["reference_gpu_value"] > cb["value"] and sec["main_line_extras"]["speedup_vs_reference_gpu"] > 10
        assert sec["cpu_baseline_detail"]["reference_gpu_path"]["no_grad"]["ms_per_step"] > 1.0
        assert "cpu_model" in sec["cpu_baseline_detail"]
    # the value IS the median of the 1 + 5 blocks of exactly --steps steps (review r05 item 4); the first block rides beside it
    r = j["roofline"]
    assert j["value"] == r["stats_median"] and r["value_above_p90"] is False and r["stats_p10"] <= j["value"] <= r["stats_p90"]
    assert abs(r["single_block_value"] - j["value"]) <= 0.10 * j["value"]
    assert sec["main_line_extras"]["speedup_vs_cpu"] > 10
    # round 6: the split-bf16 chain and the reparam pass against the box's write roof, as first-level scalars
    for k in ("split_bf16_value", "split_bf16_frac_of_bf16_peak", "split_bf16_ms_per_step", "write_roof_GBps", "reparam_frac_of_write_roof"):
        assert k in r and not isinstance(r[k], (dict, list)), k
    assert r["split_bf16_value"] > j["value"] and 0.3 < r["split_bf16_frac_of_bf16_peak"] < 1 and 0.7 < r["reparam_frac_of_write_roof"] < 1.3


def test_bench_json_contract_other_config_as_headline():
    j, _ = _run(["--config", "configs[1]", "--pipeline", "1", "--no-extras"])
    assert j["dtype"] == "bf16" and j["n_gpus"] == 1 and "3Conv3FC" in j["metric"]
    _check_roofline(j["roofline"], 2500.0)


def test_bench_n_gt_1_flow_rehearsed_on_one_device():
    """The driver's N > 1 invocation (one process per GPU under torch.distributed.run), rehearsed on ONE MI355X: both ranks pinned
    to device 0 and gloo instead of RCCL (bench.py's documented test hooks) -- launch, rendezvous on 127.0.0.1, work-unit
    sharding, one all_gather per step, max-over-ranks timing, rank 0 printing the one JSON line.  The number is meaningless
    (two processes share a GPU and gloo stages through the host); the flow is what is checked."""
    env = dict(os.environ, BBB_BENCH_DEVICE="0", BBB_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    j, _ = _split(p.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["scaling"] == "strong" and j["value"] > 0
    # the N > 1 line carries rank 0's roofline for ITS share and the CPU baseline of a rank-0 pre-pass
    _check_roofline(j["roofline"], 157.3)
    assert j["roofline"]["slabs_per_launch"] == 20 and j["cpu_baseline"]["value"] > 0          # rank 0: 20 of a group's 40 draws
    assert j["config"]["global_batch"] == 512 and j["config"]["num_ens_total"] == 10          # the metric's workload, not 2x of it
    assert j["config"]["backend_mode"].startswith("eager all_gather") and "gloo" in j["config"]["backend_mode"]
    assert "groups of 4 steps" in j["config"]["parallelism"] and j["config"]["draws_per_rank"] == [20, 20]
    assert j["weak_scaling"]["num_ens_total"] == 20
    assert abs(j["value"] - 512 * 10 / (j["ms_per_step"] * 1e-3)) <= 1e-3 * j["value"]


def test_bench_gpus_2_as_typed_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run: bench.py re-executes itself under the launcher (one process
    per GPU, rendezvous on 127.0.0.1 with a free port).  Rehearsed on one MI355X through the same hooks as above."""
    env = dict(os.environ, BBB_BENCH_DEVICE="0", BBB_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-extras",
                        "--steps-per-launch", "1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    j, _ = _split(p.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    assert j["config"]["ranks_seen"] == 2 and sum(j["config"]["units_per_rank"]) == 10 and j["config"]["backend"] == "gloo"


def test_bench_multi_rank_flow_over_rccl_with_one_rank():
    """The code path a multi-GPU box executes -- RCCL process group, the probe for recordable collectives, per-lane communicators,
    sharded lanes whose graphs hold the step's one all_gather, barriers, broadcast pre-heat, max-over-ranks timing, weak-scaling
    leg, rank 0's share roofline -- run through the REAL backend ("nccl") with a single rank (bench.py's BBB_BENCH_SELF_GROUP hook;
    the gloo rehearsals above cover two processes, this covers RCCL itself)."""
    env = dict(os.environ, BBB_BENCH_SELF_GROUP="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BBB_BENCH_BACKEND", "BBB_BENCH_DEVICE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    j, _ = _split(p.stdout)
    assert j["n_gpus"] == 1 and j["config"]["ranks_seen"] == 1 and j["config"]["backend"] == "nccl"
    assert j["config"]["backend_mode"].startswith("all_gather recorded"), j["config"]["backend_mode"]
    assert j["config"]["draws_per_rank"] == [40] and "2 lane" in j["config"]["launch"] and "4 steps per launch" in j["config"]["launch"]
    assert j["value"] > 0 and j["weak_scaling"]["value"] > 0
    _check_roofline(j["roofline"], 157.3)
    assert j["roofline"]["slabs_per_launch"] == 40


def test_bench_first_contact_watchdog_ends_a_stalled_sharded_run():
    """Review r04 item 8: a rank that never finishes its first sharded replays must not cost the driver its lease.  Two ranks on one
    device over gloo; rank 1 sleeps inside the first-contact section (test hook); both ranks' watchdogs (8 s here) fire, rank 0
    prints a line in the judged format with "error" and the processes exit with code 3 -- within seconds, not at the launcher's
    timeout."""
    import time
    env = dict(os.environ, BBB_BENCH_DEVICE="0", BBB_BENCH_BACKEND="gloo", BBB_BENCH_FIRST_CONTACT_TIMEOUT_S="8",
               BBB_BENCH_TEST_STALL_S="600", BBB_BENCH_TEST_STALL_RANK="1")
    t0 = time.time()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert time.time() - t0 < 240
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"error"' in ln]
    assert lines, p.stdout[-2000:] + p.stderr[-2000:]
    j = json.loads(lines[-1])
    assert j["value"] is None and "first sharded replays did not complete" in j["error"] and j["n_gpus"] == 2
