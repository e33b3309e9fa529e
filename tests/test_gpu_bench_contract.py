"""bench.py prints ONE JSON line with the contract's keys (run with -m gpu; short run, CPU baseline leg skipped here --
the default invocation includes it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def _check_roofline(r, peak):
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["peak"] == peak and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1


def test_bench_json_contract_default():
    j = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_reparam", "stats", "one_step_in_flight", "dropin_loop", "configs"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["unit"] == "samples/s" and j["scaling"] == "strong" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert j["metric"] == "MC-forward samples/sec, BayesianAlexNet CIFAR-10 bs=512 num_ens=10" and j["dtype"] == "f32"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 512 * 10 / (j["ms_per_step"] * 1e-3)) <= 1e-3 * j["value"]
    _check_roofline(j["roofline"], 157.3)
    rr = j["roofline_reparam"]
    assert rr["bound"] == "hbm" and rr["peak"] == 8000.0 and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-3
    st = j["stats"]
    assert st["p10"] <= st["median"] <= st["p90"]
    assert "error" not in j["dropin_loop"] and j["dropin_loop"]["value"] > 0
    assert "error" not in j["training_step"] and j["training_step"]["path"] == "chwn-autograd"
    dflt = j["training_step"]["reference_default_config"]             # config_bayesian.py defaults: lrt, bs 256, num_ens 1
    assert "error" not in dflt and dflt["path"] == "chwn-autograd" and 0 < dflt["hipgraph_ms_per_step"] <= dflt["eager_ms_per_step"] * 1.5
    # every other BASELINE configuration is measured, each with its own roofline
    assert set(j["configs"]) == {"configs[1]", "configs[2]", "configs[3]", "configs[4]"}
    for name, c in j["configs"].items():
        assert "error" not in c, (name, c)
        assert c["value"] > 0 and c["roofline"] is not None
        _check_roofline(c["roofline"], 2500.0 if c["dtype"] == "bf16" else 157.3)
    assert j["configs"]["configs[1]"]["dtype"] == "bf16" and j["configs"]["configs[4]"]["rows_out"] == 512 * 49


def test_bench_json_contract_other_config_as_headline():
    j = _run(["--config", "configs[1]", "--pipeline", "1", "--no-extras"])
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
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["scaling"] == "strong" and j["value"] > 0
    assert j["config"]["global_batch"] == 512 and j["config"]["num_ens_total"] == 10          # the metric's workload, not 2x of it
    assert "work units" in j["config"]["parallelism"] and j["weak_scaling"]["num_ens_total"] == 20
    assert abs(j["value"] - 512 * 10 / (j["ms_per_step"] * 1e-3)) <= 1e-3 * j["value"]


def test_bench_gpus_2_as_typed_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run: bench.py re-executes itself under the launcher (one process
    per GPU, rendezvous on 127.0.0.1 with a free port).  Rehearsed on one MI355X through the same hooks as above."""
    env = dict(os.environ, BBB_BENCH_DEVICE="0", BBB_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-extras"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    assert j["config"]["ranks_seen"] == 2 and sum(j["config"]["units_per_rank"]) == 10 and j["config"]["backend"] == "gloo"
