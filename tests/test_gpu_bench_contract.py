"""bench.py prints ONE JSON line with the contract's keys (run with -m gpu; short run, CPU baseline leg skipped here --
the default invocation includes it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--dtype", "bf16", "--pipeline", "1"]])
def test_bench_json_contract(extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["unit"] == "samples/s" and j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 512 * 10 / (j["ms_per_step"] * 1e-3)) <= 1e-3 * j["value"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    if not extra:
        assert j["dtype"] == "f32" and r["peak"] == 157.3 and "roofline_reparam" in j and "bf16" in j
        assert j["roofline_reparam"]["bound"] == "hbm"
    else:
        assert j["dtype"] == "bf16" and r["peak"] == 2500.0
