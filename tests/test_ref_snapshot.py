"""oracle/ref_snapshot.py: the archive of UNMODIFIED upstream files that carries the reference to the GPU box (test
infrastructure; the product never reads it -- tests/test_host_cpu.py::test_no_oracle_import_in_product)."""
import hashlib
import os
import subprocess
import sys

import pytest

import ref_snapshot as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.reference
def test_archive_is_byte_identical_to_the_checkout(tmp_path):
    if not os.path.isdir(R.UPSTREAM):
        pytest.skip("build container only: compares the archive with /root/reference")
    arc = str(tmp_path / "snap.zip")
    m = R.build_snapshot(archive=arc)
    assert m is not None and "main_bayesian.py" in m["files"] and "layers/BBB/BBBConv.py" in m["files"]
    assert "models/BayesianModels/BayesianAlexNet.py" in m["files"] and "utils.py" in m["files"] and "metrics.py" in m["files"]
    for n, meta in m["files"].items():
        with open(os.path.join(R.UPSTREAM, n), "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == meta["sha256"], n
    first = open(arc, "rb").read()
    R.build_snapshot(archive=arc)                                  # idempotent, reproducible
    assert open(arc, "rb").read() == first
    # the in-tree archive (what travels to the GPU box) is the same tree
    assert R.manifest() is not None and R.manifest()["tree_sha256"] == m["tree_sha256"]


@pytest.mark.reference
def test_unpacked_snapshot_serves_the_upstream_modules():
    """What the GPU box does: no checkout, unpack, import the upstream model on the upstream layers, one CPU forward."""
    if R.manifest() is None:
        pytest.skip("no oracle/_ref/upstream_snapshot.zip (run __graft_entry__.build() in the build container)")
    code = r'''
import sys; sys.dont_write_bytecode = True
sys.path.insert(0, %r)
import ref_snapshot as R
R.UPSTREAM = "/nonexistent"
kind, d = R.checkout()
assert kind == "snapshot", kind
sys.path.insert(0, d)
import torch
from unittest import mock
from models.BayesianModels.BayesianLeNet import BBBLeNet
import layers, config_bayesian as cfg, utils
assert layers.__file__.startswith(d)
with mock.patch("torch.cuda.is_available", return_value=False):
    net = BBBLeNet(10, 1, cfg.priors, "lrt", "softplus")
out, kl = net(torch.rand(4, 1, 32, 32))
assert out.shape == (4, 10) and kl.dim() == 0
print("OK")
''' % os.path.join(ROOT, "oracle")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]


@pytest.mark.reference
def test_bench_cpu_baseline_times_the_upstream_modules_from_the_archive():
    """bench.py's CPU leg as the GPU box runs it: no checkout, the archive unpacked, the upstream BBBAlexNet on the upstream
    layers (constructed seeing no GPU), kind "reference"."""
    if R.manifest() is None:
        pytest.skip("no oracle/_ref/upstream_snapshot.zip")
    code = r'''
import sys, json; sys.dont_write_bytecode = True
sys.path.insert(0, %r); sys.path.insert(0, %r)
import ref_snapshot as R
R.UPSTREAM = "/nonexistent"
import bench
r = bench.cpu_baseline(0.5)
assert r["kind"] == "reference" and r["value"] > 0 and r["steps_timed"] >= 1, r      # (a loaded host times one step inside the 0.5 s budget)
assert "upstream_snapshot.zip" in r["sample"] and "sha256" in r["sample"], r["sample"]
print("OK", r["value"], r["cores"])
''' % (os.path.join(ROOT, "oracle"), ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]
