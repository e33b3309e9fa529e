#!/usr/bin/env python3
"""MC-forward samples/sec of BayesianAlexNet (CIFAR-10 shape, bs=512, num_ens=10, BBB layers) on MI355X.

One step = main_bayesian.py:73-80 of the reference: num_ens x net(x) + log_softmax, then utils.logmeanexp --
forward only, no_grad, inputs resident in HBM.  Synthetic data: torch.manual_seed(0), parameters from the
layers' own reset_parameters with config_bayesian.priors, x ~ U[0,1).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Launch structure of the timed region (N = 1): hipGraph replays; a graph holds `--steps-per-launch` (default 4) consecutive steps
-- 4 batches x 10 draws = 40 slabs per layer launch, each step with its own noise calls, bit for bit what 4 separate replays
compute (tests/test_gpu_steps_per_launch.py) -- and `--pipeline` (default 2) such graphs are in flight on separate HIP streams.
A device-side call counter inside the graph advances the Philox noise on every replay.  The one-step-per-launch pipeline of
rounds 1-3 (3 lanes) is timed beside it (`roofline.one_step_per_launch_ms`).
Order of a run: CPU baseline (GPU idle, ~20 s) -> graphs built -> K steps timed COLD (`cold_first_block`, reported, never
`value`) -> `preheat_ms` of the same replays, untimed (the part clocks up under load; a 20-step region is 13 ms) -> W warm-up
steps -> barrier + sync -> EXACTLY K timed steps -> sync + barrier.

N > 1 is STRONG scaling of the metric's workload, with the same launch structure: the 4 x 10 draws of a group of four steps (each
step the same 512 images x 10 draws) are dealt to the ranks as contiguous ranges of WHOLE draws on whole batches
(ensemble.group_share; 8 ranks: 5 draws each, a step's draws on two ranks), every rank sampling only the weight sets its draws
use, ONE all_gather of [4 * B*C + 1] floats per group over RCCL, recorded into the lane's hipGraph.  value = 512 * 10 / time per
step.  `--steps-per-launch 1` shards ONE step as (draw x batch-slice) work units instead (ensemble.shard_plan; 8 ranks: 40
quarter-batch units, 5 each).  (`weak_scaling`: a short secondary run with 10 draws PER rank, reported next to it.)

Output: the LAST line is the contract's JSON object, kept under 2000 characters, with every judged number as a scalar inside
`roofline` / `cpu_baseline` (the driver's record keeps those two objects' scalars):
  roofline       the dominant kernel (fp32-MFMA implicit-GEMM conv): algorithmic FLOP / HIP-event time per launch vs the 157.3
                 TFLOP/s fp32 matrix peak (`frac`, `per_launch_us`, `traffic` from the committed rocprofv3 PMC passes), plus
                 sustained_frac (same FLOPs / ms_per_step), stats_* (5 blocks of K steps), one_step_in_flight_ms (one lane, one step
                 per launch), and the fused reparam+KL pass vs HBM 8 TB/s as reparam_frac / reparam_avg_us / reparam_bytes (also
                 nested: `reparam`).
  cpu_baseline   the reference's CPU nn.Module path timed on this host's cores (rank 0): the unmodified upstream modules (kind
                 "reference": the checkout, or on the GPU box the archive __graft_entry__.build() packed into oracle/_ref/);
                 only when neither exists the oracle's bit-identical torch-CPU port (kind "port").
A line starting with `SECONDARY ` is printed BEFORE it and carries the bulky objects: every other BASELINE configuration with its
own roofline, the drop-in loop, the split-16-bit mode, the training step, per-launch detail and notes.
"""
import argparse
import json
import os
import re
import statistics
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before HIP initialises: one hardware queue per lane (bbb_hip/__init__.py)
os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")   # before the process group exists: recorded collectives (ensemble.collective_capture_ok)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
sys.dont_write_bytecode = True

import torch  # noqa: E402

PRIORS = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense v_mfma_f32_32x32x16_bf16 peak (same guide)
FIRST_CONTACT_TIMEOUT_S = float(os.environ.get("BBB_BENCH_FIRST_CONTACT_TIMEOUT_S", "300"))   # watchdog around a sharded run's first replays
PEAK_HBM_GBS = 8000.0             # HBM3E spec (6.3 TB/s achievable)

# BASELINE.json: metric config + configs[1..4] (configs[0] is the reference's CPU-only plumbing case)
CONFIGS = {
    "metric": dict(net="alexnet", classes=10, lt="bbb", B=512, E=10, hw=32, precision="fp32",
                   what="BayesianAlexNet 3x32x32 -> 10, layer_type=bbb, softplus, bs=512, num_ens=10"),
    "configs[1]": dict(net="3conv3fc", classes=10, lt="bbb", B=256, E=1, hw=32, precision="bf16",
                       what="Bayesian3Conv3FC CIFAR-10, BBB layers, bf16 storage, batch 256, num_ens=1"),
    "configs[2]": dict(net="alexnet", classes=100, lt="lrt", B=512, E=1, hw=32, precision="fp32",
                       what="BayesianAlexNet CIFAR-100, BBB_LRT layers, batch 512, num_ens=1"),
    "configs[3]": dict(net="alexnet", classes=10, lt="bbb", B=512, E=25, hw=32, precision="fp32",
                       what="BayesianAlexNet CIFAR-10, num_ens=25 (the 8-GPU ensemble of configs[3], all 25 draws on this GPU)"),
    "configs[4]": dict(net="alexnet", classes=10, lt="bbb", B=512, E=1, hw=224, precision="fp32",
                       what="BayesianAlexNet 3x224x224, 512 images = one GPU's shard of the batch-4096 data-parallel run"),
}


def usable_cpus():
    """CPUs this process may actually use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=5).stdout
        m = re.search(r"Model name:\s*(.+)", out)
        return m.group(1).strip() if m else None
    except Exception:
        return None


def pctl(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    i = q * (len(xs) - 1)
    lo, hi = int(i), min(int(i) + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (i - lo)


def cpu_baseline(budget_s=20.0):
    """The reference's CPU nn.Module path on the metric config (bs=512, num_ens=10, bbb, fp32), a bounded sample."""
    cfg = CONFIGS["metric"]
    B, E, C = cfg["B"], cfg["E"], cfg["classes"]
    avail = usable_cpus()
    # the UNMODIFIED upstream modules: the checkout in the build container, on the GPU box the byte-for-byte archive that
    # __graft_entry__.build() packed into oracle/_ref/ (oracle/ref_snapshot.py; test infrastructure, never the product path)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_snapshot
    ref_kind, ref_dir = ref_snapshot.checkout()
    torch.manual_seed(0)
    x = torch.rand(B, 3, 32, 32)
    if ref_dir is not None:
        kind = "reference"
        saved_path, saved_mods = list(sys.path), {k: v for k, v in sys.modules.items() if k == "layers" or k.startswith("layers.")}
        for k in list(saved_mods):
            del sys.modules[k]
        sys.path.insert(0, ref_dir)
        try:
            from models.BayesianModels.BayesianAlexNet import BBBAlexNet as RefAlexNet
            import utils as ref_utils
            import torch.nn.functional as F
            # the upstream layers pin themselves to cuda:0 whenever a GPU is visible (layers/BBB/BBBConv.py:27); this leg is the
            # reference's CPU path, so the modules are CONSTRUCTED seeing what a GPU-less host shows them (nothing upstream is edited)
            from unittest import mock
            with mock.patch("torch.cuda.is_available", return_value=False):
                net = RefAlexNet(C, 3, PRIORS, "bbb", "softplus")
            assert all(p.device.type == "cpu" for p in net.parameters()) and net.conv1.device.type == "cpu"

            def step():                                   # main_bayesian.py:73-80
                outputs = torch.zeros(B, C, E)
                kl = 0.0
                for j in range(E):
                    net_out, _kl = net(x)
                    kl += _kl
                    outputs[:, :, j] = F.log_softmax(net_out, dim=1).data
                return ref_utils.logmeanexp(outputs, dim=2), kl
            src = ("unmodified upstream modules (models.BayesianModels.BayesianAlexNet on the upstream layers/, utils.logmeanexp) imported from "
                   + ("the read-only checkout" if ref_kind == "checkout" else "oracle/_ref/upstream_snapshot.zip (sha256-verified, tree %s)"
                      % ref_snapshot.manifest()["tree_sha256"][:12]))
        finally:
            sys.path[:] = saved_path
            for k in [k for k in sys.modules if k == "layers" or k.startswith("layers.")]:
                del sys.modules[k]
            sys.modules.update(saved_mods)
    else:
        kind = "port"
        import ref_port_torch as P
        params = P.init_params("alexnet", 3, C, P.CONFIG_PRIORS)

        def step():
            return P.mc_step("alexnet", params, x, C, E, "bbb", "softplus")
        src = "oracle/ref_port_torch.mc_step (same ATen ops / draw order as the upstream modules, bit-identical under a seed)"

    def run(nthreads, budget, grad):
        torch.set_num_threads(nthreads)
        times = []
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx:
            step()                                        # warm-up
            t_all = time.perf_counter()
            while True:
                t0 = time.perf_counter()
                step()
                times.append(time.perf_counter() - t0)
                el = time.perf_counter() - t_all
                if (len(times) >= 3 and el > budget) or len(times) >= 100 or el > 4 * budget:
                    break
        return times

    # mkldnn does not always scale to every core of the cgroup: time all cores and half of them, keep the faster
    best = None
    for nt in sorted({avail, max(1, avail // 2)}, reverse=True):
        ts = run(nt, budget_s * 0.3, False)
        med = statistics.median(ts)
        if best is None or med < best[0]:
            best = (med, ts, nt)
    med, ts, nt = best
    tg = run(nt, budget_s * 0.25, True)                   # validate_model does not disable autograd (main_bayesian.py:65-86)
    torch.set_num_threads(avail)
    n = B * E
    # beside the CPU path: the SAME unmodified modules on this box's GPU through PyTorch-ROCm's own kernels (ATen / MIOpen / rocBLAS; eps
    # from the CPU generator, copied over: layers/BBB/BBBConv.py:63) -- what a user of the reference gets on an MI355X today.  A
    # subprocess (oracle/ref_gpu_path.py; MIOpen picks its kernels in the first step) with a hard time limit; a reported baseline.
    ref_gpu = None
    if kind == "reference" and torch.cuda.is_available():
        import subprocess
        try:
            pr = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_gpu_path.py"), "2.5"], capture_output=True, text=True,
                                timeout=120)
            line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
            ref_gpu = json.loads(line[-1]) if (pr.returncode == 0 and line) else {"error": (pr.stderr or "no output")[-200:]}
        except Exception as exc:                         # noqa: BLE001 (a baseline must never take the bench down)
            ref_gpu = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
    return {"value": round(n / med, 1), "unit": "samples/s", "cores": nt, "kind": kind, "reference_gpu_path": ref_gpu,
            "p10": round(n / pctl(ts, 0.9), 1), "p90": round(n / pctl(ts, 0.1), 1), "steps_timed": len(ts),
            "autograd_enabled": {"value": round(n / statistics.median(tg), 1), "p10": round(n / pctl(tg, 0.9), 1),
                                 "p90": round(n / pctl(tg, 0.1), 1), "steps_timed": len(tg)},
            "cpu_model": cpu_model_name(), "torch_num_threads": nt, "usable_cpus": avail, "logical_cpus": os.cpu_count(),
            "sample": f"{len(ts)} full MC steps (bs={B}, num_ens={E}, fp32, no_grad; median) + {len(tg)} with autograd enabled, "
                      f"after a warm-up each; {src}; {avail} CPUs usable (cgroup quota / affinity), best of {{all, half}} thread counts"}


def profile_traffic(kind, suffix=""):
    """HBM bytes per launch-group from the committed PMC summaries (profiles/r06_pmc_FETCH_SIZE<suffix>.txt / _WRITE_SIZE<suffix>.txt,
    written by profiles/collect.sh; suffix "" = the metric configuration, "_configs1" ... = BASELINE configs[1..4], "_split" = the
    split-bf16 mode): (read_bytes, written_bytes, source) or None.  FETCH_SIZE x2 = the gfx950 wide-load correction.
    kind: one kernel family or a tuple of families whose STEP_TOTAL rows are added."""
    out = {}
    tag = None
    for t in ("r06", "r05", "r04", "r03", "r02"):  # the newest committed PMC passes
        if all(os.path.exists(os.path.join(ROOT, "profiles", f"{t}_pmc_{c}{suffix}.txt")) for c in ("FETCH_SIZE", "WRITE_SIZE")):
            tag = t
            break
    if tag is None:
        return None
    kinds = (kind,) if isinstance(kind, str) else tuple(kind)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_{c}{suffix}.txt")
        kb = None
        for k in kinds:
            last = None
            for line in open(path):
                if line.startswith("##"):
                    break                                # (a second section of the file: another launch mode)
                m = re.match(r"\s*STEP_TOTAL\s+%s\s+(\S+)\s+KB_per_step=([0-9.]+)" % re.escape(k), line)
                if m:
                    last = float(m.group(2))
            if last is not None:
                kb = (kb or 0.0) + last
        if kb is None:
            return None
        out[c] = kb * 1024.0
    rd = 2.0 * out["FETCH_SIZE"]
    return rd, out["WRITE_SIZE"], "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 wide-load correction) + WRITE_SIZE, separate passes: " \
        "profiles/%s_pmc_FETCH_SIZE%s.txt, profiles/%s_pmc_WRITE_SIZE%s.txt (STEP_TOTAL %s rows)" % (tag, suffix, tag, suffix, "+".join(kinds))


def build_net(cfg, dev):
    from bbb_hip import rng, zoo
    torch.manual_seed(0)
    net = zoo.getModel(cfg["net"], 3, cfg["classes"], PRIORS, cfg["lt"], "softplus").to(dev)
    rng.assign_stream_ids(net)
    x = torch.rand(cfg["B"], 3, cfg["hw"], cfg["hw"]).to(dev)
    return net, x


def mfma_loop_ceiling():
    """TFLOP/s of a loop of nothing but v_mfma_f32_32x32x2_f32 on this box (profiles/probe/mfma_ceiling.hip, built by build.sh as
    its own library -- a measurement aid, not part of libbbb_hip.so): the ceiling a kernel built on that instruction can
    approach, next to the guide's nominal 157.3.  None when the probe library is not there."""
    import ctypes
    path = os.path.join(ROOT, "profiles", "probe", "libmfma_probe.so")
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        lib.probe_mfma_f32_ceiling.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_void_p]
        v = (ctypes.c_double * 2)(0.0, 0.0)
        rc = lib.probe_mfma_f32_ceiling(v, 5, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return (round(v[0], 1), round(v[1], 3)) if rc == 0 and v[0] > 0 else None
    except Exception:
        return None


def write_roof_probe(elems_per_draw, draws):
    """GB/s of a write-only kernel with the store shape of the fused reparam+KL pass (profiles/probe/mfma_ceiling.hip,
    probe_write_roof: same bytes, 16-byte stores, draws strided by the tensor size, no reads, no arithmetic): the write roof of THIS
    box for that pattern.  (plain, non-temporal) or None."""
    import ctypes
    path = os.path.join(ROOT, "profiles", "probe", "libmfma_probe.so")
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        lib.probe_write_roof.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        v = (ctypes.c_double * 2)(0.0, 0.0)
        rc = lib.probe_write_roof(v, int(elems_per_draw), int(draws), 5, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return (round(v[0], 1), round(v[1], 1)) if rc == 0 and max(v[0], v[1]) > 0 else None
    except Exception:
        return None


def gemm_roofline(agg, timer_steps, precision, metric_cfg):
    """Dominant kernel of the step: all conv / linear launches.  FLOPs counted = in-bounds taps only (what the fp32 kernel
    multiplies; the bf16 kernel also multiplies the zero taps of first layers, which is not counted as useful work)."""
    g = agg.get("conv_gemm") or agg.get("lrt_gemm")
    if not g:
        return None
    if "conv_gemm" in agg and "lrt_gemm" in agg:
        g = {k: agg["conv_gemm"][k] + agg["lrt_gemm"][k] for k in ("ms", "n", "work", "work_im2col")}
    tf = g["work"] / (g["ms"] * 1e-3) / 1e12
    peak = PEAK_BF16_MFMA_TFLOPS if precision == "bf16" else PEAK_F32_MFMA_TFLOPS
    kern = ("pconv_bf16_* (v_mfma_f32_32x32x16_bf16: pooled first layer, strip form over channel-interleaved input, general kernel)" if precision == "bf16" else
            "pconv_gemm_kernel (fp32 v_mfma_f32_32x32x2_f32, batch-innermost, in-bounds taps only)")
    r = {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
         "traffic": None, "kernel": kern + ", all conv/linear launches of a step",
         "launches": g["n"], "avg_us": round(1e3 * g["ms"] / g["n"], 2),
         "flop_per_step": g["work"] / timer_steps, "im2col_flop_per_step": g["work_im2col"] / timer_steps,
         "timed_by": "HIP events around every launch, %d eager single-stream steps of the same workload" % timer_steps}
    if metric_cfg:
        tr = profile_traffic("pconv_gemm")
        if tr:
            r["traffic"] = round(tr[0] + tr[1], 1)
            r["traffic_read_written"] = [round(tr[0], 1), round(tr[1], 1)]
            r["traffic_source"] = tr[2] + "; bytes per step over the 6 conv/linear launches"
    return r


def preheat(fn, seconds, dev):
    """Run fn() back to back for `seconds` (untimed): a part that idled through the CPU baseline clocks up under load over the
    first tens of milliseconds; short timed regions would otherwise measure that ramp.  Returns the milliseconds spent."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            fn()
        torch.cuda.synchronize(dev)
    return round(1e3 * (time.perf_counter() - t0), 1)


class LaunchRecorder:
    """Stands in for ensemble.Timers: keeps every bracketed launch (a closure over its operands) so that it can be replayed."""
    reps = 20

    def __init__(self):
        self.calls = []
        self.per_launch_us = []

    def bracket(self, tag, info, fn):
        out = fn()
        self.calls.append((tag, info(out) if callable(info) else info, fn))
        return out

    def time_in_graphs(self, dev, heat_s=0.04):
        """Every recorded GEMM launch replayed `reps` times back to back inside its own hipGraph (the timed region's launch
        mode: no host, no event packets between kernels); the graph is replayed for heat_s seconds first (same clocks as the
        timed region), then HIP events bracket 3 replays enqueued without a host sync in between; median of 3 such brackets."""
        from bbb_hip import ops as _ops
        agg = {}
        for tag, info, fn in self.calls:
            if tag not in ("conv_gemm", "lrt_gemm"):
                continue
            for _ in range(2):
                fn()
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with _ops.graph_capture(g):
                for _ in range(self.reps):
                    fn()
            preheat(g.replay, heat_s, dev)
            ts = []
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(3):
                    g.replay()
                e.record()
                torch.cuda.synchronize(dev)
                ts.append(s.elapsed_time(e) / (3 * self.reps))
            ms = statistics.median(ts)
            del g
            a = agg.setdefault(tag, {"ms": 0.0, "n": 0, "work": 0.0, "work_im2col": 0.0})
            a["ms"] += ms
            a["n"] += 1
            w = info if isinstance(info, tuple) else (info, info)
            a["work"] += w[0]
            a["work_im2col"] += w[1]
            self.per_launch_us.append(round(ms * 1e3, 2))
        return agg


def reparam_probe(net, dev, n_params, E, hbm_probe=True):
    """Fused reparam+KL pass timed on its own: 20 back-to-back launches inside one hipGraph (no host gaps), pre-heated, HIP
    events around the replay, median of 7 replays.  (a) the model's 12 tensors, E draws per launch - what the step runs
    (E = steps_per_launch x num_ens), Infinity-Cache resident up to ~10 draws; (b) one 2^26-element tensor (0.8-3.2 GB of
    traffic, far beyond the 256 MB cache) for a genuine HBM figure next to a device copy."""
    from bbb_hip import ensemble, ops
    mus, rhos, ids = [], [], []
    for l in ensemble.bayesian_layers(net):
        m, r, i = l._param_lists()
        mus += m
        rhos += r
        ids += i

    def timed(fn, reps, heat=0.03):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with ops.graph_capture(g):
            for _ in range(reps):
                fn()
        preheat(g.replay, heat, dev)
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            torch.cuda.synchronize(dev)
            ts.append(s.elapsed_time(e) * 1e-3 / reps)
        return statistics.median(ts), min(ts), max(ts)

    with torch.no_grad():
        t_model, t_lo, t_hi = timed(lambda: ops.reparam_kl_forward(mus, rhos, 0, 0.1, ids, 1, 0, draws=E), 20)
        byts = (8 + 4 * E) * n_params
        gbs = byts / t_model / 1e9
        r = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
             "traffic": None, "draws_per_launch": E,
             "kernel": "reparam_kl_fast_kernel: 12 tensors x %d draws + the KL sum in ONE launch (Philox4x32-7, packed softplus/KL)" % E,
             "bytes_per_launch": byts, "avg_us": round(t_model * 1e6, 2), "min_us": round(t_lo * 1e6, 2), "max_us": round(t_hi * 1e6, 2)}
        if hbm_probe:
            big = 1 << 26
            mu = torch.randn(big, device=dev) * 0.1
            rho = torch.randn(big, device=dev) * 0.1 - 5
            t1 = timed(lambda: ops.reparam_kl_forward([mu], [rho], 0, 0.1, [0], 1, 0, draws=1), 3, 0.0)[0]
            t10 = timed(lambda: ops.reparam_kl_forward([mu], [rho], 0, 0.1, [0], 1, 0, draws=10), 2, 0.0)[0]
            dst = torch.empty_like(mu)
            tc = timed(lambda: dst.copy_(mu), 3, 0.0)[0]
            del mu, rho, dst
            r["hbm_resident_probe"] = {"elements": big, "E1_GBps": round(12 * big / t1 / 1e9, 1), "E10_GBps": round(48 * big / t10 / 1e9, 1),
                                       "device_copy_GBps": round(8 * big / tc / 1e9, 1)}
    tr = profile_traffic("reparam")
    if tr and E == 10:
        r["traffic"] = round(tr[0] + tr[1], 1)
        r["traffic_read_written"] = [round(tr[0], 1), round(tr[1], 1)]
        r["traffic_source"] = tr[2]
    return r


def reparam_region_probe(net, x, E, G, dev, n_params, reps=10):
    """The fused reparam + KL launch where the timed region runs it: first launch of a group of G steps (G x E draws), the group's
    GEMMs behind it.  Twenty launches back to back (reparam_probe) find the 256 MB Infinity Cache full of the previous launch's
    dirty lines -- a state the region never has: 2.4 ms of matrix-bound GEMMs separate two parameter passes of a lane.  Here the
    launch is followed by the group's first GEMM launch (conv1 + pool1: ~0.48 ms for the cache to drain), `reps` pairs inside one
    hipGraph, against a graph of the `reps` GEMM launches alone: the difference is the pass's marginal cost in its own context."""
    from bbb_hip import ensemble, ops, rng
    rec = LaunchRecorder()
    xg = x.repeat(G, 1, 1, 1) if G > 1 else x
    with torch.no_grad():
        seed, call0 = rng.next_calls(0)
        ensemble._local_lse(net, xg, E, seed, call0, E, timers=rec, **({"groups": G} if G > 1 else {}))
        torch.cuda.synchronize(dev)
        rp = [fn for tag, _, fn in rec.calls if tag == "reparam_kl"]
        gm = [fn for tag, _, fn in rec.calls if tag == "conv_gemm"]
        if len(rp) != 1 or not gm:
            return None

        def timed(fns):
            for _ in range(2):
                for f in fns:
                    f()
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with ops.graph_capture(g):
                for _ in range(reps):
                    for f in fns:
                        f()
            preheat(g.replay, 0.05, dev)
            ts = []
            for _ in range(9):
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                g.replay()
                e_.record()
                torch.cuda.synchronize(dev)
                ts.append(s_.elapsed_time(e_) * 1e-3 / reps)
            return statistics.median(ts)

        t_pair = timed([rp[0], gm[0]])
        t_gemm = timed([gm[0]])
    dt = t_pair - t_gemm
    byts = (8 + 4 * E * G) * n_params
    return {"avg_us": round(dt * 1e6, 2), "frac": round(byts / dt / 1e9 / PEAK_HBM_GBS, 4), "draws_per_launch": E * G,
            "bytes_per_launch": byts, "pair_us": round(t_pair * 1e6, 2), "gemm_alone_us": round(t_gemm * 1e6, 2),
            "timed_by": "hipGraph of %d x [reparam, conv1 launch] minus hipGraph of %d x [conv1 launch], HIP events, median of 9" % (reps, reps)}


def run_config(cfg, steps, warmup, pipeline, dev, group=None, world=1, want_roofline=True, stat_blocks=0, timer_steps=5,
               total_ens=None, steps_per_launch=1, preheat_s=0.4, cold_block=False, single_lane=True, rank=0, multi=None):
    """Throughput of one configuration: hipGraph lanes, K timed steps between device syncs (max over ranks) -> dict.
    steps_per_launch = G > 1: every lane's graph holds G consecutive steps (GraphedPipeline).  cold_block: the K steps are also
    timed once BEFORE the pre-heat (reported as cold_first_block, never the value)."""
    from bbb_hip import ensemble
    net, x = build_net(cfg, dev)
    E = cfg["E"] if total_ens is None else total_ens
    prec = cfg["precision"]
    G = int(steps_per_launch)
    if multi is None:
        multi = world > 1                                # the process-group flow (also rehearsed at world size 1: BBB_BENCH_SELF_GROUP)

    def barrier():
        if multi:
            torch.distributed.barrier(group=group)
        torch.cuda.synchronize(dev)

    out = {}

    def build_and_first_replays():
        with torch.no_grad():
            if pipeline > 1 or G > 1:
                g = ensemble.GraphedPipeline(net, x, E, depth=max(1, pipeline), group=group, precision=prec, steps_per_launch=G)
            else:
                g = ensemble.GraphedMC(net, x, E, group=group, precision=prec)
            fl = g.sync if G > 1 else (lambda: None)         # a partly filled group is launched before the clock stops
            for _ in range(max(1, pipeline) * G):   # setup: every lane's graph is replayed once (first replay = its upload)
                g.step()
            fl()
            torch.cuda.synchronize(dev)
            stall = float(os.environ.get("BBB_BENCH_TEST_STALL_S", "0"))   # test hook: this rank never finishes its first replays
            if stall and rank == int(os.environ.get("BBB_BENCH_TEST_STALL_RANK", "-1")):
                time.sleep(stall)
            return g

    if multi:
        # FIRST CONTACT of a sharded run (review r04 item 8: an 8-GPU job whose first recorded collective hangs would sit in the
        # driver's lease forever): the pipeline is built and every lane replayed once under a wall-clock watchdog.  A stall with
        # collectives recorded into the graphs -> once more with the eager-collective protocol on fresh streams; a second stall
        # (or a stall of the eager protocol: a rank is simply missing) -> a diagnostic line in the judged format and exit code 3.
        done, res = ensemble._with_watchdog(build_and_first_replays, FIRST_CONTACT_TIMEOUT_S)
        protocol = ensemble.last_protocol.get("collective")
        if not done and res is None and protocol == "recorded":
            ensemble.force_eager_collectives(group)
            out["first_contact"] = "recorded collectives stalled > %.0f s: fell back to the eager-collective protocol" % FIRST_CONTACT_TIMEOUT_S
            done, res = ensemble._with_watchdog(build_and_first_replays, FIRST_CONTACT_TIMEOUT_S)
        if not done:
            why = ("no progress within %.0f s" % FIRST_CONTACT_TIMEOUT_S) if res is None else "%s: %s" % (type(res).__name__, str(res)[:200])
            print(json.dumps({"metric": "MC-forward samples/sec, " + cfg["what"], "value": None, "unit": "samples/s", "n_gpus": world,
                              "error": "first sharded replays did not complete on rank %d (%s); collective protocol %s"
                                       % (rank, why, ensemble.last_protocol)}), flush=True)
            os._exit(3)
        gstep = res
        out["collective_protocol"] = dict(ensemble.last_protocol)
    else:
        gstep = build_and_first_replays()
    with torch.no_grad():
        step = gstep.step
        flush = gstep.sync if G > 1 else (lambda: None)

        def timed_block(n):
            barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                r = step()
            flush()
            barrier()
            return time.perf_counter() - t0, r

        if cold_block:
            dt, _ = timed_block(steps)
            if multi:
                t = torch.tensor([dt], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
                dt = t.item()
            out["cold_first_block"] = {"ms_per_step": round(1e3 * dt / steps, 4), "value": round(cfg["B"] * E * steps / dt, 1)}
        # pre-heat: the same replays, untimed (every rank runs the same number of collectives: a fixed step count from rank 0's clock)
        t0 = time.perf_counter()
        if preheat_s > 0:
            if multi:
                n_heat = torch.tensor([0], device=dev, dtype=torch.int64)
                if rank == 0:
                    dt1, _ = timed_block(max(8, steps))
                    n_heat.fill_(int(preheat_s / (dt1 / max(8, steps))))
                else:
                    timed_block(max(8, steps))
                torch.distributed.broadcast(n_heat, 0, group=group)
                for _ in range(int(n_heat.item())):
                    step()
                flush()
                torch.cuda.synchronize(dev)
            else:
                while time.perf_counter() - t0 < preheat_s:
                    for _ in range(max(1, pipeline) * G * 2):
                        step()
                    flush()
                    torch.cuda.synchronize(dev)
        out["preheat_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
        for _ in range(warmup):
            step()
        flush()
        elapsed, (lo, kl) = timed_block(steps)
        lo, kl = lo.clone(), kl.clone()                  # (checked for finiteness AFTER the statistics blocks below: a host round trip here
        if multi:                                        #  idles the part, and the block behind it measured 15 % low on every box)
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
            elapsed = t.item()
        rows = lo.shape[0]
        out["ms_per_step"] = round(1e3 * elapsed / steps, 4)
        out["value"] = round(cfg["B"] * E / (elapsed / steps), 1)
        out["rows_out"] = rows
        if stat_blocks and not multi:
            # block statistics: `stat_blocks` MORE blocks of exactly `steps` steps, each between device syncs.  The reported value is
            # the MEDIAN over all 1 + stat_blocks blocks (a single block lands anywhere in the box's clock spread -- review r05
            # item 4); the first block alone stays next to it as `single_block`.
            vals = [cfg["B"] * E * steps / elapsed]
            for _ in range(stat_blocks):
                dt, _ = timed_block(steps)
                vals.append(cfg["B"] * E * steps / dt)
            med = statistics.median(vals)
            out["single_block"] = {"ms_per_step": out["ms_per_step"], "value": out["value"]}
            out["value"] = round(med, 1)
            out["ms_per_step"] = round(1e3 * cfg["B"] * E / med, 4)
            out["stats"] = {"blocks": len(vals), "steps_per_block": steps, "median": round(med, 1),
                            "p10": round(pctl(vals, 0.1), 1), "p90": round(pctl(vals, 0.9), 1), "unit": "samples/s",
                            "value_is": "median of the blocks", "block_values": [round(v, 1) for v in vals]}
        torch.cuda.synchronize(dev)
        assert torch.isfinite(lo).all() and torch.isfinite(kl).all()
        del gstep, step, flush
        if not multi and single_lane and (pipeline > 1 or G > 1):
            g1 = ensemble.GraphedMC(net, x, E, precision=prec)
            preheat(g1.step, 0.1, dev)
            n1 = max(steps, 10)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(n1):
                g1.step()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t1) / n1
            out["one_step_in_flight"] = {"value": round(cfg["B"] * E / dt, 1), "ms_per_step": round(1e3 * dt, 4)}
            del g1
        if want_roofline and rank == 0:
            S, lo_u, hi_u = ensemble.shard_plan(net, x, E, 0, world, True, prec)
            timers = ensemble.Timers()
            # park the GPU behind a ~40 ms spin kernel so that every launch below is already queued when its turn comes:
            # the event brackets then hold kernel time only, not host launch latency
            torch.cuda._sleep(int(1.0e8))
            from bbb_hip import rng
            if G > 1 and multi:                          # rank 0's share of a group of G steps: whole draws on whole batches
                lo_u, hi_u, _g0, n_gl, g_off = ensemble.group_share(E, G, 0, world)
                xg = x.repeat(n_gl, 1, 1, 1)

                def one_pass(t):
                    seed, call0 = rng.next_calls(0)      # (not advanced: only rank 0 runs this, and the ranks' counters must stay equal)
                    ensemble._local_lse(net, xg, hi_u - lo_u, seed, call0 + lo_u, 0, timers=t, precision=prec, share=(E, g_off))
            elif G > 1:                                  # the launches of the timed region: G batches x E draws per launch
                xg = x.repeat(G, 1, 1, 1)

                def one_pass(t):
                    seed, call0 = rng.next_calls(G * E)
                    ensemble._local_lse(net, xg, E, seed, call0, E, timers=t, precision=prec, groups=G)
            elif multi:                                  # rank 0's share of the sharded step
                def one_pass(t):
                    seed, call0 = rng.next_calls(0)      # (not advanced: see above)
                    if S > 1:
                        ensemble._local_lse(net, x, E, seed, call0, 0, timers=t, precision=prec, units=(S, lo_u, hi_u))
                    else:
                        ensemble._local_lse(net, x, hi_u - lo_u, seed, call0 + lo_u, 0, timers=t, precision=prec)
            else:
                def one_pass(t):
                    ensemble.mc_forward(net, x, E, timers=t, precision=prec)
            for _ in range(timer_steps):
                one_pass(timers)
            torch.cuda.synchronize(dev)
            per = G                                      # the recorded launches cover (this rank's share of) G steps
            eager = gemm_roofline(timers.summary(), timer_steps * per, prec, cfg is CONFIGS["metric"] and not multi)
            rpk = timers.summary().get("reparam_kl")
            if rpk and rpk["n"]:
                # the parameter pass where the step runs it: first launch of a pass, the GEMMs behind it (HIP events, eager single stream)
                out["reparam_in_step"] = {"avg_us": round(1e3 * rpk["ms"] / rpk["n"], 2), "launches": rpk["n"],
                                          "draws": (G * E) if not multi else (hi_u - lo_u)}
            # the same launches in the timed region's launch mode: every GEMM launch of the step replayed 20x back to back
            # inside its own hipGraph (no host, no event packets between kernels), pre-heated, HIP events around 3 replays
            rec = LaunchRecorder()
            one_pass(rec)
            torch.cuda.synchronize(dev)
            ing = rec.time_in_graphs(dev)
            roof = gemm_roofline(ing, per, prec, cfg is CONFIGS["metric"] and not multi) if ing else None
            if roof is not None and eager is not None:
                roof["timed_by"] = ("per launch: hipGraph of %d back-to-back replays, pre-heated, HIP events around 3 graph replays, median of 3; "
                                    "summed over the %d conv/linear launches" % (rec.reps, roof["launches"]))
                roof["eager_event_brackets"] = {"achieved": eager["achieved"], "frac": eager["frac"], "avg_us": eager["avg_us"],
                                                "launches": eager["launches"], "timed_by": eager["timed_by"]}
                roof["per_launch_us"] = rec.per_launch_us
                roof["slabs_per_launch"] = (G * E) if not multi else (hi_u - lo_u)
            out["roofline"] = roof if roof is not None else eager
    return out, net, x


def dropin_loop(dev, steps):
    """The loop of main_bayesian.py:73-80 as written upstream -- a Python loop of net(x) through the drop-in `layers`, torch
    log_softmax / logmeanexp, eager launches -- with and without autograd."""
    import torch.nn.functional as F
    cfg = CONFIGS["metric"]
    net, x = build_net(cfg, dev)
    B, C, E = cfg["B"], cfg["classes"], cfg["E"]

    def step():
        outputs = torch.zeros(B, C, E, device=dev)
        kl = 0.0
        for j in range(E):
            net_out, _kl = net(x)
            kl = kl + _kl
            outputs[:, :, j] = F.log_softmax(net_out, dim=1)
        m = outputs.max(dim=2, keepdim=True).values
        return (m + torch.log(torch.mean(torch.exp(outputs - m), dim=2, keepdim=True))).squeeze(2), kl

    def timed(ctx):
        with ctx:
            for _ in range(3):
                step()
            preheat(step, 0.25, dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / steps

    dt_ng = timed(torch.no_grad())
    dt_ag = timed(torch.enable_grad())
    return {"value": round(B * E / dt_ng, 1), "unit": "samples/s", "ms_per_step": round(1e3 * dt_ng, 4),
            "autograd_enabled": {"value": round(B * E / dt_ag, 1), "ms_per_step": round(1e3 * dt_ag, 4),
                                 "note": "what the UNMODIFIED validate_model gets: it does not disable autograd"},
            "note": "for j in range(10): net(x) through the drop-in layers + torch log_softmax / logmeanexp; a call that repeats the "
                    "previous call's input is answered from one speculatively batched K-draw launch (layers/_fused.py), 0.25 s pre-heat"}


def slow_paths(dev, steps):
    """The corners of the Python layer on the scoreboard (review r04 items 6 / 9): an odd batch (B = 510: padded onto the
    batch-innermost kernels since round 5; `layout="nchw"` = the reference-layout kernels it used to drop to), and a model with a
    forward hook (the fused path does not call the children, so the drop-in loop takes the per-layer path: one launch per module)."""
    from bbb_hip import ensemble, rng
    cfg = dict(CONFIGS["metric"], B=510)
    net, x = build_net(cfg, dev)
    E, out = cfg["E"], {}

    def rate(fn, n, B):
        with torch.no_grad():
            for _ in range(3):
                fn()
            preheat(fn, 0.15, dev)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / n
        return {"value": round(B * E / dt, 1), "ms_per_step": round(1e3 * dt, 4)}

    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, E, depth=3)
    r = rate(pipe.step, steps, 510)
    pipe.sync()
    out["odd_batch_510"] = dict(r, launch="hipGraph replay, 3 lanes; 510 images padded with 2 zero images onto the batch-innermost kernels")
    del pipe

    def eager_nchw():
        seed, call0 = rng.next_calls(E)
        lg, _ = ensemble.mc_logits(net, x, E, seed, call0, layout="nchw")
        return lg

    out["odd_batch_510_reference_layout"] = dict(rate(eager_nchw, max(5, steps // 4), 510), launch="eager, reference-layout (NCHW) kernels: where B % 4 != 0 ran until round 4")
    # a forward hook on one layer: the unmodified loop through net(x)
    import torch.nn.functional as F
    cfgm = CONFIGS["metric"]
    netm, xm = build_net(cfgm, dev)
    seen = []
    h = netm.conv3.register_forward_hook(lambda m, i, o: seen.append(1))

    def loop():
        outputs = torch.zeros(cfgm["B"], cfgm["classes"], E, device=dev)
        for j in range(E):
            o, _ = netm(xm)
            outputs[:, :, j] = F.log_softmax(o, dim=1)
        return outputs

    out["forward_hook_dropin_loop"] = dict(rate(loop, max(5, steps // 4), cfgm["B"]),
                                           launch="for j in range(10): net(x), a forward hook on conv3 -> per-layer path (each conv on the batch-innermost kernel "
                                                  "between layout transposes since round 5, torch activations / pooling), eager")
    h.remove()
    assert seen
    return out


SPLIT_STEPS_PER_LAUNCH = 4


def hot_us(fn, dev, reps=10, heat_s=0.03):
    """us per call of fn: `reps` calls back to back inside one hipGraph, pre-heated, HIP events around 3 replays, median of 3."""
    from bbb_hip import ops
    for _ in range(2):
        fn()
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(reps):
            fn()
    preheat(g.replay, heat_s, dev)
    ts = []
    for _ in range(3):
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(3):
            g.replay()
        e0.record()
        torch.cuda.synchronize(dev)
        ts.append(s0.elapsed_time(e0) / (3 * reps))
    return round(statistics.median(ts) * 1e3, 1)


def fusion_ab_fp32(dev, slabs=40, B=512):
    """N3 on the fp32-MFMA chain, every pooling layer of BayesianAlexNet (BBB at 40 slabs = the metric launch shape; LRT at 16 slabs =
    configs[2]'s): us per launch of the conv launch + the pooling launch against the pooled launch (serial window: one workgroup
    walks the window's four pixels, running maximum in accumulation registers; bit for bit the two launches), and what ships.
    conv5 + pool3 has a split contraction (a second set of running registers): no pooled form."""
    from bbb_hip import ops
    out = {}
    shapes = {"pool1": ((3, 32, 32), (64, 3, 11, 11), 4, 5), "pool2": ((64, 4, 4), (192, 64, 5, 5), 1, 2)}
    with torch.no_grad():
        torch.manual_seed(0)
        for name, ((C, H, W), wshape, st, pd) in shapes.items():
            x = torch.rand(1 if name == "pool1" else slabs, C, H, W, B, device=dev)
            w = torch.randn(slabs, *wshape, device=dev) * 0.05
            b = torch.randn(slabs, wshape[0], device=dev) * 0.1
            conv = hot_us(lambda: ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus", bf16x3=False), dev)
            y = ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus", bf16x3=False)
            pool = hot_us(lambda: ops.maxpool_chwn(y, 2, 2), dev)
            fused = hot_us(lambda: ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus", bf16x3=False, pool=True), dev)
            ships = ops.pool_fusion_ok(tuple(x.shape), tuple(w.shape), st, pd, 1, slabs)
            out["bbb_" + name] = {"conv_us": conv, "pool_us": pool, "fused_us": fused, "saved_us": round(conv + pool - fused, 1),
                                  "shipped": "fused" if ships else "separate launches"}
            del x, w, b, y
        ls = 16
        for name, ((C, H, W), wshape, st, pd) in shapes.items():
            x = torch.rand(ls, C, H, W, B, device=dev)
            w_mu = torch.randn(*wshape, device=dev) * 0.05
            w_var = torch.rand(*wshape, device=dev) * 1e-4
            b_mu, b_var = torch.randn(wshape[0], device=dev) * 0.1, torch.rand(wshape[0], device=dev) * 1e-4
            f = lambda pool: ops.lrt_conv2d_chwn_forward(x, w_mu, w_var, b_mu, b_var, 1, 0, 2, st, pd, 1, act="softplus", pool=pool)[0]
            conv = hot_us(lambda: f(False), dev)
            y = f(False)
            pool = hot_us(lambda: ops.maxpool_chwn(y, 2, 2), dev)
            fused = hot_us(lambda: f(True), dev)
            out["lrt_" + name] = {"conv_us": conv, "pool_us": pool, "fused_us": fused, "saved_us": round(conv + pool - fused, 1),
                                  "shipped": "separate launches", "slabs": ls}
            del x, y
    out["bbb_pool3"] = {"shipped": "separate launch", "why": "conv5's contraction is split (ops.split_k): no pooled form"}
    out["unit"] = "us per launch (BBB: %d slabs, LRT: 16 slabs) x %d images" % (slabs, B)
    return out


def fusion_ab(dev, slabs=40, B=512):
    """SURVEY section 8(f) N3 on the split-bf16 chain (review r05 item 7): per pooling layer of BayesianAlexNet either the fused launch
    that ships (A/B measured here, us per 40-slab launch) or the separate pooling launch's cost next to the extra matrix work its
    fusion would need.  pool1: conv1 in space-to-depth form has no padding, every window walks the same taps -> the parallel-window
    form (four waves = four pixels) fuses it at no extra matrix work.  pool2 / pool3 follow PADDED layers: the four pixels of a
    window walk different in-bounds tap sets, a shared weight tile means the union (conv2: 64 taps for 49 = +31 %; conv5 on a 2 x 2
    map: 36 for 16 = +125 %), and the serial-window form quarters the workgroup count at 4x their length (measured slower on the
    fp32 kernel, profiles/r04_notes.md section 7, r05_notes.md section 4)."""
    from bbb_hip import ops
    out = {}
    with torch.no_grad():
        torch.manual_seed(0)
        x = torch.rand(4 * B, 3, 32, 32, device=dev)
        xs = ops.s2d_c8s3(x, 4, 11, 4, 5)                                     # four steps' batches
        w = ops.w_s2d_tap_major(torch.randn(slabs, 64, 3, 11, 11, device=dev) * 0.05, 4)
        b = torch.randn(slabs, 64, device=dev) * 0.1
        zb = ops.s2d_zero_border(3, 11, 4, 5, 32, 32)
        kw = dict(act="softplus", zero_border=zb, x_div=slabs // 4)
        fused = hot_us(lambda: ops.conv2d_c8x3_forward(xs, w, b, 3, 1, 0, 1, pool=True, **kw), dev)
        conv = hot_us(lambda: ops.conv2d_c8x3_forward(xs, w, b, 3, 1, 0, 1, **kw), dev)
        y = ops.conv2d_c8x3_forward(xs, w, b, 3, 1, 0, 1, **kw)
        pool = hot_us(lambda: ops.maxpool_c8s3(y, 2, 2), dev)
        assert torch.equal(ops.maxpool_c8s3(y, 2, 2), ops.conv2d_c8x3_forward(xs, w, b, 3, 1, 0, 1, pool=True, **kw))
        out["pool1"] = {"fused_us": fused, "conv_us": conv, "pool_us": pool, "shipped": "fused (bit for bit the two launches)",
                        "saved_us": round(conv + pool - fused, 1)}
        del y
        for name, (C, H), extra in (("pool2", (192, 4), "+31 % matrix work (union of the window's tap sets: 64 for 49)"),
                                    ("pool3", (128, 2), "+125 % matrix work (36 taps for 16)")):
            t = ops.c8s3_from_f32(torch.rand(slabs, C, H, H, B, device=dev))
            out[name] = {"pool_us": hot_us(lambda t=t: ops.maxpool_c8s3(t, 2, 2), dev), "shipped": "separate launch",
                         "fused_form_would_need": extra}
            del t
    out["unit"] = "us per launch of %d slabs x %d images" % (slabs, B)
    return out


def split_bf16(dev, steps, pipeline):
    """The metric step with the GEMM launches on the 16-bit matrix pipe at fp32 accuracy, range-free (ops.gemm_mode = "bf16x3",
    bbb_conv2d_chwn_bf16x3_fwd: every operand element split into three bf16 pieces while staged, six products per fp32 product,
    fp32 accumulation).  Opt-in mode, reported NEXT TO the fp32 headline, never as it: throughput, single-lane latency, per-launch
    times, the largest difference of the step's log-probabilities from the fp32 path under the same noise, and the training step."""
    from bbb_hip import ensemble, ops, rng
    cfg = CONFIGS["metric"]
    net, x = build_net(cfg, dev)
    E = cfg["E"]
    out = {}
    try:
        with torch.no_grad():
            seed_call = rng.next_calls(0)
            ref_lo = ensemble.mc_forward(net, x, E)[0].clone()
            ops.gemm_mode = "bf16x3"
            rng.rewind(seed_call)
            lo = ensemble.mc_forward(net, x, E)[0]
            out["max_abs_diff_of_log_probs_vs_fp32_path"] = float((lo - ref_lo).abs().max())
            out["max_abs_log_prob"] = float(ref_lo.abs().max())
            # the headline's launch geometry (G steps per launch x 2 lanes) first, then one step per launch x `pipeline` lanes and x 1
            G = SPLIT_STEPS_PER_LAUNCH
            for name, depth, spl in (("steps_per_launch_%d_x_2_lanes" % G, 2, G), ("steps_in_flight_%d" % pipeline, pipeline, 1),
                                     ("one_step_in_flight", 1, 1)):
                pipe = ensemble.GraphedPipeline(net, x, E, depth=depth, steps_per_launch=spl)
                n = -(-steps // spl) * spl
                for _ in range(depth * spl):
                    pipe.step()
                pipe.sync()
                t_heat = time.perf_counter()
                while time.perf_counter() - t_heat < 0.2:
                    for _ in range(depth * spl * 2):
                        pipe.step()
                    pipe.sync()
                vals = []
                for _ in range(5 if spl > 1 else 1):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(n):
                        pipe.step()
                    pipe.sync()
                    torch.cuda.synchronize(dev)
                    vals.append((time.perf_counter() - t0) / n)
                dt = statistics.median(vals)
                out[name] = {"ms_per_step": round(1e3 * dt, 4), "value": round(cfg["B"] * E / dt, 1)}
                if len(vals) > 1:
                    out[name]["blocks"] = len(vals)
                    out[name]["best_ms_per_step"] = round(1e3 * min(vals), 4)
                del pipe
            for tag, g_ in (("gemm_launches", G), ("gemm_launches_one_step", 1)):
                rec = LaunchRecorder()
                if g_ > 1:
                    seed, call0 = rng.next_calls(g_ * E)
                    ensemble._local_lse(net, x.repeat(g_, 1, 1, 1), E, seed, call0, E, timers=rec, groups=g_)
                else:
                    ensemble.mc_forward(net, x, E, timers=rec)
                torch.cuda.synchronize(dev)
                ing = rec.time_in_graphs(dev)
                g = ing.get("conv_gemm")
                if g:
                    tf = g["work"] / (g["ms"] * 1e-3) / 1e12
                    out[tag] = {"per_launch_us": rec.per_launch_us, "slabs_per_launch": g_ * E, "fp32_equivalent_TFLOPs": round(tf, 1),
                                "bf16_mfma_TFLOPs": round(6 * tf, 1), "frac_of_bf16_peak": round(6 * tf / PEAK_BF16_MFMA_TFLOPS, 4),
                                "frac_of_fp32_matrix_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                                "what": "executed (in-bounds) fp32-equivalent FLOPs of the layers x 6 bf16 products, over the summed hot per-launch "
                                        "times; conv1 runs in space-to-depth form (432 products per output against the layer's <= 363: the extra "
                                        "ones are not counted)"}
        # the mode also covers the role-swapped gradient GEMMs of the training step (range-free: 1e-6-sized operands are fine)
        try:
            from bbb_hip import train
            y = torch.randint(0, cfg["classes"], (cfg["B"],), device=dev)
            res = {}
            for mode in ("fp32", "bf16x3"):
                ops.gemm_mode = mode
                net_t, _ = build_net(cfg, dev)
                opt = train.FusedAdam(net_t.parameters(), lr=1e-3)
                for _ in range(4):
                    train.train_step(net_t, opt, x, y, E, 0.1, 50000.0, graph=False)
                torch.cuda.synchronize(dev)
                n = max(5, steps // 8)
                t0 = time.perf_counter()
                for _ in range(n):
                    train.train_step(net_t, opt, x, y, E, 0.1, 50000.0, graph=False)
                torch.cuda.synchronize(dev)
                res[mode] = round(1e3 * (time.perf_counter() - t0) / n, 4)
                del net_t, opt
            out["training_step_ms"] = {"bf16x3": res["bf16x3"], "fp32": res["fp32"],
                                       "what": "forward + backward + Adam, bs=512 num_ens=10, launch by launch; forward, wgrad and dgrad GEMMs in the mode"}
        except Exception as exc:
            out["training_step_ms"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
        try:
            out["fusion_ab"] = fusion_ab(dev)
        except Exception as exc:
            out["fusion_ab"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
        # the other fp32 configurations of BASELINE.json in the mode (same launch shapes as their fp32 rows in SECONDARY.configs;
        # configs[2]: LRT layers on the kernel's LRT form)
        oc = {}
        for name in ("configs[2]", "configs[3]", "configs[4]"):
            try:
                small = CONFIGS[name]["E"] == 1 and CONFIGS[name]["hw"] == 32        # (the launch shapes of SECONDARY.configs)
                r, n2, x2 = run_config(CONFIGS[name], 256 if small else 12, 16 if small else 3, 4 if small else 3, dev, want_roofline=False,
                                       preheat_s=0.15, single_lane=False, steps_per_launch=16 if small else 1)
                oc[name] = {"value": r["value"], "ms_per_step": r["ms_per_step"]}
                del n2, x2
            except Exception as exc:
                oc[name] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
            torch.cuda.empty_cache()
        out["other_configs"] = oc
        out["unit"] = "samples/s"
        out["note"] = ("opt-in precision mode, range-free: every GEMM operand element is the sum of hi = bf16(a), mid = bf16(a - hi), "
                       "lo = bf16(a - hi - mid) (exact), six products on v_mfma_f32_32x32x16_bf16, fp32 accumulation.  Round 6: both "
                       "operands MFMA-ready in memory (csrc/pconv_c8x3.hip) -- activations travel channel-interleaved and already split "
                       "(c8 S3, cut once by the producing epilogue), weights come tap-major from the parameter pass, conv1 runs in "
                       "space-to-depth form with its pooling inside the launch.  Against the float64 oracle 2.5-3.2e-7 of sum|w||x| on "
                       "every operand scale (the fp32 kernel: 2.8-3.7e-7); held to the SAME bounds as the fp32 path at every full-size "
                       "configuration (tests/test_gpu_parity_fullsize.py[bf16x3], tests/test_gpu_c8x3.py)")
    finally:
        ops.gemm_mode = "fp32"
    return out


def ops_pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def training_step(dev, steps):
    """N1 (training extension), not the metric: one iteration of train_model's batch loop (main_bayesian.py:40-58) at the metric
    shape -- 10 stochastic forwards of 512 images, KL, logmeanexp, ELBO, backward, Adam -- on the batch-innermost kernels with
    ONE autograd node for the batched forward (bbb_hip/fast_train.py), eager launches."""
    from bbb_hip import train, ensemble
    cfg = CONFIGS["metric"]
    net, x = build_net(cfg, dev)
    y = torch.randint(0, cfg["classes"], (cfg["B"],), device=dev)
    opt = train.FusedAdam(net.parameters(), lr=1e-3)
    res = {}
    for mode, graph in (("launch_by_launch", False), ("default", None)):      # default: train_step captures itself after 3 calls
        for _ in range(5):
            train.train_step(net, opt, x, y, cfg["E"], 0.1, 50000.0, graph=graph)
        if mode == "launch_by_launch":
            path = ensemble.stats["path"]
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            train.train_step(net, opt, x, y, cfg["E"], 0.1, 50000.0, graph=graph)
        torch.cuda.synchronize(dev)
        res[mode] = (time.perf_counter() - t0) / steps
    dt = res["default"]
    # executed matrix work of the step: forward (in-bounds taps) + weight gradients (the same products, every layer) + input gradients
    # (the same products, every layer but the first) -- over the whole step's time: how much of the fp32 matrix peak a training
    # step sustains (layout shuffles, pooling / activation backward, parameter passes and Adam included in the time)
    fwd = first = 0.0
    shape = (cfg["hw"], cfg["hw"])
    import torch.nn as nn
    from layers.misc import FlattenLayer
    for m in ensemble.flat_children(net):
        if hasattr(m, "W_mu"):
            if m.W_mu.dim() == 4:
                f = ensemble.conv_flops(cfg["B"], m.in_channels, shape[0], shape[1], m.out_channels, *m.kernel_size, m.stride, m.padding,
                                        m.dilation, cfg["E"])[0]
                sh, ph, dh = ops_pair(m.stride), ops_pair(m.padding), ops_pair(m.dilation)
                shape = tuple((shape[a] + 2 * ph[a] - dh[a] * (m.kernel_size[a] - 1) - 1) // sh[a] + 1 for a in (0, 1))
            else:
                f = 2.0 * cfg["E"] * cfg["B"] * m.in_features * m.out_features
            first = first or f
            fwd += f
        elif isinstance(m, nn.MaxPool2d):
            k, st = m.kernel_size, (m.stride if m.stride is not None else m.kernel_size)
            shape = tuple((v - k) // st + 1 for v in shape)
    flop = 3.0 * fwd - first
    out = {"ms_per_step": round(1e3 * dt, 4), "value": round(cfg["B"] * cfg["E"] / dt, 1), "unit": "samples/s (forward + backward + Adam)",
           "roofline": {"bound": "mfma", "achieved": round(flop / dt / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(flop / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), "flop_per_step": flop,
                        "what": "executed FLOPs of forward + wgrad + dgrad (in-bounds taps; 3 x forward - the first layer's dgrad) over the "
                                "whole step's wall time"},
           "launch_by_launch_ms_per_step": round(1e3 * res["launch_by_launch"], 4), "path": path,
           "note": "BayesianAlexNet bs=512 num_ens=10, fp32, train.train_step as called: steps of >= 2048 (image x draw) rows stay launch by "
                   "launch (GPU-bound; weight gradients on two side streams beside the input gradients, HIP loss tail incl. the ELBO: round 6 3.02 -> 2.47 ms), smaller "
                   "ones capture themselves as one hipGraph after 3 identical calls; round 1 (reference-layout autograd path): 8.3 ms"}
    del net, x, y, opt
    # the reference's own defaults (config_bayesian.py:1-18: layer_type 'lrt', batch_size 256, train_ens 1): eager and as one hipGraph
    try:
        from bbb_hip import rng, zoo
        torch.manual_seed(0)
        net = zoo.getModel("alexnet", 3, 10, PRIORS, "lrt", "softplus").to(dev)
        rng.assign_stream_ids(net)
        x = torch.rand(256, 3, 32, 32, device=dev)
        y = torch.randint(0, 10, (256,), device=dev)
        d = {}
        for mode in ("eager", "default", "hipgraph"):
            opt = train.FusedAdam(net.parameters(), lr=1e-3, capturable=(mode == "hipgraph"))
            if mode == "hipgraph":
                g = train.GraphedTrainStep(net, opt, x, y, 1, 0.1, 50000.0, warmup=3)
                step = g.step
            elif mode == "default":                           # train_step as a user calls it: self-capturing
                step = lambda: train.train_step(net, opt, x, y, 1, 0.1, 50000.0)
            else:
                step = lambda: train.train_step(net, opt, x, y, 1, 0.1, 50000.0, graph=False)
            for _ in range(5):
                step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(2 * steps):
                step()
            torch.cuda.synchronize(dev)
            d[mode + "_ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / (2 * steps), 4)
        d["path"] = ensemble.stats["path"]
        d["note"] = "BayesianAlexNet, layer_type lrt, bs=256, num_ens=1 = the reference's config_bayesian.py defaults; forward + backward + Adam"
        out["reference_default_config"] = d
    except Exception as exc:
        out["reference_default_config"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
    return out


def compact(obj, limit=120):
    """Strings cut to `limit` characters, floats to 6 significant digits (the judged line must stay short)."""
    if isinstance(obj, dict):
        return {k: compact(v, limit) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [compact(v, limit) for v in obj]
    if isinstance(obj, str):
        return obj[:limit]
    if isinstance(obj, float):
        return float("%.6g" % obj)
    return obj


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline measurement only (profiling runs)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches on one stream (profiling runs: rocprofv3 --pmc per launch)")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the per-launch roofline passes (profiling runs: the trace then ends with the timed region's graph replays)")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="independent graphs in flight (hipGraph lanes on separate streams); default by the draws one rank's launch "
                         "holds: 2 from 30 draws (N = 1, 4 steps per launch), 3 from 10, else 4 (8 ranks: 5 draws per launch)")
    ap.add_argument("--steps-per-launch", type=int, default=None,
                    help="consecutive Monte-Carlo steps per graph launch; default 4: N = 1 measured 0.635 ms per step with 2 lanes "
                         "against 0.650 for one step per launch x 3 lanes (profiles/r04_steps_per_launch_sweep.txt)")
    ap.add_argument("--preheat-ms", type=float, default=400.0, help="untimed replays of the timed graphs before the warm-up steps")
    ap.add_argument("--config", default="metric", choices=list(CONFIGS), help="which BASELINE configuration is the reported value")
    ap.add_argument("--gemm-mode", default="fp32", choices=["fp32", "bf16x3"],
                    help="bf16x3: the opt-in range-free split-bf16 GEMM mode for the WHOLE run (profiling it, or timing a sharded run with "
                         "it); the default run reports it as the secondary object `split_bf16` next to the fp32 headline")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    # test hook (never set by the driver): run the N > 1 FLOW -- RCCL process group, sharded lanes with the recorded collective, barriers,
    # max-over-ranks timing -- with ONE rank on one GPU, so that the code a multi-GPU box would execute runs in every suite run
    self_group = os.environ.get("BBB_BENCH_SELF_GROUP") == "1" and "WORLD_SIZE" not in os.environ and args.gpus == 1
    multi = world > 1 or self_group
    cfg = CONFIGS[args.config]
    if args.steps_per_launch is None:
        args.steps_per_launch = 4 if (cfg["hw"] == 32 and cfg["E"] <= 10) else 1
    if args.pipeline is None:
        # lanes by the draws one launch of one rank holds (profiles/r03_notes.md section 7, r04 section 2): large launches fill the
        # chip themselves, small ones need more of them in flight
        local = -(-args.steps_per_launch * cfg["E"] // max(1, args.gpus))
        args.pipeline = 2 if local >= 30 else (3 if local >= 10 else 4)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: become the launcher -- one process per GPU under torch.distributed.run
        # (rendezvous on 127.0.0.1, a free port), same arguments.  Under the driver's own torch.distributed.run
        # invocation WORLD_SIZE is already set and this branch is not taken.
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        if os.environ.get("BBB_BENCH_PRINT_LAUNCH") == "1":          # test hook (CPU suite): show the launch, do not exec
            print(json.dumps({"launch": cmd}))
            return
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one process per GPU (python bench.py --gpus N does it itself)"
                         % (args.gpus, world))
    # test hooks (single-GPU rehearsal of the N > 1 code path): BBB_BENCH_DEVICE pins every rank to one device,
    # BBB_BENCH_BACKEND=gloo replaces RCCL.  Never set by the driver.
    dev_index = int(os.environ.get("BBB_BENCH_DEVICE", local_rank))
    backend = os.environ.get("BBB_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # the reference's CPU path first, on rank 0, BEFORE the process group exists: the other ranks then
    # wait in the rendezvous (sleeping on a socket), not in a collective (spinning on the cores the baseline is being timed on)
    cpu = None
    if rank == 0 and not args.no_cpu_baseline and not args.no_extras and not args.no_graph:
        cpu = cpu_baseline(20.0 if world == 1 else 10.0)
    group = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if self_group:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ["BBB_FORCE_COMBINE"] = "1"        # ensemble: the sharded code path although one rank holds every unit
            dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": dev} if backend == "nccl" else {}))
        elif backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    from bbb_hip import ensemble, _lib, ops
    _lib.lib()   # fail loudly here if the HIP library is missing
    ops.gemm_mode = args.gemm_mode

    if args.no_graph:
        # profiling mode (rocprofv3 --pmc per launch): the timed region's launches issued eagerly on one stream, nothing else.
        # With G steps per launch a pass is G steps; --warmup / --steps are rounded up to whole passes.
        from bbb_hip import rng
        net, x = build_net(cfg, dev)
        G = args.steps_per_launch if not multi else 1     # (the eager profiling pass of a sharded run: one step, work units)
        xg = x.repeat(G, 1, 1, 1) if G > 1 else x

        def one_pass():
            if G > 1:
                seed, call0 = rng.next_calls(G * cfg["E"])
                ensemble._local_lse(net, xg, cfg["E"], seed, call0, cfg["E"], precision=cfg["precision"], groups=G)
            else:
                ensemble.mc_forward(net, x, cfg["E"], group=group, precision=cfg["precision"])
        nw, nt = -(-args.warmup // G), -(-args.steps // G)
        with torch.no_grad():
            for _ in range(nw):
                one_pass()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(nt):
                one_pass()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / (nt * G)
        if rank == 0:
            print(json.dumps({"metric": "eager single-stream step (profiling mode)", "value": round(cfg["B"] * cfg["E"] / dt, 1),
                              "unit": "samples/s", "ms_per_step": round(1e3 * dt, 4), "n_gpus": world, "steps": nt * G,
                              "warmup": nw * G, "steps_per_launch": G, "mc_steps_total": (nw + nt) * G}), flush=True)
        if multi:
            torch.distributed.destroy_process_group()
        return

    extras = rank == 0 and not multi and not args.no_extras
    G = args.steps_per_launch
    head, net, x = run_config(cfg, args.steps, args.warmup, args.pipeline, dev, group, world, want_roofline=not args.no_roofline,
                              stat_blocks=5 if extras else 0, timer_steps=min(args.steps, 10), steps_per_launch=G,
                              preheat_s=args.preheat_ms * 1e-3, cold_block=True, rank=rank, single_lane=extras, multi=multi)
    n_params = sum(p.numel() for n, p in net.named_parameters() if n.endswith("_mu"))

    weak = None
    if multi and not args.no_extras:
        w, _, _ = run_config(cfg, max(5, args.steps // 2), 3, args.pipeline, dev, group, world, want_roofline=False,
                             total_ens=cfg["E"] * max(world, 2 if self_group else 1), preheat_s=0.1, rank=rank, multi=multi)
        weak = {"value": w["value"], "unit": "samples/s", "ms_per_step": w["ms_per_step"], "num_ens_total": cfg["E"] * world}  # 10 draws per GPU

    final_lines = []
    if rank == 0:
        with torch.no_grad():
            S, lo, hi = ensemble.shard_plan(net, x, cfg["E"], 0, world, True, cfg["precision"])
        launch = "hipGraph replay, %d lane(s)" % max(1, args.pipeline) + (" x %d steps per launch" % G if G > 1 else "")
        out = {
            "metric": "MC-forward samples/sec, BayesianAlexNet CIFAR-10 bs=512 num_ens=10",
            "value": head["value"], "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if cfg["precision"] == "bf16" else
                     ("f32" if args.gemm_mode == "fp32" else "f32 tensors; GEMM products as 6 x bf16 (hi/mid/lo split), f32 accumulate"),
            "data": "synthetic",
            "config": {"workload": cfg["what"] + ", forward only",
                       "global_batch": cfg["B"], "num_ens_total": cfg["E"],
                       "parallelism": (("groups of %d steps: the %d draws of a group in %d contiguous ranges (whole draws on whole batches, "
                                        "<= %d per GPU), one all_gather per group"
                                        % (G, G * cfg["E"], world, -(-G * cfg["E"] // world))) if G > 1 else
                                       ("work units: %d slices/draw, %d units of %d images, <= %d per GPU, one all_gather per step"
                                        % (S, cfg["E"] * S, cfg["B"] // S, hi - lo)))
                       if multi else "single",
                       "launch": launch},
            "preheat_ms": head.get("preheat_ms"),
        }
        if "cold_first_block" in head:
            out["cold_first_block"] = head["cold_first_block"]
        if multi:
            out["config"]["ranks_seen"] = torch.distributed.get_world_size(group)
            if G > 1:
                out["config"]["draws_per_rank"] = [(lambda r: r[1] - r[0])(ensemble.group_share(cfg["E"], G, r, world)) for r in range(world)]
            else:
                out["config"]["units_per_rank"] = [(lambda r: r[1] - r[0])(ensemble.unit_range(cfg["E"], S, r, world)) for r in range(world)]
            out["config"]["backend"] = backend
            try:
                out["config"]["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
            except Exception:
                out["config"]["rccl_version"] = None
            proto = head.get("collective_protocol") or {}
            out["config"]["backend_mode"] = ("all_gather recorded into each lane's hipGraph (private communicator per lane)"
                                             if proto.get("collective") == "recorded" else
                                             "eager all_gather between graph replays (%s)" % (proto.get("reason") or "n/a"))
            if head.get("first_contact"):
                out["config"]["first_contact"] = head["first_contact"]
        if args.config != "metric":
            out["metric"] = "MC-forward samples/sec, " + cfg["what"]
        second = {}                                          # the bulky objects: printed on the SECONDARY line
        roof = head.get("roofline")
        if roof:
            second["roofline_detail"] = dict(roof)
            # the judged object: scalars only (the driver's record keeps first-level scalars of `roofline` and `cpu_baseline`)
            r = {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_us") if k in roof}
            r["kernel"] = "pconv_gemm fp32 MFMA x6 launches" if cfg["precision"] != "bf16" else "pconv_bf16 MFMA"
            r["per_launch_us"] = "/".join("%.1f" % v for v in roof.get("per_launch_us", []))
            r["timed_by"] = "HIP events; launch x20 in hot hipGraph" if roof.get("per_launch_us") else "HIP event brackets, eager"
            r["slabs_per_launch"] = roof.get("slabs_per_launch")
            fps = roof.get("flop_per_step")
            if fps:
                tf = fps / (head["ms_per_step"] * 1e-3) / 1e12
                r["sustained_TFLOPs"] = round(tf, 2)
                r["sustained_frac"] = round(tf / roof["peak"], 4)
            if roof.get("traffic") and fps and cfg is CONFIGS["metric"]:
                tr = profile_traffic("pconv_gemm")
                r["traffic_ratio"] = round(roof["traffic"] / ALGORITHMIC_GEMM_BYTES, 3) if tr else None
            if cfg["precision"] != "bf16" and not multi:
                ceil = mfma_loop_ceiling()
                if ceil:
                    r["mfma_loop_TFLOPs"] = ceil[0]
                    second["mfma_loop_ceiling"] = {"TFLOPs": ceil[0], "shader_clock_GHz": ceil[1]}
            out["roofline"] = r
        else:
            out["roofline"] = None
        if "stats" in head and out["roofline"] is not None:
            st = head["stats"]
            out["roofline"].update(stats_median=st["median"], stats_p10=st["p10"], stats_p90=st["p90"], stats_blocks=st["blocks"],
                                   value_above_p90=bool(out["value"] > st["p90"]),
                                   single_block_value=(head.get("single_block") or {}).get("value"))
            second["stats"] = st
        if "one_step_in_flight" in head and out["roofline"] is not None:
            out["roofline"]["one_step_in_flight_ms"] = head["one_step_in_flight"]["ms_per_step"]
            second["one_step_in_flight"] = head["one_step_in_flight"]
        if weak is not None:
            out["weak_scaling"] = weak
        if extras:
            if G > 1:
                # the launch mode of rounds 1-3 beside the headline: one step per launch, three lanes
                r1, _, _ = run_config(cfg, max(args.steps, 30), 5, 3, dev, want_roofline=not args.no_roofline, timer_steps=3,
                                      steps_per_launch=1, preheat_s=0.2, single_lane=False)
                second["one_step_per_launch"] = r1
                if out["roofline"] is not None:
                    out["roofline"]["one_step_per_launch_ms"] = r1["ms_per_step"]
                    if r1.get("roofline"):
                        out["roofline"]["one_step_per_launch_frac"] = r1["roofline"]["frac"]
            if cfg["lt"] == "bbb":
                # the fused reparam + KL pass.  `reparam_frac` is the launch shape the TIMED REGION runs (G x E draws per launch);
                # the E-draw launch of a one-step-per-launch graph (Infinity-Cache assisted: draws 2-10 land in MALL) and the
                # genuinely HBM-resident probe (2^26 elements, 10 draws: 3.2 GB of traffic) are named scalars beside it
                rp = reparam_probe(net, dev, n_params, cfg["E"])
                second["roofline_reparam"] = rp
                rpg = reparam_probe(net, dev, n_params, cfg["E"] * G, hbm_probe=False) if G > 1 else rp
                if G > 1:
                    second["roofline_reparam_steps_per_launch"] = rpg
                try:
                    rpr = reparam_region_probe(net, x, cfg["E"], G, dev, n_params)
                except Exception as exc:                          # noqa: BLE001
                    rpr = None
                    second["roofline_reparam_in_region"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
                if rpr:
                    second["roofline_reparam_in_region"] = rpr
                if out["roofline"] is not None:
                    ris = head.get("reparam_in_step")
                    if ris and ris["draws"] == cfg["E"] * G:
                        second["reparam_in_step"] = dict(ris, frac=round((8 + 4 * ris["draws"]) * n_params / (ris["avg_us"] * 1e-6) / (PEAK_HBM_GBS * 1e9), 4))
                    hb = rp.get("hbm_resident_probe", {})
                    # reparam_frac: the launch shape AND the context of the timed region (G x E draws, the group's GEMMs behind it);
                    # the same launch twenty times back to back, the E-draw launch and the HBM-resident probe are named beside it
                    main_rp = rpr or rpg
                    out["roofline"].update(reparam_frac=main_rp["frac"], reparam_avg_us=main_rp["avg_us"], reparam_draws=cfg["E"] * G,
                                           reparam_back_to_back_frac=rpg["frac"], reparam_traffic=rp.get("traffic"),
                                           reparam_10draw_frac=rp["frac"],
                                           reparam_hbm_resident_frac=(round(hb["E10_GBps"] / PEAK_HBM_GBS, 4) if "E10_GBps" in hb else None),
                                           device_copy_GBps=hb.get("device_copy_GBps"))
                    # the write roof of this box for the pass's store pattern (same bytes, no reads, no arithmetic), back to back like
                    # reparam_back_to_back_frac: how much of what the hardware gives the pass's WRITES the fused pass reaches
                    # (A-B-A, best brackets of both: the two probes must see the same clock state -- measured minutes apart the ratio
                    # wandered 0.80-1.03 on one box)
                    ra = reparam_probe(net, dev, n_params, cfg["E"] * G, hbm_probe=False)
                    wr = write_roof_probe(n_params, cfg["E"] * G)
                    rb = reparam_probe(net, dev, n_params, cfg["E"] * G, hbm_probe=False)
                    if wr:
                        roof_gbps = max(wr)
                        bb_gbps = max(p_["bytes_per_launch"] / (p_["min_us"] * 1e-6) / 1e9 for p_ in (ra, rb))
                        second["write_roof_probe"] = {"plain_GBps": wr[0], "non_temporal_GBps": wr[1], "draws": cfg["E"] * G,
                                                      "reparam_best_GBps_before_after": [round(p_["bytes_per_launch"] / (p_["min_us"] * 1e-6) / 1e9, 1) for p_ in (ra, rb)],
                                                      "elements_per_draw": n_params,
                                                      "what": "write-only kernel, the reparam pass's store shape and bytes, 10 launches back to back, best of 5"}
                        out["roofline"].update(write_roof_GBps=roof_gbps,
                                               reparam_frac_of_write_roof=round(bb_gbps / roof_gbps, 4),
                                               reparam_in_step_frac_of_write_roof=round(main_rp["frac"] * PEAK_HBM_GBS / roof_gbps, 4))
            del net, x
            try:
                second["dropin_loop"] = dropin_loop(dev, max(40, args.steps // 2))
                if out["roofline"] is not None:
                    out["roofline"]["dropin_loop_value"] = second["dropin_loop"]["value"]
                    out["roofline"]["dropin_loop_autograd_value"] = second["dropin_loop"]["autograd_enabled"]["value"]
            except Exception as exc:
                second["dropin_loop"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
            if cfg is CONFIGS["metric"]:
                try:
                    if args.gemm_mode == "fp32":
                        second["split_bf16"] = split_bf16(dev, max(20, args.steps // 2), 3)
                        sb = second["split_bf16"]
                        key = "steps_per_launch_%d_x_2_lanes" % SPLIT_STEPS_PER_LAUNCH
                        if out["roofline"] is not None and key in sb:
                            out["roofline"]["split_bf16_value"] = sb[key]["value"]
                            out["roofline"]["split_bf16_ms_per_step"] = sb[key]["ms_per_step"]
                            if "gemm_launches" in sb:
                                out["roofline"]["split_bf16_frac_of_bf16_peak"] = sb["gemm_launches"]["frac_of_bf16_peak"]
                                out["roofline"]["split_bf16_per_launch_us"] = "/".join("%.1f" % v for v in sb["gemm_launches"]["per_launch_us"])
                            out["roofline"]["split_bf16_max_abs_diff_vs_fp32"] = sb.get("max_abs_diff_of_log_probs_vs_fp32_path")
                            trs = profile_traffic("pconv_c8x3", "_split")
                            if trs and "gemm_launches" in sb:
                                sb["gemm_launches"]["traffic"] = round(trs[0] + trs[1], 1)
                                sb["gemm_launches"]["traffic_source"] = trs[2]
                                out["roofline"]["split_bf16_traffic"] = round(trs[0] + trs[1], 1)
                except Exception as exc:
                    second["split_bf16"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
            try:
                second["slow_paths"] = slow_paths(dev, max(20, args.steps // 2))
                if out["roofline"] is not None:
                    out["roofline"]["odd_batch_510_value"] = second["slow_paths"]["odd_batch_510"]["value"]
                    out["roofline"]["hooked_loop_value"] = second["slow_paths"]["forward_hook_dropin_loop"]["value"]
            except Exception as exc:
                second["slow_paths"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
            if cfg is CONFIGS["metric"]:
                try:
                    second["fusion_ab_fp32"] = fusion_ab_fp32(dev)
                except Exception as exc:
                    second["fusion_ab_fp32"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
            try:
                second["training_step"] = training_step(dev, max(12, args.steps // 5))
                ts = second["training_step"]
                if out["roofline"] is not None and "roofline" in ts:
                    out["roofline"]["training_step_ms"] = ts["ms_per_step"]
                    out["roofline"]["training_step_frac"] = ts["roofline"]["frac"]
            except Exception as exc:
                second["training_step"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
            torch.cuda.empty_cache()
            others = {}
            for name, c in CONFIGS.items():
                if name == args.config:
                    continue
                try:
                    # single-draw configurations are a dozen 10-20 us launches per step: four lanes AND sixteen consecutive steps per
                    # launch (profiles/r03_notes.md sections 4, 9); the 25-draw and 224x224 steps fill the chip: three lanes
                    small = c["E"] == 1 and c["hw"] == 32
                    depth = 4 if small else 3
                    spl = 16 if small else 1                  # (round 5 sweep, profiles/r05_notes.md section 3: 4 -> 16 steps per launch)
                    nst = max(10, args.steps // 2)
                    if spl > 1:
                        nst = max(nst, 4 * spl * depth)       # at least four groups per lane: one replay per lane is all ramp
                    nst = -(-nst // (spl * depth)) * spl * depth
                    r, n2, x2 = run_config(c, nst, 5, depth, dev, want_roofline=True, timer_steps=3, steps_per_launch=spl, preheat_s=0.15)
                    r["steps_in_flight"] = depth
                    if spl > 1:
                        r["steps_per_launch"] = spl
                        r1, _, _ = run_config(c, nst, 5, depth, dev, want_roofline=False, steps_per_launch=1, preheat_s=0.1, single_lane=False)
                        r["one_step_per_launch"] = {"ms_per_step": r1["ms_per_step"], "value": r1["value"]}
                    if c["precision"] == "bf16":
                        # what the bf16 storage path's numbers are held to (review r05): the project's own bf16 oracle states the
                        # rounding points (tests: 1e-2 of max|logit|, a self-consistency bound -- the reference has no bf16); the
                        # meaningful figure is the same draw against the fp32 path, measured here
                        try:
                            from bbb_hip import rng as _rng
                            with torch.no_grad():
                                sc = _rng.next_calls(0)
                                a32 = ensemble.mc_forward(n2, x2, c["E"])[0].clone()
                                _rng.rewind(sc)
                                a16 = ensemble.mc_forward(n2, x2, c["E"], precision="bf16")[0]
                                r["parity"] = {"max_abs_diff_of_log_probs_vs_fp32_path_same_noise": float((a16 - a32).abs().max()),
                                               "max_abs_log_prob": float(a32.abs().max()),
                                               "test_bound_vs_the_bf16_oracle": "1e-2 of max|logit| (tests/test_gpu_parity_fullsize.py::test_config1_*)"}
                        except Exception as exc:
                            r["parity"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:120])}
                    del n2, x2
                    if r.get("roofline"):
                        tr = profile_traffic(("pconv_bf16", "pconv_gemm"), "_" + name.replace("[", "").replace("]", ""))
                        if tr:
                            r["roofline"]["traffic"] = round(tr[0] + tr[1], 1)
                            r["roofline"]["traffic_read_written"] = [round(tr[0], 1), round(tr[1], 1)]
                            r["roofline"]["traffic_source"] = tr[2] + "; bytes per step over the conv/linear launches"
                    r["workload"] = c["what"]
                    r["dtype"] = "bf16" if c["precision"] == "bf16" else "f32"
                    r["unit"] = "samples/s"
                    others[name] = r
                except Exception as exc:
                    others[name] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
                torch.cuda.empty_cache()
            second["configs"] = others
        if cpu is not None:
            second["cpu_baseline_detail"] = cpu
            out["cpu_baseline"] = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                   "sample": "%d MC steps 512x10 fp32 no_grad, median; %s" % (
                                       cpu["steps_timed"], "upstream modules" if cpu["kind"] == "reference" else "bit-identical port"),
                                   "p10": cpu["p10"], "p90": cpu["p90"],           # (cpu_model, thread counts: SECONDARY cpu_baseline_detail)
                                   "autograd_value": cpu["autograd_enabled"]["value"]}
            rg = cpu.get("reference_gpu_path") or {}
            if "no_grad" in rg:
                # the upstream modules on THIS GPU (ATen / MIOpen / rocBLAS + CPU-side eps): the reference's own GPU path, same step
                out["cpu_baseline"]["reference_gpu_value"] = rg["no_grad"]["samples_per_s"]
                out["speedup_vs_reference_gpu"] = round(out["value"] / rg["no_grad"]["samples_per_s"], 1)
            out["speedup_vs_cpu"] = round(out["value"] / cpu["value"], 1)
        # the judged line must stay under 2000 characters (the driver's record keeps a 2000-character tail): scalars that only
        # repeat or qualify another one ride on the SECONDARY line instead (second["main_line_extras"])
        extras_moved = {}
        if isinstance(out.get("roofline"), dict):
            for k in ("reparam_traffic", "device_copy_GBps", "one_step_per_launch_frac", "sustained_TFLOPs", "dropin_loop_autograd_value",
                      "split_bf16_per_launch_us", "stats_blocks", "reparam_in_step_frac_of_write_roof", "split_bf16_max_abs_diff_vs_fp32",
                      "one_step_per_launch_ms", "reparam_10draw_frac", "reparam_hbm_resident_frac"):
                if k in out["roofline"]:
                    extras_moved["roofline." + k] = out["roofline"].pop(k)
        if isinstance(out.get("cpu_baseline"), dict):
            for k in ("p10", "p90", "autograd_value"):
                if k in out["cpu_baseline"]:
                    extras_moved["cpu_baseline." + k] = out["cpu_baseline"].pop(k)
        for k in ("speedup_vs_reference_gpu", "speedup_vs_cpu"):
            if k in out:
                extras_moved[k] = out.pop(k)
        if extras_moved:
            second["main_line_extras"] = extras_moved
        final_lines = (["SECONDARY " + json.dumps(second)] if second else []) + [json.dumps(compact(out))]
    def flush_c_stdio():
        # RCCL writes a version banner through C stdio, which sits in libc's buffer until exit when stdout is a pipe -- it would
        # land AFTER a line printed from Python and become the "last line" of the run.  Flush it out first: the JSON stays last.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()

    if multi:
        flush_c_stdio()                                  # every rank, while the ranks can still be ordered
        torch.distributed.barrier(group=group)
        torch.distributed.destroy_process_group()
        flush_c_stdio()
        if rank == 0 and world > 1:
            time.sleep(1.0)                              # the other ranks have nothing left to do but exit: let them (and their stdio) go first
    if rank == 0:
        flush_c_stdio()
        for ln in final_lines:
            print(ln, flush=True)


# algorithmic operand + output bytes of the six GEMM launches of one metric step: 185.0 MB read + 147.0 MB
# written -- conv1 writes its POOLED map (pooling in the launch: 21.0 MB instead of 83.9; 394.9 MB in total with a separate pool1
# launch, which then reads 83.9 MB and writes 21.0 MB more)
ALGORITHMIC_GEMM_BYTES = 332.0e6


if __name__ == "__main__":
    main()
