#!/usr/bin/env python3
"""MC-forward samples/sec of BayesianAlexNet (CIFAR-10 shape, bs=512, num_ens=10, BBB layers) on MI355X.

One step = main_bayesian.py:73-80 of the reference: num_ens x net(x) + log_softmax, then utils.logmeanexp --
forward only, no_grad, inputs resident in HBM.  Synthetic data: torch.manual_seed(0), parameters from the
layers' own reset_parameters with config_bayesian.priors, x ~ U[0,1).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Launch structure: each step is one captured hipGraph (a device-side call counter inside it advances the Philox
noise on every replay) and `--pipeline` (default 3) independent steps are in flight on separate HIP streams.

N > 1 is STRONG scaling of the metric's workload: the same 512 images x 10 draws, cut into (draw x batch-slice) work
units dealt evenly over the ranks (ensemble.shard_plan; 8 ranks: 40 quarter-batch units, 5 each), every rank sampling
only the weight sets its units touch, ONE all_gather of [B*C + 1] floats per step over RCCL.  value = 512 * 10 / step
time.  (`weak_scaling`: a short secondary run with 10 draws PER rank, reported next to it.)

Prints ONE JSON line on rank 0 (contract in the task description) with these extra objects:
  roofline          the dominant kernel (fp32-MFMA implicit-GEMM conv): algorithmic FLOP / HIP-event time vs the 157.3 TFLOP/s
                    fp32 matrix peak; `traffic` = HBM bytes per step from the committed rocprofv3 PMC passes, parsed from
                    profiles/ at run time (null when the files are absent).
  roofline_reparam  the fused reparam+KL pass vs HBM (8 TB/s).
  cpu_baseline      the reference's CPU nn.Module path timed on this host's cores (rank 0, N=1 only): the unmodified upstream
                    modules when /root/reference exists (kind "reference"), else the oracle's bit-identical torch-CPU port
                    (kind "port"); no_grad and autograd-enabled variants, median and p10/p90.
  one_step_in_flight / stats / dropin_loop / training_step / configs   single-lane latency, block-time statistics, the unmodified-loop
                    shape through the drop-in layers, and BASELINE.json's other configurations -- each with its own roofline.
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
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
sys.dont_write_bytecode = True

import torch  # noqa: E402

PRIORS = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense v_mfma_f32_32x32x16_bf16 peak (same guide)
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
    ref_dir = "/root/reference"
    torch.manual_seed(0)
    x = torch.rand(B, 3, 32, 32)
    if os.path.isdir(os.path.join(ref_dir, "layers")):
        # the UNMODIFIED upstream modules (build container only: the GPU box has no checkout)
        kind = "reference"
        saved_path, saved_mods = list(sys.path), {k: v for k, v in sys.modules.items() if k == "layers" or k.startswith("layers.")}
        for k in list(saved_mods):
            del sys.modules[k]
        sys.path.insert(0, ref_dir)
        try:
            from models.BayesianModels.BayesianAlexNet import BBBAlexNet as RefAlexNet
            import utils as ref_utils
            import torch.nn.functional as F
            net = RefAlexNet(C, 3, PRIORS, "bbb", "softplus")

            def step():                                   # main_bayesian.py:73-80
                outputs = torch.zeros(B, C, E)
                kl = 0.0
                for j in range(E):
                    net_out, _kl = net(x)
                    kl += _kl
                    outputs[:, :, j] = F.log_softmax(net_out, dim=1).data
                return ref_utils.logmeanexp(outputs, dim=2), kl
            src = "unmodified upstream modules imported from /root/reference (models.BayesianModels.BayesianAlexNet, utils.logmeanexp)"
        finally:
            sys.path[:] = saved_path
            for k in [k for k in sys.modules if k == "layers" or k.startswith("layers.")]:
                del sys.modules[k]
            sys.modules.update(saved_mods)
    else:
        kind = "port"
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
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
    return {"value": round(n / med, 1), "unit": "samples/s", "cores": nt, "kind": kind,
            "p10": round(n / pctl(ts, 0.9), 1), "p90": round(n / pctl(ts, 0.1), 1), "steps_timed": len(ts),
            "autograd_enabled": {"value": round(n / statistics.median(tg), 1), "p10": round(n / pctl(tg, 0.9), 1),
                                 "p90": round(n / pctl(tg, 0.1), 1), "steps_timed": len(tg)},
            "cpu_model": cpu_model_name(), "torch_num_threads": nt, "usable_cpus": avail, "logical_cpus": os.cpu_count(),
            "sample": f"{len(ts)} full MC steps (bs={B}, num_ens={E}, fp32, no_grad; median) + {len(tg)} with autograd enabled, "
                      f"after a warm-up each; {src}; {avail} CPUs usable (cgroup quota / affinity), best of {{all, half}} thread counts"}


def profile_traffic(kind):
    """HBM bytes per launch-group from the committed PMC summaries (profiles/r02_pmc_FETCH_SIZE.txt / _WRITE_SIZE.txt, written
    by profiles/collect.sh): (read_bytes, written_bytes, source) or None.  FETCH_SIZE x2 = the gfx950 wide-load correction."""
    out = {}
    tag = None
    for t in ("r03", "r02"):                       # the newest committed PMC passes
        if all(os.path.exists(os.path.join(ROOT, "profiles", f"{t}_pmc_{c}.txt")) for c in ("FETCH_SIZE", "WRITE_SIZE")):
            tag = t
            break
    if tag is None:
        return None
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_{c}.txt")
        kb = None
        for line in open(path):
            m = re.match(r"\s*STEP_TOTAL\s+%s\s+(\S+)\s+KB_per_step=([0-9.]+)" % kind, line)
            if m:
                kb = float(m.group(2))
        if kb is None:
            return None
        out[c] = kb * 1024.0
    rd = 2.0 * out["FETCH_SIZE"]
    return rd, out["WRITE_SIZE"], "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 wide-load correction) + WRITE_SIZE, separate passes: " \
        "profiles/%s_pmc_FETCH_SIZE.txt, profiles/%s_pmc_WRITE_SIZE.txt (STEP_TOTAL %s rows)" % (tag, tag, kind)


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
    kern = ("pconv_bf16_kernel (v_mfma_f32_32x32x16_bf16, batch-innermost, LDS transpose reads)" if precision == "bf16" else
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

    def time_in_graphs(self, dev):
        agg = {}
        for tag, info, fn in self.calls:
            if tag not in ("conv_gemm", "lrt_gemm"):
                continue
            for _ in range(2):
                fn()
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            from bbb_hip import ops as _ops
            with _ops.graph_capture(g):
                for _ in range(self.reps):
                    fn()
            g.replay()
            torch.cuda.synchronize(dev)
            ts = []
            for _ in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                g.replay()
                e.record()
                torch.cuda.synchronize(dev)
                ts.append(s.elapsed_time(e) / self.reps)
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


def reparam_probe(net, dev, n_params, E):
    """Fused reparam+KL pass timed on its own: 20 back-to-back launches inside one hipGraph (no host gaps), HIP events
    around the replay, median of 7 replays.  (a) the model's 12 tensors, E draws - what the step runs, Infinity-Cache
    resident; (b) one 2^26-element tensor (0.8-3.2 GB of traffic, far beyond the 256 MB cache) for a genuine HBM figure."""
    from bbb_hip import ensemble, ops
    mus, rhos, ids = [], [], []
    for l in ensemble.bayesian_layers(net):
        m, r, i = l._param_lists()
        mus += m
        rhos += r
        ids += i

    def timed(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        from bbb_hip import ops as _ops
        with _ops.graph_capture(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize(dev)
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
        big = 1 << 26
        mu = torch.randn(big, device=dev) * 0.1
        rho = torch.randn(big, device=dev) * 0.1 - 5
        t1 = timed(lambda: ops.reparam_kl_forward([mu], [rho], 0, 0.1, [0], 1, 0, draws=1), 3)[0]
        t10 = timed(lambda: ops.reparam_kl_forward([mu], [rho], 0, 0.1, [0], 1, 0, draws=10), 2)[0]
        dst = torch.empty_like(mu)
        tc = timed(lambda: dst.copy_(mu), 3)[0]
        del mu, rho, dst
    gbs = byts / t_model / 1e9
    r = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
         "traffic": None,
         "kernel": "reparam_kl_fast_kernel: 12 tensors x %d draws + the KL sum in ONE launch (Philox4x32-7, packed softplus/KL)" % E,
         "bytes_per_launch": byts, "avg_us": round(t_model * 1e6, 2), "min_us": round(t_lo * 1e6, 2), "max_us": round(t_hi * 1e6, 2),
         "note": "(8 + 4E) B per weight element; median of 7 graph replays of 20 back-to-back launches; the 17 MB of (mu,rho) and "
                 "87 MB of w are Infinity-Cache resident at this size, and the pass is VALU-bound (profiles/r02_notes.md)",
         "hbm_resident_probe": {
             "elements": big,
             "E1_GBps": round(12 * big / t1 / 1e9, 1), "E10_GBps": round(48 * big / t10 / 1e9, 1),
             "device_copy_GBps": round(8 * big / tc / 1e9, 1),
             "note": "single 2^26-element tensor, (8+4E) B/element; device_copy = torch copy_ of the same tensor (8 B/element), "
                     "the achievable streaming rate on this box"}}
    tr = profile_traffic("reparam")
    if tr and E == 10:
        r["traffic"] = round(tr[0] + tr[1], 1)
        r["traffic_read_written"] = [round(tr[0], 1), round(tr[1], 1)]
        r["traffic_source"] = tr[2]
    return r


def run_config(cfg, steps, warmup, pipeline, dev, group=None, world=1, want_roofline=True, stat_blocks=0, timer_steps=5,
               total_ens=None, steps_per_launch=1):
    """Throughput of one configuration: hipGraph lanes, K timed steps between device syncs (max over ranks) -> dict.
    steps_per_launch > 1 (one-draw configurations): every lane's graph holds that many consecutive steps (GraphedPipeline)."""
    from bbb_hip import ensemble
    net, x = build_net(cfg, dev)
    E = cfg["E"] if total_ens is None else total_ens
    prec = cfg["precision"]

    def barrier():
        if world > 1:
            torch.distributed.barrier(group=group)
        torch.cuda.synchronize(dev)

    out = {}
    with torch.no_grad():
        if pipeline > 1:
            gstep = ensemble.GraphedPipeline(net, x, E, depth=pipeline, group=group, precision=prec, steps_per_launch=steps_per_launch)
        else:
            gstep = ensemble.GraphedMC(net, x, E, group=group, precision=prec)
        step = gstep.step
        flush = gstep.sync if steps_per_launch > 1 else (lambda: None)      # a partly filled group is launched before the clock stops
        for _ in range(max(1, pipeline) * steps_per_launch):   # setup: every lane's graph is replayed once (first replay = its upload)
            step()
        torch.cuda.synchronize(dev)
        for _ in range(warmup):
            step()
        flush()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            lo, kl = step()
        flush()
        barrier()
        elapsed = time.perf_counter() - t0
        lo, kl = lo.clone(), kl.clone()
        assert torch.isfinite(lo).all() and torch.isfinite(kl).all()
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
            elapsed = t.item()
        rows = lo.shape[0]
        out["ms_per_step"] = round(1e3 * elapsed / steps, 4)
        out["value"] = round(cfg["B"] * E / (elapsed / steps), 1)
        out["rows_out"] = rows
        if stat_blocks and world == 1:
            # block statistics: `stat_blocks` blocks of `steps` steps, each between device syncs
            vals = []
            for _ in range(stat_blocks):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(steps):
                    step()
                torch.cuda.synchronize(dev)
                vals.append(cfg["B"] * E * steps / (time.perf_counter() - t1))
            out["stats"] = {"blocks": stat_blocks, "steps_per_block": steps, "median": round(statistics.median(vals), 1),
                            "p10": round(pctl(vals, 0.1), 1), "p90": round(pctl(vals, 0.9), 1), "unit": "samples/s"}
        if world == 1 and pipeline > 1:
            g1 = ensemble.GraphedMC(net, x, E, precision=prec)
            for _ in range(5):
                g1.step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(steps):
                g1.step()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t1) / steps
            out["one_step_in_flight"] = {"value": round(cfg["B"] * E / dt, 1), "ms_per_step": round(1e3 * dt, 4),
                                         "note": "one hipGraph lane: step i+1 starts only after step i has drained (step latency)"}
        if want_roofline and world == 1:
            timers = ensemble.Timers()
            # park the GPU behind a ~40 ms spin kernel so that every launch below is already queued when its turn comes:
            # the event brackets then hold kernel time only, not host launch latency
            torch.cuda._sleep(int(1.0e8))
            if steps_per_launch > 1:                     # the launches of the timed region: steps_per_launch batches, one draw each
                from bbb_hip import rng
                xg = x.repeat(steps_per_launch, 1, 1, 1)

                def one_pass(t):
                    seed, call0 = rng.next_calls(steps_per_launch * E)
                    ensemble._local_lse(net, xg, E, seed, call0, E, timers=t, precision=prec, groups=steps_per_launch)
            else:
                def one_pass(t):
                    ensemble.mc_forward(net, x, E, timers=t, precision=prec)
            for _ in range(timer_steps):
                one_pass(timers)
            torch.cuda.synchronize(dev)
            eager = gemm_roofline(timers.summary(), timer_steps * steps_per_launch, prec, cfg is CONFIGS["metric"])
            # the same launches in the timed region's launch mode: every GEMM launch of the step replayed 20x back to back
            # inside its own hipGraph (no host, no event packets between kernels), HIP events around the replay
            rec = LaunchRecorder()
            one_pass(rec)
            torch.cuda.synchronize(dev)
            ing = rec.time_in_graphs(dev)
            roof = gemm_roofline(ing, steps_per_launch, prec, cfg is CONFIGS["metric"]) if ing else None
            if roof is not None and eager is not None:
                roof["timed_by"] = ("per launch: a hipGraph of %d back-to-back replays of that launch (the timed region's launch mode), HIP events "
                                    "around the graph on its stream, median of 5; summed over the step's %d conv/linear launches" % (rec.reps, roof["launches"]))
                roof["eager_event_brackets"] = {"achieved": eager["achieved"], "frac": eager["frac"], "avg_us": eager["avg_us"],
                                                "launches": eager["launches"], "timed_by": eager["timed_by"]}
                roof["per_launch_us"] = rec.per_launch_us
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
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / steps

    dt_ng = timed(torch.no_grad())
    dt_ag = timed(torch.enable_grad())
    return {"value": round(B * E / dt_ng, 1), "unit": "samples/s", "ms_per_step": round(1e3 * dt_ng, 4),
            "autograd_enabled": {"value": round(B * E / dt_ag, 1), "ms_per_step": round(1e3 * dt_ag, 4),
                                 "note": "what the UNMODIFIED validate_model gets: it does not disable autograd, so every forward "
                                         "also records one autograd node (same batch-innermost kernels, bbb_hip/fast_train.py)"},
            "note": "for j in range(10): net(x) through the drop-in layers + torch log_softmax / logmeanexp, eager launches.  Under "
                    "torch.no_grad() each net(x) runs on the batch-innermost inference kernels (one draw per call); the headline "
                    "uses the batched ensemble entry point (all draws per launch, one hipGraph) instead"}


def split_fp16(dev, steps, pipeline):
    """The metric step with the GEMM launches on the 16-bit matrix pipe at fp32 accuracy (ops.gemm_mode = "fp16x2",
    bbb_conv2d_chwn_f16x2_fwd: operands split into two fp16 pieces while staged, three products per fp32 product, fp32
    accumulation).  Opt-in mode, reported NEXT TO the fp32 headline, never as it: throughput, single-lane latency, per-launch
    times, and the largest difference of the step's log-probabilities from the fp32 path under the same noise."""
    from bbb_hip import ensemble, ops, rng
    cfg = CONFIGS["metric"]
    net, x = build_net(cfg, dev)
    E = cfg["E"]
    out = {}
    try:
        with torch.no_grad():
            seed_call = rng.next_calls(0)
            ref_lo = ensemble.mc_forward(net, x, E)[0].clone()
            ops.gemm_mode = "fp16x2"
            rng.rewind(seed_call)
            lo = ensemble.mc_forward(net, x, E)[0]
            out["max_abs_diff_of_log_probs_vs_fp32_path"] = float((lo - ref_lo).abs().max())
            out["max_abs_log_prob"] = float(ref_lo.abs().max())
            for name, depth in (("steps_in_flight_%d" % pipeline, pipeline), ("one_step_in_flight", 1)):
                pipe = ensemble.GraphedPipeline(net, x, E, depth=depth)
                for _ in range(20):
                    pipe.step()
                pipe.sync()
                t0 = time.perf_counter()
                for _ in range(steps):
                    pipe.step()
                pipe.sync()
                dt = (time.perf_counter() - t0) / steps
                out[name] = {"ms_per_step": round(1e3 * dt, 4), "value": round(cfg["B"] * E / dt, 1)}
                del pipe
            rec = LaunchRecorder()
            ensemble.mc_forward(net, x, E, timers=rec)
            torch.cuda.synchronize(dev)
            ing = rec.time_in_graphs(dev)
            g = ing.get("conv_gemm")
            if g:
                tf = g["work"] / (g["ms"] * 1e-3) / 1e12
                out["gemm_launches"] = {"per_launch_us": rec.per_launch_us, "fp32_equivalent_TFLOPs": round(tf, 1),
                                        "f16_mfma_TFLOPs": round(3 * tf, 1), "frac_of_bf16_f16_peak": round(3 * tf / PEAK_BF16_MFMA_TFLOPS, 4),
                                        "frac_of_fp32_matrix_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}
        out["unit"] = "samples/s"
        out["note"] = ("opt-in precision mode: fp32 tensors in HBM, every GEMM operand element split into hi = fp16(a), lo = fp16(a - hi) "
                       "while its tile is staged, products hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulation; "
                       "against the float64 oracle 1.2-2.4e-7 of sum|w||x| (the fp32 kernel: 0.8-3.9e-7), tests/test_gpu_f16x2.py")
    finally:
        ops.gemm_mode = "fp32"
    return out


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
    out = {"ms_per_step": round(1e3 * dt, 4), "value": round(cfg["B"] * cfg["E"] / dt, 1), "unit": "samples/s (forward + backward + Adam)",
           "launch_by_launch_ms_per_step": round(1e3 * res["launch_by_launch"], 4), "path": path,
           "note": "BayesianAlexNet bs=512 num_ens=10, fp32, train.train_step as called (it captures itself as one hipGraph after 3 "
                   "identical calls; graph=False = launch_by_launch); round 1 (reference-layout autograd path): 8.3 ms"}
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
                    help="independent MC steps in flight (hipGraph lanes on separate streams); default 3, and 4 when N > 1 "
                         "(a rank's share of a step is a few small launches: measured 124 vs 140 us per step for the 8-rank share)")
    ap.add_argument("--config", default="metric", choices=list(CONFIGS), help="which BASELINE configuration is the reported value")
    ap.add_argument("--gemm-mode", default="fp32", choices=["fp32", "fp16x2"],
                    help="fp16x2: the opt-in split-fp16 GEMM mode for the WHOLE run (profiling it, or timing a sharded run with it); the "
                         "default run reports it as the secondary object `split_fp16` next to the fp32 headline")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.pipeline is None:
        args.pipeline = 3 if world == 1 else 4
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
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    from bbb_hip import ensemble, _lib, ops
    _lib.lib()   # fail loudly here if the HIP library is missing
    ops.gemm_mode = args.gemm_mode

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_extras:
        cpu = cpu_baseline()

    cfg = CONFIGS[args.config]
    if args.no_graph:
        # profiling mode: the step launched eagerly on one stream, nothing else
        net, x = build_net(cfg, dev)
        with torch.no_grad():
            for _ in range(args.warmup):
                ensemble.mc_forward(net, x, cfg["E"], group=group, precision=cfg["precision"])
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ensemble.mc_forward(net, x, cfg["E"], group=group, precision=cfg["precision"])
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / args.steps
        if rank == 0:
            print(json.dumps({"metric": "eager single-stream step (profiling mode)", "value": round(cfg["B"] * cfg["E"] / dt, 1),
                              "unit": "samples/s", "ms_per_step": round(1e3 * dt, 4), "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    extras = rank == 0 and world == 1 and not args.no_extras
    head, net, x = run_config(cfg, args.steps, args.warmup, args.pipeline, dev, group, world, want_roofline=not args.no_roofline,
                              stat_blocks=5 if extras else 0, timer_steps=min(args.steps, 10))
    n_params = sum(p.numel() for n, p in net.named_parameters() if n.endswith("_mu"))

    weak = None
    if world > 1 and not args.no_extras:
        w, _, _ = run_config(cfg, max(5, args.steps // 2), 3, args.pipeline, dev, group, world, want_roofline=False,
                             total_ens=cfg["E"] * world)
        weak = {"value": w["value"], "unit": "samples/s", "ms_per_step": w["ms_per_step"], "num_ens_total": cfg["E"] * world,
                "note": "weak scaling, secondary: %d draws per GPU (a %d-draw ensemble of the same 512 images)" % (cfg["E"], cfg["E"] * world)}

    if rank == 0:
        with torch.no_grad():
            S, lo, hi = ensemble.shard_plan(net, x, cfg["E"], 0, world, True, cfg["precision"])
        out = {
            "metric": "MC-forward samples/sec, BayesianAlexNet CIFAR-10 bs=512 num_ens=10",
            "value": head["value"], "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if cfg["precision"] == "bf16" else
                     ("f32" if args.gemm_mode == "fp32" else "f32 tensors; GEMM products as 3 x f16 (hi/lo split), f32 accumulate"),
            "data": "synthetic",
            "config": {"workload": cfg["what"] + ", forward only (main_bayesian.py:73-80)",
                       "global_batch": cfg["B"], "num_ens_total": cfg["E"],
                       "parallelism": ("mc-ensemble work units: %d batch slices per draw, %d (draw x slice) units of %d images, "
                                       "<= %d per GPU, one all_gather per step" % (S, cfg["E"] * S, cfg["B"] // S, hi - lo))
                       if world > 1 else "single",
                       "launch": "hipGraph replay, %d step(s) in flight" % max(1, args.pipeline)},
        }
        if world > 1:
            out["config"]["ranks_seen"] = torch.distributed.get_world_size(group)
            out["config"]["units_per_rank"] = [(lambda r: r[1] - r[0])(ensemble.unit_range(cfg["E"], S, r, world)) for r in range(world)]
            out["config"]["backend"] = backend
        if args.config != "metric":
            out["metric"] = "MC-forward samples/sec, " + cfg["what"]
        if head.get("roofline"):
            out["roofline"] = head["roofline"]
            fps = head["roofline"].get("flop_per_step")
            if fps and world == 1:
                peak = head["roofline"]["peak"]
                ceil = mfma_loop_ceiling() if cfg["precision"] != "bf16" else None
                if ceil:
                    tf, ghz = ceil
                    out["roofline"]["mfma_loop_ceiling"] = {
                        "value": tf, "unit": "TFLOP/s", "frac_of_it": round(head["roofline"]["achieved"] / tf, 4),
                        "shader_clock_GHz": ghz or None,
                        "peak_at_that_clock": round(PEAK_F32_MFMA_TFLOPS * ghz / 2.4, 1) if ghz else None,
                        "note": "a loop of nothing but v_mfma_f32_32x32x2_f32 (4 accumulators per wave, 8 waves per CU, no memory), "
                                "timed on this box in this run, with the shader clock the kernel measured on itself (s_memtime / 100 MHz "
                                "wall clock): the nominal 157.3 TFLOP/s assumes 2.4 GHz; under a full-chip MFMA load the part runs "
                                "below that, which is the gap between this figure and the guide's 155"}
                out["roofline"]["sustained_in_timed_region"] = {
                    "achieved": round(fps / (head["ms_per_step"] * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                    "frac": round(fps / (head["ms_per_step"] * 1e-3) / 1e12 / peak, 4),
                    "note": "the same FLOPs per step / ms_per_step of the timed region (%d steps in flight; includes every "
                            "non-GEMM kernel of the step)" % max(1, args.pipeline)}
        for k in ("stats", "one_step_in_flight"):
            if k in head:
                out[k] = head[k]
        if weak is not None:
            out["weak_scaling"] = weak
        if extras:
            if cfg["lt"] == "bbb":
                out["roofline_reparam"] = reparam_probe(net, dev, n_params, cfg["E"])
            del net, x
            try:
                out["dropin_loop"] = dropin_loop(dev, max(5, args.steps // 5))
            except Exception as exc:
                out["dropin_loop"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
            if cfg is CONFIGS["metric"]:
                try:
                    if args.gemm_mode == "fp32":
                        out["split_fp16"] = split_fp16(dev, max(20, args.steps // 2), max(1, args.pipeline))
                except Exception as exc:
                    out["split_fp16"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
            try:
                out["training_step"] = training_step(dev, max(5, args.steps // 5))
            except Exception as exc:
                out["training_step"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
            torch.cuda.empty_cache()
            others = {}
            for name, c in CONFIGS.items():
                if name == args.config:
                    continue
                try:
                    # single-draw configurations are a dozen 10-20 us launches per step: four steps in flight instead of three
                    # (measured, profiles/r03_notes.md section 4: configs[1] 0.066 -> 0.055 ms, configs[2] 0.182 -> 0.164 ms)
                    small = c["E"] == 1 and c["hw"] == 32 and args.pipeline == 3
                    depth = args.pipeline + 1 if small else args.pipeline
                    # ... and FOUR consecutive steps per launch (GraphedPipeline steps_per_launch: four batches, each with its own
                    # weight draw and noise calls, in one set of launches; same per-step results as one step per launch,
                    # tests/test_gpu_steps_per_launch.py): configs[1] 0.054 -> 0.036 ms, configs[2] 0.164 -> 0.131 ms
                    spl = 4 if small else 1
                    nst = max(10, args.steps // 2)
                    nst = -(-nst // (spl * depth)) * spl * depth
                    r, n2, x2 = run_config(c, nst, 5, depth, dev, want_roofline=True, timer_steps=3, steps_per_launch=spl)
                    r["steps_in_flight"] = depth
                    if spl > 1:
                        r["steps_per_launch"] = spl
                        r1, _, _ = run_config(c, nst, 5, depth, dev, want_roofline=False, steps_per_launch=1)
                        r["one_step_per_launch"] = {"ms_per_step": r1["ms_per_step"], "value": r1["value"]}
                    del n2, x2
                    r["workload"] = c["what"]
                    r["dtype"] = "bf16" if c["precision"] == "bf16" else "f32"
                    r["unit"] = "samples/s"
                    others[name] = r
                except Exception as exc:
                    others[name] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
                torch.cuda.empty_cache()
            out["configs"] = others
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu"] = round(out["value"] / cpu["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
