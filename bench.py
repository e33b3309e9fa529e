#!/usr/bin/env python3
"""MC-forward samples/sec of BayesianAlexNet (CIFAR-10 shape, bs=512, num_ens=10, BBB layers) on MI355X.

One step = main_bayesian.py:73-80 of the reference: num_ens x net(x) + log_softmax, then utils.logmeanexp --
forward only, no_grad, inputs resident in HBM.  Synthetic data: torch.manual_seed(0), parameters from the
layers' own reset_parameters with config_bayesian.priors, x ~ U[0,1).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Launch structure: each step is one captured hipGraph (a device-side call counter inside it advances the Philox
noise on every replay) and `--pipeline` (default 3) independent steps are in flight on separate HIP streams.

N > 1 (weak scaling): every rank runs num_ens=10 draws of the same 512-image batch -- a 10*N-draw ensemble
sharded over the GPUs (draw j is noise call call0 + j on whichever rank owns it) -- and the ranks combine
their log-sum-exp blocks and KL sums with ONE all_gather over RCCL per step.  value = B * 10 * N / step time.

Prints ONE JSON line on rank 0 (contract in the task description) with these extra objects:
  roofline      the dominant kernel (fp32-MFMA implicit-GEMM conv): algorithmic FLOP / HIP-event time vs the
                157.3 TFLOP/s fp32 matrix peak (+ its HBM traffic from the PMC passes in profiles/);
                roofline_reparam: the fused reparam+KL pass vs HBM.
  cpu_baseline  the oracle's torch-CPU port of the reference MC step (oracle/ref_port_torch.py, bit-identical
                to the upstream nn.Modules under the same seed) timed on this host's cores, rank 0, N=1 only.
  one_step_in_flight   the same workload with a single graph lane (step latency instead of throughput).
  bf16          secondary measurement of the same workload under the bf16 storage model (BASELINE.json configs[1]
                precision; `--dtype bf16` makes it the reported value instead).  Never the default headline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
sys.dont_write_bytecode = True

import torch  # noqa: E402

PRIORS = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
BATCH, NUM_ENS, CLASSES = 512, 10, 10
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense v_mfma_f32_32x32x16_bf16 peak (same guide)
PEAK_HBM_GBS = 8000.0             # HBM3E spec (6.3 TB/s achievable)


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


def cpu_baseline(budget_s=16.0):
    """Reference CPU path (oracle port: same ATen ops / order as the upstream modules), bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_port_torch as P
    avail = usable_cpus()
    torch.manual_seed(0)
    params = P.init_params("alexnet", 3, CLASSES, P.CONFIG_PRIORS)
    x = torch.rand(BATCH, 3, 32, 32)

    def run(nthreads, budget):
        torch.set_num_threads(nthreads)
        with torch.no_grad():
            P.mc_step("alexnet", params, x, CLASSES, 1, "bbb", "softplus")      # warm-up (1 draw)
            t0 = time.perf_counter()
            n = 0
            while True:
                P.mc_step("alexnet", params, x, CLASSES, NUM_ENS, "bbb", "softplus")
                n += 1
                el = time.perf_counter() - t0
                if (n >= 3 and el > budget) or n >= 100 or el > 4 * budget:
                    break
        return BATCH * NUM_ENS * n / el, n, el

    # mkldnn does not always scale to every core of the cgroup: time all cores and half of them, report the best
    best = None
    for nt in sorted({avail, max(1, avail // 2)}, reverse=True):
        v, n, el = run(nt, budget_s / 2)
        if best is None or v > best[0]:
            best = (v, n, el, nt)
    v, n, el, nt = best
    return {"value": round(v, 1), "unit": "samples/s", "cores": nt, "kind": "port",
            "sample": f"{n} full MC steps (bs={BATCH}, num_ens={NUM_ENS}, fp32, no_grad) of oracle/ref_port_torch.mc_step "
                      f"in {el:.1f}s after a warm-up; {avail} CPUs usable (cgroup quota / affinity) of "
                      f"{os.cpu_count()} logical, best of {{all, half}} thread counts"}


def reparam_probe(net, dev, n_params):
    """Fused reparam+KL pass timed on its own: 20 back-to-back launches inside one hipGraph (no host gaps), HIP events
    around the replay.  (a) the model's 12 tensors, E=10 - what the step runs, Infinity-Cache resident; (b) one
    2^26-element tensor (0.8-1.6 GB of traffic, far beyond the 256 MB cache) for a genuine HBM figure."""
    import torch
    from bbb_hip import ensemble, ops
    layers_ = ensemble.bayesian_layers(net)
    mus, rhos, ids = [], [], []
    for l in layers_:
        m, r, i = l._param_lists()
        mus += m
        rhos += r
        ids += i

    def timed(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize(dev)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize(dev)
        return s.elapsed_time(e) * 1e-3 / reps

    with torch.no_grad():
        t_model = timed(lambda: ops.reparam_kl_forward(mus, rhos, 0, 0.1, ids, 1, 0, draws=NUM_ENS), 20)
        byts = (8 + 4 * NUM_ENS) * n_params
        big = 1 << 26
        mu = torch.randn(big, device=dev) * 0.1
        rho = torch.randn(big, device=dev) * 0.1 - 5
        t1 = timed(lambda: ops.reparam_kl_forward([mu], [rho], 0, 0.1, [0], 1, 0, draws=1), 3)
        t4 = timed(lambda: ops.reparam_kl_forward([mu], [rho], 0, 0.1, [0], 1, 0, draws=4), 3)
        dst = torch.empty_like(mu)
        tc = timed(lambda: dst.copy_(mu), 3)
        del mu, rho, dst
    gbs = byts / t_model / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "traffic": 102.2e6 if abs(byts - 104445408) < 1 else None,
            "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 wide-load correction) + WRITE_SIZE, separate passes: "
                              "profiles/r01_pmc_FETCH_SIZE.txt, r01_pmc_WRITE_SIZE.txt (17.1 MB + 85.1 MB per launch)",
            "kernel": "reparam_kl_fwd_kernel + kl_finish_kernel, 12 tensors x 10 draws in one launch",
            "bytes_per_launch": byts, "avg_us": round(t_model * 1e6, 2),
            "note": "(8 + 4E) B per weight element; the 17 MB of (mu,rho) and 87 MB of w are Infinity-Cache resident here",
            "hbm_resident_probe": {
                "elements": big,
                "E1_GBps": round(12 * big / t1 / 1e9, 1), "E4_GBps": round(24 * big / t4 / 1e9, 1),
                "device_copy_GBps": round(8 * big / tc / 1e9, 1),
                "note": "single 2^26-element tensor, (8+4E) B/element; device_copy = torch copy_ of the same tensor (8 B/element), "
                        "the achievable streaming rate on this box"}}


def gemm_roofline(g, timer_steps, dtype):
    """Dominant kernel of the step: all conv / linear launches.  FLOPs counted = in-bounds taps only (what the fp32 kernel
    multiplies; the bf16 kernel also multiplies the zero taps, which is not counted as useful work)."""
    tf = g["work"] / (g["ms"] * 1e-3) / 1e12
    peak = PEAK_BF16_MFMA_TFLOPS if dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
    kern = ("pconv_bf16_kernel (v_mfma_f32_32x32x16_bf16, batch-innermost, LDS transpose reads)" if dtype == "bf16" else
            "pconv_gemm_kernel (fp32 v_mfma_f32_32x32x2_f32, batch-innermost, in-bounds taps only)")
    # HBM bytes per launch from the PMC passes of the metric workload (fp32 only; recorded, not re-measured per run)
    traffic = 88.5e6 if (dtype != "bf16" and abs(g["work"] / timer_steps - 70812958720.0) < 1.0) else None
    return {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
            "traffic": traffic,
            "traffic_source": ("rocprofv3 --pmc FETCH_SIZE (x2, gfx950 wide-load correction) + WRITE_SIZE, separate passes "
                               "(profiles/r01_pmc_FETCH_SIZE.txt, r01_pmc_WRITE_SIZE.txt): 326.3 MB read + 205.0 MB written per step "
                               "over the 6 conv/linear launches; algorithmic bytes 197 MB + 210 MB") if traffic else None,
            "kernel": kern + ", all conv/linear launches of a step",
            "launches": g["n"], "avg_us": round(1e3 * g["ms"] / g["n"], 2),
            "flop_per_step": g["work"] / timer_steps, "im2col_flop_per_step": g["work_im2col"] / timer_steps,
            "timed_by": "HIP events around every launch, %d eager single-stream steps of the same workload right after "
                        "the timed region" % timer_steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layer-type", default="bbb", choices=["bbb", "lrt"])
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="independent sub-ensembles on separate HIP streams")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--pipeline", type=int, default=3, help="independent MC steps in flight (hipGraph lanes on separate streams)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32: the reference's arithmetic on the exact fp32 matrix cores (headline).  bf16: sampled weights and "
                         "hidden activations stored as bf16, fp32 accumulate (BASELINE.json configs[1] precision)")
    ap.add_argument("--no-bf16-extra", action="store_true", help="skip the secondary bf16 measurement of the default run")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
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

    from bbb_hip import ensemble, rng, zoo, _lib
    _lib.lib()   # fail loudly here if the HIP library is missing

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    torch.manual_seed(0)
    net = zoo.BBBAlexNet(CLASSES, 3, PRIORS, args.layer_type, "softplus").to(dev)
    rng.assign_stream_ids(net)
    x = torch.rand(BATCH, 3, 32, 32).to(dev)
    total_ens = NUM_ENS * world
    n_params = sum(p.numel() for n, p in net.named_parameters() if n.endswith("_mu"))

    def barrier():
        if world > 1:
            torch.distributed.barrier(group=group)
        torch.cuda.synchronize(dev)

    use_graph = not args.no_graph
    precision = "bf16" if args.dtype == "bf16" else "fp32"
    if precision == "bf16" and args.layer_type != "bbb":
        raise SystemExit("--dtype bf16 covers BBB layers only")
    with torch.no_grad():
        launch_note = None
        if use_graph:
            # the whole step (E draws x all layers + tail) is one captured hipGraph; a device-side call counter inside the
            # graph gives every replay fresh Philox noise (no cached outputs)
            try:
                if args.pipeline > 1:
                    gstep = ensemble.GraphedPipeline(net, x, total_ens, depth=args.pipeline, streams=args.streams, group=group,
                                                     precision=precision)
                else:
                    gstep = ensemble.GraphedMC(net, x, total_ens, streams=args.streams, group=group, precision=precision)
                step = gstep.step
                step()                    # first replay (+ first collective when N > 1) inside the guarded region
                torch.cuda.synchronize(dev)
            except Exception as exc:      # launch-mode fallback only (same kernels, launched eagerly); reported in the JSON
                use_graph = False
                launch_note = "hipGraph capture failed (%s: %s); eager launches" % (type(exc).__name__, str(exc)[:120])
                torch.cuda.synchronize(dev)
        if not use_graph:
            # explicit --no-graph: exactly the requested stream count (profiling runs); capture failure: two draw streams
            eager_streams = args.streams if args.no_graph else max(2, args.streams)
            step = lambda: ensemble.mc_forward(net, x, total_ens, group=group, streams=eager_streams, precision=precision)
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            lo, kl = step()
        barrier()
        elapsed = time.perf_counter() - t0
        lo, kl = lo.clone(), kl.clone()
        # transparency: the same step with ONE step in flight (single graph lane), same process, same data
        serial = None
        if use_graph and args.pipeline > 1 and world == 1:
            g1 = ensemble.GraphedMC(net, x, total_ens, streams=args.streams, precision=precision)
            for _ in range(5):
                g1.step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                g1.step()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t1) / args.steps
            serial = {"value": round(BATCH * total_ens / dt, 1), "ms_per_step": round(1e3 * dt, 4),
                      "note": "one hipGraph lane: step i+1 starts only after step i has drained (step latency)"}
        # per-kernel HIP-event brackets: the same step launched eagerly on one stream (events cannot sit inside a graph)
        timers = None
        if not args.no_kernel_timers:
            timers = ensemble.Timers()
            timer_steps = min(args.steps, 10)
            # park the GPU behind a ~40 ms spin kernel so that every launch below is already queued when its turn comes:
            # the event brackets then hold kernel time only, not host launch latency
            torch.cuda._sleep(int(1.0e8))
            for _ in range(timer_steps):
                ensemble.mc_forward(net, x, total_ens, group=group, timers=timers, precision=precision)
            torch.cuda.synchronize(dev)
        # secondary measurement: the same workload under the bf16 storage model (never the headline value)
        bf16_extra = None
        if precision == "fp32" and args.layer_type == "bbb" and use_graph and not args.no_bf16_extra:
            try:
                gb = ensemble.GraphedPipeline(net, x, total_ens, depth=max(1, args.pipeline), streams=args.streams, group=group,
                                              precision="bf16")
                for _ in range(max(3, args.warmup)):
                    gb.step()
                barrier()
                tb = time.perf_counter()
                for _ in range(args.steps):
                    gb.step()
                barrier()
                tb = time.perf_counter() - tb
                tmb = ensemble.Timers()
                torch.cuda._sleep(int(1.0e8))
                for _ in range(5):
                    ensemble.mc_forward(net, x, total_ens, group=group, timers=tmb, precision="bf16")
                torch.cuda.synchronize(dev)
                bf16_extra = (tb, tmb.summary())
            except Exception as exc:
                bf16_extra = "bf16 measurement failed: %s: %s" % (type(exc).__name__, str(exc)[:160])

    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
        elapsed = t.item()
    assert torch.isfinite(lo).all() and torch.isfinite(kl).all()

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        out = {
            "metric": "MC-forward samples/sec, BayesianAlexNet CIFAR-10 bs=512 num_ens=10",
            "value": round(BATCH * total_ens / (elapsed / args.steps), 1),
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BayesianAlexNet 3x32x32 -> {CLASSES} classes, layer_type={args.layer_type}, "
                                   f"softplus, bs={BATCH}, num_ens={NUM_ENS} per GPU ({total_ens} draws total), "
                                   "forward only (main_bayesian.py:73-80)",
                       "global_batch": BATCH, "num_ens_total": total_ens,
                       "parallelism": f"mc-ensemble x{world}" if world > 1 else "single",
                       "launch": ("hipGraph replay, %d step(s) in flight x %d draw streams" % (max(1, args.pipeline), args.streams))
                       if use_graph else (launch_note or ("eager, %d streams" % (args.streams if args.no_graph else max(2, args.streams))))},
        }
        if timers is not None:
            agg = timers.summary()
            g = agg.get("conv_gemm") or agg.get("lrt_gemm")
            if g:
                out["roofline"] = gemm_roofline(g, timer_steps, args.dtype)
            if args.layer_type == "bbb":
                out["roofline_reparam"] = reparam_probe(net, dev, n_params)
        if serial is not None:
            out["one_step_in_flight"] = serial
        if isinstance(bf16_extra, tuple):
            tb, aggb = bf16_extra
            out["bf16"] = {"value": round(BATCH * total_ens / (tb / args.steps), 1), "unit": "samples/s",
                           "ms_per_step": round(1e3 * tb / args.steps, 4),
                           "note": "same workload, launch structure and noise streams with sampled weights + hidden activations "
                                   "stored as bf16 (fp32 accumulate / bias / activation / KL); parity: tests/test_gpu_bf16.py; "
                                   "not the headline value",
                           "roofline": gemm_roofline(aggb["conv_gemm"], 5, "bf16") if "conv_gemm" in aggb else None}
        elif bf16_extra is not None:
            out["bf16"] = {"error": bf16_extra}
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu"] = round(out["value"] / cpu["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
