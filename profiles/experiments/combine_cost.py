"""Host + device cost of the N > 1 step protocol on ONE GPU: world-size-1 RCCL with BBB_FORCE_COMBINE=1, work = what the busiest
rank of `world` ranks would run (units).  Round 4: both protocols -- the collective and the reduction over ranks recorded into the
step's hipGraph (one host call per step; per-lane communicators) vs graph replay + eager all_gather + post graph (three)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
os.environ["BBB_FORCE_COMBINE"] = "1"
if len(sys.argv) > 1 and sys.argv[1] == "cache":          # reproduces the watchdog abort (hipErrorCapturedEvent) of r04_notes.md
    os.environ["TORCH_NCCL_CUDA_EVENT_CACHE"] = "1"
import torch, torch.distributed as dist
from bbb_hip import ensemble, zoo, rng
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
group = dist.group.WORLD
PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
torch.manual_seed(0)
net = zoo.BBBAlexNet(10, 3, PRI, "bbb", "softplus").to(dev)
rng.assign_stream_ids(net)
E = 10
for B, Eloc, tag in ((128, 5, "8 ranks: 5 units of 128 images"), (256, 5, "4 ranks: 5 units of 256"), (512, 5, "2 ranks: 5 draws of 512")):
    x = torch.rand(B, 3, 32, 32, device=dev)
    for depth in (1, 3, 4):
        for fused in (True, False):
            use_group = group
            ensemble.capture_collectives = fused
            ensemble._capture_probe.clear()
            pipe = ensemble.GraphedPipeline(net, x, Eloc, depth=depth, group=use_group)
            assert all(l.fused == fused for l in pipe.lanes)
            for _ in range(30): pipe.step()
            pipe.sync()
            n = 600
            t0 = time.perf_counter()
            for _ in range(n): pipe.step()
            t1 = time.perf_counter()
            pipe.sync()
            t2 = time.perf_counter()
            print(json.dumps({"case": tag, "lanes": depth, "collective_in_graph": fused, "host_us_per_step": round((t1 - t0) / n * 1e6, 1),
                              "us_per_step": round((t2 - t0) / n * 1e6, 1)}), flush=True)
            del pipe
dist.destroy_process_group()
