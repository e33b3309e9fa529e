"""Per-rank GPU time of the strong-scaling step, measured on ONE GPU: rank r of N runs its work units (no collective).
`python rank_sim.py groups [G]`: the same for GROUPS of G steps per launch (default 4) dealt to the ranks as contiguous draw ranges
(ensemble.group_share) -- ms per STEP of the busiest rank."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ensemble, zoo, rng, ops
PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = zoo.BBBAlexNet(10, 3, PRI, "bbb", "softplus").to(dev)
rng.assign_stream_ids(net)
x = torch.rand(512, 3, 32, 32, device=dev)
E = 10
class Lane:
    def __init__(self, S, lo, hi, lane, lanes):
        self.counter = torch.full((1,), lane * E, dtype=torch.int32, device=dev)
        self.stream = ensemble._lane_streams(dev, lanes)[lane]     # the pool GraphedPipeline uses
        self.S, self.lo, self.hi = S, lo, hi
        self.stride = lanes * E
        with torch.no_grad(), torch.cuda.stream(self.stream), rng.device_call_offset(self.counter):
            for _ in range(2): self.body()
        torch.cuda.synchronize()
        self.g = torch.cuda.CUDAGraph()
        with torch.no_grad(), rng.device_call_offset(self.counter), torch.cuda.graph(self.g, stream=self.stream, capture_error_mode="thread_local"):
            self.out = self.body()
    def body(self):
        if self.S > 1:
            lse, kl = ensemble._local_lse(net, x, E, 1, 0, 0, units=(self.S, self.lo, self.hi))
        else:
            lse, kl = ensemble._local_lse(net, x, self.hi - self.lo, 1, self.lo, 0)
        self.counter.add_(self.stride)
        return lse, kl
    def step(self):
        with torch.cuda.stream(self.stream):
            self.g.replay()
class GroupLane(Lane):
    def __init__(self, G, rank, world, lane, lanes):
        self.G = G
        self.lo, self.hi, g_lo, self.n_gl, self.off = ensemble.group_share(E, G, rank, world)
        self.xg = x.repeat(self.n_gl, 1, 1, 1)
        Lane.__init__(self, 1, self.lo, self.hi, lane, lanes)
        self.stride = lanes * E * G
    def body(self):
        with ops.overlapped_launches(lanes_stride[0] > E * self.G):      # more than one lane: what GraphedMC tells the launch planner
            lse, kl = ensemble._local_lse(net, self.xg, self.hi - self.lo, 1, self.lo, 0, share=(E, self.off))
        self.counter.add_(lanes_stride[0])
        return lse, kl
lanes_stride = [0]
if len(sys.argv) > 1 and sys.argv[1] == "groups":
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ops.pool_fusion = not (len(sys.argv) > 3 and sys.argv[3] == "nofuse")
    for world in (1, 2, 4, 8):
        for depth in (1, 2, 3, 4):
            worst = 0
            lanes_stride[0] = depth * E * G
            for rank in sorted({0, world // 2, world - 1}):
                lanes = [GroupLane(G, rank, world, l, depth) for l in range(depth)]
                for i in range(12): lanes[i % depth].step()
                torch.cuda.synchronize()
                t_end = time.perf_counter() + 0.3
                while time.perf_counter() < t_end:
                    for i in range(depth): lanes[i].step()
                    torch.cuda.synchronize()
                t0 = time.perf_counter(); n = 120
                for i in range(n): lanes[i % depth].step()
                torch.cuda.synchronize()
                worst = max(worst, (time.perf_counter() - t0) / (n * G))
                del lanes
            print(json.dumps({"world": world, "steps_per_launch": G, "draws_per_rank": -(-G * E // world), "lanes": depth,
                              "ms_per_step_busiest_rank": round(worst * 1e3, 4), "projected_samples_per_s": round(5120 / worst, 0)}), flush=True)
    sys.exit(0)
for world in (1, 2, 4, 8):
    S = ensemble.plan_slices(E, world, 512)
    for depth in (1, 2, 3, 4):
        worst = 0
        for rank in sorted({0, world - 1}):
            lo, hi = ensemble.unit_range(E, S, rank, world)
            lanes = [Lane(S, lo, hi, l, depth) for l in range(depth)]
            for i in range(30): lanes[i % depth].step()
            torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 300
            for i in range(n): lanes[i % depth].step()
            torch.cuda.synchronize()
            worst = max(worst, (time.perf_counter() - t0) / n)
            del lanes
        print(json.dumps({"world": world, "S": S, "lanes": depth, "ms_per_step_busiest_rank": round(worst * 1e3, 4),
                          "projected_samples_per_s": round(5120 / worst, 0)}), flush=True)
