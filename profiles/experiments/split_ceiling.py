"""Where should the split-contraction plan stop?  One process per (BBB_SPLIT_CEIL, BBB_SPLIT_WANT) pair (experiment build of
pconv_gemm.hip that reads them): ms per step of AlexNet bs 512 for E = 1..6 (1 and 3 lanes), LRT CIFAR-100 E = 1, and the busiest
rank's share of the 8- and 4-rank step (4 lanes).

The experiment build is two lines in split_plan() (csrc/pconv_gemm.hip), not shipped:
    static const int64_t ceil_items = getenv("BBB_SPLIT_CEIL") ? atoll(getenv("BBB_SPLIT_CEIL")) : 384;
    static const int64_t want_wgs = getenv("BBB_SPLIT_WANT") ? atoll(getenv("BBB_SPLIT_WANT")) : 768;
replacing the constants 384 (`items > 384`) and 768 (`want`)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    for ceil, want in ((0, 768), (384, 768), (512, 768), (767, 768), (767, 1024), (1023, 1024), (1023, 1536)):
        env = dict(os.environ, BBB_SPLIT_CEIL=str(ceil), BBB_SPLIT_WANT=str(want))
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env)
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ensemble, ops, rng, zoo
PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
dev = torch.device("cuda:0")


def build(lt, classes):
    torch.manual_seed(0)
    net = zoo.getModel("alexnet", 3, classes, PRI, lt, "softplus").to(dev)
    rng.assign_stream_ids(net)
    return net, torch.rand(512, 3, 32, 32, device=dev)


def time_steps(net, x, E, lanes, n=300):
    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, E, depth=lanes)
        for _ in range(30):
            pipe.step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n):
                pipe.step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
    del pipe
    return round(best * 1e3, 4)


class Lane:
    def __init__(self, net, x, S, lo, hi, lane, lanes, E=10):
        self.counter = torch.full((1,), lane * E, dtype=torch.int32, device=dev)
        self.stream = ensemble._lane_streams(dev, lanes)[lane]
        self.stride = lanes * E

        def body():
            lse, kl = ensemble._local_lse(net, x, E, 1, 0, 0, units=(S, lo, hi))
            self.counter.add_(self.stride)
            return lse, kl
        with torch.no_grad(), torch.cuda.stream(self.stream), rng.device_call_offset(self.counter):
            for _ in range(2):
                body()
        torch.cuda.synchronize()
        self.g = torch.cuda.CUDAGraph()
        with torch.no_grad(), rng.device_call_offset(self.counter), torch.cuda.graph(self.g, stream=self.stream, capture_error_mode="thread_local"):
            self.out = body()

    def step(self):
        with torch.cuda.stream(self.stream):
            self.g.replay()


def share(net, x, world, depth=4):
    S = ensemble.plan_slices(10, world, 512)
    lo, hi = ensemble.unit_range(10, S, world - 1, world)
    lanes = [Lane(net, x, S, lo, hi, l, depth) for l in range(depth)]
    for i in range(40):
        lanes[i % depth].step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(400):
            lanes[i % depth].step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 400)
    return round(best * 1e3, 4)


row = {"ceil": os.environ.get("BBB_SPLIT_CEIL"), "want": os.environ.get("BBB_SPLIT_WANT")}
net, x = build("bbb", 10)
for E in (1, 2, 3, 4, 5, 6):
    row[f"E{E}"] = [time_steps(net, x, E, 1), time_steps(net, x, E, 3)]
row["share8"] = share(net, x, 8)
row["share4"] = share(net, x, 4)
net, x = build("lrt", 100)
row["lrt_E1"] = [time_steps(net, x, 1, 1), time_steps(net, x, 1, 4)]
print(json.dumps(row), flush=True)
