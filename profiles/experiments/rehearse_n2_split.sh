#!/usr/bin/env bash
# The driver's N > 1 invocation rehearsed on ONE device (both ranks on GPU 0, gloo) in split-bf16 mode and in fp32: the flow, not the number.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for m in bf16x3 fp32; do
  BBB_BENCH_DEVICE=0 BBB_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
      $R/bench.py --gpus 2 --steps 8 --warmup 4 --gemm-mode $m --no-extras 2>&1 | tail -1 | cut -c1-900
done
