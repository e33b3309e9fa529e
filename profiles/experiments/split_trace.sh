#!/usr/bin/env bash
# Kernel trace of the split-bf16 metric step's graph replays (4 steps per launch, one lane): per-kernel durations inside the timed region.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
{ for P in 1 2; do
    rm -rf /tmp/kg && mkdir -p /tmp/kg
    rocprofv3 --kernel-trace -d /tmp/kg -o kg -- python $R/bench.py --gemm-mode bf16x3 --steps 48 --warmup 8 --pipeline $P --no-extras --no-roofline --no-cpu-baseline --preheat-ms 0 > /tmp/kg/log.txt 2>&1
    echo "# rocprofv3 --kernel-trace -- python bench.py --gemm-mode bf16x3 --steps 48 --warmup 8 --pipeline $P --no-extras --no-roofline --preheat-ms 0   ($TAG; the last 12 launches per layer = the timed region's graph replays, 4 steps each)"
    grep '^{' /tmp/kg/log.txt | cut -c1-400
    python $R/profiles/summarize_rocpd.py $(find /tmp/kg -name '*.db' | head -1) --last-steps 12 --by-grid
    echo
  done; } > "$OUT/${TAG}_split_graph_kernel_stats.txt" 2>&1
cat "$OUT/${TAG}_split_graph_kernel_stats.txt" | head -60
