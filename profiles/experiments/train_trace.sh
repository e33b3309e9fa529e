#!/usr/bin/env bash
# Kernel trace of the training step (forward + backward + Adam) at the metric shape (bs 512 x 10 draws, BBB) and at the reference's default
# configuration (lrt, bs 256, 1 draw), launch by launch: per-kernel totals.   gpurun -- 'bash profiles/experiments/train_trace.sh r06'
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
{ for ARGS in "bbb 512 10" "lrt 256 1"; do
    rm -rf /tmp/kt && mkdir -p /tmp/kt
    TRAIN_STEPS_LONG=0 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/profiles/experiments/train_steps.py $ARGS > /tmp/kt/log.txt 2>&1
    echo "# rocprofv3 --kernel-trace -- python profiles/experiments/train_steps.py $ARGS   ($TAG; 5 warm-up + 10 steps, eager; totals over the 15 steps -- the weight-side launches run on two side streams beside the gradient chain, so durations add up to more than the wall time)"
    tail -2 /tmp/kt/log.txt
    python $R/profiles/summarize_rocpd.py $(find /tmp/kt -name '*.db' | head -1) --last-steps 10 --by-grid 2>&1 | head -70
    echo
  done; } > "$OUT/${TAG}_train_kernel_stats.txt" 2>&1
head -80 "$OUT/${TAG}_train_kernel_stats.txt"
