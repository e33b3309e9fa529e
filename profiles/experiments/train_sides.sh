#!/usr/bin/env bash
# Training step (bs 512 x 10, eager, warm): 1, 2 or 3 side streams for the weight-side work of the backward.
# (Round 6 also tried the input gradients' flipped weight operands prepared up front on a side stream: 2.695 against 2.677 ms: dropped.)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for rnd in 1 2; do for n in 1 2 3; do
  echo -n "side streams $n: "; BBB_TRAIN_SIDE_STREAMS=$n python $R/profiles/experiments/train_steps.py bbb 512 10 2>/dev/null | tail -1
done; done
