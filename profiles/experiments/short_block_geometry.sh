#!/usr/bin/env bash
# Launch geometry for SHORT timed blocks (the driver's --steps 20): steps per launch x lanes, three runs each, single timed block.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for spec in "4 2" "5 2" "10 2" "5 4" "4 5" "2 5" "10 1"; do
  set -- $spec
  echo -n "steps_per_launch=$1 lanes=$2: "
  for i in 1 2 3; do
    python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-roofline --steps-per-launch $1 --pipeline $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], end=' ')"
  done
  echo
done
