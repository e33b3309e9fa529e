#!/bin/bash
# The driver's flags (--steps 20 --warmup 5) with 4 vs 5 steps per launch, interleaved, three runs each: 20 steps are five groups
# of four on two lanes (3 + 2: the last group runs alone) or four groups of five (2 + 2).
for i in 1 2 3; do
  for g in 4 5; do
    python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-roofline --steps-per-launch $g 2>/dev/null | tail -1 > /tmp/line.json
    python - "$g" <<'PY'
import json, sys
j = json.load(open("/tmp/line.json"))
print(json.dumps({"steps_per_launch": int(sys.argv[1]), "value": j["value"], "ms_per_step": j["ms_per_step"]}))
PY
  done
done
