#!/usr/bin/env bash
# The driver's own invocation, three times: length of the judged line and the quantities tests/test_gpu_bench_contract.py asserts on.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for i in 1 2 3; do
  python $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $R/gpurun_out/r06_drv_$i.txt
  python - "$R/gpurun_out/r06_drv_$i.txt" <<'PY'
import json, sys
l = open(sys.argv[1]).read().strip()
d = json.loads(l); r = d["roofline"]
print(len(l), d["value"], r["single_block_value"], r["stats_p10"], r["stats_p90"], r["split_bf16_value"], r["split_bf16_frac_of_bf16_peak"],
      r["reparam_frac_of_write_roof"], r.get("training_step_frac"), d.get("cpu_baseline", {}).get("value"))
PY
done
