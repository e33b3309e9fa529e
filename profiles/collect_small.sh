#!/usr/bin/env bash
# Round 5: matrix-pipe counters of the single-draw configurations at the launch shape bench.py times them in (16 steps per launch):
#   gpurun -- 'bash profiles/collect_small.sh r05'
# --pmc passes never combine with other trace domains (only --kernel-trace).
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
{ echo "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- python bench.py --steps 16 --warmup 16 --no-graph --config <c> --steps-per-launch 16   (eager single-stream passes of 16 one-draw steps per launch)"
  for C in ${SMALL_CONFIGS:-"configs[1]" "configs[2]"}; do
    rm -rf /tmp/pm && mkdir -p /tmp/pm
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d /tmp/pm -o pm -- python $R/bench.py --steps 16 --warmup 16 --no-graph --config "$C" --steps-per-launch 16 > /tmp/pm_log.txt 2>&1
    echo "## $C"
    grep '^{' /tmp/pm_log.txt | cut -c1-300
    python $R/profiles/summarize_pmc.py $(find /tmp/pm -name '*.db' | head -1) pconv
  done; } > "$OUT/${TAG}_pmc_sq_mfma_small_configs.txt" 2>&1
ls -la "$OUT" | tail -3
