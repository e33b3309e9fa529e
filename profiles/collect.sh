#!/usr/bin/env bash
# Collect the round's rocprofv3 evidence on the GPU box:  gpurun -- 'bash profiles/collect.sh r01'
# Separate passes (kernel trace / FETCH_SIZE / WRITE_SIZE), summaries into gpurun_out/<tag>_*.txt for copying to profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline"
EAGER="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-bf16-extra --no-graph --streams 1"

rm -rf /tmp/kt && mkdir -p /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- $BENCH > /tmp/kt/log.txt 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline   ($TAG; fp32 headline + secondary bf16 block + kernel-timer and probe passes)"
  grep '^{' /tmp/kt/log.txt | cut -c1-400
  DB=$(find /tmp/kt -name '*.db' | head -1)
  python $R/profiles/summarize_rocpd.py $DB
  echo
  echo "## only the fp32 HIP-event pass (the 10 eager single-stream steps behind the first spin kernel = what roofline.avg_us was timed on)"
  python $R/profiles/summarize_rocpd.py $DB --after-spin 1 --steps 10
  echo
  echo "## only the bf16 HIP-event pass (5 eager steps behind the second spin kernel = what bf16.roofline.avg_us was timed on)"
  python $R/profiles/summarize_rocpd.py $DB --after-spin 2 --steps 5; } > "$OUT/${TAG}_bench_kernel_stats.txt" 2>&1

for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm && mkdir -p /tmp/pm
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o pm -- $EAGER > /tmp/pm/log.txt 2>&1
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-bf16-extra --no-graph --streams 1   (KB per launch; gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads)"
    python $R/profiles/summarize_pmc.py $(find /tmp/pm -name '*.db' | head -1); } > "$OUT/${TAG}_pmc_${C}.txt" 2>&1
done
EAGER16="python $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-graph --streams 1"
rm -rf /tmp/pm && mkdir -p /tmp/pm
{ echo "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- (fp32 eager step, then the bf16 eager step)"
  for CMD in "$EAGER" "$EAGER16"; do
    rm -rf /tmp/pm/* 
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d /tmp/pm -o pm -- $CMD > /tmp/pm_log.txt 2>&1
    echo "## $CMD" | sed "s#$R/##"
    python $R/profiles/summarize_pmc.py $(find /tmp/pm -name '*.db' | head -1) pconv
    python $R/profiles/summarize_pmc.py $(find /tmp/pm -name '*.db' | head -1) reparam_kl_fwd
  done; } > "$OUT/${TAG}_pmc_sq_mfma.txt" 2>&1
ls -la "$OUT" | tail -5
