#!/usr/bin/env bash
# Collect the round's rocprofv3 evidence on the GPU box:  gpurun -- 'bash profiles/collect.sh r02'
# Separate passes (kernel trace / FETCH_SIZE / WRITE_SIZE / SQ counters), summaries into gpurun_out/<tag>_*.txt for copying
# to profiles/.  --pmc passes never combine with other trace domains (only --kernel-trace).
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline"
# eager single-stream passes of the timed region's launches, nothing else in the process.  Round 4: a launch carries 4 steps
# (bench.py --steps-per-launch 4, the default), so 4 warm-up + 8 timed steps = 3 passes = 12 MC steps; the bf16 configs[1] pass and
# the one-step-per-launch comparison (EAGER1) keep 2 + 5 = 7 steps.
EAGER="python $R/bench.py --steps 8 --warmup 4 --no-graph"
EAGER1="python $R/bench.py --steps 5 --warmup 2 --no-graph --steps-per-launch 1"
EAGER16="python $R/bench.py --steps 5 --warmup 2 --no-graph --config configs[1] --steps-per-launch 1"
NSTEPS=12

rm -rf /tmp/kt && mkdir -p /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- $BENCH > /tmp/kt/log.txt 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline   ($TAG; fp32 headline, block statistics, single lane, kernel-timer pass, reparam probe, drop-in loop, BASELINE configs[1..4])"
  grep '^{' /tmp/kt/log.txt | cut -c1-600
  DB=$(find /tmp/kt -name '*.db' | head -1)
  python $R/profiles/summarize_rocpd.py $DB
  echo
  echo "## only the fp32 HIP-event pass of the metric config (the 10 eager single-stream steps behind the first spin kernel = what roofline.avg_us was timed on)"
  python $R/profiles/summarize_rocpd.py $DB --after-spin 1 --steps 10; } > "$OUT/${TAG}_bench_kernel_stats.txt" 2>&1

# the TIMED REGION's launch mode: kernel trace of the hipGraph replays themselves (no roofline / extras passes in the process), one
# lane and the default three lanes; per-layer rows (--by-grid).  With several lanes in flight kernels of different steps overlap,
# so their durations add up to more than the wall time; the single-lane table is the per-kernel in-graph duration.
{ for P in 1 3; do
    rm -rf /tmp/kg && mkdir -p /tmp/kg
    rocprofv3 --kernel-trace -d /tmp/kg -o kg -- python $R/bench.py --steps 48 --warmup 8 --pipeline $P --no-extras --no-roofline --no-cpu-baseline --preheat-ms 0 > /tmp/kg/log.txt 2>&1
    echo "# rocprofv3 --kernel-trace -- python bench.py --steps 48 --warmup 8 --pipeline $P --no-extras --no-roofline --preheat-ms 0   ($TAG; the last 12 launches per layer = the timed region's graph replays, 4 steps each)"
    grep '^{' /tmp/kg/log.txt | cut -c1-400
    python $R/profiles/summarize_rocpd.py $(find /tmp/kg -name '*.db' | head -1) --last-steps 12 --by-grid
    echo
  done; } > "$OUT/${TAG}_graph_kernel_stats.txt" 2>&1

bash $R/profiles/experiments/split_trace.sh $TAG > /dev/null 2>&1
cd /tmp
if [ "${SKIP_PMC:-0}" = "1" ]; then ls -la "$OUT" | tail -8; exit 0; fi     # kernel traces only (the PMC passes were taken earlier in the round)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm && mkdir -p /tmp/pm
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o pm -- $EAGER > /tmp/pm/log.txt 2>&1
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 8 --warmup 4 --no-graph   (KB per launch, 4 steps per launch; gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads; STEP_TOTAL = per MC step, $NSTEPS steps)"
    python $R/profiles/summarize_pmc.py $(find /tmp/pm -name '*.db' | head -1) --steps $NSTEPS
    rm -rf /tmp/pm/*
    rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o pm -- $EAGER1 > /tmp/pm/log.txt 2>&1
    echo
    echo "## the same with ONE step per launch (python bench.py --steps 5 --warmup 2 --no-graph --steps-per-launch 1; 7 steps): rows prefixed G1_"
    python $R/profiles/summarize_pmc.py $(find /tmp/pm -name '*.db' | head -1) --steps 7 | sed 's/^STEP_TOTAL/G1_STEP_TOTAL/'; } > "$OUT/${TAG}_pmc_${C}.txt" 2>&1
done
# HBM traffic of the other configurations (BASELINE configs[1], [2], [4], configs[3] = the 25-draw step) and of the split-bf16 mode,
# at the launch shapes bench.py times them in: the files bench.py's profile_traffic(kind, suffix) reads for their `traffic` fields
for C in FETCH_SIZE WRITE_SIZE; do
  for SPEC in "configs1|--config configs[1] --steps 16 --warmup 16 --steps-per-launch 16|32" "configs2|--config configs[2] --steps 16 --warmup 16 --steps-per-launch 16|32" \
              "configs3|--config configs[3] --steps 4 --warmup 2 --steps-per-launch 1|6" "configs4|--config configs[4] --steps 3 --warmup 1 --steps-per-launch 1|4" \
              "split|--gemm-mode bf16x3 --steps 8 --warmup 4|12"; do
    NAME=${SPEC%%|*}; REST=${SPEC#*|}; ARGS=${REST%|*}; NST=${REST##*|}
    rm -rf /tmp/pm && mkdir -p /tmp/pm
    rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o pm -- python $R/bench.py --no-graph $ARGS > /tmp/pm/log.txt 2>&1
    { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --no-graph $ARGS   (KB per launch; gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads; STEP_TOTAL = per MC step, $NST steps)"
      grep '^{' /tmp/pm/log.txt | cut -c1-300
      python $R/profiles/summarize_pmc.py $(find /tmp/pm -name '*.db' | head -1) --steps $NST; } > "$OUT/${TAG}_pmc_${C}_${NAME}.txt" 2>&1
  done
done
rm -rf /tmp/pm && mkdir -p /tmp/pm
{ echo "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- (fp32 eager step, then the bf16 configs[1] eager step)"
  for CMD in "$EAGER" "$EAGER16" "$EAGER --gemm-mode bf16x3"; do
    rm -rf /tmp/pm/*
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d /tmp/pm -o pm -- $CMD > /tmp/pm_log.txt 2>&1
    echo "## $CMD" | sed "s#$R/##"
    python $R/profiles/summarize_pmc.py $(find /tmp/pm -name '*.db' | head -1) pconv
  done; } > "$OUT/${TAG}_pmc_sq_mfma.txt" 2>&1
# VALU picture of the fused reparam+KL pass (the north_star's named kernel)
rm -rf /tmp/pm && mkdir -p /tmp/pm
{ echo "# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $EAGER   (reparam kernels only)" | sed "s#$R/##"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm -o pm -- $EAGER > /tmp/pm_log.txt 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pm -name '*.db' | head -1) reparam; } > "$OUT/${TAG}_pmc_reparam_valu.txt" 2>&1
ls -la "$OUT" | tail -8
