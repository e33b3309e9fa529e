#!/usr/bin/env bash
# Counter passes of the c8x3 kernel's launches (gpurun -- 'bash profiles/experiments/c8x3_pmc.sh TAG').  One --pmc pass per counter group, --kernel-trace only.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/profiles/experiments/c8x3_pmc.py"
{ echo "# c8x3 kernel, AlexNet CIFAR conv2..conv5, bs 512, 40 slabs per launch; rocprofv3 --kernel-trace --pmc <group> -- python profiles/experiments/c8x3_pmc.py"
  for G in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    rm -rf /tmp/pm && mkdir -p /tmp/pm
    rocprofv3 --kernel-trace --pmc $G -d /tmp/pm -o pm -- $CMD > /tmp/pm_log.txt 2>&1
    echo "## --pmc $G"
    grep -i "ceiling\|error\|invalid" /tmp/pm_log.txt | head -5
    DB=$(find /tmp/pm -name '*.db' | head -1)
    [ -n "$DB" ] && python $R/profiles/summarize_pmc.py $DB c8x3 || tail -5 /tmp/pm_log.txt
  done; } > "$OUT/${TAG}_pmc_c8x3.txt" 2>&1
rocprofv3 -L > "$OUT/rocprofv3_counters.txt" 2>&1 || true
tail -60 "$OUT/${TAG}_pmc_c8x3.txt"
