cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in bf16x3 fp32; do
rm -rf /tmp/kg && mkdir -p /tmp/kg
rocprofv3 --kernel-trace -d /tmp/kg -o kg -- python $R/bench.py --config "configs[4]" --gemm-mode $m --steps 8 --warmup 2 --pipeline 1 --no-extras --no-roofline --no-cpu-baseline --preheat-ms 0 > /tmp/kg/log.txt 2>&1
echo "# configs[4] $m"
grep '^{' /tmp/kg/log.txt | cut -c1-200
python $R/profiles/summarize_rocpd.py $(find /tmp/kg -name '*.db' | head -1) --last-steps 4 --by-grid | head -24
done
