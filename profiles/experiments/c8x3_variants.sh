#!/usr/bin/env bash
# Scheduling variants of the c8x3 k loop (C8X3_VARIANT: 1 = s_setprio around the matrix instructions, 2 = the next weight tile written after
# the barrier instead of before it, 3 = both).  build HERE:  bash profiles/experiments/c8x3_variants.sh build ; run on the GPU box: ... run
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
P=$R/pytorch-bayesiancnn_amd
mkdir -p $R/build_var
VARS="${VARS:-1 2 3}"
if [ "${1:-run}" = build ]; then
  for k in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DC8X3_VARIANT=$k -c $P/csrc/pconv_c8x3.hip -o $R/build_var/c8x3_v$k.o &
  done
  wait
  for k in $VARS; do
    objs=$(ls $P/build/*.o | grep -v pconv_c8x3.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/build_var/c8x3_v$k.o -o $R/build_var/libbbb_v$k.so
  done
  ls -la $R/build_var/libbbb_v*.so
  exit 0
fi
echo "# c8x3 k-loop variants (us per launch, 40 slabs; two rounds): 0 = shipped"
for round in 1 2; do
for k in 0 $VARS; do
  if [ $k = 0 ]; then unset BBB_HIP_LIB; else export BBB_HIP_LIB=$R/build_var/libbbb_v$k.so; fi
  echo -n "variant=$k "
  C8X3_ONLY=1 python $R/profiles/experiments/c8x3_layers.py 40 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:v['c8x3_us'] for k,v in d['layers'].items()}, d['total']['c8x3_us'], 'err', [v['rel_diff_vs_fp32'] for v in d['layers'].values()])"
done
done
