#!/usr/bin/env bash
# What the c8x3 kernel's time is made of: the library rebuilt with one part of the kernel removed at a time (C8X3_ABLATE, results are
# garbage, timing only), AlexNet conv2..conv5 at 40 slabs.  Build HERE (no GPU needed):  bash profiles/experiments/c8x3_ablate.sh build
# Run on the GPU box:  bash profiles/experiments/c8x3_ablate.sh run > gpurun_out/r06_c8x3_ablate.txt
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
P=$R/pytorch-bayesiancnn_amd
mkdir -p $R/build_var
if [ "${1:-run}" = build ]; then
  for k in ${ABL:-1 2 3 4 5}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DC8X3_ABLATE=$k -c $P/csrc/pconv_c8x3.hip -o $R/build_var/c8x3_ab$k.o &
  done
  wait
  for k in ${ABL:-1 2 3 4 5}; do
    objs=$(ls $P/build/*.o | grep -v pconv_c8x3.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/build_var/c8x3_ab$k.o -o $R/build_var/libbbb_ab$k.so
  done
  ls -la $R/build_var/*.so
  exit 0
fi
echo "# c8x3 ablations (us per launch, 40 slabs): 0 = the kernel; 1 no epilogue math; 2 no barriers; 3 no image loads; 4 no weight staging; 5 no MFMAs"
for k in 0 ${ABL:-1 2 3 4 5}; do
  if [ $k = 0 ]; then unset BBB_HIP_LIB; else export BBB_HIP_LIB=$R/build_var/libbbb_ab$k.so; fi
  echo -n "ablate=$k "
  C8X3_ONLY=1 C8X3_NOCHECK=1 python $R/profiles/experiments/c8x3_layers.py 40 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:v['c8x3_us'] for k,v in d['layers'].items()}, d['total']['c8x3_us'])"
done
