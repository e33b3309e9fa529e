#!/usr/bin/env bash
# Variant library for bf16_shape_sweep.py: a COPY of pconv_bf16.hip whose launcher reads BBB_BF16_FORCE, linked with the shipped objects
# into build_var/libbbb_force.so (git-ignored; loaded through BBB_HIP_LIB by the experiment only).  Run ./build.sh first.
set -euo pipefail
cd "$(dirname "$0")/../.."
mkdir -p build_var
python - <<'PY'
src = open('pytorch-bayesiancnn_amd/csrc/pconv_bf16.hip').read()
a = "    if (tiny) shape = 12;\n"
b = "    if (ws) kgs = 1;\n"
assert src.count(a) == 1 and src.count(b) == 1
src = src.replace(a, a + '    static const int force = getenv("BBB_BF16_FORCE") ? atoi(getenv("BBB_BF16_FORCE")) : 0;   // shape*100 + kgs*10 + ws\n'
                      '    if (!tiny && force) shape = force / 100;\n')
src = src.replace(b, b + '    if (!tiny && force) { kgs = (force / 10) % 10; ws = (force % 10) != 0; if (ws) kgs = 1; if (shape != 12 && kgs == 4) kgs = 2; }\n')
for h, p in (('"../../include/bbb_hip.h"', '"../include/bbb_hip.h"'), ('"bbb_common.cuh"', '"../pytorch-bayesiancnn_amd/csrc/bbb_common.cuh"'),
             ('"pconv_args.h"', '"../pytorch-bayesiancnn_amd/csrc/pconv_args.h"')):
    src = src.replace('#include ' + h, '#include ' + p)
open('build_var/pconv_bf16_force.hip', 'w').write(src)
PY
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-variable -c build_var/pconv_bf16_force.hip -o build_var/pconv_bf16_force.o
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $(ls pytorch-bayesiancnn_amd/build/*.o | grep -v '/pconv_bf16.o') build_var/pconv_bf16_force.o -o build_var/libbbb_force.so
echo "built build_var/libbbb_force.so"
