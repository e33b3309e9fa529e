#!/usr/bin/env bash
# Build libbbb_hip.so for gfx950 (hipcc cross-compiles without a GPU).  Used by __graft_entry__.build().
set -euo pipefail
cd "$(dirname "$0")/pytorch-bayesiancnn_amd"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p build
FLAGS="${BBB_EXTRA_FLAGS:-} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-variable"
objs=()
for f in csrc/*.hip; do
  o=build/$(basename "${f%.hip}").o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ csrc/bbb_common.cuh -nt "$o" ] || [ ../include/bbb_hip.h -nt "$o" ] || [ csrc/pconv_args.h -nt "$o" ]; then
    "$HIPCC" $FLAGS -c "$f" -o "$o" &
  fi
  objs+=("$o")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o bbb_hip/libbbb_hip.so
echo "built $(pwd)/bbb_hip/libbbb_hip.so"
