#!/usr/bin/env bash
# Build libbbb_hip.so for gfx950 (hipcc cross-compiles without a GPU).  Used by __graft_entry__.build().
set -euo pipefail
cd "$(dirname "$0")/pytorch-bayesiancnn_amd"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p build
FLAGS="${BBB_EXTRA_FLAGS:-} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-variable"
objs=()
pids=()
for f in csrc/*.hip; do
  o=build/$(basename "${f%.hip}").o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ csrc/bbb_common.cuh -nt "$o" ] || [ ../include/bbb_hip.h -nt "$o" ] || [ csrc/pconv_args.h -nt "$o" ]; then
    rm -f "$o"                       # a failed compile must not leave a stale object for the link step
    "$HIPCC" $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
  objs+=("$o")
done
for p in "${pids[@]:-}"; do
  [ -n "$p" ] && { wait "$p" || { echo "build.sh: a hipcc job failed" >&2; exit 1; }; }
done
# objects of sources that no longer exist must not be linked
for o in build/*.o; do
  [ -f "csrc/$(basename "${o%.o}").hip" ] || rm -f "$o"
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o bbb_hip/libbbb_hip.so
echo "built $(pwd)/bbb_hip/libbbb_hip.so"
# measurement aid for bench.py only (pure-MFMA-loop ceiling of the box); deliberately NOT linked into the product library
if [ ! -f ../profiles/probe/libmfma_probe.so ] || [ ../profiles/probe/mfma_ceiling.hip -nt ../profiles/probe/libmfma_probe.so ]; then
  "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ../profiles/probe/mfma_ceiling.hip -o ../profiles/probe/libmfma_probe.so
  echo "built $(cd ../profiles/probe && pwd)/libmfma_probe.so"
fi
