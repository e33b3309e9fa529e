#!/usr/bin/env bash
# Build libbbb_hip.so for gfx950 (hipcc cross-compiles without a GPU).  Used by __graft_entry__.build().
set -euo pipefail
cd "$(dirname "$0")/pytorch-bayesiancnn_amd"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p build
FLAGS="${BBB_EXTRA_FLAGS:-} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-variable"
objs=()
pids=()
# An object is rebuilt when its source or ANY file it included last time is newer (hipcc -MD writes build/<name>.d: the
# dependency list of the compile that produced the object -- pconv_gemm.o depends on pconv_body.cuh and pconv_bf16x3.cuh, which a
# fixed list of headers missed), when it has no dependency file yet, or when a listed file has disappeared.
stale() {  # $1 = object, $2 = source
  local o=$1 f=$2 d=${1%.o}.d dep
  [ -f "$o" ] && [ -f "$d" ] || return 0
  [ "$f" -nt "$o" ] && return 0
  for dep in $(sed -e 's/^[^:]*://' -e 's/\\$//' "$d"); do
    [ -e "$dep" ] || return 0
    [ "$dep" -nt "$o" ] && return 0
  done
  return 1
}
for f in csrc/*.hip; do
  o=build/$(basename "${f%.hip}").o
  if stale "$o" "$f"; then
    if [ -n "${BBB_BUILD_DRY_RUN:-}" ]; then echo "stale: $o"; continue; fi     # tests/test_host_cpu.py: what WOULD be rebuilt
    rm -f "$o" "${o%.o}.d"           # a failed compile must not leave a stale object for the link step
    "$HIPCC" $FLAGS -MMD -MF "${o%.o}.d" -c "$f" -o "$o" &
    pids+=($!)
  fi
  objs+=("$o")
done
[ -n "${BBB_BUILD_DRY_RUN:-}" ] && exit 0
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
