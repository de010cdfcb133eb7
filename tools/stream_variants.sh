#!/bin/bash
# Build experiment variants of the streaming kernel (compile-time switches of qbp_stream.hpp) into
# build/variants/ (git-ignored, travels with gpurun).  Usage: tools/stream_variants.sh name:flags ...
set -eu
cd "$(dirname "$0")/.."
mkdir -p build/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result"
pids=()
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -shared -o build/variants/libqbp_$name.so qldpc_amd/csrc/qbp.hip && echo "built $name ($defs)" ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
