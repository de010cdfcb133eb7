#!/bin/bash
# Build experiment variants of libqbp.so (extra -D / compiler flags) into build/variants/ (git-ignored,
# travels with gpurun); tools/ab_run.py then runs one benchmark script per variant on the same box:
#   tools/build_variants.sh "base:" "vu8:-DQBP_STREAM_VU=8"
#   gpurun -- 'python tools/ab_run.py tools/bench_stream.py'
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
