#!/bin/bash
# Build experiment variants of libqbp.so (extra -D flags) into build/variants/ (git-ignored,
# travels with gpurun); tools/ab_run.py then runs one benchmark script per variant on the same box:
#   tools/build_variants.sh "base:" "jg2:-DQBP_WIDE_JG=2"
#   gpurun -- 'python tools/ab_run.py tools/ab_wide_mc.py'
set -eu
cd "$(dirname "$0")/.."
mkdir -p build/variants
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  make -s -C qldpc_amd/csrc EXTRA="$defs" OBJ=../../build/variants/obj_$name OUT=../../build/variants/libqbp_$name.so ../../build/variants/libqbp_$name.so \
    && echo "built $name ($defs)"
  rm -rf build/variants/obj_$name
done
