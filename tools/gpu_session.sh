#!/bin/bash
# One gpurun call = tests + diagnostics + tuning + bench + rocprof (box acquisition dominates the
# charge, so batch everything).  Usage: gpurun -- 'bash tools/gpu_session.sh <tag>'
set -u
TAG=${1:-s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
echo "== pytest" ; timeout -k 10 900 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest exit=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
echo "== diag"; timeout -k 10 300 python tools/diag_llr.py > $OUT/diag_llr.log 2>&1; tail -30 $OUT/diag_llr.log
echo "== tune"; timeout -k 10 600 python tools/tune.py > $OUT/tune.log 2>&1; tail -3 $OUT/tune.log
echo "== tune early exit"; timeout -k 10 300 python tools/tune.py --early-exit --slots 2 4 7 --blocks 1 2 3 --regs 0 2 --batch 200000 > $OUT/tune_early.log 2>&1; tail -2 $OUT/tune_early.log
echo "== bench"; timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
echo "== rocprof"
export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
echo "rocprof exit=$?"; find $OUT/prof -name "*stats*" | head; 
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -8 $f; done
