#!/bin/bash
# Round-2 GPU session: tests, general-H kernel rates, the bench line.  Usage: gpurun -- 'bash tools/gpu_session2.sh <tag> [steps...]'
set -u
TAG=${1:-t}; shift || true
STEPS=${*:-"pytest generic bench"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for S in $STEPS; do
case $S in
  pytest) echo "== pytest"; timeout -k 10 1000 python -m pytest tests -m gpu -q -s -x > $OUT/pytest.log 2>&1; echo "pytest exit=$?" | tee -a $OUT/pytest.log; tail -n 15 $OUT/pytest.log ;;
  pytestall) echo "== pytest (no -x)"; timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest exit=$?" | tee -a $OUT/pytest.log; tail -n 25 $OUT/pytest.log ;;
  generic) echo "== generic"; timeout -k 10 300 python tools/bench_generic.py --kernels onchip general > $OUT/generic.json 2> $OUT/generic.err; cat $OUT/generic.json; tail -n 3 $OUT/generic.err ;;
  bench) echo "== bench"; timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit=$?"; cat $OUT/bench.json; tail -n 5 $OUT/bench.err ;;
  kernels) echo "== kernels"; (timeout -k 10 300 python tools/bench_osd.py; timeout -k 10 300 python tools/ab_early.py; timeout -k 10 300 python tools/measure_latency.py; timeout -k 10 300 python tools/bench_configs.py) > $OUT/kernels.log 2>&1; grep -v amdgpu.ids $OUT/kernels.log ;;
  fuzzmc) echo "== fuzz MC 16000"; QBP_FUZZ_CASES=16000 QBP_FUZZ_SEED=99991 timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -q -s -k "monte or mc" > $OUT/fuzz_mc.log 2>&1; echo "fuzz exit=$?"; tail -n 8 $OUT/fuzz_mc.log ;;
  fuzz) echo "== fuzz 3000"; QBP_FUZZ_CASES=3000 QBP_FUZZ_SEED=${FUZZ_SEED:-4242} timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -q -s > $OUT/fuzz.log 2>&1; echo "fuzz exit=$?"; tail -n 8 $OUT/fuzz.log ;;
  rehearsal) echo "== python bench.py --gpus 2 (self-launch; two gloo ranks on this GPU)"
    timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --share-device --batch 20000 --mode forced --no-cpu-baseline > $OUT/bench_2rank_rehearsal.json 2> $OUT/bench_2rank.err; echo "rehearsal exit=$?"; grep "^{\"metric" $OUT/bench_2rank_rehearsal.json > $OUT/tmp.json; mv $OUT/tmp.json $OUT/bench_2rank_rehearsal.json; tail -c 700 $OUT/bench_2rank_rehearsal.json; echo ;;
  config5) echo "== config 5"; timeout -k 10 300 python -m qldpc_amd.mc --code 288 --p 0.1 0.06 0.05 0.04 0.03 0.02 0.01 0.009 0.006 0.005 0.004 0.003 0.002 0.001 --trials 1000000 --osd --out $OUT/config5_288_bposd_1M.json > $OUT/config5.log 2>&1; tail -n 16 $OUT/config5.log ;;
  prof) echo "== rocprof kernel trace of the bench"
    export TMPDIR=/tmp; cd /tmp
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --mode forced > $OUT/prof_bench.log 2>&1
    echo "rocprof exit=$?"; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -n 8 $f; done; cd $R ;;
esac
done
