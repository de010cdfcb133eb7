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
echo "== tune early exit"; timeout -k 10 300 python tools/tune.py --early-exit --slots 2 4 7 --blocks 1 2 --batch 200000 > $OUT/tune_early.log 2>&1; tail -2 $OUT/tune_early.log
echo "== misc"; timeout -k 10 600 python tools/measure_misc.py > $OUT/misc.log 2>&1; cat $OUT/misc.log
echo "== kernels"; (timeout -k 10 300 python tools/bench_generic.py; timeout -k 10 300 python tools/bench_osd.py; timeout -k 10 300 python tools/ab_early.py; timeout -k 10 300 python tools/measure_latency.py) > $OUT/kernels.log 2>&1; grep -v amdgpu.ids $OUT/kernels.log
echo "== other configs"; timeout -k 10 300 python tools/bench_configs.py > $OUT/configs.log 2>&1; cat $OUT/configs.log
echo "== paper_results smoke"; timeout -k 10 300 python -m qldpc_amd.paper_results --codes 72 288 --p 0.05 0.02 --trials 50000 --max-iter 50 --out $OUT/paper_smoke > $OUT/paper_smoke.log 2>&1; tail -12 $OUT/paper_smoke.log
echo "== bench"; timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
echo "== 2-rank rehearsal of the distributed path (gloo, both ranks on cuda:0)"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --share-device --batch 20000 > $OUT/bench_2rank_rehearsal.json 2> $OUT/bench_2rank.err; echo "rehearsal exit=$?"; tail -c 600 $OUT/bench_2rank_rehearsal.json; tail -3 $OUT/bench_2rank.err
echo "== rocprof"
export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
echo "rocprof exit=$?"; find $OUT/prof -name "*stats*" | head; 
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -8 $f; done

echo "== pmc"
PMC1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
PMC2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
i=0
for C in "$PMC1" "$PMC2" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --mode forced > $OUT/pmc_$i.log 2>&1
  echo "pmc pass $i ($C) exit=$?"
done
cd $R; python tools/pmc_summary.py $OUT/pmc_* > $OUT/pmc_summary.json; cat $OUT/pmc_summary.json | head -60
