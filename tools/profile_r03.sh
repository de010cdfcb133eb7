#!/bin/bash
# Round-3 profiles, one call:   gpurun --timeout 1100 -- 'bash tools/profile_r03.sh r03'
#   1. rocprofv3 --kernel-trace --stats of the bench command (kernel durations to compare with bench.py's HIP events)
#   2. PMC passes over the headline forced-50 kernel (instruction counts by class, LDS, busy cycles; FETCH_SIZE and
#      WRITE_SIZE in passes of their own, as MI355X_MICROARCH.md prescribes)
#   3. PMC passes over the OSD-0 kernel (tools/bench_osd.py) and the Monte-Carlo kernel (tools/bench_mc.py)
# Every rocprofv3 command has the program itself after `--` (python3 ...), and --pmc runs carry no trace domain but
# --kernel-trace.  Summaries: gpurun_out/<tag>/summary_*.json  ->  copy to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
pass() {   # name, counters, program...
  local name=$1 counters=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- "$@" > $OUT/pmc_$name.log 2>&1
  echo "pmc $name exit=$?"
}
if [ "${ONLY:-}" != "legs" ]; then
echo "== kernel trace of the bench command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace_bench.log 2>&1
echo "trace exit=$?"
F="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --mode forced"
pass fused_class "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" $F
pass fused_total "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" $F
pass fused_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" $F
pass fused_clk "GRBM_GUI_ACTIVE" $F
pass fused_fetch "FETCH_SIZE" $F
pass fused_write "WRITE_SIZE" $F
O="python3 $R/tools/bench_osd.py"
pass osd_total "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" $O
pass osd_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" $O
M="python3 $R/tools/bench_mc.py"
pass mc_total "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" $M
pass mc_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" $M
fi
L="python3 $R/tools/bench_legs.py --once"
pass legs_total "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" $L
pass legs_wait "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $L
pass legs_fetch "FETCH_SIZE" $L
pass legs_write "WRITE_SIZE" $L
cd $R
python3 tools/pmc_collect.py $OUT > $OUT/summary_pmc.json
python3 tools/pmc_collect.py $OUT legs > $OUT/summary_legs.json
echo "summary: $OUT/summary_pmc.json"
