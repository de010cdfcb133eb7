#!/bin/bash
# PMC evidence for the general-H kernel on the reference's space-time matrices (forced 50), and the
# per-class VALU instruction counters of the headline on-chip kernel.
#   gpurun -- 'bash tools/profile_generic.sh <tag>'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-gen}; mkdir -p $OUT
cd $R; timeout -k 10 300 python tools/bench_generic.py --kernels general > $OUT/generic.json 2> $OUT/generic.err; cat $OUT/generic.json
export TMPDIR=/tmp; cd /tmp
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
G3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
G4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
G5="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum"
for M in "864x2592" "2592x7776"; do
  i=0
  for C in "$G1" "$G2" "$G3" "$G4" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1)); d=$OUT/pmc_${M}_$i
    timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -o pmc -- python3 $R/tools/bench_generic.py --kernels general --only $M --reps 1 > $d.log 2>&1
    echo "pmc $M pass $i exit=$?"
  done
done
# headline kernel: instruction mix by class
i=0
for C in "$G3" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); d=$OUT/pmc_fused_$i
  timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --mode forced > $d.log 2>&1
  echo "pmc fused pass $i exit=$?"
done
cd $R; python - <<PY
import csv, glob, json, collections, os
out = {}
for d in sorted(glob.glob("$OUT/pmc_*/")):
    tag = os.path.basename(d.rstrip("/")).rsplit("_", 1)[0][4:]
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "bp_generic_kernel" in k or "bp_fused_kernel" in k:
                per[(row["Counter_Name"], row["Dispatch_Id"], k[:60])] += float(row["Counter_Value"])
        by = collections.defaultdict(list)
        for (name, _, k), v in per.items():
            by[(name, k)].append(v)
        for (name, k), vs in by.items():
            out.setdefault(tag, {}).setdefault(k, {})[name] = {"last": vs[-1], "mean": sum(vs) / len(vs), "n": len(vs)}
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "bp_generic_kernel" in k or "bp_fused_kernel" in k:
                out.setdefault(tag, {}).setdefault(k[:60], {}).setdefault("durations_ns", []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
