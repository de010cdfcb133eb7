#!/bin/bash
# PMC passes over the headline (forced-mode on-chip) kernel: instruction counts by class, totals, busy
# cycles, HBM bytes (separate passes for FETCH_SIZE / WRITE_SIZE).   gpurun -- 'bash tools/profile_fused.sh <tag>'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-fused}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
G3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
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
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "bp_fused_kernel" in k:
                per[(row["Counter_Name"], row["Dispatch_Id"], k[:80])] += float(row["Counter_Value"])
        by = collections.defaultdict(list)
        for (name, _, k), v in per.items():
            by[(name, k)].append(v)
        for (name, k), vs in by.items():
            out.setdefault(k, {})[name] = {"last": vs[-1], "mean": sum(vs) / len(vs), "n": len(vs)}
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "bp_fused_kernel" in k:
                out.setdefault(k[:80], {}).setdefault("durations_ns", []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for k, v in out.items():
    if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v:
        # one launch: 125000 syndromes x 50 iterations, 7 slots per 16-wave workgroup
        wave_iters = 125000 * 50 / 7 * 16
        v["derived"] = {"valu_per_wave_iteration": v["SQ_INSTS_VALU"]["last"] / wave_iters,
                        "by_class_per_wave_iteration": {c: v[c]["last"] / wave_iters for c in v if c.startswith("SQ_INSTS_VALU_")},
                        "lds_per_wave_iteration": v["SQ_INSTS_LDS"]["last"] / wave_iters,
                        "salu_per_wave_iteration": v["SQ_INSTS_SALU"]["last"] / wave_iters}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps({k: v.get("derived") for k, v in out.items()}, indent=1))
PY
