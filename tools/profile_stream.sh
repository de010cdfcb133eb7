#!/bin/bash
# PMC evidence for the streaming kernel: physical HBM bytes per launch (FETCH_SIZE, WRITE_SIZE).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-stream}; mkdir -p $OUT
cd $R; python tools/bench_stream.py > $OUT/stream.json 2> $OUT/stream.err; cat $OUT/stream.json
export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" GRBM_GUI_ACTIVE; do
  d=$OUT/pmc_$(echo $C | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -o pmc -- python3 $R/tools/bench_stream.py --steps 2 > $d.log 2>&1
  echo "pmc $C exit=$?"
done
cd $R; python - <<PY
import csv, glob, json, collections
out = {}
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        if "bp_stream_kernel" in row["Kernel_Name"]:
            per[(row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
    by = collections.defaultdict(list)
    for (name, _), v in per.items():
        by[name].append(v)
    for name, vs in by.items():
        out[name] = sum(vs) / len(vs)
b = json.load(open("$OUT/stream.json"))
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    rd, wr = out["FETCH_SIZE"] * 1024 * 2, out["WRITE_SIZE"] * 1024   # FETCH_SIZE x2: gfx950 correction
    out["hbm_read_bytes_x2"] = rd; out["hbm_write_bytes"] = wr
    out["physical_GBps"] = (rd + wr) / b["kernel_ms"] / 1e6
    out["physical_frac_of_8TBps"] = out["physical_GBps"] / 8000
    out["physical_over_algorithmic"] = (rd + wr) / b["algorithmic_bytes_per_launch"]
out["bench"] = b
json.dump(out, open("$OUT/stream_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
