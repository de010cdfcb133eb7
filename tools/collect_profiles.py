#!/usr/bin/env python3
"""Copy the judged artefacts of one tools/gpu_session.sh run (gpurun_out/<tag>/) into profiles/.

    python tools/collect_profiles.py s9 [round-prefix, default r01]
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
pre = sys.argv[2] if len(sys.argv) > 2 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
for a, b in (("bench.json", "bench_1gpu.json"), ("pytest.log", "pytest_gpu.log"), ("tune.log", "tune.txt"),
             ("tune_early.log", "tune_early_exit.txt"), ("misc.log", "misc_measurements.txt"),
             ("configs.log", "other_configs.txt"), ("diag_llr.log", "llr_drift_vs_oracle.txt"),
             ("bench_2rank_rehearsal.json", "bench_2rank_gloo_rehearsal.json"),
             ("kernels.log", "other_kernels.txt")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, f"{pre}_{b}"))
rows = list(csv.reader(open(os.path.join(src, "prof", "trace_kernel_stats.csv"))))
with open(os.path.join(dst, f"{pre}_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:160]          # torch helper kernels have page-long names
        w.writerow(r)
d = json.load(open(os.path.join(src, "pmc_summary.json")))
b = json.load(open(os.path.join(src, "bench.json")))
ms = b["roofline"]["kernel_ms"]
B, it = b["config"]["syndromes_per_gpu_per_step"], b["config"]["max_iter"]
g = d["GRBM_GUI_ACTIVE"]["mean_per_dispatch"] / 8
d["derived"] = {
    "command": "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline --mode forced (one pass per counter group, tools/gpu_session.sh)",
    "workload": f"{B} syndromes of [[288,12,18]], {it} iterations each, one launch of {b['roofline']['kernel']}",
    "kernel_ms_unprofiled": ms,
    "shader_clock_GHz": g / (ms * 1e-3) / 1e9,
    "valu_busy_fraction": d["SQ_ACTIVE_INST_VALU"]["mean_per_dispatch"] * 4 / (1024 * g),
    "valu_insts_per_syndrome_iteration": d["SQ_INSTS_VALU"]["mean_per_dispatch"] / (B * it),
    "cycles_per_valu_inst": d["SQ_ACTIVE_INST_VALU"]["mean_per_dispatch"] * 4 / d["SQ_INSTS_VALU"]["mean_per_dispatch"],
    "lds_conflict_fraction": d["SQ_LDS_BANK_CONFLICT"]["mean_per_dispatch"] / d["SQ_LDS_IDX_ACTIVE"]["mean_per_dispatch"],
    "hbm_bytes_per_launch": d["hbm_read_bytes_x2_gfx950"] + d["hbm_write_bytes"],
    "algorithmic_io_bytes_per_launch": B * (144 + 288 + 8 * 288 + 5),
    "algorithmic_message_bytes_per_launch": B * it * 4 * 864 * 8,
    "notes": "SQ_ACTIVE_INST_*/SQ_WAVE_CYCLES count quad-cycles (x4); GRBM_GUI_ACTIVE is summed over the 8 "
             "XCDs (/8); FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (wide coalesced reads count "
             "64 B per 128-B request).",
}
json.dump(d, open(os.path.join(dst, f"{pre}_pmc_summary.json"), "w"), indent=1)
json.dump({"bytes_per_launch_forced50": d["derived"]["hbm_bytes_per_launch"],
           "source": f"profiles/{pre}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                     "FETCH_SIZE x2 gfx950 correction)", "workload": d["derived"]["workload"]},
          open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in d["derived"].items() if k not in ("command", "notes")}, indent=1))
print("value", b["value"], "frac", b["roofline"]["frac"], "early", b["early_exit"]["value"], "cpu", b.get("cpu_baseline", {}).get("value"))
