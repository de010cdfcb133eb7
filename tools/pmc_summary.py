#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs of the bench run (one directory per pass) into one JSON.

    python tools/pmc_summary.py gpurun_out/<tag>/pmc_* > profiles/r01_pmc_summary.json

Per counter: mean over the dispatches of qbp::bp_fused_kernel.  FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section) -- both the raw and the doubled figure are kept."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(list)
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if "bp_fused_kernel" not in row.get("Kernel_Name", ""):
                        continue
                    acc[row["Counter_Name"]].append((row.get("Dispatch_Id"), float(row["Counter_Value"])))
    out = {}
    for name, vals in acc.items():
        # a counter row may be split per dimension (XCC / SE): sum rows of one dispatch
        per = defaultdict(float)
        for did, v in vals:
            per[did] += v
        xs = list(per.values())
        out[name] = {"mean_per_dispatch": sum(xs) / len(xs), "dispatches": len(xs)}
    if "FETCH_SIZE" in out:
        out["hbm_read_bytes_raw"] = out["FETCH_SIZE"]["mean_per_dispatch"] * 1024
        out["hbm_read_bytes_x2_gfx950"] = out["hbm_read_bytes_raw"] * 2
    if "WRITE_SIZE" in out:
        out["hbm_write_bytes"] = out["WRITE_SIZE"]["mean_per_dispatch"] * 1024
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
