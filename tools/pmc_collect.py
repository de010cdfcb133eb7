#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/profile_r03.sh into one JSON (stdout).

Per kernel (name prefix) and counter: the value of the LAST dispatch of that kernel in the pass and the mean over its
dispatches (a counter row may be split per XCC / SE: rows of one dispatch are summed); kernel durations from the
kernel traces.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced stream (MI355X_MICROARCH.md, HBM section): `hbm` holds the raw figure, the corrected one (x 2) and the
write bytes -- this kernel's reads are the 144-byte syndromes (narrow), so the truth lies between the two read
figures; its traffic is write-dominated either way."""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = ("bp_fused_kernel", "osd0_kernel", "osd0_big_kernel", "bp_generic_kernel", "bp_stream_kernel")


def main():
    root = sys.argv[1]
    out = {}
    for d in sorted(glob.glob(os.path.join(root, "pmc_*/")) + glob.glob(os.path.join(root, "trace/"))):
        tag = os.path.basename(d.rstrip("/"))
        for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
            per = collections.defaultdict(float)
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if any(x in k for x in KERNELS):
                    per[(row["Counter_Name"], int(row["Dispatch_Id"]), k[:96])] += float(row["Counter_Value"])
            by = collections.defaultdict(list)
            for (name, did, k), v in sorted(per.items(), key=lambda t: t[0][1]):
                by[(name, k)].append(v)
            for (name, k), vs in by.items():
                out.setdefault(k, {}).setdefault("counters", {})[name] = {"last": vs[-1], "mean": sum(vs) / len(vs),
                                                                            "dispatches": len(vs), "pass": tag}
        for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if any(x in k for x in KERNELS):
                    out.setdefault(k[:96], {}).setdefault("durations_ns", {}).setdefault(tag, []).append(
                        int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in out.items():
        c = v.get("counters", {})
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            rd = c.get("FETCH_SIZE", {}).get("last", 0.0) * 1024
            wr = c.get("WRITE_SIZE", {}).get("last", 0.0) * 1024
            v["hbm"] = {"read_bytes_raw": rd, "read_bytes_x2_gfx950": 2 * rd, "write_bytes": wr,
                        "traffic_bytes_per_launch": 2 * rd + wr}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
