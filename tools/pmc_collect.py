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


def legs_summary(root):
    """Passes pmc_legs_*: `tools/bench_legs.py --once` launches the kernels of its legs once each, in the order of
    bench_legs.ORDER -- the k-th dispatch of a qbp kernel belongs to the k-th slot."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    order = None
    for ln in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_legs.py")):
        if ln.startswith("ORDER = "):
            src = ln
            break
    # (ORDER is a two-line literal: read it without importing torch)
    txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_legs.py")).read()
    lit = txt[txt.index("ORDER = ") + 8:]
    lit = lit[:lit.index("]\n") + 1]
    order = eval(lit)
    slots = [name for name, k in order for _ in range(k)]
    out = {name: [] for name, _ in order}
    for d in sorted(glob.glob(os.path.join(root, "pmc_legs_*/"))):
        tag = os.path.basename(d.rstrip("/"))
        per = collections.defaultdict(dict)
        names = {}
        for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "qbp::" in k and "permute_prior" not in k:
                    did = int(row["Dispatch_Id"])
                    names[did] = k[:96]
                    per[did][row["Counter_Name"]] = per[did].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        dur = {}
        for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "qbp::" in row["Kernel_Name"] and "permute_prior" not in row["Kernel_Name"]:
                    dur[int(row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        dids = sorted(per)
        if len(dids) != len(slots):
            raise SystemExit(f"{tag}: {len(dids)} qbp dispatches, expected {len(slots)}")
        seen = collections.defaultdict(int)
        for did, leg in zip(dids, slots):
            i = seen[leg]
            seen[leg] += 1
            while len(out[leg]) <= i:
                out[leg].append({"kernel": names[did], "counters": {}, "duration_ns": {}})
            assert out[leg][i]["kernel"] == names[did], (leg, out[leg][i]["kernel"], names[did])
            out[leg][i]["counters"].update(per[did])
            if did in dur:
                out[leg][i]["duration_ns"][tag] = dur[did]
    return out


def main():
    root = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == "legs":
        json.dump(legs_summary(root), sys.stdout, indent=1)
        print()
        return
    out = {}
    for d in sorted(glob.glob(os.path.join(root, "pmc_*/")) + glob.glob(os.path.join(root, "trace/"))):
        tag = os.path.basename(d.rstrip("/"))
        if tag.startswith("pmc_legs_"):
            continue
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
