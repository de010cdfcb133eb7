#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel of libqbp.so, from the compiler's
kernel-resource-usage remarks (`make -C qldpc_amd/csrc resources` runs this)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qldpc_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math".split()
UNITS = [("qbp_tu_fused.hip", []), ("qbp_tu_generic.hip", ["-DQBP_GENERIC_MEM=0"]),
         ("qbp_tu_generic.hip", ["-DQBP_GENERIC_MEM=1"]), ("qbp_tu_generic.hip", ["-DQBP_GENERIC_MEM=2"]),
         ("qbp_tu_stream.hip", []), ("qbp_tu_osd.hip", [])]


def demangle(sym):
    try:
        return subprocess.check_output(["c++filt", sym], text=True).strip()
    except Exception:
        return sym


rows = []
for src, extra in UNITS:
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + sys.argv[1:] +
                         ["--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src],
                         cwd=CSRC, capture_output=True, text=True).stderr
    cur = None
    for ln in out.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = {"name": demangle(m.group(1))}
            rows.append(cur)
            continue
        for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, ln)
            if m and cur is not None:
                cur[key] = int(m.group(1))
print(f"{'kernel':78s} {'SGPR':>5s} {'VGPR':>5s} {'scratch B/lane':>15s} {'waves/SIMD':>11s}")
for r in rows:
    name = re.sub(r"\(qbp::\w+(, qbp::\w+)?\)$|^void ", "", r["name"])
    print(f"{name[:78]:78s} {r.get('sgpr', 0):5d} {r.get('vgpr', 0):5d} {r.get('scratch', 0):15d} {r.get('occ', 0):11d}")
