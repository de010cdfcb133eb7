#!/usr/bin/env python3
"""Run a benchmark script once per library in build/variants/ (QBP_LIB_PATH), on the same box:
    python tools/ab_run.py tools/ab_early.py [args...]"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sorted(glob.glob(os.path.join(ROOT, "build", "variants", "libqbp_*.so"))):
    name = os.path.basename(lib)[7:-3]
    r = subprocess.run([sys.executable] + sys.argv[1:], env=dict(os.environ, QBP_LIB_PATH=lib),
                       capture_output=True, text=True, timeout=900)
    last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-400:]
    print(f"{name:20s} {last}", flush=True)
