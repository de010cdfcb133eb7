#!/usr/bin/env python3
"""GPU side of tools/stream_variants.sh: for every build/variants/libqbp_*.so run the streaming
kernel on the headline workload in a child process (QBP_LIB_PATH), check its outputs against the
default library's on-chip kernel on a small batch, and print one line per variant."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, %r)
from qldpc_amd import _lib, bp, codes
code = codes.load_code("[[288, 12, 18]]")
dec = bp.decoder_for(code.Hx)
rng = np.random.default_rng(5)
syn = ((rng.random((4096, code.n)) < 0.05).astype(np.int64) @ code.Hx.T %% 2).astype(np.uint8)
prior = np.full(code.n, np.log(0.95 / 0.05))
ref = dec.decode(syn, prior, 50)
dec.set_option(_lib.OPT_KERNEL, 3)
got = dec.decode(syn, prior, 50)
same = all(np.array_equal(a, b) for a, b in zip(ref, got))
dec.set_option(_lib.OPT_KERNEL, 0)
''' % ROOT


def main():
    libs = sorted(glob.glob(os.path.join(ROOT, "build", "variants", "libqbp_*.so")))
    batch = sys.argv[1] if len(sys.argv) > 1 else "262144"
    for lib in libs:
        name = os.path.basename(lib)[7:-3]
        env = dict(os.environ, QBP_LIB_PATH=lib)
        same = "n/a"
        if "skip" not in name:
            r = subprocess.run([sys.executable, "-c", CHILD + "print(same)"], env=env, capture_output=True,
                               text=True, timeout=600)
            same = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "ERR " + r.stderr[-300:]
        line = f"{name:28s} bit-identical={same:5s}"
        for p in ("0.01", "0.08"):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_stream.py"), "--batch", batch,
                                "--steps", "3", "--p", p], env=env, capture_output=True, text=True, timeout=900)
            try:
                j = json.loads(r.stdout.strip().splitlines()[-1])
                line += (f" | p={p}: {j['syndromes_per_s']:.3e} syn/s {j['kernel_ms']:.1f} ms "
                         f"{j['algorithmic_GBps'] / 1000:.2f} TB/s")
            except Exception:
                line += f" | p={p}: FAILED {r.stderr[-300:]}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
