#!/usr/bin/env python3
"""What the reference's own `python main.py` prints (BASELINE config 1 plumbing): tests/golden/
main_py_transcript.txt, produced by RUNNING /root/reference/main.py in the build container -- in a scratch
directory (the script loads `codes/steane.npz` and writes `media/steane_matrix.png` relative to the working
directory; the reference tree itself is read-only), MPLBACKEND=Agg.

tests/test_dropin_gpu.py runs a script written for this build that makes the same five calls through the
reference's import names -- `from decoding.beliefPropagation import performBeliefPropagation`,
`from decoding.OSD import performOSD` -- with the drop-in package on PYTHONPATH, on the GPU, and compares its
standard output with this transcript.

    python tests/golden/make_golden_transcript.py
"""
import os
import subprocess
import sys
import tempfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

with tempfile.TemporaryDirectory() as td:
    os.symlink(os.path.join(REF, "codes"), os.path.join(td, "codes"))
    os.mkdir(os.path.join(td, "media"))
    env = dict(os.environ, MPLBACKEND="Agg", PYTHONPATH=REF)
    r = subprocess.run([sys.executable, os.path.join(REF, "main.py")], cwd=td, env=env, capture_output=True, text=True,
                       timeout=600)
    if r.returncode != 0:
        sys.exit(r.stderr)
    assert os.path.exists(os.path.join(td, "media", "steane_matrix.png"))
out = os.path.join(HERE, "main_py_transcript.txt")
open(out, "w").write(r.stdout)
print(r.stdout, end="")
print("->", out)
