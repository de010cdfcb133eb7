"""GPU: the drop-in package under the reference's import names, end to end in a fresh interpreter."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(script, *args):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "qldpc_amd", "dropin")]))
    return subprocess.run([sys.executable, os.path.join(HERE, "scripts", script), *args], capture_output=True,
                          text=True, timeout=600, env=env, cwd=HERE)


def test_main_py_calls_through_the_reference_import_names():
    """`from decoding.beliefPropagation import performBeliefPropagation` / `from decoding.OSD import performOSD`
    resolve to the drop-in package; the script's output equals, character for character, what the reference's
    own main.py printed in the build container (tests/golden/main_py_transcript.txt, made by
    make_golden_transcript.py) -- including the two lines performBeliefPropagation prints by default."""
    r = _run("steane_smoke.py")
    assert r.returncode == 0, r.stderr[-2000:]
    want = open(os.path.join(HERE, "golden", "main_py_transcript.txt")).read()
    assert r.stdout == want, (r.stdout, want)


def test_batch_driver_loop_through_the_reference_import_names():
    """The import line and call pattern of paperResults_GPU.py:18-22, :95-123 on a small batch."""
    r = _run("batch_driver_smoke.py")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "GPU_AVAILABLE True" in r.stdout and "batch ok" in r.stdout, r.stdout
