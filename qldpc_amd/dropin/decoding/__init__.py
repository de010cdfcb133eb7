"""Drop-in ``decoding`` package: put ``<repo>`` and ``<repo>/qldpc_amd/dropin`` on PYTHONPATH and the
reference's ``main.py`` / ``paperResults.py`` / ``paperResults_GPU.py`` import THIS package for
``decoding.beliefPropagation`` and ``decoding.beliefPropagationGPU`` (a regular package wins
over the reference's namespace directory of the same name).  Sub-modules this build does not
replace (``decoding.beliefPropagationJAX`` ...) still resolve to the reference's own files: its
``decoding/`` directory, when present on sys.path, is appended to this package's search path.

Importing the package itself gives the rework-style names (``from decoding import
performBeliefPropagationFast, performMinSum_Symmetric, ...`` as rework/main.py:5-6 does).
"""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
for _p in list(_sys.path):
    _cand = _os.path.join(_p or _os.getcwd(), "decoding")
    if _os.path.isdir(_cand) and _os.path.abspath(_cand) != _here and _cand not in __path__:
        __path__.append(_cand)

from qldpc_amd.rework import (performBeliefPropagation_Symmetric,  # noqa: E402,F401
                              performBeliefPropagationFast, performMinSum_Symmetric,
                              performOSD_enhanced)
