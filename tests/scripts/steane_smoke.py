"""The five calls of the reference's smoke script (main.py:14-32: load the Steane matrix, inject errors on
qubits 0 and 1, form the syndrome, belief propagation with its default arguments, OSD on the result), written
for this build's test-suite and made through the REFERENCE'S import names: run with
PYTHONPATH=<repo>:<repo>/qldpc_amd/dropin (tests/test_dropin_gpu.py), `decoding` resolves to the drop-in
package and the work happens on the GPU.  The matrix comes from qldpc_amd.codes (the reference's file cannot
travel); the plot of main.py:16 is not part of the decoding path and is left out."""
import numpy as np

from decoding.beliefPropagation import performBeliefPropagation
from decoding.OSD import performOSD

from qldpc_amd import codes

H = codes.load_code("steane").Hx
p = 0.1
initialBelief = [np.log((1 - p) / p)] * len(H[0])
error = np.zeros(len(H[0]), dtype=int)
error[0] = 1
error[1] = 1
print(f"Error introduced: {error}")
syndrome = (error @ H.T) % 2
detection, isSyndromeFound, llrs = performBeliefPropagation(H, syndrome, initialBelief)
solution = performOSD(H, syndrome, llrs, detection)
print(solution)
