"""The import line and per-batch call pattern of the reference's batch driver (paperResults_GPU.py:18-22 and
:95-123), written for this build's test-suite: sample two error batches with the caller's numpy Generator,
XOR them, decode the batch, run OSD on every sample BP left unconverged; checked against the syndromes."""
import numpy as np

from decoding.beliefPropagationGPU import (GPU_AVAILABLE, generate_errors_and_syndromes_batch,
                                           performBeliefPropagationBatch)
from decoding.OSD import performOSD

from qldpc_amd import codes

print("GPU_AVAILABLE", GPU_AVAILABLE)
code = codes.load_code("[[72, 12, 6]]").Hx
rng = np.random.default_rng(0)
p, B = 0.04, 400
prior = np.array([np.log((1 - p) / p)] * code.shape[1])
e1, s1 = generate_errors_and_syndromes_batch(code, p, B, rng)
e2, s2 = generate_errors_and_syndromes_batch(code, p, B, rng)
errors, syndromes = (e1 + e2) % 2, (s1 + s2) % 2
detections, converged, llrs = performBeliefPropagationBatch(code, syndromes, prior, maxIter=150)
assert detections.dtype == np.int8 and converged.dtype == bool and llrs.shape == (B, code.shape[1])
n_osd = 0
for i in range(B):
    detection = detections[i]
    if not converged[i]:
        detection = performOSD(code, syndromes[i], llrs[i], detection)
        n_osd += 1
    assert np.array_equal((detection @ code.T) % 2, syndromes[i])
assert 0 < n_osd < B
print("batch ok", B, "samples,", n_osd, "through OSD")
