"""Replaces the reference's decoding/beliefPropagationGPU.py (same names and signatures)."""
from qldpc_amd.bp import (generate_errors_and_syndromes_batch, gpu_available,  # noqa: F401
                          performBeliefPropagationBatch, performBeliefPropagationGPU)

GPU_AVAILABLE = gpu_available()
# the reference prints a one-line banner at import (beliefPropagationGPU.py:12/16)
print("MI355X HIP decoder (libqbp) - GPU acceleration enabled" if GPU_AVAILABLE
      else "libqbp found no GPU - decoding calls will raise (no CPU fallback)")
