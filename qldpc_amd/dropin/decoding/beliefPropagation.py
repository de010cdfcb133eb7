"""Replaces the reference's decoding/beliefPropagation.py (same names and signatures)."""
from qldpc_amd.bp import performBeliefPropagation, performBeliefPropagationFast  # noqa: F401
