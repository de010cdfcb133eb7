"""Replaces the reference's decoding/OSD.py (performOSD: OSD-0) with the GPU implementation."""
from qldpc_amd.osd import performOSD  # noqa: F401
