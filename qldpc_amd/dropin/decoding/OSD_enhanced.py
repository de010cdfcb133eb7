"""Replaces the reference's decoding/OSD_enhanced.py (see qldpc_amd/osd.py for the order semantics)."""
from qldpc_amd.osd import performOSD_enhanced  # noqa: F401
