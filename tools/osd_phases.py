#!/usr/bin/env python3
"""Where a workgroup of osd0_blocked_kernel spends its cycles (a library built with -DQBP_OSD_TIMING:
tools/build_variants.sh "osdt:-DQBP_OSD_TIMING"; QBP_LIB_PATH=build/variants/libqbp_osdt.so)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch
from scipy.sparse import block_diag, csr_matrix

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes, mc  # noqa: E402

NAMES = ["sort", "build", "bytes", "columns", "pivot_rows", "table", "update", "finish"]


def space_time(H, T):
    m = H.shape[0]
    return np.hstack([np.kron(np.eye(T, dtype=np.int64), H),
                      (np.eye(m * T, dtype=np.int64) + np.eye(m * T, k=-m, dtype=np.int64)) % 2])


L = _lib.load()
L.qbp_debug_osd_timing.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
st144 = space_time(codes.load_code("[[144, 12, 12]]").Hx, 12)
for name, H in (("1728x5184", block_diag([csr_matrix(st144), csr_matrix(st144)]).tocsr()),
                ("2592x7776", csr_matrix(space_time(codes.load_code("[[288, 12, 18]]").Hx, 18)))):
    dec = _lib.Decoder(*bp.csr_from_H(H), bp.DEVICE)
    m, n = H.shape
    rng = np.random.default_rng(5)
    p = 0.03
    err = (rng.random((600, n)) < p).astype(np.uint8)
    syn = np.asarray((csr_matrix(err.astype(np.int64)) @ H.T.astype(np.int64)).todense() % 2).astype(np.uint8)
    hard, conv, iters, llr = dec.decode(syn, mc.prior_of(p, n), 12)
    f = np.tile(np.flatnonzero(~conv)[:256], 4)
    dec.osd0(syn[f[:2]], llr[f[:2]], hard[f[:2]])
    buf = (ctypes.c_ulonglong * 16)()
    L.qbp_debug_osd_timing(buf, 1)
    dec.osd0(syn[f], llr[f], hard[f])
    L.qbp_debug_osd_timing(buf, 1)
    t = np.array(list(buf)[:8], dtype=np.float64) / len(f)
    st = np.array(list(buf)[8:], dtype=np.float64) / len(f)
    print(json.dumps({"matrix": name, "cycles_per_syndrome": {k: int(v) for k, v in zip(NAMES, t)},
                      "total": int(t.sum()), "blocks_with_pivots": float(st[0]), "active_rows_per_block": float(st[1] / max(st[0], 1)),
                      "words_per_active_row": float(st[2] / max(st[1], 1)), "mean_k0_at_end_of_sweeps": float(st[4]), "mean_rank_at_end": float(st[5]), "sweeps_ended_early": float(st[6]), "sweeps": float(st[7]), "max_k0_at_end": float(st[3] * len(f)), "share": {k: round(float(v / t.sum()), 3) for k, v in zip(NAMES, t)}}))
    dec.close()
