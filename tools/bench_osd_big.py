#!/usr/bin/env python3
"""OSD-0 through the workgroup-per-syndrome kernel (matrix copy in global memory) on matrices beyond the LDS limit:
two copies of the 864 x 2592 space-time matrix of [[144,12,12]] (1728 x 5184, tests/test_gpu_osd.py) and the
2592 x 7776 one of [[288,12,18]]; BP failures of a p = 0.03 batch.  One JSON line (QBP_LIB_PATH-aware)."""
import json
import os
import sys
import time

import numpy as np
from scipy.sparse import block_diag, csr_matrix

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes, mc  # noqa: E402


def space_time(H, T):
    m = H.shape[0]
    return np.hstack([np.kron(np.eye(T, dtype=np.int64), H),
                      (np.eye(m * T, dtype=np.int64) + np.eye(m * T, k=-m, dtype=np.int64)) % 2])


out = {}
st144 = space_time(codes.load_code("[[144, 12, 12]]").Hx, 12)
for name, H in (("1728x5184", block_diag([csr_matrix(st144), csr_matrix(st144)]).tocsr()),
                ("2592x7776", csr_matrix(space_time(codes.load_code("[[288, 12, 18]]").Hx, 18)))):
    dec = _lib.Decoder(*bp.csr_from_H(H), bp.DEVICE)
    m, n = H.shape
    rng = np.random.default_rng(5)
    p = 0.03
    B = 600
    err = (rng.random((B, n)) < p).astype(np.uint8)
    syn = np.asarray((csr_matrix(err.astype(np.int64)) @ H.T.astype(np.int64)).todense() % 2).astype(np.uint8)
    hard, conv, iters, llr = dec.decode(syn, mc.prior_of(p, n), 12)
    f = np.flatnonzero(~conv)[:256]
    kind = int(os.environ.get("QBP_OSD_KIND", "0"))
    if kind:
        dec.set_option(_lib.OPT_OSD_BIG, kind)
    rep = int(os.environ.get("QBP_OSD_REP", "1"))
    f = np.tile(f, rep)
    dec.osd0(syn[f[:2]], llr[f[:2]], hard[f[:2]])          # tables, rank (host, once)
    import torch
    ds, dl, dh = (torch.from_numpy(np.ascontiguousarray(x[f])).cuda() for x in (syn, llr, hard))
    dsol = torch.empty((len(f), n), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    dec.osd0_device(ds.data_ptr(), dl.data_ptr(), dh.data_ptr(), len(f), dsol.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec.osd0_device(ds.data_ptr(), dl.data_ptr(), dh.data_ptr(), len(f), dsol.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    got = dsol.cpu().numpy()
    ok = bool((np.asarray(csr_matrix(got.astype(np.int64)) @ H.T.astype(np.int64).todense() if False else
                          (csr_matrix(got.astype(np.int64)) @ H.T.astype(np.int64)).todense()) % 2 == syn[f]).all())
    out[name] = {"failures": int(len(f)), "seconds": round(dt, 4), "osd_per_s": round(len(f) / dt, 1), "all_valid": ok, "kind": kind,
                 "checksum": int(got.astype(np.int64).sum())}
    dec.close()
print(json.dumps(out))
