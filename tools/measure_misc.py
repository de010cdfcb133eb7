#!/usr/bin/env python3
"""GPU box: (1) accuracy of the raw v_rcp_f64 seed and of div_nr; (2) general-H kernel throughput;
(3) OSD-0 throughput; (4) Monte-Carlo throughput incl. sampling + classification."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes, mc  # noqa: E402

code = codes.load_code("[[288, 12, 18]]")
dec = bp.decoder_for(code.Hx)
rng = np.random.default_rng(0)
x = np.concatenate([rng.uniform(1, 2, 2_000_000), 10.0 ** rng.uniform(-7, 2, 1_000_000)])
r = dec.debug_math(2, x)
rel = np.abs(r * x - 1.0)
print(f"v_rcp_f64 seed: max |x*rcp(x)-1| = {rel.max():.3e} = 2^{np.log2(rel.max()):.2f}; "
      f"fraction exact RN(1/x): {np.mean(r == 1.0 / x):.4f}")
d = dec.debug_math(3, x)
print(f"div_nr(1, x) == 1/x (IEEE): {np.mean(d == 1.0 / x) * 100:.5f}% of {x.size}")

# general-H kernel on the 288 code, forced 50
p = 0.01
B = 20000
syn = ((rng.random((B, code.n)) < p).astype(np.uint8) @ code.Hx.T % 2).astype(np.uint8)
prior = mc.prior_of(p, code.n)
for force in (0, 1):
    dec.set_option(_lib.OPT_FORCE_GENERIC, force)
    dec.decode(syn[:100], prior, 50, flags=_lib.FLAG_FORCE_FULL)
    t0 = time.perf_counter()
    dec.decode(syn, prior, 50, flags=_lib.FLAG_FORCE_FULL, want_llr=False)
    dt = time.perf_counter() - t0
    print(f"{'general-H' if force else 'fused'} kernel via host API (incl. PCIe): {B / dt:.3e} syndromes/s forced 50")
dec.set_option(_lib.OPT_FORCE_GENERIC, 0)

# OSD-0 throughput
hard, conv, iters, llr = dec.decode(syn, prior, 3)
f = np.flatnonzero(~conv)[:5000]
if len(f) < 1000:
    p2 = 0.08
    syn2 = ((rng.random((5000, code.n)) < p2).astype(np.uint8) @ code.Hx.T % 2).astype(np.uint8)
    hard, conv, iters, llr = dec.decode(syn2, mc.prior_of(p2, code.n), 10)
    f = np.flatnonzero(~conv)
    syn = syn2
dec.osd0(syn[f[:64]], llr[f[:64]], hard[f[:64]])
t0 = time.perf_counter()
dec.osd0(syn[f], llr[f], hard[f])
dt = time.perf_counter() - t0
print(f"OSD-0 via host API: {len(f) / dt:.3e} syndromes/s ({len(f)} BP failures)")

# Monte-Carlo end to end (sampling + decode + classify), early exit
for pp, osd in ((0.01, False), (0.05, False), (0.05, True)):
    T = 1_000_000 if not osd else 1 << 20
    pr = mc.prior_of(pp, code.n)
    dec.mc_run(code.Lx, code.distance, pp, pr, 0, 10000, flags=_lib.FLAG_OSD0 if osd else 0)
    t0 = time.perf_counter()
    c = dec.mc_run(code.Lx, code.distance, pp, pr, 0, T, flags=_lib.FLAG_OSD0 if osd else 0)
    dt = time.perf_counter() - t0
    print(f"qbp_mc_run p={pp} osd={osd}: {T / dt:.3e} trials/s, LER {c[1] / c[0]:.5f}, "
          f"not converged {c[6] / c[0]:.4f}, mean iters {c[7] / c[0] + 1:.2f}")

# space-time matrix of [[144,12,12]] over 12 cycles: (8, 4) on-chip shape vs general-H kernel
from scipy.sparse import csr_matrix
H = codes.load_code("[[144, 12, 12]]").Hx
mm, T = H.shape[0], 12
Hst = np.hstack([np.kron(np.eye(T, dtype=np.int64), H),
                 (np.eye(mm * T, dtype=np.int64) + np.eye(mm * T, k=-mm, dtype=np.int64)) % 2])
dst = bp.decoder_for(csr_matrix(Hst))
err = (rng.random((4000, Hst.shape[1])) < 0.005).astype(np.int64)
sst = (err @ Hst.T % 2).astype(np.uint8)
pst = mc.prior_of(0.005, Hst.shape[1])
for force in (0, 1):
    dst.set_option(_lib.OPT_FORCE_GENERIC, force)
    dst.decode(sst[:64], pst, 50, flags=_lib.FLAG_FORCE_FULL)
    t0 = time.perf_counter()
    dst.decode(sst, pst, 50, flags=_lib.FLAG_FORCE_FULL, want_llr=False)
    dt = time.perf_counter() - t0
    print(f"space-time 864x2592 {'general-H' if force else 'fused (8,4)'} kernel: {len(sst) / dt:.3e} syndromes/s forced 50")

# streaming kernel (lane per syndrome, messages in HBM): device-resident timing on both matrices
import torch
dev = torch.device("cuda", 0)
for label, Hm, pp, Bs in (("[[288,12,18]]", code.Hx, 0.01, 262144), ("space-time 864x2592", Hst, 0.005, 65536)):
    d = bp.decoder_for(csr_matrix(Hm))
    mm2, nn2 = Hm.shape
    g = torch.Generator(device=dev); g.manual_seed(3)
    er = torch.rand((Bs, nn2), generator=g, device=dev) < pp
    sy = (er.float() @ torch.from_numpy(np.asarray(Hm).T.astype(np.float32)).to(dev)).remainder_(2).to(torch.uint8)
    pr = torch.full((nn2,), float(np.log((1 - pp) / pp)), dtype=torch.float64, device=dev)
    hd = torch.empty((Bs, nn2), dtype=torch.uint8, device=dev); cv = torch.empty((Bs,), dtype=torch.uint8, device=dev)
    itr = torch.empty((Bs,), dtype=torch.int32, device=dev); ll = torch.empty((Bs, nn2), dtype=torch.float64, device=dev)
    stq = torch.cuda.current_stream(dev)
    E2 = int(np.asarray(Hm).sum())
    d.set_option(_lib.OPT_FORCE_GENERIC, 0)
    for kern, name in ((_lib.KERNEL_STREAM, "streaming"), (_lib.KERNEL_GENERAL, "general-H"), (_lib.KERNEL_AUTO, "default")):
        d.set_option(_lib.OPT_KERNEL, kern)
        def run():
            d.decode_device(sy.data_ptr(), pr.data_ptr(), Bs, 50, 0, 1.0, 1.0, 20.0, _lib.FLAG_FORCE_FULL,
                            hd.data_ptr(), cv.data_ptr(), itr.data_ptr(), ll.data_ptr(), stq.cuda_stream)
        run(); torch.cuda.synchronize()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b_.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b_)
        alg = Bs * 50 * 4 * E2 * 8
        print(f"{label} {name} kernel (kind {d.info('kernel_kind')}): {Bs / ms * 1e3:.3e} syndromes/s forced 50, "
              f"{ms:.1f} ms, algorithmic message traffic {alg / ms / 1e9:.2f} TB/s")
    d.set_option(_lib.OPT_KERNEL, _lib.KERNEL_AUTO)
