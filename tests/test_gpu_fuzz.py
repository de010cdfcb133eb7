"""GPU: randomised cross-check of every device kernel against the CPU oracle and against each other.

Random matrices (shapes the on-chip kernel takes, shapes only the general-H / streaming kernels
take, empty rows, isolated variables, wide rows and columns), random priors (some of them 0, +-inf
or negative), variants, iteration
limits, batch sizes and flags.  Bars:

* the device kernels (automatic choice, general-H, streaming) return identical bits on EVERY input;
* device == oracle in EVERY output bit (hard decision, converged flag, iteration, posterior LLRs) on EVERY
  syndrome, in both column-sum orders (round 3: oracle and device run numpy's own tanh / arctanh kernels;
  rounds 1-2 had to set aside exact cancellations, whose rounding residue the reference scales by 1e15 at
  beliefPropagation.py:122-123, and trajectories that run long without converging);
* OSD-0 and the Monte-Carlo counters: identical to the oracle pipeline, no exceptions.

The default run is short (part of ``pytest -m gpu``); ``QBP_FUZZ_CASES=2000 python -m pytest
tests/test_gpu_fuzz.py -s`` is the long campaign (profiles/r01_fuzz.txt holds one).
"""
import os

import numpy as np
import pytest

from oracle import oracle
from qldpc_amd import _lib, bp

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("QBP_FUZZ_CASES", "60"))


def capped_matrix(rng, m, n, max_row, max_col, fill):
    """Random H with row weights <= max_row and column weights <= max_col."""
    H = np.zeros((m, n), np.int64)
    colw = np.zeros(n, np.int64)
    for c in range(m):
        want = int(rng.integers(0 if rng.random() < 0.03 else 1, max_row + 1))
        free = np.flatnonzero(colw < max_col)
        if len(free) == 0 or want == 0:
            continue
        pick = rng.choice(free, min(want, len(free)), replace=False)
        if rng.random() < fill:
            H[c, pick] = 1
            colw[pick] += 1
        else:
            H[c, pick[:max(1, len(pick) // 2)]] = 1
            colw[pick[:max(1, len(pick) // 2)]] += 1
    return H


def random_matrix(rng):
    kind = rng.choice(["cap63", "cap63", "cap84", "wide", "tiny"])
    if kind == "cap63":
        m = int(rng.integers(2, 260)); n = int(rng.integers(max(2, m // 2), 2 * m + 8))
        return kind, capped_matrix(rng, m, n, 6, 3, rng.uniform(0.5, 1.0))
    if kind == "cap84":
        m = int(rng.integers(2, 200)); n = int(rng.integers(max(2, m // 2), 3 * m + 8))
        return kind, capped_matrix(rng, m, n, 8, 4, rng.uniform(0.5, 1.0))
    if kind == "wide":
        m = int(rng.integers(1, 60)); n = int(rng.integers(2, 120))
        return kind, (rng.random((m, n)) < rng.uniform(0.03, 0.3)).astype(np.int64)
    m = int(rng.integers(1, 6)); n = int(rng.integers(1, 10))
    return kind, (rng.random((m, n)) < 0.5).astype(np.int64)


def test_fuzz_decode_kernels_vs_oracle():
    rng = np.random.default_rng(int(os.environ.get("QBP_FUZZ_SEED", "20261004")))
    stats = dict(cases=0, syndromes=0, kinds={}, kernels={1: 0, 2: 0, 3: 0})
    for case in range(CASES):
        kind, H = random_matrix(rng)
        m, n = H.shape
        B = int(rng.choice([1, 3, 17, 64, 200, 600]))
        if rng.random() < 0.5:
            pv = np.full(n, rng.uniform(0.005, 0.2))
        else:
            pv = rng.uniform(0.005, 0.3, n)
        prior = np.log((1 - pv) / pv)
        if rng.random() < 0.15:                                   # extreme priors: p = 0.5, 0, 1, > 0.5
            k = max(1, n // 8)
            prior[rng.choice(n, k, replace=False)] = rng.choice([0.0, np.inf, -np.inf, -1.5, 700.0])
        err = (rng.random((B, n)) < pv * rng.uniform(0.3, 1.5)).astype(np.int64)
        syn = (err @ H.T % 2).astype(np.uint8)
        if rng.random() < 0.2:
            syn[rng.integers(0, B)] = rng.random(m) < 0.5          # a syndrome outside the model
        variant = int(rng.choice([_lib.SUM_PRODUCT, _lib.SUM_PRODUCT, _lib.DAMPED_SP, _lib.MIN_SUM]))
        kw = dict(alpha=1.0, damping=1.0, clip_llr=20.0)
        if variant != _lib.SUM_PRODUCT:
            kw = dict(alpha=float(rng.uniform(0.6, 1.0)), damping=float(rng.uniform(0.5, 1.0)),
                      clip_llr=float(rng.uniform(10, 30)))
        max_iter = int(rng.choice([1, 2, 7, 25, 50]))
        flags = _lib.FLAG_FORCE_FULL if rng.random() < 0.3 else 0
        tag = f"case {case} {kind} {m}x{n} B={B} variant={variant} it={max_iter} flags={flags} {kw}"

        dec = bp.decoder_for(H)
        outs = {}
        for k in (_lib.KERNEL_AUTO, _lib.KERNEL_GENERAL, _lib.KERNEL_STREAM):
            dec.set_option(_lib.OPT_KERNEL, k)
            try:
                outs[k] = dec.decode(syn, prior, max_iter, variant, flags=flags, **kw)
                stats["kernels"][dec.info("last_kernel")] += 1
            finally:
                dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_AUTO)
        a = outs[_lib.KERNEL_AUTO]
        for k in (_lib.KERNEL_GENERAL, _lib.KERNEL_STREAM):
            for x, y in zip(a, outs[k]):
                assert np.array_equal(x, y, equal_nan=True), f"kernel {k} differs from the default: {tag}"
        # the general-H kernel's other memory modes (messages in its global workspace; Q there and
        # half of R in LDS), which large matrices select by themselves
        if case % 3 == 0:
            for mem in (1, 2):
                dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_GENERAL)
                dec.set_option(_lib.OPT_GENERAL_MEM, mem)
                try:
                    alt = dec.decode(syn, prior, max_iter, variant, flags=flags, **kw)
                finally:
                    dec.set_option(_lib.OPT_GENERAL_MEM, 0)
                    dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_AUTO)
                for x, y in zip(a, alt):
                    assert np.array_equal(x, y, equal_nan=True), f"general-H memory mode {mem} differs: {tag}"

        # device == oracle in EVERY output bit -- hard decision, converged flag, iteration, LLR -- on every
        # syndrome, however long it ran (round 3: both sides evaluate numpy's own tanh / arctanh kernels, so
        # exact cancellations, the 1e15-scaled residues of beliefPropagation.py:122 and chaotic non-converging
        # trajectories all come out the same; rounds 1-2 had to set those aside).  Half of the cases in the
        # column order of a Fortran-ordered H, where the matrix allows it.
        d_flags, o_flags = flags, (oracle.FLAG_FORCE_FULL if flags else 0)
        if case % 2 and H.sum(0).max() <= 3:
            d_flags |= _lib.FLAG_DENSE_F_COLSUM
            o_flags |= oracle.FLAG_DENSE_F_COLSUM
            a = dec.decode(syn, prior, max_iter, variant, flags=d_flags, **kw)
            stats["f_order"] = stats.get("f_order", 0) + 1
        o = oracle.decode_batch(H, syn, prior, max_iter, variant, flags=o_flags, threads=8, **kw)
        for x, y, what in zip(a, o, ("hard decision", "converged flag", "iteration", "LLR")):
            assert np.array_equal(x, y, equal_nan=True), f"{what} differs from the oracle: {tag}"
        assert ((a[3].view(np.uint64) == o[3].view(np.uint64)) | np.isnan(o[3])).all(), f"LLR bits: {tag}"
        stats["cases"] += 1
        stats["syndromes"] += B
        stats["kinds"][kind] = stats["kinds"].get(kind, 0) + 1
        if (case + 1) % 500 == 0:
            print(f"  ... {case + 1} cases, {stats['syndromes']} syndromes", flush=True)
    print(f"fuzz decode: {stats}")


def test_fuzz_osd_vs_oracle():
    rng = np.random.default_rng(77)
    done = big = 0
    for case in range(max(CASES // 4, 8)):
        kind, H = random_matrix(rng)
        m, n = H.shape
        dec = bp.decoder_for(H)
        B = int(rng.choice([1, 5, 40]))
        llr = rng.normal(0, 5, (B, n))
        if rng.random() < 0.5:
            llr = np.round(llr)                                   # many ties in |llr|
        hard = (llr < 0).astype(np.uint8)
        err = (rng.random((B, n)) < 0.1).astype(np.int64)
        syn = (err @ H.T % 2).astype(np.uint8)                    # in the column space ...
        consistent = rng.random() < 0.8
        if not consistent:                                        # ... or anywhere (the sweep then runs to the rank)
            syn = (rng.random((B, m)) < 0.5).astype(np.uint8)
        elif rng.random() < 0.3:                                  # light residuals, as BP leaves them: early end
            hard = ((err + (rng.random((B, n)) < 0.02)) % 2).astype(np.uint8)
        sol = dec.osd0(syn, llr, hard)
        for i in range(B):
            ref = oracle.osd0(H, syn[i], llr[i], hard[i])
            assert np.array_equal(sol[i], ref), f"OSD-0 differs: case {case} {kind} {m}x{n}"
            if consistent:
                assert np.array_equal(sol[i].astype(np.int64) @ H.T % 2, syn[i])
        if case % 3 == 0:
            # the kernels for matrices beyond the LDS limit, forced: eight pivots per pass (1), one pivot at a
            # time (2), eight per pass with several sweeps (3) -- a decoder of its own (options must not leak)
            own = _lib.Decoder(*bp.csr_from_H(H), bp.DEVICE)
            for k in (1, 2, 3):
                own.set_option(_lib.OPT_OSD_BIG, k)
                assert np.array_equal(own.osd0(syn, llr, hard), sol), f"OSD kernel {k}: case {case} {kind} {m}x{n}"
            own.close()
            big += B
        done += B
        if (case + 1) % 500 == 0:
            print(f"  ... {case + 1} OSD cases", flush=True)
    print(f"fuzz OSD-0: {done} solutions identical to the oracle; {big} of them through the three "
          f"workgroup-per-syndrome kernels as well")


def test_fuzz_mc_counters_vs_oracle():
    """qbp_mc_run (sampling, decoding, classification on the device) against the oracle pipeline:
    on-chip kernel (with and without OSD-0) and the general-H kernel's Monte-Carlo mode, the latter
    also on matrices only it can take."""
    rng = np.random.default_rng(int(os.environ.get("QBP_FUZZ_SEED", "20261004")) + 99)
    done = {1: 0, 2: 0}
    for case in range(max(CASES // 4, 10)):
        if rng.random() < 0.6:
            m = int(rng.integers(8, 120)); n = int(rng.integers(m, 2 * m + 4))
            H = capped_matrix(rng, m, n, 6, 3, 1.0)
        else:
            m = int(rng.integers(4, 50)); n = int(rng.integers(m, 3 * m))
            H = (rng.random((m, n)) < rng.uniform(0.05, 0.25)).astype(np.int64)
        k = int(rng.integers(1, 9))
        Lx = (rng.random((k, n)) < 0.3).astype(np.uint8)
        p = float(rng.uniform(0.01, 0.08))
        prior = np.full(n, np.log((1 - p) / p))
        T = int(rng.integers(200, 3000)); t0 = int(rng.integers(0, 10**9))
        draws = int(rng.integers(1, 3)); seed = int(rng.integers(0, 2**40))
        distance = int(rng.integers(2, 12))
        variant = int(rng.choice([_lib.SUM_PRODUCT, _lib.MIN_SUM]))
        kw = dict(alpha=0.8, damping=0.7, clip_llr=25.0) if variant == _lib.MIN_SUM else {}
        dec = bp.decoder_for(H)
        general = dec.info("kernel_kind") == 2 or rng.random() < 0.3
        osd = bool(rng.random() < 0.5) and not general
        flags = (_lib.FLAG_OSD0 if osd else 0) | (_lib.FLAG_FORCE_FULL if rng.random() < 0.2 else 0)
        dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_GENERAL if general else _lib.KERNEL_AUTO)
        try:
            got = dec.mc_run(Lx, distance, p, prior, t0, t0 + T, draws=draws, seed=seed, max_iter=30,
                             variant=variant, flags=flags, **kw)
            used = dec.info("last_kernel")
        finally:
            dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_AUTO)
        assert used == (2 if general else 1)
        ref = oracle.mc_counters(H, Lx, distance, p, prior, t0, t0 + T, draws=draws, seed=seed,
                                 max_iter=30, variant=variant, osd=osd, **kw)
        assert np.array_equal(np.asarray(got)[:len(ref)], np.asarray(ref)), \
            f"MC counters differ: case {case} {m}x{n} T={T} kernel={used} osd={osd} {got} vs {ref}"
        done[used] += T
        if (case + 1) % 100 == 0:
            print(f"  ... {case + 1} Monte-Carlo cases", flush=True)
    print(f"fuzz Monte-Carlo: {done[1]} trials on the on-chip kernel, {done[2]} on the general-H kernel, "
          f"counters identical to the oracle pipeline")
