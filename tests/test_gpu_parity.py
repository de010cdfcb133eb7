"""GPU: the HIP decoder, called through the C ABI, against (1) the golden vectors produced by
the real reference and (2) the CPU oracle on fresh seeded inputs."""
import numpy as np
import pytest

import golden_util
from oracle import oracle
from qldpc_amd import _lib, bp, codes, mc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", golden_util.TAGS + golden_util.IRREGULAR_TAGS)
def test_hip_matches_reference_goldens(tag):
    """Every reference-generated vector: hard decision, converged flag, iteration and LLR BITS identical
    (golden_util.compare), through the kernel the library picks -- and through the other kernels."""
    n_cases = n_syn = 0
    for case in golden_util.load(tag):
        dec = bp.decoder_for(case["H"])
        # 1 = fused on-chip kernel ((6,3) or (8,4) shape), 2 = general-H kernel (anything wider)
        assert dec.info("kernel_kind") == (2 if tag == "rand" else 1)
        args = (case["syndromes"], case["prior"], case["max_iter"], case["variant"], case["alpha"],
                case["damping"], case["clip_llr"], golden_util.device_flags(case))
        n_syn += golden_util.compare(case, *dec.decode(*args), "hip")
        for kernel in (2, 3):
            dec.set_option(_lib.OPT_KERNEL, kernel)
            try:
                golden_util.compare(case, *dec.decode(*args), f"hip kernel {kernel}")
            finally:
                dec.set_option(_lib.OPT_KERNEL, 0)
        n_cases += 1
    assert n_cases >= (4 if tag in golden_util.IRREGULAR_TAGS else 10)
    print(f"{tag}: {n_cases} cases, {n_syn} syndromes identical to the reference in every bit (3 kernels)")


def test_drop_in_functions_pick_the_reference_column_order():
    """The module-level functions look at the memory order of the H they are given, as numpy does: the
    Fortran-ordered Hx of the code files and a C-ordered copy of it give DIFFERENT last bits in the dense
    single-syndrome forms of the reference (goldens: Fortran order), the same bits in the batch form."""
    from qldpc_amd import rework
    case = next(c for c in golden_util.load("288") if c["fn"] == "fast4" and c["note"] == "p=0.05")
    H_f = case["H"]
    assert H_f.flags.f_contiguous and not H_f.flags.c_contiguous
    H_c = np.ascontiguousarray(H_f)
    n_diff = 0
    for i in range(len(case["syndromes"])):
        h, c, llr, it = rework.performBeliefPropagationFast(H_f, case["syndromes"][i], case["prior"], maxIter=case["max_iter"])
        assert golden_util.same_bits(llr, case["llr"][i]).all() and it == case["iters"][i]
        h2, c2, llr2, it2 = rework.performBeliefPropagationFast(H_c, case["syndromes"][i], case["prior"], maxIter=case["max_iter"])
        assert c2 == c and it2 == it and np.array_equal(h, h2)
        n_diff += int(not golden_util.same_bits(llr, llr2).all())
    assert n_diff > 0
    b_f = bp.performBeliefPropagationBatch(H_f, case["syndromes"], case["prior"], maxIter=case["max_iter"])
    b_c = bp.performBeliefPropagationBatch(H_c, case["syndromes"], case["prior"], maxIter=case["max_iter"])
    assert golden_util.same_bits(b_f[2], b_c[2]).all()


@pytest.mark.parametrize("name,p,B", [("[[72, 12, 6]]", 0.01, 10000),      # BASELINE config 2
                                      ("[[144, 12, 12]]", 0.05, 4000),
                                      ("[[288, 12, 18]]", 0.05, 3000),
                                      ("[[288, 12, 18]]", 0.01, 3000)])
def test_hip_vs_oracle_fresh(name, p, B):
    code = codes.load_code(name)
    rng = np.random.default_rng([code.n, int(p * 1e6)])      # fixed inputs
    errors = (rng.random((B, code.n)) < p).astype(np.uint8)
    syn = (errors @ code.Hx.T % 2).astype(np.uint8)
    prior = np.full(code.n, np.log((1 - p) / p))
    dec = bp.decoder_for(code.Hx)
    for d_flags, o_flags in ((0, 0), (_lib.FLAG_DENSE_F_COLSUM, oracle.FLAG_DENSE_F_COLSUM)):
        hard, conv, iters, llr = dec.decode(syn, prior, 50, flags=d_flags)
        o_hard, o_conv, o_iters, o_llr = oracle.decode_batch(code.Hx, syn, prior, 50, flags=o_flags, threads=8)
        assert np.array_equal(conv, o_conv)
        assert np.array_equal(iters, o_iters)
        assert np.array_equal(hard, o_hard)
        # device and oracle run the same arithmetic (numpy's tanh / arctanh kernels, numpy's summation order):
        # every LLR bit of every syndrome, late convergers and non-converged ones included
        assert golden_util.same_bits(llr, o_llr).all()
        # converged => H . hard == syndrome (size-independent property)
        assert np.array_equal((hard[conv].astype(np.int64) @ code.Hx.T) % 2, syn[conv])
    print(f"{name} p={p}: {int(conv.sum())}/{B} converged, mean iters {iters.mean():.2f}, "
          f"{int((conv & (iters > 20)).sum())} after iteration 20, {int((~conv).sum())} not at all: "
          "LLR bits identical to the oracle in both column orders")


def test_force_full_same_outputs_and_determinism():
    code = codes.load_code("[[144, 12, 12]]")
    rng = np.random.default_rng(3)
    p = 0.06
    syn = ((rng.random((2000, code.n)) < p).astype(np.uint8) @ code.Hx.T % 2).astype(np.uint8)
    prior = np.full(code.n, np.log((1 - p) / p))
    dec = bp.decoder_for(code.Hx)
    a = dec.decode(syn, prior, 50)
    b = dec.decode(syn, prior, 50, flags=_lib.FLAG_FORCE_FULL)
    c = dec.decode(syn, prior, 50)
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)


@pytest.mark.parametrize("name", ["[[72, 12, 6]]", "[[144, 12, 12]]", "[[288, 12, 18]]", "steane"])
def test_forced_mode_one_barrier_kernel_equals_two_barrier_kernel_and_early_exit(name):
    """QBP_FLAG_FORCE_FULL launches of the (6, 3) shape run the kernel with ONE workgroup barrier per iteration
    (two copies of the messages in LDS, the convergence test of iteration k read one phase later:
    qbp_kernels.hpp, ONE_BAR).  Its outputs must equal, bit for bit, those of the two-barrier forced kernel
    (QBP_OPT_FORCED_TWO_BARRIERS) and of the early-exit kernel -- all three variants of the update, several
    iteration limits (1 and 2: the last iteration of a syndrome is also its first / second), batches larger
    than the resident slots (work fetch) and smaller (idle slots)."""
    code = codes.load_code(name)
    H = code.Hx
    dec = bp.decoder_for(H)
    rng = np.random.default_rng(11)
    for p, B in ((0.02, 300), (0.07, 9000)):
        syn = ((rng.random((B, code.n)) < p).astype(np.int64) @ H.T % 2).astype(np.uint8)
        prior = np.full(code.n, np.log((1 - p) / p)) * rng.uniform(0.8, 1.2, code.n)
        for variant, kw in ((0, {}), (1, dict(alpha=0.9, damping=0.8, clip_llr=20.0)),
                            (2, dict(alpha=0.8, damping=0.7, clip_llr=25.0))):
            for max_iter in (1, 2, 3, 17, 50):
                early = dec.decode(syn, prior, max_iter, variant, **kw)
                one = dec.decode(syn, prior, max_iter, variant, flags=_lib.FLAG_FORCE_FULL, **kw)
                used_one = dec.info("one_barrier")
                dec.set_option(_lib.OPT_FORCED_TWO_BARRIERS, 1)
                try:
                    two = dec.decode(syn, prior, max_iter, variant, flags=_lib.FLAG_FORCE_FULL, **kw)
                    assert dec.info("one_barrier") == 0
                finally:
                    dec.set_option(_lib.OPT_FORCED_TWO_BARRIERS, 0)
                assert used_one == 1, (name, "the one-barrier kernel should fit this shape")
                for x, y, z in zip(early, one, two):
                    assert np.array_equal(x, y, equal_nan=True) and np.array_equal(x, z, equal_nan=True), \
                        (name, p, variant, max_iter)


@pytest.mark.parametrize("name", ["[[72, 12, 6]]", "[[288, 12, 18]]", "steane", "irregular"])
def test_first_check_step_table_changes_nothing(name):
    """Early-exit launches take a syndrome's first check step from a per-workgroup LDS table (its messages
    depend on the priors and the syndrome bit only: qbp_kernels.hpp, r0_lds).  Outputs with the table
    must equal those without (QBP_OPT_NO_FIRST_STEP_TABLE), bit for bit: three variants, non-uniform and
    extreme priors (0, negative, +-inf), an irregular matrix with padded rows, decode and Monte-Carlo."""
    rng = np.random.default_rng(5)
    if name == "irregular":
        H = (rng.random((40, 90)) < 0.05).astype(np.int64)
        H[:, 0] = 0
        H[3] = 0
        H[3, :2] = 1                         # a weight-2 check, an isolated variable
        H = H[(H.sum(1) <= 6)]
        H = H[:, H.sum(0) <= 3]
        Lx, dist = (rng.random((3, H.shape[1])) < 0.3).astype(np.uint8), 4
    else:
        code = codes.load_code(name)
        H = code.Hx
        Lx, dist = (code.Lx, code.distance) if code.Lx is not None else ((rng.random((2, H.shape[1])) < 0.3).astype(np.uint8), 3)
    n = H.shape[1]
    dec = bp.decoder_for(H)
    assert dec.info("kernel_kind") == 1
    p = 0.04
    syn = ((rng.random((5000, n)) < p).astype(np.int64) @ H.T % 2).astype(np.uint8)
    priors = [np.full(n, np.log((1 - p) / p)),
              np.log((1 - p) / p) * rng.uniform(0.2, 2.0, n),
              np.where(rng.random(n) < 0.1, rng.choice([0.0, -1.5, np.inf, -np.inf], n), 3.0)]
    for prior in priors:
        for variant, kw in ((0, {}), (1, dict(alpha=0.9, damping=0.8, clip_llr=20.0)),
                            (2, dict(alpha=0.8, damping=0.7, clip_llr=25.0))):
            for max_iter in (1, 4, 30):
                with_table = dec.decode(syn, prior, max_iter, variant, **kw)
                mc_a = dec.mc_run(Lx, dist, p, prior, 0, 20000, seed=9, max_iter=max_iter, variant=variant, **kw)
                dec.set_option(_lib.OPT_NO_FIRST_STEP_TABLE, 1)
                try:
                    without = dec.decode(syn, prior, max_iter, variant, **kw)
                    mc_b = dec.mc_run(Lx, dist, p, prior, 0, 20000, seed=9, max_iter=max_iter, variant=variant, **kw)
                finally:
                    dec.set_option(_lib.OPT_NO_FIRST_STEP_TABLE, 0)
                for x, y in zip(with_table, without):
                    assert np.array_equal(x, y, equal_nan=True), (name, variant, max_iter)
                assert np.array_equal(mc_a, mc_b), (name, variant, max_iter)


def test_min_sum_and_damped_vs_oracle():
    code = codes.load_code("[[144, 12, 12]]")       # BASELINE config 3 parameterisation
    rng = np.random.default_rng(4)
    p = 0.05
    syn = ((rng.random((3000, code.n)) < p).astype(np.uint8) @ code.Hx.T % 2).astype(np.uint8)
    prior = np.full(code.n, np.log((1 - p) / p))
    dec = bp.decoder_for(code.Hx)
    for variant, kw in ((_lib.MIN_SUM, dict(alpha=0.8, damping=0.7, clip_llr=25.0)),
                        (_lib.DAMPED_SP, dict(alpha=1.0, damping=0.8, clip_llr=20.0))):
        hard, conv, iters, llr = dec.decode(syn, prior, 50, variant, **kw)
        o = oracle.decode_batch(code.Hx, syn, prior, 50, variant, **kw)
        assert np.array_equal(conv, o[1]) and np.array_equal(iters, o[2])
        assert np.array_equal(hard, o[0])
        assert golden_util.same_bits(llr, o[3]).all()
        # the order the reference's rework functions use on the code files' Hx (Fortran order at iteration 0)
        fh, fc, fi, fl = dec.decode(syn, prior, 50, variant, flags=bp.dense_colsum_flags(code.Hx, damped=True), **kw)
        fo = oracle.decode_batch(code.Hx, syn, prior, 50, variant,
                                 flags=oracle.colsum_flags("minsum", code.Hx), **kw)
        assert np.array_equal(fh, fo[0]) and np.array_equal(fc, fo[1]) and np.array_equal(fi, fo[2])
        assert golden_util.same_bits(fl, fo[3]).all()


def test_edge_cases():
    code = codes.load_code("[[72, 12, 6]]")
    dec = bp.decoder_for(code.Hx)
    prior = np.full(code.n, np.log(0.95 / 0.05))
    # empty batch
    hard, conv, iters, llr = dec.decode(np.zeros((0, 36), np.uint8), prior, 50)
    assert hard.shape == (0, 72) and conv.shape == (0,)
    # zero syndrome -> converged at iteration 0, hard = 0
    hard, conv, iters, llr = dec.decode(np.zeros((5, 36), np.uint8), prior, 50)
    assert conv.all() and not hard.any() and (iters == 0).all()
    # B = 1 and a ragged batch size (not a multiple of the slots per workgroup)
    syn = ((np.random.default_rng(0).random((37, 72)) < 0.05).astype(np.uint8) @ code.Hx.T % 2)
    a = dec.decode(syn.astype(np.uint8), prior, 50)
    for i in (0, 36):
        b = dec.decode(syn[i:i + 1].astype(np.uint8), prior, 50)
        assert np.array_equal(a[0][i], b[0][0]) and a[2][i] == b[2][0]
        assert np.array_equal(a[3][i], b[3][0])
    # argument errors
    with pytest.raises(_lib.QbpError):
        dec.decode(syn.astype(np.uint8), prior, 0)
    with pytest.raises(ValueError):
        dec.decode(syn[:, :10].astype(np.uint8), prior, 50)


def test_reference_signatures_on_gpu(capsys):
    """The shim functions: names, arity, dtypes, printed lines (beliefPropagation.py:28,82,141)."""
    case = next(c for c in golden_util.load("steane") if c["note"] == "main.py")
    H, s, prior = case["H"], case["syndromes"][0], list(case["prior"])
    det, ok, llrs = bp.performBeliefPropagation(H, s, prior)
    out = capsys.readouterr().out
    assert "Initial syndrome: [1 1 0]" in out and "Error found at iteration 0: [0 0 1 0 0 0 0]" in out
    assert det.dtype == np.int8 and ok is True and llrs.dtype == np.float64
    assert det.tolist() == [0, 0, 1, 0, 0, 0, 0]
    assert golden_util.same_bits(llrs, case["llr"][0]).all()
    det2, ok2, llr2 = bp.performBeliefPropagationFast(H, s, prior, verbose=False, maxIter=50)
    assert np.array_equal(det, det2) and np.array_equal(llrs, llr2)
    from scipy.sparse import csr_matrix
    det3, ok3, llr3 = bp.performBeliefPropagation(csr_matrix(H), s, prior, verbose=False)
    assert np.array_equal(det, det3) and np.array_equal(llrs, llr3)
    detb, convb, llrb = bp.performBeliefPropagationBatch(H, case["syndromes"], np.array(prior))
    assert detb.dtype == np.int8 and convb.dtype == bool and detb.shape == (1, 7)
    from qldpc_amd import rework
    out4 = rework.performMinSum_Symmetric(H, s, prior, maxIter=50, alpha=0.8, damping=0.7,
                                          clip_llr=25)
    assert len(out4) == 4 and isinstance(out4[3], int)
    with pytest.raises(ValueError):
        bp.performBeliefPropagationFast(H, s, prior, verbose=False, maxIter=0)


@pytest.mark.parametrize("name", ["[[72, 12, 6]]", "[[288, 12, 18]]", "steane"])
def test_general_kernel_equals_fused_kernel_bitwise(name):
    """Both kernels run the same arithmetic in the same order: every output bit must agree."""
    code = codes.load_code(name)
    rng = np.random.default_rng(21)
    p = 0.06
    syn = ((rng.random((700, code.n)) < p).astype(np.uint8) @ code.Hx.T % 2).astype(np.uint8)
    prior = np.log((1 - p) / p) * rng.uniform(0.8, 1.2, code.n)
    dec = bp.decoder_for(code.Hx)
    for variant, kw in ((_lib.SUM_PRODUCT, {}), (_lib.MIN_SUM, dict(alpha=0.8, damping=0.7, clip_llr=25.0)),
                        (_lib.DAMPED_SP, dict(alpha=0.9, damping=0.8, clip_llr=20.0))):
        dec.set_option(_lib.OPT_FORCE_GENERIC, 0)
        a = dec.decode(syn, prior, 40, variant, **kw)
        dec.set_option(_lib.OPT_FORCE_GENERIC, 1)
        assert dec.info("kernel_kind") == 2
        b = dec.decode(syn, prior, 40, variant, **kw)
        c = dec.decode(syn, prior, 40, variant, flags=_lib.FLAG_FORCE_FULL, **kw)
        dec.set_option(_lib.OPT_FORCE_GENERIC, 0)
        dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_STREAM)     # lane per syndrome, messages in HBM
        assert dec.info("kernel_kind") == 3
        d = dec.decode(syn, prior, 40, variant, **kw)
        e = dec.decode(syn, prior, 40, variant, flags=_lib.FLAG_FORCE_FULL, **kw)
        dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_AUTO)
        for x, y, z, u, w in zip(a, b, c, d, e):
            assert np.array_equal(x, y) and np.array_equal(x, z)
            assert np.array_equal(x, u) and np.array_equal(x, w)


@pytest.mark.parametrize("name,T,kind", [("[[144, 12, 12]]", 12, 1), ("[[288, 12, 18]]", 6, 1),
                                         ("[[288, 12, 18]]", 7, 2), ("[[288, 12, 18]]", 8, 2),
                                         ("[[288, 12, 18]]", 12, 2), ("[[288, 12, 18]]", 18, 2)])
def test_large_spacetime_matrix_vs_oracle(name, T, kind):
    """Space-time matrices (spaceTime.py:4-18), row weight 8, column weight 3: [[144,12,12]] over 12
    cycles (864 x 2592) and [[288,12,18]] over 6 cycles (864 x 2592, one syndrome per workgroup)
    run on the (8, 4) instantiation of the on-chip kernel; [[288,12,18]] over 7 cycles (m = 1008)
    needs more than 160 KiB of LDS and over 8 cycles has m = 1152 > 1024: general-H kernel.  Over 12 cycles
    (1728 x 5184, E = 13 680) the messages fit the LDS only as the ONE in-place array of the undamped
    sum-product variant; over 18 cycles (2592 x 7776) all but 112 of its slots do."""
    H = codes.load_code(name).Hx
    m, n = H.shape
    Hs = np.kron(np.eye(T, dtype=np.int64), H)
    Ht = (np.eye(m * T, dtype=np.int64) + np.eye(m * T, k=-m, dtype=np.int64)) % 2
    Hst = np.hstack([Hs, Ht])
    rng = np.random.default_rng(8)
    err = (rng.random((60, Hst.shape[1])) < 0.01).astype(np.int64)
    syn = (err @ Hst.T % 2).astype(np.uint8)
    prior = np.full(Hst.shape[1], np.log(0.99 / 0.01))
    from scipy.sparse import csr_matrix
    dec = bp.decoder_for(csr_matrix(Hst))
    assert dec.info("kernel_kind") == kind
    hard, conv, iters, llr = dec.decode(syn, prior, 50)
    if kind == 1:                                    # the general-H kernel gives the same bits
        dec.set_option(_lib.OPT_FORCE_GENERIC, 1)
        g = dec.decode(syn, prior, 50)
        dec.set_option(_lib.OPT_FORCE_GENERIC, 0)
        for x, y in zip((hard, conv, iters, llr), g):
            assert np.array_equal(x, y)
    dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_STREAM)       # and so does the streaming kernel
    st = dec.decode(syn, prior, 50)
    assert dec.info("last_kernel") == 3
    dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_AUTO)
    for x, y in zip((hard, conv, iters, llr), st):
        assert np.array_equal(x, y)
    o = oracle.decode_batch(Hst, syn, prior, 50, threads=8)
    assert np.array_equal(conv, o[1]) and np.array_equal(iters, o[2]) and np.array_equal(hard, o[0])
    assert golden_util.same_bits(llr, o[3]).all()
    if T >= 12:
        # the damped variant keeps two message arrays (other memory split of the same matrix): a few syndromes
        kw = dict(alpha=1.0, damping=0.8, clip_llr=20.0)
        d = dec.decode(syn[:12], prior, 30, _lib.DAMPED_SP, **kw)
        od = oracle.decode_batch(Hst, syn[:12], prior, 30, _lib.DAMPED_SP, threads=8, **kw)
        assert np.array_equal(d[1], od[1]) and np.array_equal(d[2], od[2]) and np.array_equal(d[0], od[0])
        assert golden_util.same_bits(d[3], od[3]).all()


@pytest.mark.parametrize("name,B,variant,kw", [
    ("[[144, 12, 12]]", 100_000, _lib.MIN_SUM, dict(alpha=0.8, damping=0.7, clip_llr=25.0)),   # config 3
    ("[[288, 12, 18]]", 125_000, _lib.SUM_PRODUCT, {}),                                       # config 4 shard
])
def test_full_size_batches_properties(name, B, variant, kw):
    """BASELINE.json batch sizes, checked through size-independent properties plus an oracle
    comparison on a random subset."""
    code = codes.load_code(name)
    rng = np.random.default_rng(99)
    p = 0.04
    err = (rng.random((B, code.n)) < p).astype(np.uint8)
    syn = (err.astype(np.float32) @ code.Hx.T.astype(np.float32) % 2).astype(np.uint8)
    prior = mc.prior_of(p, code.n)
    dec = bp.decoder_for(code.Hx)
    hard, conv, iters, llr = dec.decode(syn, prior, 50, variant, **kw)
    # (1) converged  <=>  H . hard == syndrome   (non-converged outputs must NOT satisfy it)
    ok = ((hard.astype(np.float32) @ code.Hx.T.astype(np.float32)) % 2 == syn).all(1)
    assert np.array_equal(ok, conv)
    # (2) iteration index range and hard decision == sign of the LLR
    assert iters.min() >= 0 and iters.max() <= 49 and (iters[~conv] == 49).all()
    assert np.array_equal(hard, (llr < 0).astype(np.uint8))
    # (3) batch order does not matter, and a second run gives the same bits
    perm = rng.permutation(B)
    h2, c2, i2, l2 = dec.decode(syn[perm], prior, 50, variant, **kw)
    assert np.array_equal(h2, hard[perm]) and np.array_equal(i2, iters[perm])
    assert np.array_equal(l2, llr[perm])
    # (4) the all-zero syndrome rows converge at iteration 0 with hard = 0
    z = ~syn.any(1)
    assert conv[z].all() and (iters[z] == 0).all() and not hard[z].any()
    # (5) oracle on a random subset
    sub = rng.choice(B, 1500, replace=False)
    o = oracle.decode_batch(code.Hx, syn[sub], prior, 50, variant, **kw)
    assert np.array_equal(hard[sub], o[0]) and np.array_equal(conv[sub], o[1])
    assert np.array_equal(iters[sub], o[2])
    assert golden_util.same_bits(llr[sub], o[3]).all()
    print(f"{name}: B={B}, converged {conv.mean():.4f}, mean iterations {iters.mean() + 1:.2f}")


def test_random_small_matrices_general_kernel_vs_oracle():
    """Fuzz: random sparse matrices of arbitrary shape (empty rows / columns, wide rows) through
    whichever kernel fits, against the oracle."""
    rng = np.random.default_rng(2024)
    for trial in range(25):
        m, n = int(rng.integers(1, 40)), int(rng.integers(2, 80))
        H = (rng.random((m, n)) < rng.uniform(0.03, 0.3)).astype(np.int64)
        for r in np.flatnonzero(H.sum(1) == 1):     # a single-edge check makes min-sum emit inf
            H[r, rng.choice(np.flatnonzero(H[r] == 0))] = 1   # (min2 of an empty set) -> NaN soup
        pv = rng.uniform(0.01, 0.3, n)
        err = (rng.random((40, n)) < pv).astype(np.int64)
        syn = (err @ H.T % 2).astype(np.uint8)
        prior = np.log((1 - pv) / pv)
        dec = bp.decoder_for(H)
        dec.set_option(_lib.OPT_KERNEL, trial % 4 if (trial % 4 != 1 or dec.info("kernel_kind") == 1) else 0)
        for variant, kw in ((_lib.SUM_PRODUCT, {}), (_lib.MIN_SUM, dict(alpha=0.75, damping=0.8, clip_llr=30.0))):
            hard, conv, iters, llr = dec.decode(syn, prior, 25, variant, **kw)
            o = oracle.decode_batch(H, syn, prior, 25, variant, **kw)
            assert np.array_equal(conv, o[1]), (trial, m, n)
            assert np.array_equal(iters, o[2]) and np.array_equal(hard, o[0])
            assert golden_util.same_bits(llr, o[3]).all(), (trial, m, n, variant)


def test_c_example_runs_on_gpu(tmp_path):
    """The plain-C program of examples/ (no Python, no torch) reproduces main.py's known answer."""
    import subprocess
    from test_host_cpu import _build_c_example
    r = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "converged 1 at iteration 0: 0010000" in r.stdout
    assert "1.06635143 1.06635143 -0.06452172 3.32809773 2.19722458 2.19722458 1.06635143" in r.stdout


def test_handle_lifecycle_and_input_types():
    """Many handles created / destroyed, two codes interleaved, and the input spellings the
    reference's callers use: int64 syndromes from (e @ H.T) % 2, bool detection
    events (studyComplete.py:94-97), int8, Python lists; prior as a list of np.float64 (main.py:18)."""
    c72, c144 = codes.load_code("[[72, 12, 6]]"), codes.load_code("[[144, 12, 12]]")
    rng = np.random.default_rng(31)
    e72 = (rng.random((50, 72)) < 0.04).astype(np.int64)
    s72 = (e72 @ c72.Hx.T) % 2
    prior72 = [np.log((1 - 0.04) / 0.04)] * 72
    ref = oracle.decode_batch(c72.Hx, s72, prior72, 50)
    for i in range(40):                         # fresh handle every time (cache bypassed)
        d = _lib.Decoder(*bp.csr_from_H(c72.Hx if i % 2 == 0 else c144.Hx))
        if i % 2 == 0:
            out = d.decode(s72.astype(np.uint8), np.asarray(prior72), 50)
            assert np.array_equal(out[0], ref[0]) and np.array_equal(out[2], ref[2])
        d.close()
        d.close()                               # idempotent
    for k, s in enumerate(s72[:6]):
        want = (ref[0][k].astype(np.int8), bool(ref[1][k]))
        for spelled in (s, s.astype(bool), s.astype(np.int8), s.tolist(), s.astype(np.float64)):
            det, ok, llr = bp.performBeliefPropagationFast(c72.Hx, spelled, prior72, verbose=False)
            assert np.array_equal(det, want[0]) and ok == want[1]
        det, ok, llr, it = __import__("qldpc_amd.rework", fromlist=["x"]).performBeliefPropagationFast(
            c72.Hx.astype(np.float64), s, np.asarray(prior72))
        assert it == ref[2][k] and np.array_equal(det, want[0])
    with pytest.raises(ValueError):
        bp.performBeliefPropagationFast(c72.Hx, s72[0][:10], prior72, verbose=False)
    with pytest.raises(ValueError):
        bp.performBeliefPropagationFast(c72.Hx, s72[0], prior72[:5], verbose=False)
    with pytest.raises(ValueError):
        bp.performBeliefPropagationFast(c72.Hx, s72[0] * 2, prior72, verbose=False)


def test_auto_kernel_choice_for_wide_matrices():
    """A matrix too wide for the on-chip kernel goes to the general-H kernel (workgroup per
    syndrome); batches of >= 131072 syndromes of a small graph to the streaming kernel; same bits."""
    rng = np.random.default_rng(77)
    m, n = 48, 96
    H = np.zeros((m, n), np.int64)
    for c in range(m):
        H[c, rng.choice(n, rng.integers(3, 11), replace=False)] = 1
    dec = bp.decoder_for(H)
    assert dec.info("kernel_kind") == 2
    pv = rng.uniform(0.01, 0.1, n)
    prior = np.log((1 - pv) / pv)
    syn = (((rng.random((140000, n)) < pv).astype(np.int64) @ H.T) % 2).astype(np.uint8)
    big = dec.decode(syn, prior, 30)
    assert dec.info("last_kernel") == 3
    small = dec.decode(syn[:20000], prior, 30)
    assert dec.info("last_kernel") == 2
    for x, y in zip(big, small):
        assert np.array_equal(x[:20000], y)
    o = oracle.decode_batch(H, syn[:1500], prior, 30)
    assert np.array_equal(big[0][:1500], o[0]) and np.array_equal(big[2][:1500], o[2])


def test_distinct_handles_from_concurrent_threads():
    """include/qbp.h: a handle is not thread-safe, distinct handles are independent -- four host
    threads, each with its own handle (ctypes releases the GIL during the call), decode and run
    the Monte-Carlo loop at the same time; results equal the serial ones."""
    import threading
    code = codes.load_code("[[144, 12, 12]]")
    row_ptr, col_idx, m, n = bp.csr_from_H(code.Hx)
    rng = np.random.default_rng(123)
    prior = mc.prior_of(0.04, n)
    jobs = [((rng.random((3000 + 500 * i, n)) < 0.04).astype(np.int64) @ code.Hx.T % 2).astype(np.uint8)
            for i in range(4)]
    ref_dec = bp.decoder_for(code.Hx)
    want = [ref_dec.decode(s, prior, 50) for s in jobs]
    want_mc = [ref_dec.mc_run(code.Lx, code.distance, 0.04, prior, 1000 * i, 1000 * i + 20000, seed=3)
               for i in range(4)]
    got, got_mc, errors = [None] * 4, [None] * 4, []

    def work(i):
        try:
            dec = _lib.Decoder(row_ptr, col_idx, m, n, 0)
            for _ in range(3):
                got[i] = dec.decode(jobs[i], prior, 50)
                got_mc[i] = dec.mc_run(code.Lx, code.distance, 0.04, prior, 1000 * i, 1000 * i + 20000, seed=3)
            dec.close()
        except Exception as e:   # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(4):
        for x, y in zip(got[i], want[i]):
            assert np.array_equal(x, y)
        assert np.array_equal(got_mc[i], want_mc[i])


def test_module_functions_from_a_thread_pool():
    """The reference's functions are pure, so callers may run them from a thread pool: the cached
    Decoder of a matrix is shared by all threads and its host-array methods are serialised
    (qldpc_amd/_lib.py: _locked).  Eight threads decode interleaved batches and single syndromes
    of the SAME matrix through the module-level functions; results equal the serial ones."""
    from concurrent.futures import ThreadPoolExecutor
    from qldpc_amd import osd
    code = codes.load_code("[[144, 12, 12]]")
    H = np.ascontiguousarray(code.Hx)     # (C order: the single-syndrome form then sums like the batch form)
    rng = np.random.default_rng(77)
    prior = mc.prior_of(0.05, code.n)
    jobs = [((rng.random((400 + 37 * i, code.n)) < 0.05).astype(np.int64) @ H.T % 2) for i in range(8)]
    want = [bp.performBeliefPropagationBatch(H, s, prior, 50) for s in jobs]

    def work(i):
        out = None
        for _ in range(3):
            out = bp.performBeliefPropagationBatch(H, jobs[i], prior, 50)
            d, ok, llr = bp.performBeliefPropagationFast(H, jobs[i][0], prior, verbose=False, maxIter=50)
            assert np.array_equal(d, out[0][0]) and ok == out[1][0] and np.array_equal(llr, out[2][0])
            if not ok:
                fixed = osd.performOSD(H, jobs[i][0], llr, d)
                assert np.array_equal(fixed @ H.T % 2, jobs[i][0])
        return out

    with ThreadPoolExecutor(8) as pool:
        got = list(pool.map(work, range(8)))
    for g, w in zip(got, want):
        for x, y in zip(g, w):
            assert np.array_equal(x, y)
