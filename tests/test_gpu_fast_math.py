"""GPU: QBP_FLAG_FAST_MATH -- the opt-in approximations of tanh / arctanh (round 2's functions, 2.3 / 1.2 ulp) on the
on-chip kernel, against the REFERENCE's vectors.  The default arithmetic reproduces every bit
(tests/test_wide_golden.py, tests/test_late_golden.py); this flag trades that for throughput.  What is asserted
is what the flag promises (include/qbp.h): hard decision, converged flag and iteration of every stored syndrome;
posterior LLRs of early convergers within 1e-5 relative (the north star's tolerance); the drift on late
convergers is printed next to the spread a change of libm alone produces (the fixture's glibc column)."""
import numpy as np
import pytest

import test_late_golden as late
import test_wide_golden as wide
from qldpc_amd import _lib, bp, codes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,p", wide.CASES)
def test_fast_math_decisions_equal_the_reference_on_the_wide_set(tag, p):
    code, syn, hard, conv, iters, llr_sum = wide.load(tag, p)
    prior = np.full(code.n, np.log((1 - p) / p))
    dec = bp.decoder_for(code.Hx)
    flags = bp.dense_colsum_flags(code.Hx)
    h, c, it, llr = dec.decode(syn.astype(np.uint8), prior, 50, flags=flags | _lib.FLAG_FAST_MATH)
    assert dec.info("last_kernel") == 1
    assert np.array_equal(c, conv) and np.array_equal(it, iters) and np.array_equal(h, hard)
    # against the default arithmetic (= the reference, bit for bit): early convergers within the north star's 1e-5
    h0, c0, it0, llr0 = dec.decode(syn.astype(np.uint8), prior, 50, flags=flags)
    rel = np.abs(llr - llr0).max(1) / np.maximum(np.abs(llr0).max(1), 1e-300)
    early = c0 & (it0 <= 10)
    print(f"fast math [[{tag}]] p={p}: {len(c)} syndromes, decisions identical; LLR vs exact: converged within 10 "
          f"iterations max {rel[early].max():.1e} ({int(early.sum())}), all {np.median(rel):.1e} median / {rel.max():.1e} max; "
          f"{int((llr != llr0).any(1).sum())} vectors differ in some bit")
    assert rel[early].max() <= 1e-5
    assert (llr != llr0).any()                       # (the flag did select the other arithmetic)


@pytest.mark.parametrize("tag,p", late.POINTS)
def test_fast_math_drift_on_late_convergers(tag, p):
    code, g = late.load(tag, p)
    prior = np.full(code.n, np.log((1 - p) / p))
    flags = bp.dense_colsum_flags(code.Hx) | _lib.FLAG_FAST_MATH
    hard, conv, iters, llr = bp.decoder_for(code.Hx).decode(g["syndromes"].astype(np.uint8), prior, 50, flags=flags)
    assert np.array_equal(conv, g["converged"]) and np.array_equal(iters, g["iters"]) and np.array_equal(hard, g["hard"])
    rel = late.rel_rows(llr, g["llr"])
    print(f"\nfast math vs REFERENCE, [[{tag}]] p={p}  (bucket | n | p50 / max relative LLR difference | "
          f"glibc's tanh/atanh in the same formula: p50 / max)")
    for name, sel in late.buckets_of(g):
        if sel.any():
            print(f"  {name:13s} | {int(sel.sum()):3d} | {np.median(rel[sel]):.1e} / {rel[sel].max():.1e} | "
                  f"{np.median(g['libm_rel'][sel]):.1e} / {g['libm_rel'][sel].max():.1e}")
    # a bound, not a fit: the drift stays within two orders of magnitude of a mere change of libm and below 5e-2
    assert rel.max() < 5e-2


def test_fast_math_flag_is_ignored_by_the_other_kernels():
    code = codes.load_code("[[72, 12, 6]]")
    rng = np.random.default_rng(3)
    p = 0.05
    syn = ((rng.random((200, code.n)) < p).astype(np.int64) @ code.Hx.T % 2).astype(np.uint8)
    prior = np.full(code.n, np.log((1 - p) / p))
    dec = _lib.Decoder(*bp.csr_from_H(code.Hx), bp.DEVICE)
    dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_GENERAL)
    a = dec.decode(syn, prior, 50)
    b = dec.decode(syn, prior, 50, flags=_lib.FLAG_FAST_MATH)
    assert dec.info("last_kernel") == 2
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    dec.close()


def test_fast_math_in_the_monte_carlo_loop():
    """qbp_mc_run with the flag: same trials (the sampler does not depend on it), and -- decisions being what they
    are above -- the same classification of all but a handful of marginal trials."""
    code = codes.load_code("[[72, 12, 6]]")
    dec = bp.decoder_for(code.Hx)
    p = 0.05
    prior = np.full(code.n, np.log((1 - p) / p))
    a = dec.mc_run(code.Lx, code.distance, p, prior, 0, 40000, seed=4, max_iter=50)
    b = dec.mc_run(code.Lx, code.distance, p, prior, 0, 40000, seed=4, max_iter=50, flags=_lib.FLAG_FAST_MATH)
    print(dict(zip(_lib.COUNTER_NAMES, a.tolist())), dict(zip(_lib.COUNTER_NAMES, b.tolist())), sep="\n")
    assert a[0] == b[0] == 40000
    assert np.abs(a - b)[:7].max() <= 3 and abs(int(a[7]) - int(b[7])) <= 200      # ([7]: sum of iterations)
