"""The wide golden set (tests/golden/wide.npz: 17 500 syndromes decoded by the real reference, 3 700 of
them not converged after 50 iterations): hard decision, converged flag, iteration index and the sum of the
posterior LLRs must be IDENTICAL on every syndrome (round 3: numpy's own tanh / arctanh kernels and numpy's
column-sum order; rounds 1-2 allowed a tolerance on the LLRs).  Oracle on CPU; device on GPU."""
import os

import numpy as np
import pytest

from oracle import oracle
from qldpc_amd import codes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wide.npz")
CASES = [(t, p) for t in ("72", "144", "288") for p in (0.03, 0.06, 0.09)]
# [[144,12,12]] with the other update rules: (key prefix, variant, kwargs)
VARIANT_CASES = [("144minsum", 2, dict(alpha=0.8, damping=0.7, clip_llr=25.0), p) for p in (0.04, 0.08)] + \
                [("144sym", 1, dict(alpha=1.0, damping=0.8, clip_llr=20.0), p) for p in (0.04, 0.08)]


def load(tag, p):
    d = np.load(GOLD)
    code = codes.load_code("144" if tag.startswith("144") else tag)
    m, n = code.Hx.shape
    k = f"{tag}/p{p}"
    syn = np.unpackbits(d[f"{k}/syndromes"], axis=1)[:, :m]
    hard = np.unpackbits(d[f"{k}/hard"], axis=1)[:, :n]
    return code, syn, hard, d[f"{k}/converged"].astype(bool), d[f"{k}/iters"], d[f"{k}/llr_sum"]


def check(tag, p, decode, who):
    code, syn, hard, conv, iters, llr_sum = load(tag, p)
    prior = np.full(code.n, np.log((1 - p) / p))
    h, c, it, llr = decode(code.Hx, syn, prior)
    bad_conv = int((c != conv).sum())
    bad_iter = int((it != iters).sum())
    bad_hard = int((h != hard).any(1).sum())
    # the fixture stores `values.sum()` per syndrome (numpy's pairwise sum of the n posteriors): the same
    # call on bit-identical values gives the same bits
    s = np.array([row.sum() for row in llr])
    bad_llr = int((s != llr_sum).sum())
    print(f"{who} {tag} p={p}: {len(conv)} syndromes, {int(conv.sum())} converged; mismatches vs the "
          f"reference: converged {bad_conv}, iteration {bad_iter}, hard decision {bad_hard}, LLR sum {bad_llr}")
    assert bad_conv == 0 and bad_iter == 0 and bad_hard == 0 and bad_llr == 0


@pytest.mark.parametrize("tag,p", CASES)
def test_oracle_wide(tag, p):
    check(tag, p, lambda H, s, pr: oracle.decode_batch(H, s, pr, 50, flags=oracle.colsum_flags("fast4", H)),
          "oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("tag,p", CASES)
def test_device_wide(tag, p):
    from qldpc_amd import bp
    check(tag, p, lambda H, s, pr: bp.decoder_for(H).decode(s.astype(np.uint8), pr, 50,
                                                            flags=bp.dense_colsum_flags(H)), "device")


@pytest.mark.parametrize("tag,variant,kw,p", VARIANT_CASES)
def test_oracle_wide_variants(tag, variant, kw, p):
    fn = "minsum" if variant == 2 else "sym"
    check(tag, p, lambda H, s, pr: oracle.decode_batch(H, s, pr, 50, variant,
                                                       flags=oracle.colsum_flags(fn, H), **kw), "oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("tag,variant,kw,p", VARIANT_CASES)
def test_device_wide_variants(tag, variant, kw, p):
    from qldpc_amd import bp
    check(tag, p, lambda H, s, pr: bp.decoder_for(H).decode(
        s.astype(np.uint8), pr, 50, variant, kw["alpha"], kw["damping"], kw["clip_llr"],
        bp.dense_colsum_flags(H, damped=True)), "device")
