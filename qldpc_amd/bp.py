"""Host-side mirror of the reference's BP operator API, backed by ``libqbp.so`` (HIP, gfx950).

Same names, positional order, defaults, return arity, dtypes and printed lines as

* ``decoding/beliefPropagation.py``     performBeliefPropagation (:6), performBeliefPropagationFast (:88)
* ``decoding/beliefPropagationGPU.py``  performBeliefPropagationGPU (:22), performBeliefPropagationBatch (:81),
                                        generate_errors_and_syndromes_batch (:181), GPU_AVAILABLE
* ``rework/decoding.py``                see ``qldpc_amd/rework.py`` (4-tuple variants)

All decoding runs on the MI355X; there is no CPU path in this package.  Deviations from the
reference, all on inputs the reference does not handle either:
``maxIter < 1`` raises ``ValueError`` (reference: ``UnboundLocalError``), shape mismatches raise
``ValueError`` (reference: numpy ``IndexError``/broadcast errors), syndrome entries must be 0/1,
priors must not be NaN (+-inf are fine).
"""
from __future__ import annotations

import hashlib
import os
import threading

import numpy as np

from . import _lib

try:  # fast content hash for the per-call decoder cache
    import xxhash

    def _digest(buf) -> bytes:
        return xxhash.xxh3_128_digest(buf)
except Exception:  # pragma: no cover
    def _digest(buf) -> bytes:
        return hashlib.blake2b(buf, digest_size=16).digest()

_DECODERS: dict = {}
_BY_ID: dict = {}          # id(H) -> (weakref to H or None, fingerprint, key): skips the content hash
_MAX_CACHED = 16
_CACHE_LOCK = threading.RLock()   # the reference's functions are pure: callable from thread pools
# HIP device used by the module-level functions (one process per GPU: set QBP_DEVICE per rank, or
# assign qldpc_amd.bp.DEVICE before the first call)
DEVICE = int(os.environ.get("QBP_DEVICE", "0"))


def csr_from_H(H):
    """CSR (row_ptr, col_idx, m, n) of the non-zeros of H, columns ascending -- the order in
    which the reference multiplies / sums (np.prod(axis=1), csr indices: beliefPropagation.py:22)."""
    try:
        from scipy.sparse import issparse
    except Exception:  # pragma: no cover
        def issparse(_):
            return False
    if issparse(H):
        S = H.tocsr().copy()
        S.sum_duplicates()
        S.eliminate_zeros()
        S.sort_indices()
        return (np.ascontiguousarray(S.indptr, np.int32), np.ascontiguousarray(S.indices, np.int32),
                int(S.shape[0]), int(S.shape[1]))
    A = np.asarray(H)
    if A.ndim != 2:
        raise ValueError(f"H must be 2-D, got shape {A.shape}")
    rows, cols = np.nonzero(A)             # row-major order: ascending columns within a row
    row_ptr = np.zeros(A.shape[0] + 1, np.int32)
    np.cumsum(np.bincount(rows, minlength=A.shape[0]), out=row_ptr[1:])
    return row_ptr, np.ascontiguousarray(cols, np.int32), int(A.shape[0]), int(A.shape[1])


# The decoder cache is keyed by the CONTENT of H: every call hashes the matrix (0.2 ms for the BB codes; 30 ms
# for the reference's 2592 x 7776 float64 space-time matrix, against a 0.1 - 0.9 ms decode).  Two ways around
# the hash, both explicit:
#   * keep the handle:  dec = qldpc_amd.bp.decoder_for(H); dec.decode(...)        (no lookup at all), or
#   * QBP_TRUST_IDENTITY=1 (or qldpc_amd.bp.TRUST_IDENTITY = True): an array that is the very object and
#     buffer seen before is trusted -- large writeable ones after a checksum of ~64k evenly spaced
#     elements, which an in-place edit between two calls can miss: call forget(H) after editing such a matrix.
# Without the switch only arrays that cannot change are recognised by identity: read-only arrays whose whole
# chain of bases is read-only too (a read-only VIEW of a writeable array can change under it).
_FULL_HASH_LIMIT = 1 << 20
TRUST_IDENTITY = os.environ.get("QBP_TRUST_IDENTITY", "0") not in ("0", "")
_STRICT = os.environ.get("QBP_STRICT_HASH", "0") not in ("0", "")     # (overrides TRUST_IDENTITY)


def _immutable(A):
    """A read-only ndarray none of whose bases is writeable."""
    while isinstance(A, np.ndarray):
        if A.flags.writeable:
            return False
        A = A.base
    return A is None or isinstance(A, (bytes, memoryview)) and getattr(A, "readonly", True)


def _fingerprint(H, sparse):
    """(buffer address, shape, dtype, strides) of an ndarray, else None."""
    if sparse or not isinstance(H, np.ndarray):
        return None
    return (H.__array_interface__["data"][0], H.shape, H.dtype.str, H.strides)


def _sample_digest(A):
    """Checksum of ~64k evenly spaced elements (large arrays only)."""
    flat = A.reshape(-1) if A.flags.c_contiguous else A.ravel()
    step = max(1, flat.size // 65536)
    return _digest(np.ascontiguousarray(flat[::step]))


def forget(H=None):
    """Drop the cached decoder of H (all of them when H is None): call it after changing a large
    matrix IN PLACE, which the identity fast path cannot notice."""
    with _CACHE_LOCK:
        if H is None:
            _DECODERS.clear()
            _BY_ID.clear()
            return
        for k in [k for k in _BY_ID if k[0] == id(H)]:
            _BY_ID.pop(k, None)


def decoder_for(H, device=None) -> _lib.Decoder:
    """Decoder handle for H, cached on the matrix content (the reference re-derives the graph on
    every call; here that happens once per code).

    Lookup: the matrix is hashed in full on every call, so an in-place change is always noticed; only an
    array that cannot change (read-only down its whole chain of bases) is recognised by identity.
    QBP_TRUST_IDENTITY=1 extends that to large writeable arrays after a sampled checksum (see the comment
    above _FULL_HASH_LIMIT: opt-in, because an edit can slip through the sample).

    The cache only holds references: an evicted Decoder stays usable by whoever still holds it and
    its device handle is destroyed when the last reference goes (``Decoder.__del__``).  Every
    caller gets the same Decoder; its host-array methods are serialised (``_lib._locked``), so the
    module-level functions may be called from several threads -- threads that want to decode the
    same matrix CONCURRENTLY should each build their own ``_lib.Decoder``."""
    with _CACHE_LOCK:
        return _decoder_for(H, DEVICE if device is None else device)


def _decoder_for(H, device):
    try:
        from scipy.sparse import issparse
        sparse = issparse(H)
    except Exception:  # pragma: no cover
        sparse = False
    fp = _fingerprint(H, sparse)
    trust = TRUST_IDENTITY and not _STRICT
    big = trust and fp is not None and H.nbytes > _FULL_HASH_LIMIT
    frozen = fp is not None and _immutable(H)
    sample = None
    if fp is not None and (big or frozen):
        hit = _BY_ID.get((id(H), device))
        if hit is not None and hit[0]() is H and hit[1] == fp and hit[2] in _DECODERS:
            if frozen:
                return _DECODERS[hit[2]]
            sample = _sample_digest(H)
            if sample == hit[3]:
                return _DECODERS[hit[2]]
    if sparse:
        S = H.tocsr()
        key = ("s", S.shape, _digest(np.ascontiguousarray(S.indptr)),
               _digest(np.ascontiguousarray(S.indices)), _digest(np.ascontiguousarray(S.data)),
               device)
    else:
        A = np.asarray(H)
        if A.ndim == 2 and A.flags.f_contiguous and not A.flags.c_contiguous:
            # (the reference's codes/*.npz matrices are Fortran-ordered: hash the buffer as it lies, no copy;
            # the layout is part of the key -- numpy's summation order, hence the decode flags, depend on it)
            key = ("f", A.shape, A.dtype.str, _digest(A.T), device)
        else:
            A = np.ascontiguousarray(A)
            key = ("d", A.shape, A.dtype.str, _digest(A), device)
    dec = _DECODERS.get(key)
    if dec is None:
        row_ptr, col_idx, m, n = csr_from_H(H)
        dec = _lib.Decoder(row_ptr, col_idx, m, n, device)
        if len(_DECODERS) >= _MAX_CACHED:
            _DECODERS.pop(next(iter(_DECODERS)))       # drop the reference only (see docstring)
        _DECODERS[key] = dec
    if fp is not None and (big or frozen):
        import weakref
        if len(_BY_ID) > 4 * _MAX_CACHED:
            _BY_ID.clear()
        try:
            if big and not frozen and sample is None:
                sample = _sample_digest(H)
            _BY_ID[(id(H), device)] = (weakref.ref(H), fp, key, sample)
        except TypeError:  # pragma: no cover
            pass
    return dec


def _syndromes(s, m, batch):
    a = np.asarray(s)
    if a.dtype == bool:
        a = a.astype(np.int8)
    a = np.asarray(a, dtype=np.int8)            # as the reference casts (:94)
    want = 2 if batch else 1
    if a.ndim != want or a.shape[-1] != m:
        raise ValueError(f"syndrome has shape {a.shape}, expected {'(B, ' if batch else '('}{m})")
    if (a.view(np.uint8) > 1).any():            # (int8: negative values are > 1 as bytes)
        raise ValueError("syndrome entries must be 0 or 1")
    return a


def _prior(initialBelief, n):
    p = np.asarray(initialBelief, dtype=np.float64)
    if p.shape != (n,):
        raise ValueError(f"initialBelief has shape {p.shape}, expected ({n},)")
    if np.isnan(p).any():
        # (+-inf are legal: p = 0 or 1.  The reference turns a NaN prior into NaN messages on the
        # whole connected component; the device kernels clip with min/max, which drop NaNs.)
        raise ValueError("initialBelief contains NaN")
    return p


def _check_iter(maxIter):
    if int(maxIter) < 1:
        raise ValueError(f"maxIter must be >= 1, got {maxIter}")
    return int(maxIter)


# numpy's temporary elision (an operand that is an unreferenced temporary of at least this size is reused as the
# output of the next operation, keeping ITS memory order): numpy/_core/src/multiarray/temp_elide.c
_NPY_MIN_ELIDE_BYTES = 256 * 1024
_warned_order = set()


def dense_colsum_flags(H, damped=False):
    """Order in which a DENSE single-syndrome form of the reference adds up a column of check->variable
    messages (``np.sum(R, axis=0)``, decoding/beliefPropagation.py:129, rework/decoding.py:61 / :119 / :173),
    as a ``qbp_decode_batch`` flag.  It depends on the memory order of the caller's H, because every (m, n)
    temporary of those functions inherits the order of ``mask = H != 0``:

    * C order (lists, scipy-sparse input, ``np.hstack`` / ``np.kron`` products such as the space-time
      matrices): rows are added one after the other -- 0;
    * Fortran order -- the ``Hx`` of the reference's ``codes/*.npz`` is stored that way: a column is contiguous
      and numpy adds it with its pairwise sum over all m entries -- ``FLAG_DENSE_F_COLSUM``;
    * the damped variants (``damped=True``, rework/decoding.py:5 and :131) copy Q to C order
      (``Q_old = Q.copy()``), after which C order wins in ``damping * Q_new + (1 - damping) * Q_old`` from
      iteration 1 on -- ``FLAG_DENSE_F_COLSUM_ITER0`` -- unless the arrays reach numpy's temporary-elision size,
      where the F-ordered temporary ``damping * Q_new`` is reused as the result.

    The loop form sums gathered columns (``FLAG_PAIRWISE_COLSUM``) and the batch form works on C-ordered
    (B, m, n) arrays: neither depends on H's order.  Measured on numpy 2.2.6; DESIGN.md section 2."""
    if not isinstance(H, np.ndarray) or H.ndim != 2 or H.shape[0] < 2 or H.shape[1] < 2:
        return 0
    if not abs(H.strides[0]) < abs(H.strides[1]):
        return 0
    if damped and H.shape[0] * H.shape[1] * 8 < _NPY_MIN_ELIDE_BYTES:
        return _lib.FLAG_DENSE_F_COLSUM_ITER0
    return _lib.FLAG_DENSE_F_COLSUM


# QBP_FAST_MATH=1 (or bp.FAST_MATH = True): the module-level functions pass QBP_FLAG_FAST_MATH -- +28 % on the
# on-chip kernel, LLRs no longer bit-identical to the reference's (include/qbp.h).  Off by default.
FAST_MATH = os.environ.get("QBP_FAST_MATH", "0") not in ("0", "")


def _math_flag():
    return _lib.FLAG_FAST_MATH if FAST_MATH else 0


def decode_one(H, syndrome, initialBelief, maxIter, variant=_lib.SUM_PRODUCT, alpha=1.0,
               damping=1.0, clip_llr=20.0, flags=0):
    """(hard int8[n], converged bool, llr float64[n], iteration int) for one syndrome."""
    dec = decoder_for(H)
    flags |= _math_flag()
    syn = _syndromes(syndrome, dec.m, batch=False)
    args = (syn[None, :].view(np.uint8), _prior(initialBelief, dec.n), _check_iter(maxIter), variant,
            alpha, damping, clip_llr)
    try:
        hard, conv, iters, llr = dec.decode(*args, flags)
    except _lib.QbpError as e:
        order = flags & (_lib.FLAG_DENSE_F_COLSUM | _lib.FLAG_DENSE_F_COLSUM_ITER0)
        if not order or e.code != _lib.E_UNSUPPORTED:
            raise
        # a column-sum association the kernels cannot express (include/qbp.h): row-by-row sums instead --
        # hard decisions are unaffected in every case tested, LLRs may differ from the reference's in the last
        # ulps.  Said once per matrix shape.
        key = (dec.m, dec.n, order)
        if key not in _warned_order:
            _warned_order.add(key)
            import warnings
            warnings.warn(f"qldpc_amd: {e}; using row-by-row column sums (LLRs may differ from the "
                          "reference's in the last ulps)", RuntimeWarning, stacklevel=3)
        hard, conv, iters, llr = dec.decode(*args, flags & ~order)
    return hard[0].astype(np.int8), bool(conv[0]), llr[0], int(iters[0])


def performBeliefPropagationFast(H, syndrome, initialBelief, verbose=True, maxIter=50):
    """decoding/beliefPropagation.py:88-144."""
    hard, conv, llr, it = decode_one(H, syndrome, initialBelief, maxIter, flags=dense_colsum_flags(H))
    if conv and verbose:
        print(f"Error found at iteration {it}: {hard}")                      # :141
    return hard, conv, llr


def performBeliefPropagation(H, syndrome, initialBelief, verbose=True, plotPath=None, maxIter=50):
    """decoding/beliefPropagation.py:6-85 (accepts scipy-sparse H, :8-10)."""
    if verbose:
        print(f"Initial syndrome: {np.asarray(syndrome, dtype=np.int8)}")    # :28
    if plotPath is not None:                                                 # :30-31
        from drawUtils import plotGraph   # the reference's plotting helper, if importable
        try:
            from scipy.sparse import issparse
            dense = H.toarray() if issparse(H) else np.asarray(H, dtype=np.float64)
        except Exception:  # pragma: no cover
            dense = np.asarray(H, dtype=np.float64)
        plotGraph(dense, path=plotPath)
    # the loop form sums each gathered column with np.sum (:68): numpy's pairwise order from 8
    # entries per column on -- the dense forms accumulate row by row
    hard, conv, llr, it = decode_one(H, syndrome, initialBelief, maxIter,
                                     flags=_lib.FLAG_PAIRWISE_COLSUM)
    if conv and verbose:
        print(f"Error found at iteration {it}: {hard}")                      # :82
    return hard, conv, llr


def performBeliefPropagationGPU(H, syndrome, initialBelief, verbose=False, maxIter=50):
    """decoding/beliefPropagationGPU.py:22-78."""
    hard, conv, llr, it = decode_one(H, syndrome, initialBelief, maxIter, flags=dense_colsum_flags(H))
    if conv and verbose:
        print(f"Error found at iteration {it}")                              # :73
    return hard, conv, llr


class _LastBatch:
    """What `performOSD` needs to serve the reference driver's loop (paperResults_GPU.py:113-123: one
    OSD call per sample BP did not converge on, with rows of the arrays returned below) from ONE
    batched OSD launch: see qldpc_amd/osd.py.

    Lifetime: the record is per THREAD (a thread pool's workers do not overwrite each other's
    batch); it holds the returned LLR array (8 n bytes per syndrome, the bulk) only WEAKLY -- rows are recognised
    by their address in it, which is only meaningful while the caller still holds the array; once it is gone the
    record is dead and nothing big is kept alive -- plus the hard decisions and the syndromes (n + m bytes per
    syndrome) and, once the first OSD call arrives, its own gathered copies of the failing rows and their
    solutions; it is dropped when every failing row has been
    served, at the next batch call of the thread, or never created with QBP_NO_LAST_BATCH=1
    (qldpc_amd.bp.REMEMBER_LAST_BATCH = False)."""
    __slots__ = ("dec", "syn", "llr_ref", "hard", "conv", "addr", "rowbytes", "rows", "pos", "inputs",
                 "solutions", "lock", "served", "n_fail")

    def __init__(self, dec, syn, llr, hard, conv):
        import weakref
        self.dec, self.syn, self.conv = dec, syn, conv.copy()
        self.llr_ref, self.hard = weakref.ref(llr), hard
        self.addr = llr.__array_interface__["data"][0]
        self.rowbytes, self.rows = llr.strides[0], llr.shape[0]
        self.pos = self.inputs = self.solutions = None
        self.lock = threading.Lock()
        self.served, self.n_fail = set(), int((~conv).sum())

    @property
    def llr(self):
        return self.llr_ref()


REMEMBER_LAST_BATCH = os.environ.get("QBP_NO_LAST_BATCH", "0") in ("0", "")
_LAST_BATCH_LIMIT = 256 << 20          # larger batches are not remembered (their failing rows alone are big)
_TLS = threading.local()


def _last_batch():
    return getattr(_TLS, "last_batch", None)


def _set_last_batch(lb):
    _TLS.last_batch = lb


def performBeliefPropagationBatch(H, syndromes, initialBelief, maxIter=50):
    """decoding/beliefPropagationGPU.py:81-178 -> (int8[B, n], bool[B], float64[B, n])."""
    dec = decoder_for(H)
    syn = _syndromes(syndromes, dec.m, batch=True)
    hard, conv, _, llr = dec.decode(syn.view(np.uint8), _prior(initialBelief, dec.n),
                                    _check_iter(maxIter), flags=_math_flag())
    hard = hard.view(np.int8)                   # (0/1 bytes of a fresh array: no copy)
    keep = REMEMBER_LAST_BATCH and 1 < len(conv) and llr.nbytes <= _LAST_BATCH_LIMIT and not conv.all()
    _set_last_batch(_LastBatch(dec, syn, llr, hard, conv) if keep else None)
    return hard, conv, llr


def generate_errors_and_syndromes_batch(H, error_rate, batch_size, rng=None):
    """decoding/beliefPropagationGPU.py:181-200.  Host numpy on purpose: callers pass a numpy
    Generator (paperResults_GPU.py:34) and rely on its stream; the device Monte-Carlo path
    (qldpc_amd.mc) has its own counter-based sampler."""
    if rng is None:
        rng = np.random.default_rng()
    H = np.asarray(H)
    num_vars = H.shape[1]
    errors = (rng.random((batch_size, num_vars)) < error_rate).astype(np.int8)
    return errors, _syndromes_of(errors, H)


def _syndromes_of(errors, H):
    """``((errors @ H.T) % 2).astype(np.int8)`` (:198-200).  numpy multiplies integer matrices with a
    scalar loop over all m n entries (int8 x int64, 5 000 x 288 x 144: 0.16 s -- fifty times the
    decode of that batch); for an integer or bool H the same integer product over the non-zeros of H
    only (scipy CSR: 5 ms) has the same parity -- exactly: integer arithmetic, and a wrap-around of
    the reference's narrower accumulator (int8 for a bool H) never changes a parity."""
    if H.dtype.kind in "iub":
        from scipy.sparse import csr_matrix
        sums = np.asarray(errors @ csr_matrix(H.T.astype(np.int64)))
        return (sums & 1).astype(np.int8)
    return ((errors @ H.T) % 2).astype(np.int8)


def gpu_available() -> bool:
    """True when libqbp.so loads and sees a device (the shim's GPU_AVAILABLE)."""
    try:
        import ctypes as C
        lib = _lib.load()
        h = C.c_void_p()
        rp = np.array([0, 1], np.int32)
        ci = np.array([0], np.int32)
        rc = lib.qbp_create(rp.ctypes.data, ci.ctypes.data, 1, 1, DEVICE, C.byref(h))
        if rc == 0:
            lib.qbp_destroy(h)
        return rc == 0
    except Exception:
        return False
