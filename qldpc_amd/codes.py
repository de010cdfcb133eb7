"""Parity-check matrices of the codes the reference ships in ``codes/*.npz``.

``Hx``/``Hz`` of the five bivariate-bicycle (BB) codes are rebuilt from the
polynomials in the reference's ``generateCodeMatrices.py:5-46`` (closed form
``Hx = [A | B]``, ``Hz = [B^T | A^T]``); the Steane matrix
is the literal at ``generateCodeMatrices.py:64-68``.  The logical operators
``Lx`` have no closed form (they come from ``qldpc``'s ``get_logical_ops()``,
``generateCodeMatrices.py:52-57``), so they ship bit-packed in
``qldpc_amd/data/logicals.npz`` (made by ``tools/make_code_fixtures.py``).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "logicals.npz")

# name -> (l, m, A terms, B terms); a term is (x power, y power)
BB_CODES = {
    "[[72, 12, 6]]": (6, 6, [(3, 0), (0, 1), (0, 2)], [(0, 3), (1, 0), (2, 0)]),
    "[[90, 8, 10]]": (15, 3, [(9, 0), (0, 1), (0, 2)], [(0, 0), (2, 0), (7, 0)]),
    "[[108, 8, 10]]": (9, 6, [(3, 0), (0, 1), (0, 2)], [(0, 3), (1, 0), (2, 0)]),
    "[[144, 12, 12]]": (12, 6, [(3, 0), (0, 1), (0, 2)], [(0, 3), (1, 0), (2, 0)]),
    "[[288, 12, 18]]": (12, 12, [(3, 0), (0, 2), (0, 7)], [(0, 3), (1, 0), (2, 0)]),
}
DISTANCES = {
    "[[72, 12, 6]]": 6,
    "[[90, 8, 10]]": 10,
    "[[108, 8, 10]]": 10,
    "[[144, 12, 12]]": 12,
    "[[288, 12, 18]]": 18,
}
ALIASES = {"72": "[[72, 12, 6]]", "90": "[[90, 8, 10]]", "108": "[[108, 8, 10]]",
           "144": "[[144, 12, 12]]", "288": "[[288, 12, 18]]"}

STEANE_H = np.array(
    [[1, 0, 1, 0, 1, 0, 1],
     [0, 1, 1, 0, 0, 1, 1],
     [0, 0, 0, 1, 1, 1, 1]], dtype=np.int64)


def _shift(k: int) -> np.ndarray:
    return np.roll(np.eye(k, dtype=np.int64), 1, axis=1)


def _poly(l: int, m: int, terms) -> np.ndarray:
    x = np.kron(_shift(l), np.eye(m, dtype=np.int64))
    y = np.kron(np.eye(l, dtype=np.int64), _shift(m))
    out = np.zeros((l * m, l * m), dtype=np.int64)
    for px, py in terms:
        out = out + np.linalg.matrix_power(x, px) @ np.linalg.matrix_power(y, py)
    return out % 2


def bb_matrices(name: str):
    """Return ``(Hx, Hz)`` (int64 0/1) of a BB code by its reference name."""
    l, m, a_terms, b_terms = BB_CODES[name]
    A = _poly(l, m, a_terms)
    B = _poly(l, m, b_terms)
    return np.hstack([A, B]), np.hstack([B.T, A.T])


@dataclass
class Code:
    name: str
    Hx: np.ndarray          # (m, n) int64 0/1
    Hz: np.ndarray
    Lx: np.ndarray | None   # (k, n) uint8, None for Steane (steane.npz has no logicals)
    Lz: np.ndarray | None
    distance: int | None

    @property
    def n(self) -> int:
        return self.Hx.shape[1]


def load_code(name: str) -> Code:
    """Same content as ``np.load('codes/<name>.npz')`` in the reference."""
    name = ALIASES.get(name, name)
    if name == "steane":
        return Code("steane", STEANE_H.copy(), STEANE_H.copy(), None, None, None)
    Hx, Hz = bb_matrices(name)
    # memory order as in the reference's files (Hx Fortran-ordered, Hz C-ordered): the order in which the
    # reference's dense decoders add up column sums follows it (qldpc_amd.bp.dense_colsum_flags)
    Hx = np.asfortranarray(Hx)
    n = Hx.shape[1]
    with np.load(_DATA) as d:
        k = int(d[f"{name}/k"])
        Lx = np.unpackbits(d[f"{name}/Lx"], axis=1)[:k, :n].astype(np.uint8)
        Lz = np.unpackbits(d[f"{name}/Lz"], axis=1)[:k, :n].astype(np.uint8)
    return Code(name, Hx, Hz, Lx, Lz, DISTANCES[name])


def code_names():
    return list(BB_CODES)
