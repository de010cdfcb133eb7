#!/usr/bin/env python3
"""Build ``qldpc_amd/data/logicals.npz`` from the reference's ``codes/*.npz``.

Run in the build container only (``/root/reference`` does not travel):

    python tools/make_code_fixtures.py

It (1) checks that the closed-form ``Hx``/``Hz`` of ``qldpc_amd.codes`` equal the
matrices the reference ships, element for element, and (2) stores the logical
operators ``Lx``/``Lz`` (data with no closed form) bit-packed.  Only data is
read from the reference; no reference code is imported here.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qldpc_amd import codes  # noqa: E402

REF = os.environ.get("QLDPC_REFERENCE", "/root/reference")


def main():
    out = {}
    for name in codes.code_names():
        d = np.load(os.path.join(REF, "codes", f"{name}.npz"))
        Hx, Hz = codes.bb_matrices(name)
        assert np.array_equal(Hx, d["Hx"]), name
        assert np.array_equal(Hz, d["Hz"]), name
        assert int(d["distance"]) == codes.DISTANCES[name]
        Lx, Lz = d["Lx"].astype(np.uint8), d["Lz"].astype(np.uint8)
        assert set(np.unique(Lx)) <= {0, 1} and set(np.unique(Lz)) <= {0, 1}
        # logicals commute with the opposite-type checks
        assert not ((Hz @ Lx.T) % 2).any(), name
        out[f"{name}/k"] = np.int64(Lx.shape[0])
        out[f"{name}/Lx"] = np.packbits(Lx, axis=1)
        out[f"{name}/Lz"] = np.packbits(Lz, axis=1)
        print(name, "Hx", Hx.shape, "row wt", set(Hx.sum(1)), "col wt", set(Hx.sum(0)),
              "Lx", Lx.shape)
    st = np.load(os.path.join(REF, "codes", "steane.npz"))
    assert np.array_equal(st["Hx"], codes.STEANE_H) and np.array_equal(st["Hz"], codes.STEANE_H)
    path = os.path.join(ROOT, "qldpc_amd", "data", "logicals.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
