#!/usr/bin/env python3
"""Extract the constants of numpy's float64 tanh / arctanh kernels and write them as C headers.

The reference's BP arithmetic calls np.tanh and np.arctanh (decoding/beliefPropagation.py:114,126).
On x86-64 hosts with AVX512_SKX numpy 2.2.6 dispatches both ufuncs to the SVML routines it vendors
(numpy/_core/src/umath/svml/linux/avx512/svml_z0_tanh_d_la.s and svml_z0_atanh_d_ha.s, BSD-3-Clause,
(c) Intel): np.tanh == __svml_tanh8, np.arctanh == __svml_atanh8_ha, bit for bit (checked by
tests/test_np_math.py).  The numpy sources are not present offline, so the polynomial / table constants
are read from the .rodata of the installed numpy binary (symbol __svml_dtanh_data_internal and
__svml_datanh_ha_data_internal_avx512), and the one hardware-defined step of the arctanh kernel --
VRCP14PD followed by rounding to 4 mantissa bits -- is tabulated by executing VRCP14PD on this CPU
over every 30-bit mantissa prefix (it turns out to be a step function with 16 thresholds on a
2^-16 grid).

Outputs (both generated, both committed):
  oracle/np_svml_tables.h            -- for the CPU oracle (test infrastructure)
  qldpc_amd/csrc/qbp_np_tables.hpp   -- for the HIP kernels (product)

Run in the build container only:  python tools/extract_np_svml.py
"""
import ctypes
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def numpy_so():
    return glob.glob(os.path.join(os.path.dirname(np.__file__), "_core", "_multiarray_umath*.so"))[0]


def symbol_table(so):
    out = subprocess.check_output(["nm", "-D", "-S", "--defined-only", so]).decode()
    syms = {}
    for line in out.splitlines():
        p = line.split()
        if len(p) == 4:
            syms[p[3]] = (int(p[0], 16), int(p[1], 16))
    return syms


def sections(so):
    out = subprocess.check_output(["readelf", "-S", "-W", so]).decode()
    secs = []
    for line in out.splitlines():
        line = line.replace("[ ", "[")
        p = line.split()
        if len(p) > 6 and p[0].startswith("["):
            try:
                secs.append((p[1], int(p[3], 16), int(p[4], 16), int(p[5], 16)))
            except ValueError:
                pass
    return secs


def read_va(so, secs, va, nbytes):
    for _, addr, off, size in secs:
        if addr and addr <= va < addr + size:
            with open(so, "rb") as f:
                f.seek(off + va - addr)
                return f.read(nbytes)
    raise KeyError(hex(va))


def data_addresses(so, syms):
    """Addresses of the two constant tables: the targets of the first rip-relative loads of the kernels."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    res = {}
    for name, key in (("__svml_tanh8", "__svml_dtanh_data_internal"),
                      ("__svml_atanh8_ha", "__svml_datanh_ha_data_internal_avx512")):
        a, s = syms[name]
        dis = subprocess.check_output([objdump, "-d", "--no-show-raw-insn", f"--start-address={a:#x}",
                                       f"--stop-address={a + s:#x}", so]).decode()
        base = None
        for line in dis.splitlines():
            if f"<{key}" in line and "#" in line:
                tgt = line.split("#")[1].split()[0]
                sym = line.split("<")[1].split(">")[0]
                off = int(sym.split("+")[1], 16) if "+" in sym else 0
                base = int(tgt, 16) - off
                break
        res[key] = base
    return res


RCP_SRC = r"""
#include <immintrin.h>
#include <stdint.h>
#include <string.h>
uint64_t rcp14_rounded(uint64_t bits) {
    double x; memcpy(&x, &bits, 8);
    double y[8]; _mm512_storeu_pd(y, _mm512_rcp14_pd(_mm512_set1_pd(x)));
    uint64_t r; memcpy(&r, &y[0], 8);
    return (r + (1ull << 47)) & 0xffff000000000000ull;      /* RndAdd / RndMask of the kernel */
}
int scan(int bits, uint64_t *where, uint64_t *what, int cap) {
    uint64_t prev = rcp14_rounded(0x3ff0000000000000ull); int n = 0;
    for (uint64_t i = 1; i < (1ull << bits); ++i) {
        uint64_t mb = 0x3ff0000000000000ull | (i << (52 - bits));
        uint64_t r = rcp14_rounded(mb);
        if (r != prev) { if (n < cap) { where[n] = mb; what[n] = r; } ++n; prev = r; }
    }
    return n;
}
"""


def rcp14_thresholds():
    """Mantissa thresholds at which round4(VRCP14PD(x)) steps down, x in [1, 2)."""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "r.c")
        open(src, "w").write(RCP_SRC)
        so = os.path.join(d, "r.so")
        subprocess.check_call(["gcc", "-O2", "-mavx512f", "-shared", "-fPIC", "-o", so, src])
        lib = ctypes.CDLL(so)
        where = (ctypes.c_uint64 * 64)()
        what = (ctypes.c_uint64 * 64)()
        lib.scan.restype = ctypes.c_int
        n = lib.scan(30, where, what, 64)
        assert n == 16, n
        thr = [int(where[i]) for i in range(n)]
        for i in range(n):       # steps go 1.0 -> 31/32 -> ... -> 16/32, one at a time
            assert int(what[i]) == 0x3ff0000000000000 - ((i + 1) << 48), hex(what[i])
            assert thr[i] & ((1 << 36) - 1) == 0          # on the 2^-16 mantissa grid
        # random full-mantissa inputs agree with the step function (the low bits do not matter)
        lib.rcp14_rounded.restype = ctypes.c_uint64
        lib.rcp14_rounded.argtypes = [ctypes.c_uint64]
        rng = np.random.default_rng(5)
        tests = list(rng.integers(0, 1 << 52, size=20000, dtype=np.uint64))
        for t in thr:
            for dlt in (-3, -2, -1, 0, 1, 2, 3):
                tests.append(np.uint64((t & ((1 << 52) - 1)) + dlt))
        for mant in tests:
            bits = 0x3ff0000000000000 | int(mant)
            k = sum(1 for t in thr if bits >= t)
            assert lib.rcp14_rounded(bits) == 0x3ff0000000000000 - (k << 48)
            for e in (-30, -7, -1, 3):                        # scale invariance (exponent only)
                b2 = bits + (e << 52)
                assert lib.rcp14_rounded(b2) == 0x3ff0000000000000 - (k << 48) - (e << 52)
        return [(t >> 36) & 0xffff for t in thr]


def hexf(u):
    return float.hex(np.uint64(u).view(np.float64).item())


def main():
    so = numpy_so()
    syms = symbol_table(so)
    secs = sections(so)
    addr = data_addresses(so, syms)
    t = np.frombuffer(read_va(so, secs, addr["__svml_dtanh_data_internal"], 0x2980), dtype="<u8")
    a = np.frombuffer(read_va(so, secs, addr["__svml_datanh_ha_data_internal_avx512"], 0x500), dtype="<u8")
    t32 = t.view("<u4")

    # ---- tanh: 16 intervals x (shifter, c0, c1 .. c16) ------------------------------------------------
    assert t32[0x980 // 4] == 0x7ff80000 and t32[0x9c0 // 4] == 0x3fc00000 and t32[0xa00 // 4] == 0x780000
    assert t[0x2880 // 8] == 0x8000000000000000 and t[0x28c0 // 8] == 0x7fffffffffffffff
    assert t32[0x2940 // 4] == 0x7fe00000
    shifter = t[0:16]
    coef_offsets = [0x80] + [0x180 + 0x80 * i for i in range(16)]           # c0, c1 .. c16 (0x100 unused)
    coefs = [t[o // 8:o // 8 + 16] for o in coef_offsets]
    # ---- arctanh ----------------------------------------------------------------------------------------
    assert a[0x100 // 8] == 0x3ff0000000000000 and a[0x180 // 8] == 1 << 47 and a[0x1c0 // 8] == 0xffff000000000000
    assert a[0x4c0 // 8] == 0x3fe0000000000000
    log_hi = a[0:16]
    log_lo = a[0x80 // 8:0x80 // 8 + 16]
    poly = [a[o // 8] for o in range(0x200, 0x440, 0x40)]                   # 9 coefficients, highest first
    ln2_hi, ln2_lo = a[0x440 // 8], a[0x480 // 8]
    thr16 = rcp14_thresholds()

    def body(ns_open, ns_close, const, comment):
        L = []
        L.append("// GENERATED by tools/extract_np_svml.py -- do not edit.")
        L.append("// Constants of numpy " + np.__version__ + "'s float64 tanh / arctanh kernels on AVX512_SKX hosts (the SVML routines")
        L.append("// numpy vendors: svml_z0_tanh_d_la.s / svml_z0_atanh_d_ha.s, BSD-3-Clause (c) Intel Corporation), read from the")
        L.append("// installed numpy binary; RCP14_THR16: where round-to-4-bits(VRCP14PD(x)) steps, tabulated on the build host.")
        L.append(comment)
        L.append("#pragma once")
        L.append("#include <stdint.h>")
        L.append(ns_open)
        L.append(f"{const} uint64_t NP_TANH_SHIFTER[16] = {{")
        L.append("    " + ", ".join(f"0x{int(v):016x}ull" for v in shifter) + "};   // interval midpoints")
        L.append(f"// NP_TANH_COEF[k][i]: coefficient of r^k in interval i, r = |x| - shifter[i]")
        L.append(f"{const} uint64_t NP_TANH_COEF[17][16] = {{")
        for k, row in enumerate(coefs):
            L.append("    {" + ", ".join(f"0x{int(v):016x}ull" for v in row) + "},")
        L.append("};")
        L.append(f"{const} uint64_t NP_ATANH_LOG_HI[16] = {{" + ", ".join(f"0x{int(v):016x}ull" for v in log_hi) + "};   // log(1 + j/16), high part")
        L.append(f"{const} uint64_t NP_ATANH_LOG_LO[16] = {{" + ", ".join(f"0x{int(v):016x}ull" for v in log_lo) + "};")
        L.append(f"{const} uint64_t NP_ATANH_POLY[9] = {{" + ", ".join(f"0x{int(v):016x}ull" for v in poly) + "};   // log1p(r) = r + r^2 P(r), highest first")
        L.append(f"{const} uint64_t NP_ATANH_LN2_HI = 0x{int(ln2_hi):016x}ull, NP_ATANH_LN2_LO = 0x{int(ln2_lo):016x}ull;")
        L.append(f"// round4(VRCP14PD(x)) = (32 - k) / 32 for x in [1, 2), k = number of thresholds <= (top 16 mantissa bits of x)")
        L.append(f"{const} uint32_t NP_RCP14_THR16[16] = {{" + ", ".join(f"0x{v:04x}" for v in thr16) + "};")
        L.append(ns_close)
        return "\n".join(L) + "\n"

    open(os.path.join(ROOT, "oracle", "np_svml_tables.h"), "w").write(
        body("", "", "static const", "// Test infrastructure (CPU oracle)."))
    open(os.path.join(ROOT, "qldpc_amd", "csrc", "qbp_np_tables.hpp"), "w").write(
        body("namespace qbp {", "}  // namespace qbp", "static constexpr", "// Product side (HIP kernels)."))
    print("wrote oracle/np_svml_tables.h and qldpc_amd/csrc/qbp_np_tables.hpp")
    print("poly", [hexf(v) for v in poly])


if __name__ == "__main__":
    sys.exit(main())
