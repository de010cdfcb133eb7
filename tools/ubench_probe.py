#!/usr/bin/env python3
"""Prints what tools/ubench/valu_rates.hip measures on this GPU: per-class issue rates and cycles per
instruction (4 waves per SIMD), the shader clock from `s_nop` timing and from the cycle counter."""
import ctypes
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libvalu_rates.so"))
L.ubench_class_name.restype = ctypes.c_char_p
out = {}
c = 0
while True:
    name = L.ubench_class_name(c)
    if not name:
        break
    for w in (1, 2, 4):
        r, cy = ctypes.c_double(), ctypes.c_double()
        rc = L.ubench_valu_rate2(0, c, w, ctypes.byref(r), ctypes.byref(cy))
        out[f"{name.decode()} w{w}"] = {"rc": rc, "Gwave_insts_per_s": round(r.value / 1e9, 1),
                                        "cycles_per_inst_per_simd": round(cy.value, 3),
                                        "implied_clock_GHz": round(r.value / 1024 * cy.value / 1e9, 3)}
    c += 1
a, b = ctypes.c_double(), ctypes.c_double()
L.ubench_clock_ghz2(0, ctypes.byref(a), ctypes.byref(b))
out["clock_GHz_snop"] = a.value
out["clock_GHz_cycle_counter_over_wallclock"] = b.value
print(json.dumps(out, indent=1))
