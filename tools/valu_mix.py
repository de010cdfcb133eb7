#!/usr/bin/env python3
"""Static vector-instruction mix of a kernel's steady-state loop, read from the BUILT code object.

    python tools/valu_mix.py [--so qldpc_amd/csrc/libqbp.so] [--kernel SUBSTR ...] [--list]

bench.py prices the headline kernel against the FP64 vector-ALU issue roofline: it needs to know how
many vector instructions of which kind one wave executes per BP iteration.  That number is a
property of the machine code, so it is read from the machine code of the very library the run
loads, not from a committed profile:

1. the gfx950 code objects are cut out of the library's ``.hip_fatbin`` section (one clang offload
   bundle per translation unit) and disassembled with ROCm's ``llvm-objdump``;
2. the kernel's main loop is the backward branch with the largest span;
3. inside the loop, forward conditional branches delimit if-regions; a region is COLD when its own
   instructions (not those of nested regions) touch global memory or the kernel-argument segment --
   in these kernels that is exactly the once-per-syndrome work (loading a syndrome, emitting outputs,
   drawing the next work item), which is skipped by its branch in all but one of max_iter iterations;
   the kernels' one rarely taken ARITHMETIC path -- the |t| < 1e-15 form of the check step, entered only when
   a row's product is below 1e-15 -- is bracketed by explicit markers instead (`s_nop 9` ... `s_nop 10`);
4. every other vector-ALU instruction in the loop is counted, by class (the classes the in-run
   microbenchmark tools/ubench/valu_rates.hip measures issue rates for).

Limits: a build that re-reads a kernel argument inside the hot loop (the damped and min-sum instantiations do,
for alpha / damping, under scalar-register pressure) defeats rule 3 -- the numbers are used, and checked, for the
sum-product kernels only.

The result is checked against hardware counters in profiles/ (SQ_INSTS_VALU and the per-class
SQ_INSTS_VALU_* counters of the same kernel: see DESIGN.md section 4).
"""
import argparse
import collections
import hashlib
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

# issue classes (one microbenchmark each); order matters: first match wins
CLASSES = [
    ("trans_f64", re.compile(r"^v_(rcp|rsq|sqrt)_f64")),
    ("fma_f64", re.compile(r"^v_fma_f64|^v_fmac_f64")),
    ("mul_f64", re.compile(r"^v_mul_f64")),
    ("add_f64", re.compile(r"^v_add_f64")),
    ("minmax_f64", re.compile(r"^v_(min|max)_f64")),
    ("cmp_f64", re.compile(r"^v_cmp[x]?_\w+_f64")),
    ("cvt_f64", re.compile(r"^v_(rndne|trunc|floor|ceil|fract)_f64|^v_cvt_\w*f64|^v_(ldexp|frexp_\w+)_f64")),
    ("other_f64", re.compile(r"^v_\w+_f64")),
    ("lane_b32", re.compile(r"^v_(readlane|writelane|readfirstlane)_b32|^v_accvgpr")),
    ("alu_b32", re.compile(r"^v_")),
]
COLD = re.compile(r"^(global_|flat_|buffer_|s_load_|s_buffer_load|s_atomic|s_store|s_scratch)")


def extract_code_objects(so_path):
    """All gfx950 code objects of the library (one clang offload bundle per translation unit)."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fatbin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}",
                               so_path, os.path.join(td, "copy.so")])
        d = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, at = [], d.find(magic)
    if at < 0:
        raise RuntimeError("no uncompressed clang offload bundle in .hip_fatbin")
    while at >= 0:
        n = struct.unpack_from("<Q", d, at + 24)[0]
        off = at + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", d, off)
            off += 24
            triple = d[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and sz:
                out.append(d[at + o:at + o + sz])
        at = d.find(magic, at + len(magic))
    if not out:
        raise RuntimeError("no gfx950 code object in the bundles")
    return out


def disassemble(so_path):
    """{symbol: [(addr, mnemonic, operands, branch_target_or_None)]}"""
    funcs = {}
    head = re.compile(r"^([0-9a-f]+) <(\S+)>:")
    line = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<\S+?\+0x([0-9a-f]+)>)?\s*$")
    for co in extract_code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", f.name], text=True)
        cur, base = None, 0
        for ln in txt.splitlines():
            mh = head.match(ln)
            if mh:
                base, cur = int(mh.group(1), 16), mh.group(2)
                funcs[cur] = []
                continue
            ml = line.match(ln)
            if ml and cur is not None:
                mnem, ops, addr, tgt = ml.group(1), ml.group(2), int(ml.group(3), 16), ml.group(4)
                target = None
                if mnem.startswith("s_cbranch") or mnem == "s_branch":
                    if tgt is not None:
                        target = base + int(tgt, 16)
                    elif re.search(r"<\S+>", ln):            # branch to the symbol itself (+0x0)
                        target = base
                funcs[cur].append((addr, mnem, ops, target))
    return funcs


def classify(mnem):
    for name, rx in CLASSES:
        if rx.match(mnem):
            return name
    return None


def demangle(sym):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            return subprocess.check_output([tool, sym], text=True, stderr=subprocess.DEVNULL).strip()
        except Exception:
            continue
    return sym


def loop_mix(insts):
    """VALU instructions per execution of the main loop's always-executed part, by class."""
    # the main loop: the phase loop of the kernel is the one that synchronises -- of all backward branches whose
    # span contains a workgroup barrier (inner loops -- sampling, emission, rare paths -- hold none), the one
    # whose span holds the most FP64 arithmetic, among equals the tightest (the compiler rotates the phase loop
    # into a steady-state loop and an outer path taken once per syndrome; out-of-line cold blocks behind the
    # loop jump back over it too); kernels without barriers (the streaming kernel): the same without the
    # barrier condition
    addrs = [a for a, m, o, t in insts if m.startswith("v_") and "_f64" in m]
    bars = [a for a, m, o, t in insts if m.startswith("s_barrier")]
    back = []
    for a, m, o, t in insts:
        if t is not None and t <= a:
            nbar = sum(1 for x in bars if t <= x <= a)
            back.append((1 if nbar else 0, sum(1 for x in addrs if t <= x <= a), -(a - t), t, a))
    if not back:
        return None
    _, _, _, lo, hi = max(back)
    body = [(a, m, o, t) for a, m, o, t in insts if lo <= a <= hi]
    # if-regions from forward conditional branches inside the loop: (start, end) half-open
    # (a branch to an out-of-line block behind the loop guards everything up to the loop's end)
    regions = sorted((a, min(t, hi + 4)) for a, m, o, t in body
                     if t is not None and m.startswith("s_cbranch") and a < t)
    # nesting by containment; direct coldness
    def innermost(addr):
        best = None
        for s, e in regions:
            if s < addr < e and (best is None or (e - s) < (best[1] - best[0])):
                best = (s, e)
        return best
    cold_regions = set()
    for a, m, o, t in body:
        if COLD.match(m):
            r = innermost(a)
            if r is not None:
                cold_regions.add(r)
    # explicit markers: the kernels bracket arithmetic that does not run every iteration (the |t| < 1e-15 path
    # of the check step) with `s_nop 9` ... `s_nop 10` (qbp_kernels.hpp: QBP_COLD_BEGIN / QBP_COLD_END)
    marks = sorted((a, o.strip()) for a, m, o, t in body if m == "s_nop" and o.strip() in ("9", "10"))
    marked = []
    open_at = None
    for a, which in marks:
        if which == "9":
            if open_at is not None:
                raise RuntimeError(f"cold marker at {open_at:#x} not closed before {a:#x}")
            open_at = a
        else:
            if open_at is None:
                raise RuntimeError(f"cold end marker at {a:#x} without a begin")
            marked.append((open_at, a))
            open_at = None
    if open_at is not None:
        raise RuntimeError(f"cold marker at {open_at:#x} not closed")
    def in_cold(addr):
        return any(s < addr < e for s, e in cold_regions) or any(s <= addr <= e for s, e in marked)
    mix = collections.Counter()
    other = collections.Counter()
    mnems = collections.Counter()
    for a, m, o, t in body:
        if in_cold(a):
            continue
        c = classify(m)
        if c:
            mix[c] += 1
            mnems[m] += 1
        elif m.startswith("ds_"):
            other["lds"] += 1
        elif m.startswith("s_barrier"):
            other["barrier"] += 1
        elif m.startswith("scratch_"):
            other["scratch"] += 1
        elif m.startswith("s_"):
            other["salu"] += 1
    cold_valu = sum(1 for a, m, o, t in body if in_cold(a) and classify(m))
    return {"loop": [hex(lo), hex(hi)], "valu_by_class": dict(mix), "valu_total": sum(mix.values()),
            "valu_in_cold_regions": cold_valu, "other": dict(other),
            "mnemonics": dict(mnems.most_common())}


def analyse(so_path, patterns):
    funcs = disassemble(so_path)
    out = {}
    for sym, insts in funcs.items():
        name = demangle(sym)
        if patterns and not any(p in name or p in sym for p in patterns):
            continue
        r = loop_mix(insts)
        if r:
            out[name] = r
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "qldpc_amd", "csrc", "libqbp.so"))
    ap.add_argument("--kernel", nargs="*", default=["bp_fused_kernel<6, 3, 0, false, true"])
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--mnemonics", action="store_true")
    args = ap.parse_args()
    if args.list:
        for sym in disassemble(args.so):
            print(demangle(sym))
        return
    res = analyse(args.so, args.kernel)
    if not args.mnemonics:
        for v in res.values():
            v.pop("mnemonics", None)
    res["_library_sha256"] = hashlib.sha256(open(args.so, "rb").read()).hexdigest()
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
