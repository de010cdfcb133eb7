#!/usr/bin/env python3
"""Headline benchmark: syndromes/s of the [[288,12,18]] sum-product BP decode at 50 iterations.

    python bench.py [--gpus N --steps K --warmup W]

``--gpus N`` with N > 1 starts itself: unless a launcher already set WORLD_SIZE, the process spawns
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`` as a
child BEFORE touching torch or HIP (qldpc_amd/launch.py) and exits with its status.  Under an external
launcher (the driver's torch.distributed.run) it is simply one of the ranks.

One "step" = one pass of the hot path (qbp_decode_batch_device, include/qbp.h) over one batch of
synthetic syndromes already resident in HBM: BASELINE.json configs[3], [[288,12,18]], sum-product,
max_iter 50, 1M trials over 8 GPUs = 125 000 syndromes per GPU (weak scaling).  `value` is measured
in mode M2 of SURVEY.md 8(d): every syndrome runs all 50 iterations (QBP_FLAG_FORCE_FULL; outputs are
still those of the first converged iteration).  Everything else in the line is measured by this run:

* ``roofline``       the binding resource of the on-chip kernel is FP64 vector-ALU issue, not HBM.
                     achieved = vector instructions per wave and iteration, counted in the machine
                     code of the library this process loaded (tools/valu_mix.py), x executions / kernel
                     time; peak = issue rate of a stall-free stream with the same instruction-class mix,
                     measured now on this chip (tools/ubench/valu_rates.hip); frac = achieved / peak.
* ``hbm_effective``  SURVEY 8(d)'s algorithmic bytes over the kernel time (exceeds the HBM peak: the
                     messages it counts never leave the CU) -- kept as the north star's own yardstick.
* ``early_exit``     reference semantics (mode M1); ``stress_p010`` forced-50 on a p = 0.10 batch BP
                     cannot decode (M2 stress of SURVEY 8(d)); ``fast_math_flag`` the same launch with the opt-in
                     QBP_FLAG_FAST_MATH (not bit-exact: never the headline); ``sustained`` >= 10 s of back-to-back
                     forced-50 launches and the shader clock right after; ``hbm_streamed_variant`` the
                     one-lane-per-syndrome kernel whose messages do go through HBM; ``dropin_api`` the
                     reference's own calling pattern (paperResults_GPU.py:95-144) through the Python
                     mirror, host arrays and PCIe included; ``cpu_baseline`` the CPU oracle on this
                     box's cores (+ the reference's Python figure, a labelled constant measured in the
                     build container: the reference cannot travel to the GPU box).
PyTorch only provides device memory, the stream, events and torch.distributed.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CODE = "[[288, 12, 18]]"
BATCH_PER_GPU = 125_000
MAX_ITER = 50
P_ERR = 0.01
HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
# decoding/beliefPropagation.py:88 performBeliefPropagationFast on [[288,12,18]], forced 50
# iterations, one core of the build container's 8-vCPU Xeon @ 2.1 GHz (re-measured in round 1 with
# time.perf_counter over 200 syndromes; SURVEY.md 6.2 had 24.1): the reference is single-threaded
# Python and cannot run on the GPU box, so this is a constant with its provenance, not a measurement
# of this run.
REFERENCE_PYTHON_SYN_PER_S = 19.7


def algorithmic_bytes(E, m, n, iters_total, B):
    """per iteration every edge message is read and written once in each
    direction (4 * E * 8 B, FP64); per syndrome m + n + 8n + 5 B of I/O."""
    return iters_total * 4 * E * 8 + B * (m + n + 8 * n + 5)


def cpu_baseline(code, syndromes, prior, budget_s=12.0):
    """The CPU oracle (oracle/bp_oracle.c: a port of decoding/beliefPropagation.py:88-144), forced
    50 iterations, on a bounded sample of the same syndromes: one host thread (the reference is
    single-threaded), and the same scalar code spread over this GPU's share of the host cores."""
    from oracle import oracle
    t0 = time.perf_counter()
    oracle.decode_batch(code.Hx, syndromes[:32], prior, MAX_ITER, flags=oracle.FLAG_FORCE_FULL)
    per = (time.perf_counter() - t0) / 32
    nsamp = int(max(64, min(len(syndromes), budget_s / per)))
    t0 = time.perf_counter()
    oracle.decode_batch(code.Hx, syndromes[:nsamp], prior, MAX_ITER, flags=oracle.FLAG_FORCE_FULL)
    dt = time.perf_counter() - t0
    out = {"value": nsamp / dt, "unit": "syndromes/s", "cores": 1, "kind": "port",
           "sample": f"first {nsamp} syndromes of the same batch, forced {MAX_ITER} iterations, "
                     f"{dt:.1f} s on 1 of {os.cpu_count()} host cores (oracle/bp_oracle.c: numpy's own tanh / "
                     "arctanh kernels restated in scalar C, the same bits as the reference)"}
    host = host_cpu_info()
    out["host"] = host
    # OpenMP over syndromes (the decoder itself stays the scalar restatement): this job's share of the host (a
    # 1-GPU box is given 16 cores' worth) and -- SURVEY 8(d)(ii) -- one worker per physical core this process may
    # run on
    for key, threads in (("multi_thread", max(1, min(16, host["usable_logical_cpus"]))),
                         ("all_physical_cores", max(1, min(host["physical_cores"], host["usable_logical_cpus"])))):
        if threads <= 1 or (key == "all_physical_cores" and threads == out.get("multi_thread", {}).get("cores")):
            continue
        nmt = int(min(len(syndromes), max(2 * threads, nsamp * min(threads, 8) // (2 if key == "all_physical_cores" else 1))))
        t0 = time.perf_counter()
        oracle.decode_batch(code.Hx, syndromes[:nmt], prior, MAX_ITER, flags=oracle.FLAG_FORCE_FULL,
                            threads=threads)
        dtm = time.perf_counter() - t0
        out[key] = {"value": nmt / dtm, "unit": "syndromes/s", "cores": threads,
                    "sample": f"{nmt} syndromes, {dtm:.1f} s, OpenMP over syndromes, {host['model']}"}
    out["reference_python"] = {
        "value": REFERENCE_PYTHON_SYN_PER_S, "unit": "syndromes/s", "cores": 1, "kind": "reference",
        "measured_by_this_run": False,
        "sample": "decoding/beliefPropagation.py:88 performBeliefPropagationFast, [[288,12,18]], forced "
                  "50 iterations, 200 syndromes, 1 core of the BUILD container (8-vCPU Xeon @ 2.1 GHz); "
                  "the reference cannot travel to the GPU box"}
    return out


def host_cpu_info():
    """CPU model and core counts of this host (lscpu), and how many logical CPUs this process may use."""
    import subprocess
    info = {"model": "unknown", "physical_cores": os.cpu_count() or 1, "logical_cpus": os.cpu_count() or 1,
            "usable_logical_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)}
    try:
        kv = {}
        for ln in subprocess.check_output(["lscpu"], text=True, timeout=10).splitlines():
            if ":" in ln:
                a, b = ln.split(":", 1)
                kv[a.strip()] = b.strip()
        info["model"] = kv.get("Model name", "unknown")
        info["physical_cores"] = int(kv.get("Core(s) per socket", "1")) * int(kv.get("Socket(s)", "1"))
        info["logical_cpus"] = int(kv.get("CPU(s)", info["logical_cpus"]))
    except Exception:
        pass
    return info


# ---- FP64 vector-ALU roofline, measured by this run -------------------------------------------------
MAX_CLOCK_HZ = 2.4e9        # MI355X_MICROARCH.md: max shader clock (the in-run probe reads 2.39-2.41 GHz)


def _ubench():
    so = os.path.join(ROOT, "tools", "ubench", "libvalu_rates.so")
    if not os.path.exists(so):
        raise RuntimeError(f"{so} missing: run __graft_entry__.build()")
    L = ctypes.CDLL(so)
    L.ubench_class_name.restype = ctypes.c_char_p
    L.ubench_valu_rate2.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_double)]
    L.ubench_clock_ghz2.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    return L


def valu_issue_costs(device, waves_per_simd=4):
    """Per instruction class, measured now (tools/ubench/valu_rates.hip, every wave of a chip-filling grid
    issuing independent instructions of that class): shader-clock cycles ONE SIMD needs per
    wave-instruction (from the hardware cycle counter, so independent of the clock the chip chose for that
    stream), and the wave-instructions/s the whole chip reached."""
    L = _ubench()
    cycles, rates, c = {}, {}, 0
    while True:
        name = L.ubench_class_name(c)
        if not name:
            break
        r, cy = ctypes.c_double(), ctypes.c_double()
        rc = L.ubench_valu_rate2(device, c, waves_per_simd, ctypes.byref(r), ctypes.byref(cy))
        if rc != 0:
            raise RuntimeError(f"ubench_valu_rate2({name.decode()}) failed: {rc}")
        cycles[name.decode()], rates[name.decode()] = cy.value, r.value
        c += 1
    return cycles, rates


def shader_clock_ghz(device):
    """Shader clock from a one-wave probe on its own stream (hardware cycle counter against the
    100 MHz wall clock): called while other streams are busy it reads the clock held under that load."""
    L = _ubench()
    g, g2 = ctypes.c_double(), ctypes.c_double()
    return g2.value if L.ubench_clock_ghz2(device, ctypes.byref(g), ctypes.byref(g2)) == 0 else None


PMC_SUMMARY = os.path.join(ROOT, "profiles", "r03_pmc_summary.json")     # tools/profile_r03.sh -> tools/pmc_collect.py


def committed_pmc(kernel_name):
    """Counters of `kernel_name` from the committed rocprofv3 PMC summary (collected with tools/profile_r03.sh on
    the same tree: PMC passes cannot run inside this process), or None."""
    try:
        allk = json.load(open(PMC_SUMMARY))
    except Exception:
        return None
    short = kernel_name.replace("void ", "").split("(qbp::")[0]
    for k, v in allk.items():
        if short in k and "counters" in v:
            return v
    return None


IDEAL_ISSUE_CYCLES = {"trans_f64": 16.0}         # every other class: 4 cycles per wave-instruction per SIMD
FLOPS_PER_LANE = {"fma_f64": 2, "mul_f64": 1, "add_f64": 1}      # "useful" FP64 arithmetic (min/max, compares,
#                                                                     conversions and reciprocal seeds not counted)


def valu_roofline(lib_path, kernel_symbol, device, wave_iterations, kernel_s, num_cu, iters_per_trip=1,
                  costs=None):
    """Vector-ALU issue roofline.  needed = issue cycles of the kernel's own instructions: per wave
    and BP iteration, sum over classes of (instructions counted in the machine code of the loaded
    library) x (cycles one SIMD needs per instruction of that class, measured now), times the
    wave-iterations of a launch.  available = SIMDs x kernel time x the maximum shader clock.
    frac = needed / available <= 1; achieved / peak are the same statement in lane-instructions/s.
    Beside it: the same fraction with the IDEAL costs (4 cycles, 16 for v_rcp_f64), the useful FP64 flop rate
    against the 78.6 TFLOP/s vector peak, and -- from the committed PMC summary of the same kernel -- the
    hardware's own instruction count and the HBM bytes per launch."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import valu_mix
    res = valu_mix.analyse(lib_path, [kernel_symbol])
    if len(res) != 1:
        raise RuntimeError(f"{len(res)} kernels match {kernel_symbol!r} in {lib_path}")
    name, mix = next(iter(res.items()))
    # (the one-barrier forced kernel's loop is unrolled by two: one trip = two BP iterations)
    per_class = {c: n / iters_per_trip for c, n in mix["valu_by_class"].items()}
    mix = dict(mix, valu_total=mix["valu_total"] / iters_per_trip)
    if "other_f64" in per_class:                                   # priced like an FMA
        per_class["fma_f64"] = per_class.get("fma_f64", 0) + per_class.pop("other_f64")
    n_total = mix["valu_total"]
    cycles, rates = costs if costs is not None else valu_issue_costs(device)
    issue_cycles = sum(n * cycles[c] for c, n in per_class.items())          # per wave-iteration
    ideal_cycles = sum(n * IDEAL_ISSUE_CYCLES.get(c, 4.0) for c, n in per_class.items())
    n_simd = num_cu * 4
    avail = n_simd * kernel_s * MAX_CLOCK_HZ
    frac = issue_cycles * wave_iterations / avail
    achieved = n_total * wave_iterations * 64 / kernel_s                      # lane-instructions / s
    flops = sum(FLOPS_PER_LANE.get(c, 0) * n for c, n in per_class.items()) * 64 * wave_iterations / kernel_s
    spec_peak = num_cu * 4 * 16 * 2 * MAX_CLOCK_HZ
    out = {
        "bound": "fp64_valu", "achieved": achieved / 1e12, "peak": achieved / frac / 1e12,
        "unit": "Tlane-instr/s", "frac": frac, "traffic": None,
        "self_measured": True,
        "frac_ideal_pricing": ideal_cycles * wave_iterations / avail,
        "flops_frac": flops / spec_peak, "useful_fp64_TFLOPs": flops / 1e12,
        "kernel": name, "kernel_ms": kernel_s * 1e3,
        "valu_insts_per_wave_iteration": n_total, "valu_by_class": per_class,
        "issue_cycles_per_inst_by_class": cycles,
        "issue_cycles_per_wave_iteration": issue_cycles,
        "wave_iterations_per_launch": wave_iterations, "simds": n_simd, "max_clock_GHz": MAX_CLOCK_HZ / 1e9,
        "single_class_rates_Gwave_insts_per_s": {k: v / 1e9 for k, v in rates.items()},
        "fma_f64_stream_TFLOPs": rates["fma_f64"] * 64 * 2 / 1e12,
        "spec_peak_TFLOPs_fp64_vector": spec_peak / 1e12,
        "how": "achieved: vector-ALU instructions per wave and BP iteration counted in the machine code of "
               "the loaded libqbp.so (tools/valu_mix.py: the kernel's main loop minus its once-per-syndrome "
               "regions and the explicitly marked rare path) x wave-iterations per launch / kernel time (HIP "
               "events on the launch stream).  peak: the rate at which this instruction mix would issue if "
               "every issue cycle of every SIMD were used at the maximum shader clock; the per-class issue "
               "costs (cycles per wave-instruction per SIMD) are measured in this run with the hardware cycle "
               "counter (tools/ubench/valu_rates.hip).  frac_ideal_pricing: the same with 4 cycles per "
               "instruction and 16 per v_rcp_f64.  flops_frac: useful FP64 flops (2 per fma, 1 per mul / add) "
               "over the 78.6 TFLOP/s vector peak.  traffic and pmc: from the committed rocprofv3 PMC summary "
               "of this kernel (PMC passes cannot run inside the bench process)",
    }
    pm = committed_pmc(name)
    if pm:
        c = pm["counters"]
        prov = os.path.relpath(PMC_SUMMARY, ROOT)
        if "SQ_INSTS_VALU" in c:
            per_wi = c["SQ_INSTS_VALU"]["last"] / wave_iterations
            out["pmc"] = {"file": prov, "SQ_INSTS_VALU_per_wave_iteration": per_wi,
                          "static_count_over_pmc_count": n_total / per_wi,
                          "frac_with_pmc_count_ideal_pricing": (
                              (c["SQ_INSTS_VALU"]["last"] - c.get("SQ_INSTS_VALU_TRANS_F64", {}).get("last", 0.0)) * 4.0
                              + c.get("SQ_INSTS_VALU_TRANS_F64", {}).get("last", 0.0) * 16.0) / avail,
                          "by_class_per_wave_iteration": {k: v["last"] / wave_iterations for k, v in c.items()
                                                          if k.startswith("SQ_INSTS_")},
                          "lds_bank_conflict_share": (c["SQ_LDS_BANK_CONFLICT"]["last"] / c["SQ_LDS_IDX_ACTIVE"]["last"]
                                                      if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", {}).get("last") else None)}
        if "hbm" in pm:
            out["traffic"] = pm["hbm"]["traffic_bytes_per_launch"]
            out["traffic_detail"] = dict(pm["hbm"], file=prov, unit="bytes per launch",
                                         note="FETCH_SIZE x 2 (gfx950 correction for wide reads) + WRITE_SIZE, separate "
                                              "PMC passes; against the algorithmic figure of hbm_effective this is "
                                              "the syndrome / LLR I/O only")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="syndromes per GPU per step")
    ap.add_argument("--p", type=float, default=P_ERR)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=("both", "forced"), default="both",
                    help="forced: only the headline mode (profiling runs)")
    ap.add_argument("--sustained-seconds", type=float, default=10.0,
                    help="length of the back-to-back forced-50 leg (0 = skip; skipped in --mode forced)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 "
                         "path on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    args = ap.parse_args()

    from qldpc_amd import launch
    launch.maybe_self_launch(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:])

    import torch
    import torch.distributed as dist

    from qldpc_amd import _lib, bp, codes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_device:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    full = args.mode == "both"

    code = codes.load_code(CODE)
    m, n = code.Hx.shape
    E = int(code.Hx.sum())
    B = args.batch
    dec = bp.decoder_for(code.Hx, device=local_rank)
    if args.slots:
        dec.set_option(_lib.OPT_SLOTS_PER_BLOCK, args.slots)
    if args.blocks_per_cu:
        dec.set_option(_lib.OPT_BLOCKS_PER_CU, args.blocks_per_cu)

    # synthetic data of the reference's shape: i.i.d. Bernoulli(p) errors
    # (beliefPropagationGPU.py:195), syndrome = H e mod 2 (:198), prior = log((1-p)/p) (main.py:18).
    # Each rank draws its own shard (seed = rank): no data-path collective.
    Ht = torch.from_numpy(code.Hx.T.astype(np.float32)).to(dev)

    def draw(p, seed):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        e = (torch.rand((B, n), generator=g, device=dev) < p)
        return (e.float() @ Ht).remainder_(2).to(torch.uint8).contiguous()

    syndromes = draw(args.p, 1234 + rank)
    prior = torch.full((n,), float(np.log((1 - args.p) / args.p)), dtype=torch.float64, device=dev)
    hard = torch.empty((B, n), dtype=torch.uint8, device=dev)
    conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    llr = torch.empty((B, n), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step(flags, syn=None, pr=None):
        dec.decode_device((syndromes if syn is None else syn).data_ptr(),
                          (prior if pr is None else pr).data_ptr(), B, MAX_ITER, _lib.SUM_PRODUCT,
                          1.0, 1.0, 20.0, flags, hard.data_ptr(), conv.data_ptr(),
                          iters.data_ptr(), llr.data_ptr(), stream.cuda_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(flags, steps, warmup, syn=None, pr=None):
        for _ in range(warmup):
            step(flags, syn, pr)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for a, b in evs:
            a.record(stream)             # HIP events on the stream the kernel is launched on
            step(flags, syn, pr)
            b.record(stream)
        counts = torch.stack([conv.sum(dtype=torch.int64), iters.sum(dtype=torch.int64)])
        if world > 1:
            # the Monte-Carlo reduce of the north star: failure / iteration counts over RCCL
            dist.all_reduce(counts)
        barrier()
        wall = time.perf_counter() - t0
        timed.local_wall = wall                      # this rank's own clock around the same region
        if world > 1:
            w = torch.tensor([wall], dtype=torch.float64, device=dev)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            wall = float(w.item())
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        return wall, kernel_ms, counts.tolist()

    # ---- M2: forced 50 iterations (headline) ------------------------------------------------
    wall, kernel_ms, counts = timed(_lib.FLAG_FORCE_FULL, args.steps, args.warmup)
    headline_local_wall = timed.local_wall
    clock_after_headline = shader_clock_ghz(local_rank)
    value = world * B * args.steps / wall
    bytes_per_launch = algorithmic_bytes(E, m, n, B * MAX_ITER, B)
    geometry = {"threads_per_block": dec.info("threads"), "grid": dec.info("grid"),
                "lds_bytes": dec.info("lds_bytes")}
    kernel_kind = dec.info("last_kernel")
    one_bar = int(dec.info("one_barrier"))       # the headline launches ran the one-barrier forced kernel
    achieved = bytes_per_launch / (kernel_ms * 1e-3)

    # per-rank kernel times and the reduce on its own
    multi = None
    if world > 1:
        km = torch.zeros(world, dtype=torch.float64, device=dev)
        km[rank] = kernel_ms
        dist.all_reduce(km)
        t = torch.zeros(2, dtype=torch.int64, device=dev)
        for _ in range(3):
            dist.all_reduce(t)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(t)
        torch.cuda.synchronize(dev)
        ar_us = (time.perf_counter() - t0) / 20 * 1e6
        wl = torch.zeros(world, dtype=torch.float64, device=dev)
        wl[rank] = headline_local_wall
        dist.all_reduce(wl)
        ms_step = 1e3 * wall / args.steps
        # per rank: its kernel time, the rate it alone sustained over the timed region, and the share of a step
        # the one collective of the path (an int64 all-reduce of the counts) takes -- so that the N = 1 line can be
        # checked against the single-GPU bench and N = 8 against 8 x N = 1
        multi = {"n_ranks_seen": dist.get_world_size(), "backend": args.backend,
                 "kernel_ms_per_rank": [float(x) for x in km.tolist()],
                 "wall_s_per_rank": [float(x) for x in wl.tolist()],
                 "value_per_rank": [B * args.steps / float(x) for x in wl.tolist()],
                 "sum_of_value_per_rank": float(sum(B * args.steps / float(x) for x in wl.tolist())),
                 "all_reduce_us": ar_us,
                 "all_reduce_share_of_step": ar_us * 1e-3 / ms_step,
                 "kernel_share_of_step_per_rank": [float(x) / ms_step for x in km.tolist()]}

    # ---- M1: reference semantics (early exit) -----------------------------------------------
    # ---- the HBM-streamed design point (qbp_stream.hpp), same workload, forced 50 ----------------
    streamed = None
    if full and world == 1:
        # one lane per syndrome needs >= 256 CUs x 16 waves x 64 lanes to fill the chip: its own batch
        Bs = 262144
        reps = -(-Bs // B)
        syn_s = syndromes.repeat((reps, 1))[:Bs].contiguous()
        hard_s = torch.empty((Bs, n), dtype=torch.uint8, device=dev)
        conv_s = torch.empty((Bs,), dtype=torch.uint8, device=dev)
        iters_s = torch.empty((Bs,), dtype=torch.int32, device=dev)
        llr_s = torch.empty((Bs, n), dtype=torch.float64, device=dev)
        dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_STREAM)
        ms_s = []
        for i in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            dec.decode_device(syn_s.data_ptr(), prior.data_ptr(), Bs, MAX_ITER, _lib.SUM_PRODUCT, 1.0, 1.0,
                              20.0, _lib.FLAG_FORCE_FULL, hard_s.data_ptr(), conv_s.data_ptr(),
                              iters_s.data_ptr(), llr_s.data_ptr(), stream.cuda_stream)
            b.record(stream)
            torch.cuda.synchronize(dev)
            if i:
                ms_s.append(a.elapsed_time(b))
        dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_AUTO)
        kernel_ms_s = float(np.mean(ms_s))
        step(_lib.FLAG_FORCE_FULL)
        torch.cuda.synchronize(dev)
        same = bool(torch.equal(hard_s[:B], hard) and torch.equal(iters_s[:B], iters))
        del syn_s, hard_s, conv_s, iters_s, llr_s
        bytes_s = algorithmic_bytes(E, m, n, Bs * MAX_ITER, Bs)
        ach_s = bytes_s / (kernel_ms_s * 1e-3)
        streamed = {"kernel": "qbp::bp_stream_kernel<0>", "syndromes_per_launch": Bs,
                    "value": Bs / (kernel_ms_s * 1e-3), "unit": "syndromes/s per GPU",
                    "kernel_ms": kernel_ms_s, "same_results_as_default_kernel": same,
                    "roofline": {"bound": "hbm", "achieved": ach_s / 1e9, "peak": HBM_PEAK / 1e9,
                                 "unit": "GB/s", "frac": ach_s / HBM_PEAK, **stream_traffic()},
                    "note": "one lane per syndrome, messages streamed through HBM ([edge][syndrome] SoA): "
                            "here the algorithmic bytes ARE the physical traffic (PMC: profiles/); not the "
                            "default kernel"}

    early = stress = sustained = fast_math = None
    if full:
        wall1, kernel_ms1, counts1 = timed(0, args.steps, 1)
        bytes1 = algorithmic_bytes(E, m, n, counts1[1] / world + B, B)
        early = {"value": world * B * args.steps / wall1, "unit": "syndromes/s",
                 "mean_iterations": counts1[1] / (world * B) + 1.0, "kernel_ms": kernel_ms1,
                 "converged_fraction": counts1[0] / (world * B),
                 "effective_GBps": bytes1 / (kernel_ms1 * 1e-3) / 1e9}
        # ---- M2 stress: p = 0.10 syndromes BP does not decode -----------
        syn10 = draw(0.10, 4321 + rank)
        prior10 = torch.full((n,), float(np.log(0.9 / 0.1)), dtype=torch.float64, device=dev)
        steps10 = max(3, args.steps // 4)
        wall10, kernel_ms10, counts10 = timed(_lib.FLAG_FORCE_FULL, steps10, 1, syn10, prior10)
        stress = {"value": world * B * steps10 / wall10, "unit": "syndromes/s", "p": 0.10,
                  "kernel_ms": kernel_ms10, "converged_fraction": counts10[0] / (world * B),
                  "ratio_to_headline_kernel_ms": kernel_ms10 / kernel_ms,
                  "note": "same launch on a batch BP cannot decode: no data-dependent shortcut in forced mode"}
        del syn10
        # ---- QBP_FLAG_FAST_MATH: the opt-in approximate tanh / arctanh (NOT the headline: LLRs then differ from
        # the reference's in the last digits; decisions identical on every stored vector) ----------------
        wallf, kernel_msf, countsf = timed(_lib.FLAG_FORCE_FULL | _lib.FLAG_FAST_MATH, max(3, args.steps // 4), 1)
        fast_math = {"value": world * B * max(3, args.steps // 4) / wallf, "unit": "syndromes/s",
                     "kernel_ms": kernel_msf, "ratio_to_headline": kernel_ms / kernel_msf,
                     "converged_fraction": countsf[0] / (world * B),
                     "note": "forced-50 with QBP_FLAG_FAST_MATH (round 2's 2.3 / 1.2-ulp functions); "
                             "tests/test_gpu_fast_math.py states what it keeps and what it gives up"}
        # ---- sustained: back-to-back forced-50 launches for >= N seconds -------------------------
        if args.sustained_seconds > 0:
            n_launch = max(args.steps, int(np.ceil(args.sustained_seconds / (kernel_ms * 1e-3))))
            barrier()
            t0 = time.perf_counter()
            marks, sclk = [], []
            for i in range(n_launch):
                step(_lib.FLAG_FORCE_FULL)
                if i % max(1, n_launch // 10) == 0:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(stream)
                    marks.append((i, ev))
                    # stay about two tenths ahead of the GPU, not the whole leg: the clock probes below
                    # then fall at ~20 / 50 / 80 % of the leg's wall time, with work still queued
                    if len(marks) > 2:
                        marks[-3][1].synchronize()
                    if len(marks) in (4, 7, 10):
                        sclk.append(shader_clock_ghz(local_rank))
            barrier()
            wall_s = time.perf_counter() - t0
            clock_s = shader_clock_ghz(local_rank)
            if world > 1:
                w = torch.tensor([wall_s], dtype=torch.float64, device=dev)
                dist.all_reduce(w, op=dist.ReduceOp.MAX)
                wall_s = float(w.item())
            seg = [(marks[j + 1][0] - marks[j][0]) * B / (marks[j][1].elapsed_time(marks[j + 1][1]) * 1e-3)
                   for j in range(len(marks) - 1)]
            sustained = {"value": world * B * n_launch / wall_s, "unit": "syndromes/s", "seconds": wall_s,
                         "launches": n_launch, "ratio_to_headline": world * B * n_launch / wall_s / value,
                         "rank0_rate_by_tenth": [float(x) for x in seg],
                         "shader_clock_GHz_during": [x for x in sclk if x],
                         "shader_clock_GHz_right_after": clock_s}

    # ---- the reference's calling pattern through the drop-in API (host arrays, PCIe included) -----
    dropin = None
    if full and world == 1:
        dropin = dropin_leg(code, local_rank)
    costs = None
    others = None
    if rank == 0:
        try:
            costs = valu_issue_costs(local_rank)
        except Exception:
            costs = None
    if full and world == 1:
        try:
            others = other_config_legs(local_rank, dec.info("num_cu"))
        except Exception as ex:
            others = {"error": f"{type(ex).__name__}: {ex}"}

    if rank == 0:
        dc = 6 if (dec.info("max_row_deg") <= 6 and dec.info("max_col_deg") <= 3) else 8
        symbol = f"bp_fused_kernelILi{dc}ELi{3 if dc == 6 else 4}ELi0ELb0ELb1ELi1024ELi1ELb{one_bar}EE"
        S = max(1, geometry["threads_per_block"] // m)
        waves = (geometry["threads_per_block"] + 63) // 64
        wave_iters = B * MAX_ITER / S * waves
        try:
            if kernel_kind != 1:
                raise RuntimeError(f"the headline ran kernel kind {kernel_kind}, not the on-chip kernel")
            roof = valu_roofline(_lib.LIB_PATH, symbol, local_rank, wave_iters, kernel_ms * 1e-3,
                                 dec.info("num_cu"), iters_per_trip=2 if one_bar else 1, costs=costs)
            roof["bp_iterations_per_loop_trip"] = 2 if one_bar else 1
            roof["shader_clock_GHz_right_after"] = clock_after_headline
        except Exception as ex:             # a bench line without a roofline is still a bench line
            roof = {"bound": "fp64_valu", "achieved": None, "peak": None, "unit": "Tlane-instr/s",
                    "frac": None, "traffic": None, "error": f"{type(ex).__name__}: {ex}"}
        out = {
            "metric": "syndromes/sec at 50 BP iters, [[288,12,18]] code",
            "value": value, "unit": "syndromes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[3]: [[288,12,18]] BB code, sum-product BP, "
                                   "max_iter 50, every syndrome runs all 50 iterations (mode M2)",
                       "syndromes_per_gpu_per_step": B, "p": args.p, "max_iter": MAX_ITER,
                       "sharding": f"{world} x independent syndrome shards, one all-reduce of counts "
                                   f"({'RCCL' if args.backend == 'nccl' else args.backend})",
                       **geometry},
            "roofline": roof,
            "hbm_effective": {"achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK, "kernel_ms": kernel_ms,
                              "algorithmic_bytes_per_launch": bytes_per_launch,
                              "note": "SURVEY 8(d) algorithmic bytes / kernel time: NOT a roofline for this "
                                      "kernel (messages stay in LDS/registers, so the figure may exceed the "
                                      "HBM peak); physical HBM traffic is syndrome/LLR I/O only (PMC: "
                                      "profiles/)"},
            "multi_gpu": multi,
            "early_exit": early,
            "stress_p010": stress,
            "fast_math_flag": fast_math,
            "sustained": sustained,
            "hbm_streamed_variant": streamed,
            "other_configs": others,
            "dropin_api": dropin,
            "converged_fraction": counts[0] / (world * B),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(code, syndromes[:100000].cpu().numpy(),
                                               prior.cpu().numpy())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


PMC_LEGS = os.path.join(ROOT, "profiles", "r03_pmc_legs.json")       # tools/profile_r03.sh, tools/bench_legs.py --once


def stream_traffic():
    """HBM bytes per launch of the streaming kernel from the committed PMC summary (tools/profile_stream.sh: same
    script, same batch; FETCH_SIZE x 2 for gfx950 + WRITE_SIZE, passes of their own)."""
    path = os.path.join(ROOT, "profiles", "r03_stream_pmc_summary.json")
    try:
        d = json.load(open(path))
        return {"traffic": d["hbm_read_bytes_x2"] + d["hbm_write_bytes"],
                "traffic_detail": {"file": os.path.relpath(path, ROOT), "read_bytes_x2_gfx950": d["hbm_read_bytes_x2"],
                                   "write_bytes": d["hbm_write_bytes"],
                                   "physical_over_algorithmic": d["physical_over_algorithmic"],
                                   "physical_GBps_in_that_run": d["physical_GBps"]}}
    except Exception as ex:
        return {"traffic": None, "traffic_error": f"{type(ex).__name__}: {ex}"}


def other_config_legs(device, num_cu):
    """The other BASELINE.json configurations on one GPU (tools/bench_legs.py defines the launches): config 2
    ([[72,12,6]], p = 0.01, 10 000 syndromes) and config 3 ([[144,12,12]], min-sum alpha 0.8 / damping 0.7 / clip
    25, 100 000 syndromes), early exit (reference semantics) and forced 50; the device-resident Monte-Carlo loop of
    configs 4 / 5 ([[288,12,18]]: sample + decode + classify, with and without OSD-0); the OSD-0 kernel alone.
    Timed live (HIP events on the launch stream, inputs resident in HBM).  Each leg's `roofline` prices the
    vector instructions the HARDWARE counted for the very same launch (committed PMC summary: a static count of
    the loop does not work for kernels whose iterations differ -- early exit -- or re-read kernel arguments) at
    4 issue cycles each, 16 per v_rcp_f64, against SIMDs x time x the maximum shader clock."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_legs
    dev = torch.device("cuda", device)
    st = torch.cuda.current_stream(dev)
    try:
        pmc = json.load(open(PMC_LEGS))
    except Exception:
        pmc = {}
    out = {}
    for name, info, run, after in bench_legs.legs(device):
        run(); torch.cuda.synchronize(dev)
        ms = []
        for _ in range(5 if "config" in name else 3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); run(); b.record(st); torch.cuda.synchronize(dev)
            ms.append(a.elapsed_time(b))
        r = dict(info, **after())
        r["kernel_ms"] = float(np.median(ms))
        r["value"] = r["units"] / r["kernel_ms"] * 1e3
        if "iterations_total" in r:
            r["mean_iterations"] = r["iterations_total"] / r["units"]
            if "E" in info:
                r["hbm_effective_GBps"] = algorithmic_bytes(info["E"], info["m"], info["n"], r["iterations_total"],
                                                            r["units"]) / (r["kernel_ms"] * 1e-3) / 1e9
        disp = pmc.get(name)
        if disp and name == "osd0_288":
            disp = disp[1:]                     # (its first dispatch is the BP decode that makes the inputs: untimed;
                                                #  then the sweep and its -- empty -- redo pass)
        if disp:
            avail = num_cu * 4 * r["kernel_ms"] * 1e-3 * MAX_CLOCK_HZ
            valu = sum(d["counters"].get("SQ_INSTS_VALU", 0.0) for d in disp)
            trans = sum(d["counters"].get("SQ_INSTS_VALU_TRANS_F64", 0.0) for d in disp)
            wc = sum(d["counters"].get("SQ_WAVE_CYCLES", 0.0) for d in disp)
            main = next((d for d in reversed(disp) if "osd0_big_kernel" not in d["kernel"]), disp[-1])["counters"]
            roof = {"bound": "fp64_valu" if "osd0" not in name else "latency (dependent LDS / scalar chains)",
                    "achieved": valu * 64 / (r["kernel_ms"] * 1e-3) / 1e12,
                    "peak": num_cu * 4 * 16 * MAX_CLOCK_HZ / 1e12, "unit": "Tlane-instr/s",
                    "frac": ((valu - trans) * 4.0 + trans * 16.0) / avail, "traffic": None,
                    "pricing": "PMC instruction count of this launch x 4 issue cycles (16 per v_rcp_f64)",
                    "pmc_file": os.path.relpath(PMC_LEGS, ROOT), "kernels": [d["kernel"] for d in disp],
                    "SQ_INSTS_VALU": valu,
                    "pmc_kernel_ms": [next(iter(d["duration_ns"].values())) / 1e6 if d["duration_ns"] else None for d in disp]}
            if all("FETCH_SIZE" in d["counters"] and "WRITE_SIZE" in d["counters"] for d in disp):
                # KiB -> bytes; reads x 2 (gfx950: FETCH_SIZE counts 64 B per 128-B request), as for the headline
                rd = sum(d["counters"]["FETCH_SIZE"] for d in disp) * 1024.0
                wr = sum(d["counters"]["WRITE_SIZE"] for d in disp) * 1024.0
                roof["traffic"] = 2.0 * rd + wr
                roof["traffic_detail"] = {"read_bytes_raw": rd, "read_bytes_x2_gfx950": 2.0 * rd, "write_bytes": wr,
                                          "unit": "bytes per launch (PMC passes of their own, same launches)",
                                          "HBM_GBps_at_this_rate": (2.0 * rd + wr) / (r["kernel_ms"] * 1e-3) / 1e9}
            if main.get("SQ_WAVE_CYCLES"):
                w = main["SQ_WAVE_CYCLES"]
                roof["wave_cycle_shares_last_kernel"] = {
                    "waiting (s_waitcnt / barrier)": main.get("SQ_WAIT_ANY", 0) / w,
                    "issue stalled": main.get("SQ_WAIT_INST_ANY", 0) / w,
                    "issuing any instruction": main.get("SQ_ACTIVE_INST_ANY", 0) / w,
                    "issuing VALU": main.get("SQ_ACTIVE_INST_VALU", 0) / w,
                    "issuing LDS": main.get("SQ_ACTIVE_INST_LDS", 0) / w}
                if main.get("SQ_LDS_IDX_ACTIVE"):
                    roof["lds_bank_conflict_share"] = main.get("SQ_LDS_BANK_CONFLICT", 0) / main["SQ_LDS_IDX_ACTIVE"]
            r["roofline"] = roof
        else:
            r["roofline"] = {"bound": "fp64_valu", "frac": None, "traffic": None,
                             "error": f"no entry for {name} in {os.path.relpath(PMC_LEGS, ROOT)}"}
        out[name] = r
    return out


def dropin_leg(code, device, p=0.05, batch=5000, batches=4, max_iter=150):
    """paperResults_GPU.py:95-144 as its author would run it after swapping the import: host numpy
    sampling (two draws XORed), performBeliefPropagationBatch(code, syndromes[5000], prior,
    maxIter=150) with host arrays in and out, performOSD per BP failure, classification in numpy --
    next to the device-resident loop (qbp_mc_run) on the same trial count."""
    import torch

    from qldpc_amd import _lib, bp, mc, osd
    H, Lx = code.Hx, code.Lx.astype(np.int64)
    n = code.n
    rng = np.random.default_rng(0)
    prior = np.array([np.log((1 - p) / p)] * n)
    # warm-up with a full-size batch: the handle's device buffers and pinned staging are allocated on
    # the first call of a given size (one-time cost, not part of the steady-state rate)
    bp.performBeliefPropagationBatch(H, np.zeros((batch, H.shape[0]), np.int8), prior, maxIter=2)
    t_gen = t_bp = t_osd = t_cls = t_osd_batched = 0.0
    logical = osd_calls = 0
    for _ in range(batches):
        t0 = time.perf_counter()
        e1, s1 = bp.generate_errors_and_syndromes_batch(H, p, batch, rng)
        e2, s2 = bp.generate_errors_and_syndromes_batch(H, p, batch, rng)
        errors, syndromes = (e1 + e2) % 2, (s1 + s2) % 2
        t1 = time.perf_counter()
        det, conv, llrs = bp.performBeliefPropagationBatch(H, syndromes, prior, maxIter=max_iter)
        t2 = time.perf_counter()
        det = det.astype(np.int64)
        fails = np.flatnonzero(~conv)
        tb = time.perf_counter()
        batched = osd.performOSD_batch(H, syndromes[fails], llrs[fails], det[fails]) if len(fails) else None
        t_osd_batched += time.perf_counter() - tb
        tl = time.perf_counter()
        for i in fails:
            det[i] = osd.performOSD(H, syndromes[i], llrs[i], det[i])
            osd_calls += 1
        t3 = time.perf_counter()
        assert batched is None or np.array_equal(batched, det[fails])
        residual = (det + errors) % 2
        logical += int(((residual @ Lx.T) % 2).any(1).sum())
        t4 = time.perf_counter()
        t_gen += t1 - t0; t_bp += t2 - t1; t_osd += t3 - tl; t_cls += t4 - t3
    T = batch * batches
    # the same number of trials through the device-resident loop (sampling + BP + OSD-0 + classification)
    dec = bp.decoder_for(H, device=device)
    dec.mc_run(code.Lx, code.distance, p, prior, 0, 1000, draws=2, max_iter=max_iter, flags=_lib.FLAG_OSD0)
    t0 = time.perf_counter()
    c = dec.mc_run(code.Lx, code.distance, p, prior, 0, T, draws=2, max_iter=max_iter, flags=_lib.FLAG_OSD0)
    t_mc = time.perf_counter() - t0
    Tbig = 1 << 20
    t0 = time.perf_counter()
    dec.mc_run(code.Lx, code.distance, p, prior, 0, Tbig, draws=2, max_iter=max_iter, flags=_lib.FLAG_OSD0)
    t_mc_big = time.perf_counter() - t0
    torch.cuda.synchronize()
    return {"pattern": "paperResults_GPU.py:95-144 (two draws, batch 5000, maxIter 150, OSD on BP failures)",
            "p": p, "trials": T, "max_iter": max_iter,
            "trials_per_s_decode_call_only": T / t_bp,
            "trials_per_s_bp_plus_osd_calls": T / (t_bp + t_osd),
            "trials_per_s_bp_plus_one_batched_osd_call": T / (t_bp + t_osd_batched),
            "trials_per_s_whole_loop": T / (t_gen + t_bp + t_osd + t_cls),
            "seconds": {"host_sampling": t_gen, "performBeliefPropagationBatch": t_bp,
                        "performOSD_calls": t_osd, "performOSD_batch (one call per batch)": t_osd_batched,
                        "host_classification": t_cls},
            "osd_calls": osd_calls, "ler": logical / T,
            "device_resident_qbp_mc_run": {"trials_per_s_same_trial_count": T / t_mc,
                                           "trials_per_s_1M_trials": Tbig / t_mc_big,
                                           "ler": int(c[1]) / T},
            "note": "host-array signature: H2D of syndromes and D2H of hard/converged/LLR (8n bytes per trial) "
                    "inside the timed call; the per-failure performOSD calls are rows of the last batch and are "
                    "served from one batched OSD launch per batch (qldpc_amd/osd.py::_from_last_batch)"}


if __name__ == "__main__":
    main()
