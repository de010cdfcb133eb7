#!/usr/bin/env python3
"""Headline benchmark: syndromes/s of the [[288,12,18]] sum-product BP decode at 50 iterations.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (qbp_decode_batch_device, include/qbp.h) over one batch of
synthetic syndromes that is already resident in HBM: BASELINE.json configs[3], [[288,12,18]],
sum-product, max_iter 50, 1M trials over 8 GPUs = 125 000 syndromes per GPU (weak scaling).
`value` is measured in mode M2 of SURVEY.md 8(d): every syndrome runs all 50 iterations
(QBP_FLAG_FORCE_FULL; outputs are still those of the first converged iteration), which is what
"at 50 BP iters" means and what the algorithmic-byte roofline is defined on.  The reference's own
semantics (return at the first syndrome match, mode M1) is timed too and reported under
"early_exit".  PyTorch only provides device memory, the stream, events and torch.distributed.
"""
import argparse
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
                     f"{dt:.1f} s on 1 of {os.cpu_count()} host cores (oracle/bp_oracle.c)"}
    threads = max(1, min(16, os.cpu_count() or 1))      # a 1-GPU box's CPU share is 16 cores
    if threads > 1:
        nmt = int(min(len(syndromes), nsamp * min(threads, 8)))
        t0 = time.perf_counter()
        oracle.decode_batch(code.Hx, syndromes[:nmt], prior, MAX_ITER, flags=oracle.FLAG_FORCE_FULL,
                            threads=threads)
        dtm = time.perf_counter() - t0
        out["multi_thread"] = {"value": nmt / dtm, "unit": "syndromes/s", "cores": threads,
                               "sample": f"{nmt} syndromes, {dtm:.1f} s, OpenMP over syndromes"}
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
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 "
                         "path on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from qldpc_amd import _lib, bp, codes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

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
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    errors = (torch.rand((B, n), generator=g, device=dev) < args.p)
    Ht = torch.from_numpy(code.Hx.T.astype(np.float32)).to(dev)
    syndromes = (errors.float() @ Ht).remainder_(2).to(torch.uint8).contiguous()
    prior = torch.full((n,), float(np.log((1 - args.p) / args.p)), dtype=torch.float64, device=dev)
    hard = torch.empty((B, n), dtype=torch.uint8, device=dev)
    conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    llr = torch.empty((B, n), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step(flags):
        dec.decode_device(syndromes.data_ptr(), prior.data_ptr(), B, MAX_ITER, _lib.SUM_PRODUCT,
                          1.0, 1.0, 20.0, flags, hard.data_ptr(), conv.data_ptr(),
                          iters.data_ptr(), llr.data_ptr(), stream.cuda_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(flags, steps, warmup):
        for _ in range(warmup):
            step(flags)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for a, b in evs:
            a.record(stream)
            step(flags)
            b.record(stream)
        counts = torch.stack([conv.sum(dtype=torch.int64), iters.sum(dtype=torch.int64)])
        if world > 1:
            # the Monte-Carlo reduce of the north star: failure / iteration counts over RCCL
            dist.all_reduce(counts)
        barrier()
        wall = time.perf_counter() - t0
        if world > 1:
            w = torch.tensor([wall], dtype=torch.float64, device=dev)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            wall = float(w.item())
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        return wall, kernel_ms, counts.tolist()

    # ---- M2: forced 50 iterations (headline) ------------------------------------------------
    wall, kernel_ms, counts = timed(_lib.FLAG_FORCE_FULL, args.steps, args.warmup)
    value = world * B * args.steps / wall
    bytes_per_launch = algorithmic_bytes(E, m, n, B * MAX_ITER, B)
    geometry = {"threads_per_block": dec.info("threads"), "grid": dec.info("grid"),
                "lds_bytes": dec.info("lds_bytes")}
    achieved = bytes_per_launch / (kernel_ms * 1e-3)
    # ---- M1: reference semantics (early exit) -----------------------------------------------
    early = None
    if args.mode == "both":
        wall1, kernel_ms1, counts1 = timed(0, args.steps, 1)
        bytes1 = algorithmic_bytes(E, m, n, counts1[1] / world + B, B)
        early = {"value": world * B * args.steps / wall1, "unit": "syndromes/s",
                 "mean_iterations": counts1[1] / (world * B) + 1.0, "kernel_ms": kernel_ms1,
                 "converged_fraction": counts1[0] / (world * B),
                 "effective_GBps": bytes1 / (kernel_ms1 * 1e-3) / 1e9}

    # ---- the HBM-streamed design point (qbp_stream.hpp), same workload, forced 50 ----------------
    streamed = None
    if args.mode == "both":
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
        same = bool(torch.equal(hard_s[:B], hard) and torch.equal(iters_s[:B], iters))
        del syn_s, hard_s, conv_s, iters_s, llr_s
        bytes_s = algorithmic_bytes(E, m, n, Bs * MAX_ITER, Bs)
        ach_s = bytes_s / (kernel_ms_s * 1e-3)
        tr = None
        pf = os.path.join(ROOT, "profiles", "r01_stream_pmc_summary.json")
        if os.path.exists(pf):
            try:
                dd = json.load(open(pf))
                tr = {"physical_over_algorithmic": dd["physical_over_algorithmic"],
                      "physical_GBps_profiled": dd["physical_GBps"],
                      "source": "profiles/r01_stream_pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE, 262144-syndrome launch)"}
            except Exception:
                tr = None
        streamed = {"kernel": "qbp::bp_stream_kernel<0>", "syndromes_per_launch": Bs,
                    "value": Bs / (kernel_ms_s * 1e-3), "unit": "syndromes/s per GPU",
                    "kernel_ms": kernel_ms_s, "same_results_as_default_kernel": same,
                    "roofline": {"bound": "hbm", "achieved": ach_s / 1e9, "peak": HBM_PEAK / 1e9,
                                 "unit": "GB/s", "frac": ach_s / HBM_PEAK, "traffic": tr},
                    "note": "one lane per syndrome, messages streamed through HBM ([edge][syndrome] SoA): "
                            "here the algorithmic bytes ARE the physical traffic; not the default kernel"}

    if rank == 0:
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("bytes_per_launch_forced50")
            except Exception:
                traffic = None
        valu = None
        pf = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pf):
            try:
                dd = json.load(open(pf))["derived"]
                valu = {"note": "the physically binding resource (FP64 vector ALU), from the PMC passes in "
                                "profiles/r01_pmc_summary.json -- not measured by this run",
                        "valu_busy_fraction": dd["valu_busy_fraction"],
                        "valu_insts_per_syndrome_iteration": dd["valu_insts_per_syndrome_iteration"],
                        "shader_clock_GHz": dd["shader_clock_GHz"]}
            except Exception:
                valu = None
        out = {
            "metric": "syndromes/sec at 50 BP iters, [[288,12,18]] code",
            "value": value, "unit": "syndromes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[3]: [[288,12,18]] BB code, sum-product BP, "
                                   "max_iter 50, every syndrome runs all 50 iterations (mode M2)",
                       "syndromes_per_gpu_per_step": B, "p": args.p, "max_iter": MAX_ITER,
                       "sharding": f"{world} x independent syndrome shards, RCCL all-reduce of counts",
                       **geometry},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "kernel": "qbp::bp_fused_kernel<6,3,0,false,true,1024,1>",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "effective bandwidth: messages stay in LDS/registers, physical HBM "
                                 "traffic is only syndrome/LLR I/O; the physical limiter is FP64 VALU"},
            "early_exit": early,
            "valu_f64": valu,
            "hbm_streamed_variant": streamed,
            "converged_fraction": counts[0] / (world * B),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(code, syndromes[:100000].cpu().numpy(),
                                               prior.cpu().numpy())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
