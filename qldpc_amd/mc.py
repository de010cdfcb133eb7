"""Monte-Carlo logical-error-rate driver: the outer loop of ``paperResults_GPU.py`` (:61-160),
sharded over the GPUs of one node.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI).  Trials are
independent, so each rank runs the contiguous slice ``[rank*T/R, (rank+1)*T/R)`` of the global
trial index range of every sweep point through ``qbp_mc_run_device`` (sampling, decoding and
classification all on the device; include/qbp.h) and the only communication is ONE all-reduce
(sum, int64) of the ``[points, 12]`` counter table at the end of the sweep.  Errors are drawn
from a counter-based generator keyed by the GLOBAL trial index, so the counters are identical
for every world size -- that is the multi-GPU correctness test (tests/test_mc_distributed.py).

    python -m qldpc_amd.mc --code 288 --p 0.06 0.05 0.04 --trials 1000000
    python -m torch.distributed.run --nproc-per-node 8 -m qldpc_amd.mc --code 288 ...
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np

from . import _lib, codes

NUM_COUNTERS = _lib.NUM_COUNTERS


def shard_range(trials: int, rank: int, world: int):
    """Contiguous slice of [0, trials) owned by `rank`; slices tile the range exactly."""
    return rank * trials // world, (rank + 1) * trials // world


def prior_of(p: float, n: int) -> np.ndarray:
    return np.full(n, np.log((1 - p) / p))          # paperResults_GPU.py:79


def summarize(counters: np.ndarray) -> dict:
    """Counter row -> the per-point quantities the reference stores / prints."""
    c = {k: int(v) for k, v in zip(_lib.COUNTER_NAMES, counters)}
    t = max(c["trials"], 1)
    c["ler"] = c["logical_error"] / t                                   # :146 (BP+OSD when osd)
    # BP-only convention of notebooks/data/BP.npz: a non-converged trial counts as a failure
    c["ler_bp_only"] = (c["logical_error"] - c["logical_error_not_converged"]
                        + c["not_converged"]) / t
    c["mean_iterations"] = c["sum_iterations"] / t + 1.0
    return c


def run_sweep(code_name, ps, trials, *, draws=1, seed=0, max_iter=50, variant=_lib.SUM_PRODUCT,
              alpha=1.0, damping=1.0, clip_llr=20.0, osd=False, rank=0, world=1, device=0,
              runner=None, all_reduce=None):
    """Returns the GLOBAL counter table int64[len(ps), 12] (after the reduce).

    `runner(code, p, begin, end) -> int64[12]` and `all_reduce(int64 array) -> int64 array`
    are injection points for the CPU tests; by default the HIP library and torch.distributed."""
    code = codes.load_code(code_name)
    table = np.zeros((len(ps), NUM_COUNTERS), np.int64)
    if runner is None:
        import torch

        from . import bp
        dec = bp.decoder_for(code.Hx, device=device)
        dev = torch.device("cuda", device)
        d_table = torch.zeros((len(ps), NUM_COUNTERS), dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream(dev)
        priors = [torch.from_numpy(prior_of(p, code.n)).to(dev) for p in ps]
        flags = _lib.FLAG_OSD0 if osd else 0
        step = dec.mc_osd_step() if osd else 1 << 40         # OSD keeps per-trial records
        for i, p in enumerate(ps):
            begin, end = shard_range(trials, rank, world)
            for a in range(begin, end, step):
                dec.mc_run_device(code.Lx, code.distance, p, priors[i].data_ptr(), a,
                                  min(a + step, end), d_table[i].data_ptr(), draws=draws,
                                  seed=seed, max_iter=max_iter, variant=variant, alpha=alpha,
                                  damping=damping, clip_llr=clip_llr, flags=flags,
                                  stream=stream.cuda_stream)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(d_table)                 # the one RCCL collective of the sweep
        torch.cuda.synchronize(dev)
        return d_table.cpu().numpy()
    for i, p in enumerate(ps):
        begin, end = shard_range(trials, rank, world)
        table[i] = runner(code, p, begin, end)
    return all_reduce(table) if all_reduce is not None else table


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--code", default="[[288, 12, 18]]")
    ap.add_argument("--p", type=float, nargs="+",
                    default=[0.05, 0.04, 0.03, 0.02, 0.01, 0.009, 0.008, 0.007])   # :39
    ap.add_argument("--trials", type=int, default=10000)                          # :36
    ap.add_argument("--max-iter", type=int, default=50)
    ap.add_argument("--draws", type=int, default=1, choices=(1, 2))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--variant", choices=("sum-product", "damped", "min-sum"), default="sum-product")
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--damping", type=float, default=1.0)
    ap.add_argument("--clip-llr", type=float, default=20.0)
    ap.add_argument("--osd", action="store_true", help="OSD-0 on the trials BP does not converge on")
    ap.add_argument("--out", default=None, help="write the counter table as JSON")
    ap.add_argument("--gpus", type=int, default=0,
                    help="N > 1 without a launcher: start N ranks (one per GPU) and reduce over RCCL")
    ap.add_argument("--backend", default="nccl", help="nccl = RCCL; gloo only for rehearsals")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank on cuda:0")
    args = ap.parse_args(argv)

    import sys
    from . import launch
    if argv is None:          # (a caller passing argv runs in-process, whatever --gpus says)
        launch.maybe_self_launch(args.gpus, ["-m", "qldpc_amd.mc"] + sys.argv[1:])

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    variant = {"sum-product": _lib.SUM_PRODUCT, "damped": _lib.DAMPED_SP,
               "min-sum": _lib.MIN_SUM}[args.variant]
    # one-time setup, timed apart from the sweep: HIP context, the decoder of this code (tables, device
    # buffers, kernel images) and a first small launch of the kernels the sweep uses (0.3 - 0.4 s)
    t0 = time.perf_counter()
    run_sweep(args.code, args.p[:1], min(args.trials, 4096), draws=args.draws, seed=args.seed,
              max_iter=args.max_iter, variant=variant, alpha=args.alpha, damping=args.damping,
              clip_llr=args.clip_llr, osd=args.osd, rank=0, world=1, device=local)
    t_setup = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    table = run_sweep(args.code, args.p, args.trials, draws=args.draws, seed=args.seed,
                      max_iter=args.max_iter, variant=variant, alpha=args.alpha,
                      damping=args.damping, clip_llr=args.clip_llr, osd=args.osd, rank=rank,
                      world=world, device=local)
    dt = time.perf_counter() - t0
    if rank == 0:
        rows = []
        for p, row in zip(args.p, table):
            s = summarize(row)
            s["p"] = p
            rows.append(s)
            print(f"  p={p}: LER={s['ler']:.6f}, BP-only LER={s['ler_bp_only']:.6f}, "
                  f"degeneracies={s['degenerateErrors']}, not converged={s['not_converged']}, "
                  f"mean iters={s['mean_iterations']:.2f}")
        print(f"{len(args.p)} points x {args.trials} trials on {world} GPU(s): {dt:.3f} s "
              f"({len(args.p) * args.trials / dt:.3e} trials/s); one-time setup {t_setup:.2f} s")
        if args.out:
            with open(args.out, "w") as f:
                json.dump({"code": args.code, "trials": args.trials, "max_iter": args.max_iter,
                           "draws": args.draws, "seed": args.seed, "variant": args.variant,
                           "osd": args.osd,
                           "world_size": world, "seconds": dt, "setup_seconds": t_setup, "points": rows},
                          f, indent=1)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
