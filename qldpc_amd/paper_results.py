"""The experiment of ``paperResults_GPU.py`` / ``paperResults.py`` on MI355X GPUs, writing the
reference's result schema.

For every code and physical error rate: ``trials`` Monte-Carlo trials (errors = XOR of two
Bernoulli(p) draws, paperResults_GPU.py:96-105), BP, optional OSD-0 on the non-converged trials
(paperResults.py:73-77; the GPU script's OSD-w(7) is not accelerated), classification, counters
(:113-151).  One process per GPU; trials are sharded and the counter table is reduced once with
RCCL (``qldpc_amd.mc.run_sweep``).

Output ``<out>.npz`` holds ``results`` = ``{code: {'ler', 'BPs_fault', 'BPs_miscorrected',
'incorrectable', 'degeneracies'}}`` with one list entry per error rate, the pickled-dict layout of
``np.savez('data/BPOSD_GPU.npz', results=results_OSD)`` (:156-166) that ``loadResults.py:5-11``
reads back -- plus ``meta`` (JSON: p grid, trials, maxIter, noise model, decoder), which the
reference's files lack.

    python -m qldpc_amd.paper_results --out data/BPOSD_MI355X
    python -m torch.distributed.run --nproc-per-node 8 -m qldpc_amd.paper_results --trials 1000000
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np

from . import _lib, codes, mc

CODES = list(codes.BB_CODES)                                                   # :24-30
DEFAULT_RATES = [0.05, 0.04, 0.03, 0.02, 0.01, 0.009, 0.008, 0.007]          # :39 (last assignment wins)
KEYS = ("ler", "BPs_fault", "BPs_miscorrected", "incorrectable", "degeneracies")


def results_from_tables(tables: dict) -> dict:
    """{code: int64[points, 12]} -> the reference's results_OSD dict (:146-160)."""
    out = {}
    for name, table in tables.items():
        t = np.maximum(table[:, 0], 1)
        out[name] = {
            "ler": [float(x) for x in table[:, 1] / t],                       # :146-147
            "BPs_fault": [int(x) for x in table[:, 2]],
            "BPs_miscorrected": [int(x) for x in table[:, 3]],
            "incorrectable": [int(x) for x in table[:, 4]],
            "degeneracies": [int(x) for x in table[:, 5]],
        }
    return out


def save_results(path: str, results: dict, meta: dict) -> str:
    if not path.endswith(".npz"):
        path += ".npz"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, results=np.array(results, dtype=object), meta=np.array(json.dumps(meta)))
    return path


def load_results(path: str):
    """As loadResults.py:5-11: ``np.load(..., allow_pickle=True)['results'].item()``."""
    with np.load(path, allow_pickle=True) as d:
        return d["results"].item(), json.loads(str(d["meta"]))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--codes", nargs="+", default=CODES)
    ap.add_argument("--p", type=float, nargs="+", default=DEFAULT_RATES)
    ap.add_argument("--trials", type=int, default=10000)                       # :36
    ap.add_argument("--max-iter", type=int, default=150)                       # :109
    ap.add_argument("--draws", type=int, default=2, choices=(1, 2))            # :96-105
    ap.add_argument("--osd", type=int, default=0, choices=(-1, 0),
                    help="0: OSD-0 on BP failures (paperResults.py:77); -1: BP only")
    ap.add_argument("--seed", type=int, default=0)                             # :33-34
    ap.add_argument("--out", default="data/BPOSD_MI355X")
    ap.add_argument("--plot", action="store_true", help="two-panel figure as :169-185")
    ap.add_argument("--gpus", type=int, default=0,
                    help="N > 1 without a launcher: start N ranks (one per GPU) and reduce over RCCL")
    ap.add_argument("--backend", default="nccl", help="nccl = RCCL; gloo only for rehearsals")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank on cuda:0")
    args = ap.parse_args(argv)

    import sys
    from . import launch
    if argv is None:          # (a caller passing argv runs in-process, whatever --gpus says)
        launch.maybe_self_launch(args.gpus, ["-m", "qldpc_amd.paper_results"] + sys.argv[1:])

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
    if rank == 0:
        print(f"GPU Available: True ({world} x MI355X)")
        print(f"Running {args.trials} trials per point, BP maxIter {args.max_iter}, "
              f"{'OSD-0' if args.osd == 0 else 'no OSD'}")
        print("=" * 60)
    tables = {}
    total_start = time.time()
    for name in args.codes:
        name = codes.ALIASES.get(name, name)
        if rank == 0:
            print(f"\nProcessing code: {name}")
        t0 = time.time()
        tables[name] = mc.run_sweep(name, args.p, args.trials, draws=args.draws, seed=args.seed,
                                    max_iter=args.max_iter, osd=args.osd == 0, rank=rank,
                                    world=world, device=local)
        if rank == 0:
            dt = time.time() - t0
            for p, row in zip(args.p, tables[name]):
                print(f"  p={p}: LER={row[1] / max(row[0], 1):.6f}, degeneracies={int(row[5])}, "
                      f"time={dt / len(args.p):.1f}s")                      # :154
    if rank == 0:
        total = time.time() - total_start
        print(f"\n{'=' * 60}\nTotal time: {total:.1f}s ({total / 60:.1f} min)")
        meta = dict(physicalErrorRates=args.p, trials=args.trials, maxIter=args.max_iter,
                    draws=args.draws, osd=args.osd, seed=args.seed, world_size=world,
                    noise="XOR of two Bernoulli(p) draws" if args.draws == 2 else "Bernoulli(p)",
                    decoder="sum-product BP (libqbp, MI355X)", seconds=total,
                    counters={n: tables[n].tolist() for n in tables},
                    counter_names=list(_lib.COUNTER_NAMES))
        results = results_from_tables(tables)
        path = save_results(args.out, results, meta)
        print(f"Results saved to {path}")
        if args.plot:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            fig, axes = plt.subplots(1, 2, figsize=(14, 5))
            for n in results:
                axes[0].plot(args.p, results[n]["degeneracies"], label=n, marker="o")
                axes[1].plot(args.p, results[n]["ler"], label=n, marker="o")
            axes[0].grid(True); axes[0].legend(); axes[0].set_title("Degeneracies")
            axes[1].grid(True); axes[1].legend(); axes[1].set_yscale("log"); axes[1].set_xscale("log")
            axes[1].set_title("Logical errors")
            png = path[:-4] + "_results.png"
            plt.savefig(png, dpi=300)
            print(f"Plot saved to {png}")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
