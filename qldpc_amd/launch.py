"""Self-launch of the one-process-per-GPU programs (bench.py, ``python -m qldpc_amd.mc``,
``python -m qldpc_amd.paper_results``).

``python bench.py --gpus 8`` must work as typed.  When a program is asked for N > 1 GPUs and is not
already running under a launcher (no ``WORLD_SIZE`` in the environment), ``spawn_ranks`` starts

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
           --master-port <free port> <the same program and arguments>

as a CHILD process and returns its exit status; the children are the ranks (they see WORLD_SIZE and
do the work), rank 0 prints the result on the inherited stdout.  This is decided BEFORE the parent
imports torch or touches HIP: a process that has initialised the GPU must neither exec nor fork
workers, so the parent stays a plain supervisor that never creates a device context.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def under_launcher() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(nproc: int, program: list[str], port: int | None = None) -> list[str]:
    """``program`` = ["bench.py", args...] or ["-m", "qldpc_amd.mc", args...]."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port())] + list(program)


def spawn_ranks(nproc: int, program: list[str], env: dict | None = None) -> int:
    """Run `program` on `nproc` ranks in child processes; returns the launcher's exit status."""
    if "torch" in sys.modules:
        import torch
        if torch.cuda.is_initialized():
            raise RuntimeError("spawn_ranks must be called before this process touches the GPU")
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    e.setdefault("OMP_NUM_THREADS", "2")
    e.setdefault("MASTER_ADDR", "127.0.0.1")
    return subprocess.call(launcher_command(nproc, program), env=e)


def maybe_self_launch(gpus: int, program: list[str]) -> None:
    """Call first thing in main(): if N > 1 ranks are wanted and no launcher started us, become the
    supervisor of N child ranks and exit with their status."""
    if gpus > 1 and not under_launcher():
        sys.exit(spawn_ranks(gpus, program))
