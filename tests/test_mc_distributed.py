"""CPU, world_size 2 over gloo: the Monte-Carlo driver's sharding + single all-reduce gives the
same counter table as one rank (the per-trial RNG is keyed by the GLOBAL trial index).  The
device runner is replaced by the CPU oracle here; tests/test_gpu_mc.py checks the device runner
against the same oracle."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle
from qldpc_amd import mc

PS = [0.06, 0.03]
TRIALS = 301          # odd on purpose: ragged shards


def _oracle_runner(code, p, begin, end):
    return oracle.mc_counters(code.Hx, code.Lx, code.distance, p, mc.prior_of(p, code.n), begin,
                              end, draws=2, seed=11, max_iter=30)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def all_reduce(table):
        t = torch.from_numpy(table.copy())
        dist.all_reduce(t)
        return t.numpy()

    table = mc.run_sweep("[[72, 12, 6]]", PS, TRIALS, rank=rank, world=world,
                         runner=_oracle_runner, all_reduce=all_reduce)
    if rank == 0:
        np.save(out, table)
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "table.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    two = np.load(out)
    one = mc.run_sweep("[[72, 12, 6]]", PS, TRIALS, runner=_oracle_runner)
    assert np.array_equal(one, two)
    assert (one[:, 0] == TRIALS).all()
    s = mc.summarize(one[0])
    assert 0 < s["ler"] < 1 and s["mean_iterations"] >= 1


def _oracle_runner_osd(code, p, begin, end):
    return oracle.mc_counters(code.Hx, code.Lx, code.distance, p, mc.prior_of(p, code.n), begin,
                              end, draws=1, seed=5, max_iter=20, osd=True)


def _worker8(rank, world, port, out, trials):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def all_reduce(table):
        t = torch.from_numpy(table.copy())
        dist.all_reduce(t)
        return t.numpy()

    table = mc.run_sweep("[[72, 12, 6]]", [0.08, 0.05, 0.02], trials, rank=rank, world=world,
                         runner=_oracle_runner_osd, all_reduce=all_reduce)
    if rank == 0:
        np.save(out, table)
    dist.destroy_process_group()


def test_eight_ranks_ragged_with_osd_equal_one_rank(tmp_path):
    """The layout of BASELINE configs 4 / 5 -- 8 ranks, BP + OSD-0, one all-reduce -- rehearsed over gloo with
    a trial count that does not divide by 8 (shards of 37 and 38 trials) and one smaller than the world
    size (some ranks own nothing): the reduced table equals the single-rank one, counter for counter."""
    for trials in (301, 5):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = str(tmp_path / f"table8_{trials}.npy")
        mp.spawn(_worker8, args=(8, port, out, trials), nprocs=8, join=True)
        eight = np.load(out)
        one = mc.run_sweep("[[72, 12, 6]]", [0.08, 0.05, 0.02], trials, runner=_oracle_runner_osd)
        assert np.array_equal(one, eight)
        assert (one[:, 0] == trials).all()
        if trials > 100:
            assert one[0, 6] > 0                     # some trials did go through OSD-0


def test_shard_ranges_tile():
    for trials in (0, 1, 7, 1000, 125001):
        for world in (1, 2, 3, 8):
            cuts = [mc.shard_range(trials, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == trials
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors): the oracle's generator is the
    published algorithm, so device == oracle (test_gpu_mc) pins the device to it as well."""
    # counter (0,0,0,0), key (0,0) -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    # exposed indirectly: threshold compare on word v&3 of counter (trial, trial>>32, v>>2, draw)
    L = oracle.lib()
    e = oracle.mc_errors(4, 0.5, 1, 0, 0, 1)[0]         # words of Philox((0,0,0,0),(0,0)) < 2^31
    words = [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert e.tolist() == [int(w < 2**31) for w in words]
    assert L.oracle_mc_threshold(0.5) == 2**31
    # Bernoulli rate sanity
    e = oracle.mc_errors(288, 0.05, 1, 3, 0, 2000)
    assert abs(e.mean() - 0.05) < 0.002
    e2 = oracle.mc_errors(288, 0.05, 2, 3, 0, 2000)
    assert abs(e2.mean() - 2 * 0.05 * 0.95) < 0.003
