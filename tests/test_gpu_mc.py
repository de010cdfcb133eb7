"""GPU: the on-device Monte-Carlo path (qbp_mc_run) against the CPU oracle's statement of it."""
import numpy as np
import pytest

from oracle import oracle
from qldpc_amd import _lib, bp, codes, mc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["[[72, 12, 6]]", "[[288, 12, 18]]", "steane"])
def test_device_sampler_bit_exact(name):
    code = codes.load_code(name)
    dec = bp.decoder_for(code.Hx)
    for draws, seed, begin in ((1, 0, 0), (2, 0xDEADBEEF12345, 2**33 + 5)):
        got = dec.mc_sample_errors(0.07, begin, 777, draws=draws, seed=seed)
        want = oracle.mc_errors(code.n, 0.07, draws, seed, begin, 777)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("name,p,T,variant,kw", [
    ("[[72, 12, 6]]", 0.05, 3000, _lib.SUM_PRODUCT, {}),
    ("[[72, 12, 6]]", 0.02, 3000, _lib.SUM_PRODUCT, {}),
    ("[[144, 12, 12]]", 0.05, 2000, _lib.MIN_SUM, dict(alpha=0.8, damping=0.7, clip_llr=25.0)),
    ("[[288, 12, 18]]", 0.06, 1500, _lib.SUM_PRODUCT, {}),
])
def test_mc_counters_match_oracle(name, p, T, variant, kw):
    code = codes.load_code(name)
    dec = bp.decoder_for(code.Hx)
    prior = mc.prior_of(p, code.n)
    got = dec.mc_run(code.Lx, code.distance, p, prior, 10, 10 + T, draws=2, seed=5, max_iter=50,
                     variant=variant, **kw)
    want = oracle.mc_counters(code.Hx, code.Lx, code.distance, p, prior, 10, 10 + T, draws=2,
                              seed=5, max_iter=50, variant=variant, **kw)
    print(dict(zip(_lib.COUNTER_NAMES, got.tolist())))
    assert np.array_equal(got, want)


def test_mc_shard_invariance_and_force_full():
    code = codes.load_code("[[144, 12, 12]]")
    dec = bp.decoder_for(code.Hx)
    p, T = 0.05, 20000
    prior = mc.prior_of(p, code.n)
    whole = dec.mc_run(code.Lx, code.distance, p, prior, 0, T, seed=9)
    parts = sum(dec.mc_run(code.Lx, code.distance, p, prior, a, b, seed=9)
                for a, b in [mc.shard_range(T, r, 8) for r in range(8)])
    assert np.array_equal(whole, parts)
    forced = dec.mc_run(code.Lx, code.distance, p, prior, 0, T, seed=9, flags=_lib.FLAG_FORCE_FULL)
    assert np.array_equal(whole, forced)
    table = mc.run_sweep("[[144, 12, 12]]", [p], T, seed=9)
    assert np.array_equal(table[0], whole)


def test_non_convergence_rate_matches_reference_data():
    """Fraction of trials BP(50) fails to converge on, against the `osd` (OSD invocation) rates
    stored in the reference's rework/simulation_results10k.npz (sum-product BP, single Bernoulli
    draw, 10 000 trials per point; values in BASELINE.md).  That file does not record maxIter;
    50 is inferred: the oracle reproduces its rates at maxIter = 50 and not at 30 or 100 (e.g.
    [[288,12,18]] p = 0.05: 0.089 / 0.059 / 0.029 at 30 / 50 / 100 vs 0.0652 stored).  The other
    BP-only files (data/CC-50k-LERS-BP.npz, notebooks/data/BP.npz) match maxIter of about 30 and
    20 by the same test and are therefore not used as maxIter = 50 targets."""
    ref = {"[[288, 12, 18]]": {0.06: 0.1296, 0.05: 0.0652, 0.04: 0.0311},
           "[[144, 12, 12]]": {0.06: 0.1466, 0.05: 0.0733, 0.04: 0.0284, 0.03: 0.0105},
           "[[72, 12, 6]]": {0.06: 0.1481, 0.05: 0.0833, 0.04: 0.045, 0.03: 0.018}}
    T = 200000
    for name, pts in ref.items():
        code = codes.load_code(name)
        dec = bp.decoder_for(code.Hx)
        for p, r in pts.items():
            c = dec.mc_run(code.Lx, code.distance, p, mc.prior_of(p, code.n), 0, T, seed=1)
            rate = c[6] / c[0]
            sigma = np.hypot(np.sqrt(r * (1 - r) / 10000), np.sqrt(r * (1 - r) / T))
            print(f"{name} p={p}: not converged {rate:.4f} (reference {r}, {abs(rate - r) / sigma:.1f} sigma), "
                  f"LER {c[1] / c[0]:.4f}")
            assert abs(rate - r) <= 4.0 * sigma


def test_paper_results_driver_end_to_end(tmp_path, capsys):
    """The paperResults_GPU.py replacement: runs a small sweep on the device and writes the
    reference's result schema (results[code][key] = list over the error rates)."""
    from qldpc_amd import paper_results
    out = str(tmp_path / "BPOSD_MI355X")
    paper_results.main(["--codes", "72", "144", "--p", "0.05", "0.02", "--trials", "20000",
                        "--max-iter", "50", "--out", out])
    printed = capsys.readouterr().out
    assert "Processing code: [[72, 12, 6]]" in printed and "LER=" in printed
    res = np.load(out + ".npz", allow_pickle=True)["results"].item()
    assert list(res) == ["[[72, 12, 6]]", "[[144, 12, 12]]"]
    for name in res:
        assert set(res[name]) == set(paper_results.KEYS)
        assert len(res[name]["ler"]) == 2 and res[name]["ler"][0] > res[name]["ler"][1] > 0
        assert res[name]["BPs_fault"] == [0, 0]
    # double draw at p = 0.05 on [[72,12,6]] with OSD-0: LER around 0.6 (effective p = 0.095)
    assert 0.45 < res["[[72, 12, 6]]"]["ler"][0] < 0.75


def test_bench_two_rank_path(tmp_path):
    """`python bench.py --gpus 2` as typed: the process starts its own two ranks (qldpc_amd/launch.py)
    and relays rank 0's JSON line.  Two ranks share this GPU over gloo here (RCCL refuses two ranks on
    one device); on a multi-GPU node the same command runs with backend nccl = RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--share-device", "--batch", "8000", "--mode", "forced",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["multi_gpu"]["n_ranks_seen"] == 2 and len(line["multi_gpu"]["kernel_ms_per_rank"]) == 2
    assert line["multi_gpu"]["all_reduce_us"] > 0
    # (two ranks time-share one GPU and synchronise over gloo: not a performance number)
    assert line["value"] > 1e5 and line["config"]["syndromes_per_gpu_per_step"] == 8000
    # the accounting the driver's N-GPU lines can be checked with: value = n_gpus x syndromes per GPU and step
    # x steps / wall (max over ranks), per-rank rates and the share of a step the one all-reduce takes
    wall = line["ms_per_step"] * 1e-3 * line["steps"]
    assert abs(line["n_gpus"] * 8000 * line["steps"] / wall - line["value"]) <= 1e-6 * line["value"]
    mg = line["multi_gpu"]
    assert len(mg["value_per_rank"]) == 2 and all(v >= 0.99 * line["value"] / 2 for v in mg["value_per_rank"])
    assert 0 < mg["all_reduce_share_of_step"] < 1 and len(mg["kernel_share_of_step_per_rank"]) == 2


def test_mc_two_rank_self_launch(tmp_path):
    """`python -m qldpc_amd.mc --gpus 2` starts its own ranks; counters equal the one-rank run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    outs = []
    for gpus in (1, 2):
        f = str(tmp_path / f"mc{gpus}.json")
        cmd = [sys.executable, "-m", "qldpc_amd.mc", "--code", "72", "--p", "0.05", "0.02", "--trials", "30001",
               "--osd", "--gpus", str(gpus), "--backend", "gloo", "--share-device", "--out", f]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.load(open(f)))
    assert outs[1]["world_size"] == 2
    for a, b in zip(outs[0]["points"], outs[1]["points"]):
        for k in ("trials", "logical_error", "not_converged", "sum_iterations", "BPs_miscorrected",
                  "incorrectable"):
            assert a[k] == b[k], k
