"""CPU: host logic and the C-ABI library surface (no compute calls without a GPU)."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from qldpc_amd import _lib, bp, codes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qldpc_amd", "csrc"), "libqbp.so"])
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "qbp.h")).read()
    declared = set(re.findall(r"\b(qbp_[a-z0-9_]+)\s*\(", header))
    declared.discard("qbp_handle")
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.qbp_version()


def test_create_fails_loudly_without_gpu(lib):
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    code = codes.load_code("[[72, 12, 6]]")
    with pytest.raises(_lib.QbpError, match="no CPU fallback"):
        bp.performBeliefPropagationFast(code.Hx, np.zeros(36, int), [1.0] * 72, verbose=False)


def test_argument_validation_before_device(lib):
    h = C.c_void_p()
    rp = np.array([0, 2], np.int32)
    assert lib.qbp_create(rp.ctypes.data, np.array([1, 0], np.int32).ctypes.data, 1, 2, 0,
                          C.byref(h)) == -1          # columns not ascending
    assert b"ascending" in lib.qbp_last_error()
    assert lib.qbp_create(rp.ctypes.data, np.array([0, 5], np.int32).ctypes.data, 1, 2, 0,
                          C.byref(h)) == -1          # column out of range
    assert lib.qbp_create(None, None, 1, 2, 0, C.byref(h)) == -1


def test_no_exception_crosses_the_abi(lib):
    """Every extern "C" entry point of qbp.hip is a function-try-block (checked in the source), and what the
    catch clauses turn an exception into is observed through QBP_OPT_DEBUG_THROW, which raises inside one:
    std::bad_alloc -> QBP_E_NOMEM, anything else -> QBP_E_INVALID; the process survives."""
    src = open(os.path.join(ROOT, "qldpc_amd", "csrc", "qbp.hip")).read()
    body = src[src.index('extern "C" {'):]
    entries = re.findall(r"^(?:int|int64_t|void|const char\*) (qbp_\w+)\(([^)]*)\)\n(try )?\{", body, re.M)
    trivial = {"qbp_last_error", "qbp_version", "qbp_destroy"}        # return a pointer / free: nothing to throw
    for name, _, is_try in entries:
        assert is_try or name in trivial, f"{name} is not a function-try-block"
    assert {e[0] for e in entries if e[2]} >= set(_lib.SIGNATURES) - trivial
    assert lib.qbp_set_option(None, 99, 1) == -5 and b"memory" in lib.qbp_last_error()       # QBP_E_NOMEM
    assert lib.qbp_set_option(None, 99, 2) == -1 and b"QBP_OPT_DEBUG_THROW" in lib.qbp_last_error()
    assert lib.qbp_set_option(None, 99, 3) == -1
    assert lib.qbp_set_option(None, 99, 0) == 0
    assert lib.qbp_set_option(None, 1, 0) == -1                        # (null handle otherwise)


def test_csr_from_H_matches_scipy():
    from scipy.sparse import csr_matrix
    for name in ("steane", "[[72, 12, 6]]", "[[288, 12, 18]]"):
        H = codes.load_code(name).Hx
        rp, ci, m, n = bp.csr_from_H(H)
        S = csr_matrix(H)
        S.sort_indices()
        assert np.array_equal(rp, S.indptr) and np.array_equal(ci, S.indices)
        rp2, ci2, _, _ = bp.csr_from_H(csr_matrix(H.astype(float)))
        assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    with pytest.raises(ValueError):
        bp.csr_from_H(np.zeros(5))


def test_input_checks():
    with pytest.raises(ValueError):
        bp._syndromes([0, 2, 1], 3, batch=False)
    with pytest.raises(ValueError):
        bp._syndromes([0, 1], 3, batch=False)
    assert bp._syndromes(np.array([True, False, True]), 3, batch=False).tolist() == [1, 0, 1]
    with pytest.raises(ValueError):
        bp._prior([1.0, 2.0], 3)
    with pytest.raises(ValueError):
        bp._check_iter(0)


def test_code_fixtures():
    for name in codes.code_names():
        c = codes.load_code(name)
        assert set(c.Hx.sum(1)) == {6} and set(c.Hx.sum(0)) == {3}
        assert not ((c.Hx @ c.Hz.T) % 2).any()              # CSS condition
        assert not ((c.Hz @ c.Lx.T) % 2).any()              # logicals commute with Z checks
        n, k = c.n, c.Lx.shape[0]
        assert f"[[{n}, {k}, {c.distance}]]" == name


def test_dropin_package_resolution(tmp_path):
    """`decoding.beliefPropagation` / `decoding.OSD` / `decoding.OSD_enhanced` must resolve to this
    build, and a module this build does not replace (`decoding.beliefPropagationJAX`) to a
    reference-like namespace directory that sits earlier on sys.path (as when running the
    reference's scripts from its root)."""
    ref = tmp_path / "ref"
    (ref / "decoding").mkdir(parents=True)
    (ref / "decoding" / "beliefPropagationJAX.py").write_text("def marker():\n    return 'ref-jax'\n")
    for name in ("OSD", "OSD_enhanced", "beliefPropagation"):
        (ref / "decoding" / f"{name}.py").write_text(f"raise RuntimeError('reference {name} imported')\n")
    script = ("import decoding.beliefPropagation as b, decoding.OSD as o, decoding.OSD_enhanced as w\n"
              "import decoding.beliefPropagationJAX as j\n"
              "from decoding import performOSD_enhanced, performMinSum_Symmetric\n"
              "print(b.performBeliefPropagationFast.__module__, o.performOSD.__module__, "
              "w.performOSD_enhanced.__module__, performOSD_enhanced.__module__, j.marker())\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "qldpc_amd", "dropin")]))
    (ref / "run.py").write_text(script)
    out = subprocess.check_output([sys.executable, str(ref / "run.py")], env=env, cwd=ref, text=True)
    assert out.split() == ["qldpc_amd.bp", "qldpc_amd.osd", "qldpc_amd.osd", "qldpc_amd.osd", "ref-jax"]


def test_results_writer_schema(tmp_path):
    """The result file keeps the reference's layout (paperResults_GPU.py:156-166) so that
    loadResults.py-style readers work: results[code][key] is a list over the error rates."""
    from qldpc_amd import paper_results
    tables = {"[[72, 12, 6]]": np.array([[1000, 12, 0, 1, 11, 30, 40, 900, 10, 800, 0, 0],
                                          [1000, 2, 0, 0, 2, 5, 3, 100, 1, 950, 0, 0]], np.int64)}
    res = paper_results.results_from_tables(tables)
    path = paper_results.save_results(str(tmp_path / "out"), res, dict(physicalErrorRates=[0.05, 0.01]))
    loaded = np.load(path, allow_pickle=True)["results"].item()          # loadResults.py:5-7
    assert set(loaded["[[72, 12, 6]]"]) == set(paper_results.KEYS)
    assert loaded["[[72, 12, 6]]"]["ler"] == [0.012, 0.002]
    assert loaded["[[72, 12, 6]]"]["degeneracies"] == [30, 5]
    assert loaded["[[72, 12, 6]]"]["BPs_fault"] == [0, 0]
    back, meta = paper_results.load_results(path)
    assert back == loaded and meta["physicalErrorRates"] == [0.05, 0.01]


def _build_c_example(tmp_path):
    exe = str(tmp_path / "decode_steane")
    csrc = os.path.join(ROOT, "qldpc_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "decode_steane.c"), "-L", csrc, "-lqbp", "-lm",
                           f"-Wl,-rpath,{csrc}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    return exe


def test_header_is_plain_c_and_library_links_from_c(lib, tmp_path):
    """include/qbp.h must be usable from C (no C++, no torch types): build examples/decode_steane.c
    with gcc -std=c99 -pedantic -Werror and link it against libqbp.so."""
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    if os.path.exists("/dev/kfd"):
        assert r.returncode == 0 and "converged 1 at iteration 0: 0010000" in r.stdout
    else:
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


def _plan(lib, H):
    rp, ci, m, n = bp.csr_from_H(H)
    info = np.zeros(8, np.int32)
    assert lib.qbp_plan(rp.ctypes.data, ci.ctypes.data, m, n, info.ctypes.data, None, None, None) == 0
    kind, dc, dv = int(info[0]), int(info[1]), int(info[2])
    tabs = None
    if kind == 1:
        tv = np.zeros((dc, m), np.int32)
        tn = np.zeros((dc, dv, m), np.uint16)
        tw = np.zeros(m, np.uint32)
        assert lib.qbp_plan(rp.ctypes.data, ci.ctypes.data, m, n, info.ctypes.data, tv.ctypes.data,
                            tn.ctypes.data, tw.ctypes.data) == 0
        tabs = (tv, tn, tw)
    return info, tabs, (rp, ci, m, n)


def test_host_tables_of_the_on_chip_kernel(lib):
    """qbp_plan (host only): the gather tables the fused kernel runs on.  For every edge (c, j) of
    variable v the DV offsets must list v's column in ascending check order, padded with the zero
    word; exactly one edge per variable is its writer; padding edges have no variable."""
    from scipy.sparse import csr_matrix
    c144 = codes.load_code("[[144, 12, 12]]").Hx
    st = np.hstack([np.kron(np.eye(3, dtype=np.int64), c144),
                    (np.eye(216, dtype=np.int64) + np.eye(216, k=-72, dtype=np.int64)) % 2])
    for H, want in ((codes.load_code("steane").Hx, (1, 6, 3)), (codes.load_code("[[288, 12, 18]]").Hx, (1, 6, 3)),
                    (st, (1, 8, 4))):
        info, tabs, (rp, ci, m, n) = _plan(lib, H)
        assert tuple(info[:3]) == want
        tv, tn, tw = tabs
        dc, dv = want[1], want[2]
        S = csr_matrix(H)
        Hc = S.tocsc()
        Hc.sort_indices()
        writers = np.zeros(n, int)
        for c in range(m):
            row = ci[rp[c]:rp[c + 1]]
            assert np.array_equal(tv[:len(row), c], row) and (tv[len(row):, c] == -1).all()
            for j, v in enumerate(row):
                col_checks = Hc.indices[Hc.indptr[v]:Hc.indptr[v + 1]]          # ascending checks
                for k in range(dv):
                    off = int(tn[j, k, c])
                    if k < len(col_checks):
                        cc = int(col_checks[k])
                        jj = int(np.searchsorted(ci[rp[cc]:rp[cc + 1]], v))
                        assert off == jj * m + cc
                    else:
                        assert off == dc * m                                        # the zero word
                writers[v] += (int(tw[c]) >> j) & 1
                assert ((int(tw[c]) >> j) & 1) == int(col_checks[0] == c)
            assert (tn[len(row):, :, c] == dc * m).all()
        assert np.array_equal(writers, (np.asarray(H) != 0).any(0).astype(int))
        assert info[6] == int((np.asarray(H) != 0).sum(1).min() < dc)
    # shapes that do not fit: wide rows -> general-H kernel; m > 1024 -> general-H kernel
    rng = np.random.default_rng(0)
    wide = (rng.random((20, 60)) < 0.4).astype(np.int64)
    assert _plan(lib, wide)[0][0] == 2
    big = np.kron(np.eye(8, dtype=np.int64), c144)                                   # m = 576 fits
    assert _plan(lib, big)[0][0] == 1
    big = np.kron(np.eye(16, dtype=np.int64), c144)                                  # m = 1152
    assert _plan(lib, csr_matrix(big))[0][0] == 2


def test_nan_prior_rejected_by_the_shim():
    """NaN priors are an input error here (bp._prior); +-inf (p = 0 or 1) stay legal."""
    from qldpc_amd import bp
    with pytest.raises(ValueError, match="NaN"):
        bp._prior([0.5, float("nan"), 1.0], 3)
    assert np.isinf(bp._prior([np.inf, -np.inf, 1.0], 3)[:2]).all()


def test_bench_accounting_and_committed_line():
    """bench.py: the algorithmic-byte formula of SURVEY 8(d) and the contract keys of the JSON line
    (checked on the line committed under profiles/)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # [[288,12,18]]: E = 864, m = 144, n = 288 -> 27 648 B per iteration, 2 741 B of I/O per syndrome
    assert bench.algorithmic_bytes(864, 144, 288, 1, 0) == 27648
    assert bench.algorithmic_bytes(864, 144, 288, 50, 1) == 1382400 + 2741
    line = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_1gpu.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline",
                "hbm_effective", "early_exit", "stress_p010", "sustained", "dropin_api"):
        assert key in line, key
    assert line["scaling"] == "weak" and line["dtype"] == "f64" and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    # the binding resource of the on-chip kernel, measured by the run: a fraction of a real ceiling
    assert r["bound"] == "fp64_valu" and 0.0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    needed = r["issue_cycles_per_wave_iteration"] * r["wave_iterations_per_launch"]
    available = r["simds"] * r["kernel_ms"] * 1e-3 * r["max_clock_GHz"] * 1e9
    assert abs(r["frac"] - needed / available) < 1e-9
    assert sum(r["valu_by_class"].values()) == r["valu_insts_per_wave_iteration"]
    # round 3: the same fraction with ideal issue costs, the useful-flop share of the 78.6 TFLOP/s vector peak,
    # the HBM bytes per launch and the hardware's own instruction count from the committed PMC summary
    assert 0.0 < r["frac_ideal_pricing"] <= r["frac"] and 0.0 < r["flops_frac"] < r["frac"]
    assert r["traffic"] and abs(r["traffic"] / (125000 * 2741) - 1.0) < 0.15      # = the syndrome / LLR I/O
    assert r["pmc"]["file"] == "profiles/r03_pmc_summary.json"
    assert abs(r["pmc"]["static_count_over_pmc_count"] - 1.0) < 0.05
    for leg in ("config2_early_exit", "config2_forced_50", "config3_early_exit", "config3_forced_50",
                "mc288_p0.01", "mc288_p0.05", "mc288_p0.05_osd0", "osd0_288"):
        lr = line["other_configs"][leg]["roofline"]
        assert lr["frac"] is not None and 0.0 < lr["frac"] <= 1.0 and lr["pmc_file"] == "profiles/r03_pmc_legs.json"
    assert line["hbm_effective"]["unit"] == "GB/s"       # the north star's yardstick, kept beside it
    assert line["sustained"]["seconds"] >= 9.5 and 0.9 < line["sustained"]["ratio_to_headline"] < 1.1
    assert 0.95 < line["stress_p010"]["ratio_to_headline_kernel_ms"] < 1.05
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and "sample" in c
    assert c["reference_python"]["measured_by_this_run"] is False
    assert line["value"] > 1e6            # the north star's floor
    two = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_2rank_gloo_rehearsal.json")))
    assert two["n_gpus"] == 2 and two["multi_gpu"]["n_ranks_seen"] == 2


def test_static_instruction_count_of_the_built_library_matches_the_committed_pmc_count(lib):
    """The headline roofline prices a STATIC count of the kernel's loop (tools/valu_mix.py on the library as built).
    It must stay within a few percent of what the hardware counted for the same kernel (SQ_INSTS_VALU of
    profiles/r03_pmc_summary.json): a compiler or source change that moves code across the tool's cold-region
    markers shows up here, not as a silently shifted efficiency figure."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import valu_mix
    res = valu_mix.analyse(_lib.LIB_PATH, ["bp_fused_kernelILi6ELi3ELi0ELb0ELb1ELi1024ELi1ELb1EE"])
    assert len(res) == 1
    mix = next(iter(res.values()))
    static = mix["valu_total"] / 2                      # the one-barrier loop holds two BP iterations
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_summary.json")))
    k = next(v for name, v in pmc.items() if "bp_fused_kernel<6, 3, 0, false, true, 1024, 1, true>" in name)
    wave_iterations = 125000 * 50 / 7 * 16              # the bench launch: 7 slots per 16-wave workgroup
    counted = k["counters"]["SQ_INSTS_VALU"]["last"] / wave_iterations
    assert abs(static / counted - 1.0) < 0.05, (static, counted)
    per_class = {c: n / 2 for c, n in mix["valu_by_class"].items()}
    for cls, ctr in (("fma_f64", "SQ_INSTS_VALU_FMA_F64"), ("add_f64", "SQ_INSTS_VALU_ADD_F64"),
                     ("mul_f64", "SQ_INSTS_VALU_MUL_F64"), ("trans_f64", "SQ_INSTS_VALU_TRANS_F64")):
        assert abs(per_class[cls] - k["counters"][ctr]["last"] / wave_iterations) < 0.51, cls


def test_self_launch_command_and_supervisor(tmp_path):
    """qldpc_amd/launch.py: `--gpus N` without a launcher becomes a supervisor of N child ranks
    (torch.distributed.run on 127.0.0.1) BEFORE torch is imported; under a launcher it does nothing."""
    import subprocess
    import sys
    from qldpc_amd import launch
    cmd = launch.launcher_command(3, ["x.py", "--gpus", "3"], port=29999)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=3" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-3:] == ["x.py", "--gpus", "3"]
    prog = tmp_path / "prog.py"
    prog.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from qldpc_amd import launch\n"
        "launch.maybe_self_launch(2, [os.path.abspath(__file__)])\n"
        "assert 'torch' not in sys.modules\n"
        "sys.stdout.write('rank %s of %s\\n' % (os.environ['RANK'], os.environ['WORLD_SIZE'])); sys.stdout.flush()\n"
        "sys.exit(7 if os.environ['RANK'] == '1' else 0)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, timeout=300, env=env)
    assert "rank 0 of 2" in r.stdout and "rank 1 of 2" in r.stdout
    assert r.returncode != 0          # a failing rank fails the supervisor


def test_decoder_cache_lookup_rules(monkeypatch):
    """qldpc_amd.bp.decoder_for without a device (Decoder stubbed).  Default: every matrix is hashed in full on
    every call, so an in-place edit ANYWHERE in a large matrix is noticed (the old default
    trusted identity + a sampled checksum, which an edit off the sample grid slipped through); only arrays
    that cannot change are recognised by identity -- a read-only view of a writeable array is not one of
    them.  QBP_TRUST_IDENTITY restores the fast path as an explicit choice; `forget` drops an entry; eviction
    only drops the cache's reference."""
    from qldpc_amd import _lib, bp
    made = []

    class Stub:
        def __init__(self, row_ptr, col_idx, m, n, device):
            made.append((m, n))
            self.closed = False

        def close(self):
            self.closed = True

    monkeypatch.setattr(_lib, "Decoder", Stub)
    monkeypatch.setattr(bp, "TRUST_IDENTITY", False)
    bp.forget()
    try:
        big = np.zeros((600, 600), np.float64)                    # 2.9 MB
        big[np.arange(600), np.arange(600)] = 1
        d1 = bp.decoder_for(big)
        assert bp.decoder_for(big) is d1 and len(made) == 1       # same content: found by its hash
        step = max(1, big.size // 65536)                           # the old sample grid: flat[::step]
        assert step > 1
        big.reshape(-1)[step * 1000 + 1] = 1                       # an edit OFF that grid, in place, no forget()
        d2 = bp.decoder_for(big)
        assert d2 is not d1 and len(made) == 2
        # a read-only VIEW of a writeable array: its base can change under it -> hashed, and noticed
        view = big[:]
        view.setflags(write=False)
        v1 = bp.decoder_for(view)
        assert v1 is d2
        big.reshape(-1)[step * 2000 + 1] = 1
        assert bp.decoder_for(view) is not v1
        # an array that cannot change: recognised by identity (no hash)
        ro = np.eye(30, dtype=np.int64)
        ro.setflags(write=False)
        r1 = bp.decoder_for(ro)
        calls = []
        monkeypatch.setattr(bp, "_digest", lambda buf: calls.append(1) or b"x" * 16)
        assert bp.decoder_for(ro) is r1 and not calls
        monkeypatch.undo()
        monkeypatch.setattr(_lib, "Decoder", Stub)
        # the opt-in fast path: identity + sampled checksum for large writeable arrays (documented hazard)
        monkeypatch.setattr(bp, "TRUST_IDENTITY", True)
        big2 = np.zeros((600, 600), np.float64)
        big2[np.arange(600), np.arange(600)] = 1
        t1 = bp.decoder_for(big2)
        big2.reshape(-1)[step * 1000 + 1] = 1                      # off the sample grid: NOT noticed in this mode
        assert bp.decoder_for(big2) is t1
        bp.forget(big2)                                            # ... until the caller says so
        assert bp.decoder_for(big2) is not t1
        monkeypatch.setattr(bp, "TRUST_IDENTITY", False)
        small = np.eye(20, dtype=np.int64)
        e1 = bp.decoder_for(small)
        small[0, 3] = 1                                           # in place: noticed without any call
        assert bp.decoder_for(small) is not e1
        for k in range(20):                                       # more matrices than the cache holds
            bp.decoder_for(np.eye(40 + k, dtype=np.int64))
        assert not d1.closed and not e1.closed                    # evicted, never closed behind a holder's back
    finally:
        bp.forget()


def test_host_array_methods_of_a_decoder_are_serialised():
    """The reference's functions are pure (callable from a thread pool); a qbp_handle is not
    thread-safe.  Every Decoder method that takes host arrays runs under the Decoder's lock, the
    cache lookup under the cache's (no device needed: the lock is observed through a stub)."""
    import threading
    from qldpc_amd import _lib, bp
    for name in ("decode", "mc_run", "check_messages", "message_histograms", "osd0",
                 "mc_sample_errors", "debug_math", "set_option"):
        assert getattr(_lib.Decoder, name).__wrapped__ is not None, name
    for name in ("decode_device", "mc_run_device", "osd0_device"):     # enqueue only: caller-ordered
        assert not hasattr(getattr(_lib.Decoder, name), "__wrapped__"), name

    inside, overlaps = [0], [0]

    class Fake:
        _lock = threading.RLock()

        @_lib._locked
        def call(self):
            inside[0] += 1
            if inside[0] > 1:
                overlaps[0] += 1
            threading.Event().wait(0.002)
            inside[0] -= 1

    f = Fake()
    ts = [threading.Thread(target=lambda: [f.call() for _ in range(5)]) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert overlaps[0] == 0
    assert isinstance(bp._CACHE_LOCK, type(threading.RLock()))


def test_sampler_mirror_matches_the_reference_formula():
    """generate_errors_and_syndromes_batch (beliefPropagationGPU.py:181-200): same random stream, and the
    syndromes of the sparse integer product equal ((errors @ H.T) % 2).astype(int8) for every dtype of H
    the reference's formula accepts -- including where its narrow accumulators wrap around."""
    from qldpc_amd import bp, codes
    H = codes.load_code("[[144, 12, 12]]").Hx
    e, s = bp.generate_errors_and_syndromes_batch(H, 0.07, 500, np.random.default_rng(5))
    rng = np.random.default_rng(5)
    e_ref = (rng.random((500, H.shape[1])) < 0.07).astype(np.int8)
    assert e.dtype == np.int8 and s.dtype == np.int8
    assert np.array_equal(e, e_ref) and np.array_equal(s, ((e_ref @ H.T) % 2).astype(np.int8))
    rng = np.random.default_rng(6)
    for dt in (bool, np.int8, np.uint8, np.int32, np.int64, np.uint64, np.float64):
        if dt is bool:
            Hx = rng.integers(0, 2, (30, 300)).astype(dt)       # dense: int8 sums wrap in the reference
        elif dt in (np.uint8, np.uint64):
            Hx = rng.integers(0, 200, (30, 300)).astype(dt)
        else:
            Hx = rng.integers(-100, 100, (30, 300)).astype(dt)
        err = (rng.random((200, 300)) < 0.6).astype(np.int8)
        assert np.array_equal(bp._syndromes_of(err, Hx), ((err @ Hx.T) % 2).astype(np.int8)), dt


def test_performOSD_serves_the_driver_loop_from_one_batched_call(monkeypatch):
    """qldpc_amd/osd.py::_from_last_batch without a device: after a batch decode, performOSD on rows of
    the returned arrays (the reference driver's loop) runs ONE osd0 call for all failing rows; changed
    contents, copies, converged rows and other matrices take the one-syndrome path; results are those of
    the (stubbed) device function on the arguments actually passed."""
    from qldpc_amd import bp, osd
    m, n, B = 6, 10, 40
    calls = []

    class FakeDec:
        def __init__(self):
            self.m, self.n = m, n

        def osd0(self, syn, llr, hard):                       # a pure function of its inputs
            calls.append(len(syn))
            return ((hard.astype(np.int64) + (llr < 0) + syn.sum(1, keepdims=True)) % 2).astype(np.uint8)

    dec = FakeDec()
    monkeypatch.setattr(osd, "decoder_for", lambda H: dec)
    rng = np.random.default_rng(0)
    syn = rng.integers(0, 2, (B, m)).astype(np.int8)
    llr = rng.normal(size=(B, n))
    hard = rng.integers(0, 2, (B, n)).astype(np.int8)
    conv = rng.random(B) < 0.4
    bp._set_last_batch(bp._LastBatch(dec, syn, llr, hard, conv))

    def want(i, l=None, h=None):
        l = llr[i] if l is None else l
        h = hard[i] if h is None else h
        return (h.astype(np.int64) + (l < 0) + int(syn[i].sum())) % 2

    fails = np.flatnonzero(~conv)
    for i in fails[:5]:
        out = osd.performOSD(None, syn[i].astype(np.int64), llr[i], hard[i].astype(np.int64))
        assert out.dtype == np.int64 and np.array_equal(out, want(i))
    assert calls == [len(fails)]                              # one launch for all failing rows
    i = int(fails[5])
    assert np.array_equal(osd.performOSD(None, syn[i], llr[i].copy(), hard[i]), want(i))
    assert calls[-1] == 1 and len(calls) == 2                 # a copy is not a row of the batch
    llr[i, 3] = -llr[i, 3]                                    # contents changed after the batch call
    assert np.array_equal(osd.performOSD(None, syn[i], llr[i], hard[i]), want(i))
    assert len(calls) == 3
    h2 = 1 - hard[i]
    assert np.array_equal(osd.performOSD(None, syn[int(fails[6])], llr[int(fails[6])], h2),
                          want(int(fails[6]), h=h2))
    assert len(calls) == 4                                    # another hard decision than the batch's
    j = int(np.flatnonzero(conv)[0])
    assert np.array_equal(osd.performOSD(None, syn[j], llr[j], hard[j]), want(j))
    assert len(calls) == 5                                    # converged rows are not precomputed
    k = int(fails[7])
    assert np.array_equal(osd.performOSD(None, syn[k], llr[k], hard[k]), want(k)) and len(calls) == 5
    other = FakeDec()
    monkeypatch.setattr(osd, "decoder_for", lambda H: other)  # another matrix: never from the cache
    assert np.array_equal(osd.performOSD(None, syn[k], llr[k], hard[k]), want(k)) and len(calls) == 6
    monkeypatch.setattr(osd, "decoder_for", lambda H: dec)
    big = bp._LAST_BATCH_LIMIT // 8 // n + 1                  # LLR arrays above the limit are not remembered
    assert bp._last_batch().llr is llr and big * n * 8 > bp._LAST_BATCH_LIMIT
    # the record is the calling thread's own, and holds the returned arrays weakly
    import threading
    seen = []
    t = threading.Thread(target=lambda: seen.append(bp._last_batch()))
    t.start(); t.join()
    assert seen == [None]
    lb = bp._last_batch()
    del llr
    import gc
    gc.collect()
    assert lb.llr is None                                     # nothing was kept alive
    assert osd._from_last_batch(dec, syn[0].view(np.uint8), np.zeros(n), hard[0].view(np.uint8)) is None
    assert bp._last_batch() is None                           # dead record dropped at the next look


def test_python_constants_equal_the_header():
    """Flags, variants, options and counters of qldpc_amd/_lib.py against the enum values of include/qbp.h
    (parsed as text: the header is the contract of the ABI)."""
    import re
    text = open(os.path.join(ROOT, "include", "qbp.h")).read()
    enum = {k: int(v.rstrip("u"), 0) for k, v in re.findall(r"\b(QBP_[A-Z0-9_]+)\s*=\s*(-?(?:0x[0-9a-fA-F]+|\d+)u?)", text)}
    for py, c in (("FLAG_FORCE_FULL", "QBP_FLAG_FORCE_FULL"), ("FLAG_OSD0", "QBP_FLAG_OSD0"),
                  ("FLAG_PAIRWISE_COLSUM", "QBP_FLAG_PAIRWISE_COLSUM"), ("FLAG_DENSE_F_COLSUM", "QBP_FLAG_DENSE_F_COLSUM"),
                  ("FLAG_DENSE_F_COLSUM_ITER0", "QBP_FLAG_DENSE_F_COLSUM_ITER0"), ("FLAG_FAST_MATH", "QBP_FLAG_FAST_MATH"),
                  ("SUM_PRODUCT", "QBP_SUM_PRODUCT"), ("DAMPED_SP", "QBP_DAMPED_SP"), ("MIN_SUM", "QBP_MIN_SUM"),
                  ("OPT_OSD_BIG", "QBP_OPT_OSD_BIG"), ("OPT_KERNEL", "QBP_OPT_KERNEL"),
                  ("OPT_FORCE_GENERIC", "QBP_OPT_FORCE_GENERIC"), ("OPT_GENERAL_THREADS", "QBP_OPT_GENERAL_THREADS"),
                  ("E_UNSUPPORTED", "QBP_E_UNSUPPORTED")):
        assert getattr(_lib, py) == enum[c], (py, c)
    flags = [v for k, v in enum.items() if k.startswith("QBP_FLAG_")]
    assert len(set(flags)) == len(flags) and all(v & (v - 1) == 0 for v in flags)      # distinct single bits
    # the drop-in's switch for the approximate arithmetic is off unless asked for
    from qldpc_amd import bp
    assert bp.FAST_MATH == (os.environ.get("QBP_FAST_MATH", "0") not in ("0", "")) and (bp._math_flag() != 0) == bp.FAST_MATH
