"""End-to-end on the GPU: MPS file -> solve_lp / CLI, against the reference's recorded afiro runs and HiGHS."""
import csv
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import torchpdlp_amd as tp
from tests.conftest import GOLDEN

MPS_DIR = os.path.join(GOLDEN, "mps")
AFIRO_OPT = -464.7531428571


@pytest.mark.parametrize("mode", ["fixed_nopw_noruiz", "adaptive_pw_noruiz", "fixed_pw_ruiz", "adaptive_pw_ruiz"])
def test_afiro_like_the_reference(golden, mode):
    """BASELINE.json configs[0]: Netlib afiro through mps_to_standard_form + the solver, tol 1e-4."""
    r = golden("afiro.npz").group(f"afiro/{mode}")
    adaptive, pw, ruiz = mode.startswith("adaptive"), "_pw_" in mode, mode.endswith("_ruiz")
    c, K, q, m_ineq, l, u = tp.mps_to_standard_form(os.path.join(MPS_DIR, "afiro.mps"), device="cuda:0")
    dp = None
    Ks, cs, qs, ls, us = K, c, q, l, u
    if ruiz:
        Ks, cs, qs, ls, us, dp, _ = tp.ruiz_precondition(c, K, q, l, u, device="cuda:0")
    b0 = torch.tensor(r["b0"], device="cuda:0")
    x, obj, k, n, j, status, _ = tp.pdlp_algorithm(Ks, m_ineq, cs, qs, ls, us, "cuda:0", max_kkt=400_000, tol=1e-4, verbose=False,
                                                   precondition=ruiz, primal_update=pw, adaptive=adaptive, data_precond=dp, b0=b0)
    assert status == "Solved" == str(r["status"])
    assert abs(obj - AFIRO_OPT) <= 1e-3 * (1 + abs(AFIRO_OPT))                  # SURVEY 8d cfg 1
    assert abs(obj - float(r["obj"])) <= 2e-3 * (1 + abs(AFIRO_OPT))
    assert abs(k - int(r["k"])) <= 0.6 * int(r["k"]) + 80
    if mode == "fixed_nopw_noruiz":
        assert (k, n, j) == (int(r["k"]), int(r["n"]), int(r["j"]))             # same restart trace as the reference
        np.testing.assert_allclose(x.cpu().numpy().ravel(), r["x"], rtol=2e-3, atol=2e-3)


def test_solve_lp_unscales_and_is_feasible():
    res = tp.solve_lp(os.path.join(MPS_DIR, "afiro.mps"), precondition=True, primal_weight_update=True, adaptive_stepsize=True, seed=3)
    assert res.status == "Solved" and abs(res.objective - AFIRO_OPT) <= 1e-3 * (1 + abs(AFIRO_OPT))
    c, K, q, m_ineq, l, u = tp.mps_to_standard_form(os.path.join(MPS_DIR, "afiro.mps"), device="cuda:0")
    x = res.x.double()
    assert abs(float((c.double() * x).sum()) - res.objective) <= 1e-2           # x is the ORIGINAL problem's point
    r = K.to_dense().double() @ x - q.double()
    viol = torch.cat([r[:m_ineq].clamp(max=0), r[m_ineq:]])
    assert float(viol.norm()) <= 2e-4 * (1 + float(q.norm()))
    assert res.as_tuple()[2:6] == (res.iterations, res.restarts, res.kkt_passes, res.status)
    # arrays instead of a path, dense K as the reference passes it
    res2 = tp.solve_lp((c, K.to_dense(), q, m_ineq, l, u), adaptive_stepsize=True, seed=3)
    assert res2.status == "Solved" and abs(res2.objective - AFIRO_OPT) <= 1e-3 * (1 + abs(AFIRO_OPT))


def test_float64_reaches_a_tolerance_below_float32_resolution():
    res = tp.solve_lp(os.path.join(MPS_DIR, "afiro.mps"), tol=1e-8, precondition=True, primal_weight_update=True,
                      adaptive_stepsize=True, dtype=torch.float64, seed=3, max_kkt=2_000_000)
    assert res.status == "Solved" and abs(res.objective - AFIRO_OPT) <= 2e-8 * (1 + 2 * abs(AFIRO_OPT))   # gap test: 1e-8 (1+|p|+|d|)


def test_mixed_precision_with_ruiz_reaches_1e_8_on_afiro():
    """BASELINE configs[4]'s combination at a tolerance float32 cannot certify: Ruiz + adaptive step + primal weight, mixed precision.
    The Ruiz-scaled matrix is not float32-valued: delta mode iterates on its float32 rounding and evaluates the anchors and the
    (un-scaled) termination test with the true float64 matrix."""
    c, K, q, m_ineq, l, u = tp.mps_to_standard_form(os.path.join(MPS_DIR, "afiro.mps"), device="cuda:0", dtype=torch.float64)
    Ks, cs, qs, ls, us, dp, _ = tp.ruiz_precondition(c, K, q, l, u, device="cuda:0")
    res = {}
    for kw in (dict(), dict(precision="mixed")):
        x, obj, k, n, j, status, _ = tp.pdlp_algorithm(Ks, m_ineq, cs, qs, ls, us, "cuda:0", tol=1e-8, verbose=False, precondition=True,
                                                       primal_update=True, adaptive=True, data_precond=dp, seed=3, max_kkt=2_000_000, **kw)
        assert status == "Solved", (kw, status)
        assert abs(obj - AFIRO_OPT) <= 2e-8 * (1 + 2 * abs(AFIRO_OPT))
        res[bool(kw)] = (obj, k)
    assert 0.3 * res[False][1] <= res[True][1] <= 3 * res[False][1]


def test_solve_lp_and_cli_in_mixed_precision(tmp_path):
    """`precision="mixed"` of solve_lp / `--dtype mixed` of the CLI: 1e-8 on afiro with and without Ruiz, from the MPS file"""
    from torchpdlp_amd.__main__ import main
    for ruiz in (False, True):
        res = tp.solve_lp(os.path.join(MPS_DIR, "afiro.mps"), tol=1e-8, precondition=ruiz, primal_weight_update=True,
                          adaptive_stepsize=True, precision="mixed", seed=3, max_kkt=2_000_000)
        assert res.status == "Solved" and abs(res.objective - AFIRO_OPT) <= 2e-8 * (1 + 2 * abs(AFIRO_OPT))
        assert res.x.dtype == torch.float64
    with pytest.raises(ValueError):
        tp.solve_lp(os.path.join(MPS_DIR, "afiro.mps"), precision="half")
    one = tmp_path / "in"
    one.mkdir()
    import shutil
    shutil.copy(os.path.join(MPS_DIR, "afiro.mps"), one / "afiro.mps")
    rc = main(["--instance_path", str(one), "--output_path", str(tmp_path / "out"), "--adaptive_stepsize", "--primal_weight_update",
               "--precondition", "--dtype", "mixed", "--tolerance", "1e-8", "--seed", "1", "--max_kkt", "2000000"])
    assert rc == 0
    row = list(csv.DictReader(open(tmp_path / "out" / "solver_results.csv")))[0]
    assert row["Status"] == "Solved" and abs(float(row["Objective"]) - AFIRO_OPT) <= 1e-5


def test_cli_writes_the_reference_csv_schema(tmp_path):
    from torchpdlp_amd.__main__ import COLUMNS, main
    rc = main(["--instance_path", MPS_DIR, "--output_path", str(tmp_path), "--adaptive_stepsize", "--primal_weight_update",
               "--seed", "1", "--max_kkt", "200000"])
    assert rc == 0
    rows = list(csv.DictReader(open(tmp_path / "solver_results.csv")))
    assert list(rows[0].keys()) == COLUMNS
    by = {r["File"]: r for r in rows}
    assert sorted(by) == ["afiro.mps", "all_eq.mps", "all_ineq.mps", "features.mps", "marker.mps"]
    assert by["afiro.mps"]["Status"] == "Solved" and abs(float(by["afiro.mps"]["Objective"]) - AFIRO_OPT) < 0.5
    # a file that does not load: the reference's own row for it (main.py:92-103; short messages carry no prefix) -- and the run went on
    assert by["marker.mps"]["Status"].startswith("could not convert string to float") and by["marker.mps"]["Objective"] == "N/A"
    assert by["all_eq.mps"]["Status"] == "Solved" and by["all_ineq.mps"]["Status"] == "Solved"


def test_fishnet_warm_start_matches_the_oracle_restatement(golden):
    """spectral_casting.py:65-159 (opt-in --fishnet): same population, same breeding weights -> same survivor"""
    from oracle import oracle as orc
    from torchpdlp_amd import _native as N
    g = golden("solve_trace.npz")
    a = g.group("mixed_400x300")
    o = orc.OracleLP(a["m"], a["n"], a["m_ineq"], a["rowptr"], a["colidx"], a["val"], a["c"], a["q"], a["l"], a["u"])
    orc.set_threads(1)
    t = lambda v, dt=torch.float32: torch.tensor(np.asarray(v), dtype=dt, device="cuda:0")
    K = tp.CsrPair(int(a["m"]), int(a["n"]), t(a["rowptr"], torch.int32), t(a["colidx"], torch.int32), t(a["val"]))
    eng = tp.PdlpEngine.from_full(K, t(a["c"]), t(a["q"]), t(a["l"]), t(a["u"]), int(a["m_ineq"]))
    gen = torch.Generator().manual_seed(5)
    pts, r = tp.sample_points(eng, 3, gen)                     # 8 points
    assert pts.shape == (300, 8) and abs(r - float(g.group("mixed_400x300/fixed_nopw")["sigma"])) < 0.05 * r
    np.testing.assert_allclose(torch.norm(pts - r / 300 ** 0.5, dim=0).numpy(), r, rtol=1e-4)   # all on the sphere
    eta = 0.9 / r
    gen2 = torch.Generator().manual_seed(6)
    x, y = tp.fishnet(eng, pts.clone(), s=2, k=8, eta=eta, generator=gen2)
    gen3 = torch.Generator().manual_seed(6)

    def weights():
        n_keep = [2]          # 8 -> 4 (round 0, no breeding) -> 2 (round 1: breed 4-2-1 = 1 point from 2 survivors) ...
        while True:
            yield torch.rand(n_keep[0], generator=gen3).numpy()
    xo, yo = orc.fishnet(o, pts.numpy(), s=2, k=8, eta=eta, weights=weights())
    np.testing.assert_allclose(x.cpu().numpy(), xo, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(y.cpu().numpy(), yo, rtol=2e-4, atol=2e-4)
    # as a warm start it is accepted by the solver (x_init / y_init of pdlp_algorithm, pdhg.py:31-33)
    xs, obj, k, n, j, status, _ = tp.pdlp_algorithm(K, int(a["m_ineq"]), t(a["c"]), t(a["q"]), t(a["l"]), t(a["u"]), "cuda:0",
                                                    verbose=False, adaptive=True, primal_update=True, x_init=x, y_init=y, seed=1)
    assert status == "Solved" and abs(obj - float(g.group("mixed_400x300/fixed_nopw")["opt_obj"])) < 0.2


@pytest.mark.parametrize("j", [8, 16, 32])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_population_kernels_match_the_single_vector_path(golden, j, dtype):
    """pdlp_mv_steps / pdlp_mv_gap (PDHG_step spectral_casting.py:254-293, get_best_pts :215-234): every column of the
    population equals the same point taken through fixed_one_step_pdhg and the KKT pass, and the oracle's"""
    from oracle import oracle as orc
    from torchpdlp_amd import _native as N
    g = golden("solve_trace.npz")
    a = g.group("mixed_400x300")
    npd = np.float32 if dtype == torch.float32 else np.float64
    o = orc.OracleLP(a["m"], a["n"], a["m_ineq"], a["rowptr"], a["colidx"], a["val"], a["c"], a["q"], a["l"], a["u"], dtype=npd)
    orc.set_threads(1)
    t = lambda v, dt=dtype: torch.tensor(np.asarray(v), dtype=dt, device="cuda:0")
    K = tp.CsrPair(int(a["m"]), int(a["n"]), t(a["rowptr"], torch.int32), t(a["colidx"], torch.int32), t(a["val"]))
    eng = tp.PdlpEngine.from_full(K, t(a["c"]), t(a["q"]), t(a["l"]), t(a["u"]), int(a["m_ineq"]))
    gen = torch.Generator().manual_seed(3 + j)
    X = torch.randn(300, j, generator=gen, dtype=torch.float64).to("cuda:0", dtype)
    Y = torch.randn(400, j, generator=gen, dtype=torch.float64).to("cuda:0", dtype)
    X0, Y0 = X.clone(), Y.clone()
    eta, omega = 0.07, 1.3
    for steps in (1, 6):                                    # odd and even: the buffers alternate inside the call
        X, Y = X0.clone(), Y0.clone()
        eng.mv_steps(X, Y, steps, eta, omega, 1.0)
        gaps = eng.mv_gap(X, Y)
        tol = 2e-5 if dtype == torch.float32 else 1e-12
        for p in (0, j // 2, j - 1):
            eng.set_iterate(X0[:, p].contiguous(), Y0[:, p].contiguous())
            eng.set_step(eta, omega, 1.0, 0)
            eng.iterate(steps, False)
            xs, ys = eng.get_iterate(N.CUR)
            np.testing.assert_allclose(X[:, p].cpu().numpy(), xs.cpu().numpy(), rtol=tol, atol=tol)
            np.testing.assert_allclose(Y[:, p].cpu().numpy(), ys.cpu().numpy(), rtol=tol, atol=tol)
            xo, yo = X0[:, p].cpu().numpy(), Y0[:, p].cpu().numpy()
            for _ in range(steps):
                xo, yo = o.step_fixed(xo, yo, eta, omega, 1.0)
            np.testing.assert_allclose(X[:, p].cpu().numpy(), xo, rtol=5 * tol, atol=5 * tol)
            ref = o.kkt(xo, yo, omega)
            scale = 1 + abs(float(ref["p"])) + abs(float(ref["d_adj"]))
            assert abs(gaps[p] - float(ref["gap"])) <= 20 * tol * scale
            assert abs(gaps[p] - eng.kkt(N.CUR, omega)["gap"]) <= 20 * tol * scale
    with pytest.raises(ValueError):
        eng.mv_steps(X[:, :5].contiguous(), Y[:, :5].contiguous(), 1, eta, omega)


def test_fishnet_population_path_equals_point_by_point(golden):
    """32 points through the population kernels and through the single-vector kernels: same survivor"""
    g = golden("solve_trace.npz")
    a = g.group("mixed_400x300")
    t = lambda v, dt=torch.float32: torch.tensor(np.asarray(v), dtype=dt, device="cuda:0")
    K = tp.CsrPair(int(a["m"]), int(a["n"]), t(a["rowptr"], torch.int32), t(a["colidx"], torch.int32), t(a["val"]))
    eng = tp.PdlpEngine.from_full(K, t(a["c"]), t(a["q"]), t(a["l"]), t(a["u"]), int(a["m_ineq"]))
    pts, r = tp.sample_points(eng, 5, torch.Generator().manual_seed(11))
    assert pts.shape == (300, 32)
    outs = [tp.fishnet(eng, pts.clone(), s=2, k=8, eta=0.9 / r, generator=torch.Generator().manual_seed(12), multi_vector=mv)
            for mv in (True, False)]
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("name", ["mixed_400x300", "box_200x150", "mixed_27x32"])
@pytest.mark.parametrize("multi_vector", [True, False])
def test_fishnet_vs_reference_golden(golden, name, multi_vector):
    """spectral_cast (spectral_casting.py:5-29) = sample_points (:32-63) + fishnet (:65-159) with get_best_pts (:191-252),
    init_PDHG_vars (:161-189) and PDHG_step (:254-293), against the reference run recorded in tests/golden/fishnet.npz: a CPU
    generator with the recorded seed replays the reference's random stream draw for draw (power-iteration starts, points,
    breeding weights), so points, radius, eta, every round's gaps and survivor order, and the final (x, y) are comparable"""
    g = golden("fishnet.npz")
    a, r = g.group(name), g.group(name)
    t = lambda v, dt=torch.float32: torch.tensor(np.asarray(v), dtype=dt, device="cuda:0")
    K = tp.CsrPair(int(a["m"]), int(a["n"]), t(a["rowptr"], torch.int32), t(a["colidx"], torch.int32), t(a["val"]))
    eng = tp.PdlpEngine.from_full(K, t(a["c"]), t(a["q"]), t(a["l"]), t(a["u"]), int(a["m_ineq"]))
    gen = torch.Generator().manual_seed(int(r["seed"]))
    pts, radius = tp.sample_points(eng, int(r["i"]), gen)
    np.testing.assert_allclose(radius, float(r["radius"]), rtol=5e-5)
    np.testing.assert_allclose(pts.numpy(), r["pts0"], rtol=3e-5, atol=3e-5 * float(np.abs(r["pts0"]).max()))
    trace = []
    x, y = tp.fishnet(eng, torch.tensor(r["pts0"]), s=int(r["s"]), k=int(r["k"]), generator=gen, multi_vector=multi_vector, trace=trace)
    assert len(trace) == int(r["nrounds"])
    for rnd, (gaps, order) in enumerate(trace):
        ref = g.group(f"{name}/round{rnd}")
        scale = float(np.max(np.abs(ref["gaps"]))) + 1e-30
        np.testing.assert_allclose(gaps, ref["gaps"], rtol=3e-4, atol=3e-4 * scale, err_msg=f"round {rnd}")
        assert list(order) == [int(v) for v in ref["order"]], f"survivor order, round {rnd}"
    np.testing.assert_allclose(x.cpu().numpy(), r["x"], rtol=3e-4, atol=3e-4 * float(np.abs(r["x"]).max()))
    np.testing.assert_allclose(y.cpu().numpy(), r["y"], rtol=3e-4, atol=3e-4 * float(np.abs(r["y"]).max()))
    # the generator now stands where the reference's global RNG stood at the end: every draw was replayed
    assert gen.get_state().equal(_state_after(int(r["seed"]), a, r, g, name))
    # the top-level entry point does the same from the seed alone
    if multi_vector:
        x2, y2 = tp.spectral_cast(K, t(a["c"]), t(a["q"]), t(a["l"]), t(a["u"]), int(a["m_ineq"]), int(r["k"]), 2, int(r["i"]), "cuda:0",
                                  generator=torch.Generator().manual_seed(int(r["seed"])))
        np.testing.assert_allclose(x2.cpu().numpy(), r["x"], rtol=3e-4, atol=3e-4 * float(np.abs(r["x"]).max()))
        np.testing.assert_allclose(y2.cpu().numpy(), r["y"], rtol=3e-4, atol=3e-4 * float(np.abs(r["y"]).max()))


def _state_after(seed, a, r, g, name):
    """the CPU generator's state after the reference's sequence of draws (shapes as recorded)"""
    gen = torch.Generator().manual_seed(seed)
    n = int(a["n"])
    torch.randn(n, generator=gen)
    torch.randn(n, 2 ** int(r["i"]), generator=gen)
    torch.randn(n, generator=gen)
    for w in range(int(r["nweights"])):
        torch.rand(len(g[f"{name}/weights{w}"]), generator=gen)
    return gen.get_state()


def _instance_files():
    d = os.environ.get("PDLP_MPS_DIR")
    if not d or not os.path.isdir(d):
        return []
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.lower().endswith(".mps"))


@pytest.mark.parametrize("path", _instance_files() or [None], ids=lambda p: os.path.basename(p) if p else "no-PDLP_MPS_DIR")
def test_user_supplied_instances(path):
    """BASELINE configs[2] (a Mittelmann instance such as neos3) cannot be fetched offline; this is the hook that picks one up:
    every ``*.mps`` under ``$PDLP_MPS_DIR`` -- the reference's driver walks a folder the same way (main.py:83-85) -- is solved at
    the reference's default tolerance with Ruiz + adaptive step + primal weight, its first KKT pass is checked against the CPU
    oracle, and the objective against scipy's HiGHS.  Skipped when the variable is unset."""
    if path is None:
        pytest.skip("PDLP_MPS_DIR is not set")
    from oracle import oracle as orc                        # the checker
    from torchpdlp_amd import _native as N
    dev = torch.device("cuda", 0)
    c, K, q, m_ineq, l, u = tp.mps_to_standard_form(path, device=dev, compat=False)       # standard bound semantics: HiGHS reads it so
    # (1) one KKT pass at a seeded point, HIP vs oracle on the very same arrays
    h = lambda t: t.detach().cpu().numpy().reshape(-1)
    o = orc.OracleLP(K.m, K.n, m_ineq, h(K.rowptr), h(K.colidx), h(K.val), h(c), h(q), h(l), h(u))
    eng = tp.PdlpEngine.from_full(K, c, q, l, u, m_ineq)
    g = torch.Generator().manual_seed(1)
    x0 = torch.minimum(torch.maximum(torch.randn(K.n, generator=g), l.cpu().view(-1)), u.cpu().view(-1))
    y0 = torch.randn(K.m, generator=g)
    y0[:m_ineq].clamp_(min=0)
    eng.set_iterate(x0.to(dev), y0.to(dev))
    got, ref = eng.kkt(N.CUR, 1.0), o.kkt(x0.numpy(), y0.numpy(), 1.0)
    for key in ("pr", "dr", "p", "d_adj", "kkt"):
        np.testing.assert_allclose(got[key], float(ref[key]), rtol=2e-4, atol=2e-4 * (1 + abs(float(ref["kkt"]))))
    del eng
    # (2) the solve, against HiGHS on the same standard form
    res = tp.solve_lp((c, K, q, m_ineq, l, u), device=dev, tol=1e-4, precondition=True, primal_weight_update=True, adaptive_stepsize=True,
                      seed=0, max_kkt=int(os.environ.get("PDLP_MPS_MAX_KKT", "2000000")), time_limit=float(os.environ.get("PDLP_MPS_TIME_LIMIT", "600")))
    assert res.status == "Solved", (os.path.basename(path), res.status, res.iterations)
    import scipy.sparse as sp
    from scipy.optimize import linprog
    A = sp.csr_matrix((h(K.val).astype(np.float64), h(K.colidx), h(K.rowptr)), shape=(K.m, K.n))
    qn, cn = h(q).astype(np.float64), h(c).astype(np.float64)
    bounds = [(None if np.isinf(a) else float(a), None if np.isinf(b) else float(b)) for a, b in zip(h(l), h(u))]
    ref = linprog(cn, A_ub=-A[:m_ineq] if m_ineq else None, b_ub=-qn[:m_ineq] if m_ineq else None,
                  A_eq=A[m_ineq:] if m_ineq < K.m else None, b_eq=qn[m_ineq:] if m_ineq < K.m else None, bounds=bounds, method="highs")
    assert ref.status == 0, ref.message
    assert abs(res.objective - ref.fun) <= 2e-3 * (1 + abs(ref.fun)), (res.objective, ref.fun)


@pytest.mark.parametrize("direct", [False, True])
def test_cli_sharded_over_two_ranks(tmp_path, direct):
    """(``direct``: with ``--direct_exchange`` -- the iterations of every instance run without collectives, the ranks' handles
    connected over HIP IPC anew per instance; same rows.)
    The batch driver under a launcher (main.py:83-172 over two ranks that share this box's GPU, collectives over gloo): every rank
    parses on the host and holds only its blocks; rank 0 writes the reference's CSV.  A file that fails on every rank alike (it does
    not load) gets its row and the loop goes on (the sharded run used to leave at the first failure)."""
    import shutil
    import socket
    import subprocess
    import sys
    inst = tmp_path / "inst"
    inst.mkdir()
    for f in ("afiro.mps", "all_eq.mps", "marker.mps"):          # marker.mps: integer MARKER lines make the load fail (reference quirk)
        shutil.copy(os.path.join(MPS_DIR, f), inst / f)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PDLP_SHARE_GPU="1", PDLP_DIST_BACKEND="gloo", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs.append(subprocess.Popen([sys.executable, "-m", "torchpdlp_amd", "--instance_path", str(inst), "--output_path", str(tmp_path / "out"),
                                       "--adaptive_stepsize", "--primal_weight_update", "--precondition", "--seed", "3"]
                                      + (["--direct_exchange", "--verbose"] if direct else []),
                                      env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    rows = list(csv.DictReader(open(tmp_path / "out" / "solver_results.csv")))
    by = {r["File"]: r for r in rows}
    assert sorted(by) == ["afiro.mps", "all_eq.mps", "marker.mps"]
    assert by["afiro.mps"]["Status"] == "Solved" and abs(float(by["afiro.mps"]["Objective"]) - AFIRO_OPT) < 0.5
    assert by["all_eq.mps"]["Status"] == "Solved"
    assert by["marker.mps"]["Objective"] == "N/A"                 # the failure row, and the run went on past it
    assert "x 2 ranks" in outs[0]
    if direct:
        assert outs[0].count("direct exchange: on") == 2, outs[0]           # afiro and all_eq (marker.mps never gets that far)


def test_mixed_precision_with_ruiz_on_a_plus_minus_one_matrix():
    """Ruiz on a matrix of +-1 entries is the identity, so the scaled matrix IS float32-valued and needs no separate exact
    matrix -- `--dtype mixed --precondition` on such an LP (set cover, network LPs) used to be refused with a ValueError."""
    from torchpdlp_amd.synthetic import gen_lp
    lp = gen_lp(301, 403, 4, seed=21, recipe="mixed", ineq_frac=0.6, device="cuda:0", dtype=torch.float64)
    K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, torch.sign(lp.val) + (lp.val == 0))
    x_f = lp.x_feas.double()
    rows = torch.repeat_interleave(torch.arange(lp.m, device="cuda:0"), (lp.rowptr[1:] - lp.rowptr[:-1]).long())
    kx = torch.zeros(lp.m, dtype=torch.float64, device="cuda:0").index_add_(0, rows, K.val * x_f[lp.colidx.long()])
    q = kx.clone()
    q[:lp.m_ineq] -= 0.5                                        # x_feas stays feasible: the LP is feasible and (boxed or not) solvable
    res = tp.solve_lp((lp.c, K, q, lp.m_ineq, torch.maximum(lp.l, x_f - 5), torch.minimum(lp.u, x_f + 5)), tol=1e-6, precondition=True,
                      primal_weight_update=True, adaptive_stepsize=True, seed=1, precision="mixed", max_kkt=2_000_000)
    assert res.status == "Solved", res.status
    xs = res.x.view(-1).double()
    r = torch.zeros(lp.m, dtype=torch.float64, device="cuda:0").index_add_(0, rows, K.val * xs[lp.colidx.long()]) - q
    r[:lp.m_ineq].clamp_(max=0)
    assert float(r.norm()) <= 2e-6 * (1 + float(q.norm()))


@pytest.mark.gpu
def test_bench_line_contract_on_a_small_workload():
    """bench.py's one JSON line (the driver's contract): metric / value / unit / n_gpus / steps / warmup / ms_per_step / scaling / dtype /
    data / config.workload, the roofline object measured live (bound, achieved, peak, frac, traffic key, the kernel's launch time), the
    cpu_baseline object (value, cores, kind, sample) and the time-to-tolerance record -- on a 300k x 300k LP so that it takes seconds.
    The line is the LAST line of stdout and the only one that parses as JSON."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PDLP_BENCH_N="300000", PDLP_BENCH_NNZ="20")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "40", "--warmup", "5", "--cpu-sample-rows", "100000",
                        "--solve-limit", "30"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    d = json.loads(lines[-1])
    assert sum(1 for ln in lines if ln.lstrip().startswith("{")) == 1
    assert d["metric"] == "PDHG iterations/sec" and d["unit"] == "iterations/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 40, 5) and d["data"] == "synthetic" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] - 1000.0) < 1.0
    assert "300000x300000" in d["config"]["workload"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and "traffic" in rf
    assert rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["launch_ms"] > 0 and rf["algorithmic_bytes"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "oracle" in cb["sample"] and cb["machine_cores"] >= cb["cores"]
    assert d["time_to_tol"]["tol"] == 1e-4 and d["time_to_tol"]["status"] in ("Solved", "Unsolved (time limit exceeded)")
    assert d["timing"]["checks_in_timed_region"] == 1
