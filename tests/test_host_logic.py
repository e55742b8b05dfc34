"""The host side of the solver (torchpdlp_amd/solver.py: restart decisions, counters, primal weight, termination)
without a GPU: a stand-in engine with the PdlpEngine interface whose arithmetic is the CPU ORACLE, driven by the
product's own ``run_pdlp`` / ``PdhgDriver``, against the reference's recorded runs (tests/golden/solve_trace.npz).
This checks the control flow the HIP engine plugs into; the kernels themselves are checked on the GPU."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from torchpdlp_amd import _native as N
from torchpdlp_amd.solver import PdhgDriver, run_pdlp


class OracleEngine:
    """PdlpEngine's interface (what solver.py uses) on top of oracle.OracleLP -- tests only."""

    def __init__(self, o: orc.OracleLP, unscaled=None):
        self.o, self.unscaled = o, unscaled          # unscaled = (D_col, D_row, OracleLP of the original problem)
        self.dtype, self.device, self.comm = torch.float32, torch.device("cpu"), None
        self.n, self.m, self.nl, self.ml = o.n, o.m, o.n, o.m
        self.q, self.c = torch.from_numpy(o.q), torch.from_numpy(o.c)
        self.t = np.float32

    def set_iterate(self, x, y):
        self.x, self.y = x.numpy().astype(np.float32).copy(), y.numpy().astype(np.float32).copy()
        self.xp, self.yp = self.x, self.y
        self._reset()
        self.x_last, self.y_last = self.x.copy(), self.y.copy()

    def _reset(self):
        self.xs, self.ys, self.es = np.zeros_like(self.x), np.zeros_like(self.y), self.t(0)

    def set_step(self, eta, omega, theta=1.0, iteration=0):
        self.eta, self.omega, self.theta, self.k = self.t(eta), self.t(omega), theta, int(iteration)

    def set_omega(self, omega):
        self.omega = self.t(omega)

    def iterate(self, iters, adaptive):
        for _ in range(iters):
            self.k += 1
            self._undo = (self.x, self.y, self.xp, self.yp, self.xs.copy(), self.ys.copy(), self.es)
            self.xp, self.yp = self.x, self.y
            if adaptive:
                self.x, self.y, w, self.eta, info = self.o.step_adaptive(self.x, self.y, self.eta, self.omega, self.theta, self.k)
                self.accepted = info["accepted"]
            else:
                self.x, self.y = self.o.step_fixed(self.x, self.y, self.eta, self.omega, self.theta)
                w = self.eta
            self.xs += w * self.x
            self.ys += w * self.y
            self.es = self.t(self.es + w)

    def scalars(self):
        return dict(accepted=float(getattr(self, "accepted", True)), eta=float(self.eta), k=self.k, eta_sum=float(self.es))

    def adaptive_retry(self):
        """the trial just taken is discarded; eta keeps the rule's eta' (pdlp_adaptive_retry)"""
        self.x, self.y, self.xp, self.yp, self.xs, self.ys, self.es = self._undo
        self.k -= 1

    def flush_average(self, adaptive=True):
        pass

    def compute_average(self):
        self.xa, self.ya = self.xs / self.es, self.ys / self.es

    def kkt(self, which, omega, unscaled=False):
        x, y = {N.CUR: (self.x, self.y), N.AVG: (getattr(self, "xa", None), getattr(self, "ya", None)), N.PREV: (self.xp, self.yp)}[which]
        if unscaled:
            D_col, D_row, ou = self.unscaled
            r = ou.kkt(D_col * x, D_row * y, omega)
        else:
            r = self.o.kkt(x, y, omega)
        return {k: float(v) for k, v in r.items()}

    def restart(self, which):
        if which == N.AVG:
            self.x, self.y = self.xa, self.ya
        self._reset()

    def restart_distance(self):
        return (float(np.sum((self.x_last - self.x).astype(np.float64) ** 2)), float(np.sum((self.y_last - self.y).astype(np.float64) ** 2)))

    def mark_restart_point(self):
        self.x_last, self.y_last = self.x.copy(), self.y.copy()

    def power_iteration(self, b0, iters):
        return float(self.o.power_iter(b0.numpy(), iters))

    def get_iterate(self, which=N.CUR):
        return torch.from_numpy(self.x.copy()), torch.from_numpy(self.y.copy())

    def synchronize(self):
        pass

    def infeas_reset(self):
        self.lam_prev = np.zeros(self.o.n, np.float32)

    def detect_infeasibility(self, tol):
        st, lam, _ = self.o.detect_infeasibility(self.x, self.y, self.xp, self.yp, self.lam_prev, tol)
        self.lam_prev = lam
        return st


def _lp(g, name):
    a = g.group(name)
    return orc.OracleLP(a["m"], a["n"], a["m_ineq"], a["rowptr"], a["colidx"], a["val"], a["c"], a["q"], a["l"], a["u"])


@pytest.fixture(scope="module", autouse=True)
def _one_thread():
    orc.set_threads(1)


RUNS = [(n, f"{a}_{p}") for n in ("mixed_27x32", "mixed_400x300", "box_200x150") for a in ("fixed", "adaptive") for p in ("nopw", "pw")]


@pytest.mark.parametrize("name,mode", RUNS)
def test_driver_follows_the_reference_run(golden, name, mode):
    g = golden("solve_trace.npz")
    o = _lp(g, name)
    r = g.group(f"{name}/{mode}")
    adaptive, pw = mode.startswith("adaptive"), mode.endswith("_pw")
    trace = dict(kkt=[], omega=[], restarts=[])
    x, obj, k, n, j, status, _ = run_pdlp(OracleEngine(o), tol=1e-4, verbose=False, primal_update=pw, adaptive=adaptive,
                                          sigma=float(r["sigma"]), trace=trace)
    assert status == "Solved" == str(r["status"])
    checks = len(trace["kkt"]) - n
    assert checks % 3 == 0 and j == k + checks + 2 * n                       # pdhg.py:86/93,128,154,165
    # the product's loop and the oracle's restatement of the reference loop take the same path, step for step
    xo, objo, ko, no, jo, so, _, to = orc.pdlp_algorithm(o, tol=1e-4, adaptive=adaptive, primal_update=pw, sigma=np.float32(r["sigma"]))
    assert (k, n, j, status) == (ko, no, jo, so)
    assert trace["restarts"] == to["restarts"]
    np.testing.assert_allclose(trace["kkt"], to["kkt"], rtol=1e-6)
    np.testing.assert_allclose(trace["omega"], to["omega"], rtol=1e-6)
    np.testing.assert_allclose(x.numpy(), xo, rtol=1e-6, atol=1e-7)
    # and the reference's own record: identical decisions while rounding has not split the runs
    nfirst = 1 if adaptive else min(len(trace["restarts"]), len(r["restarts"]), 6)
    assert [tuple(v) for v in trace["restarts"][:nfirst]] == [tuple(int(t) for t in v) for v in r["restarts"][:nfirst]]
    if not adaptive and not pw:
        assert (k, n, j) == (int(r["k"]), int(r["n"]), int(r["j"]))
    assert abs(obj - float(r["obj"])) <= 2e-3 * (1 + abs(float(r["obj"])))


def test_kkt_pass_cap_and_time_limit(golden):
    g = golden("solve_trace.npz")
    o = _lp(g, "mixed_400x300")
    sig = float(g.group("mixed_400x300/fixed_nopw")["sigma"])
    x, obj, k, n, j, status, _ = run_pdlp(OracleEngine(o), max_kkt=100, verbose=False, sigma=sig)
    # pdhg.py:54,67: both loops stop once j >= max_kkt; the post-loop work (n += 1, two more passes) still runs
    xo, _, ko, no, jo, so, _, _ = orc.pdlp_algorithm(o, max_kkt=100, sigma=np.float32(sig))
    assert (k, n, j, status) == (ko, no, jo, so)
    assert status == "Unsolved (KKT passes limit exceeded)" and 100 <= j <= 106 and k < 100 and n >= 1
    x, obj, k, n, j, status, _ = run_pdlp(OracleEngine(o), time_limit=0, verbose=False, sigma=sig)
    assert status == "Unsolved (Time limit exceeded)" and (k, n, j) == (0, 0, 0) and np.isnan(obj)


def test_lazy_previous_kkt_changes_nothing(golden):
    """without a trace the driver evaluates KKT(previous) only when the 'necessary' test can fire"""
    g = golden("solve_trace.npz")
    o = _lp(g, "box_200x150")
    sig = float(g.group("box_200x150/fixed_pw")["sigma"])
    a = run_pdlp(OracleEngine(o), verbose=False, primal_update=True, sigma=sig, trace=dict(kkt=[], omega=[], restarts=[]))
    calls = []
    eng = OracleEngine(o)
    orig = eng.kkt
    eng.kkt = lambda which, omega, unscaled=False: (calls.append(which), orig(which, omega, unscaled))[1]
    b = run_pdlp(eng, verbose=False, primal_update=True, sigma=sig)
    assert a[1:6] == b[1:6] and np.array_equal(a[0].numpy(), b[0].numpy())
    assert calls.count(N.PREV) < calls.count(N.CUR)


def test_preconditioned_run_terminates_on_unscaled_residuals(golden):
    g = golden("solve_trace.npz")
    ou = _lp(g, "mixed_400x300")
    os_, D_col, D_row, _ = ou.ruiz()
    sig = float(os_.power_iter(np.ones(os_.n, np.float32), 50))
    drv_calls = []
    eng = OracleEngine(os_, unscaled=(D_col, D_row, ou))
    orig = eng.kkt
    eng.kkt = lambda which, omega, unscaled=False: (drv_calls.append(unscaled), orig(which, omega, unscaled))[1]
    x, obj, k, n, j, status, _ = run_pdlp(eng, verbose=False, precondition=True, primal_update=True, adaptive=True, sigma=sig)
    assert status == "Solved" and drv_calls.count(True) == n                    # one un-scaled evaluation per restart (pdhg.py:157-161)
    xo, objo, ko, no, jo, so, _, _ = orc.pdlp_algorithm(os_, precondition=True, primal_update=True, adaptive=True,
                                                          data_precond=(D_col, D_row, ou), sigma=np.float32(sig))
    assert (k, n, j, status) == (ko, no, jo, so) and abs(obj - objo) <= 1e-5 * (1 + abs(objo))
    opt = float(g.group("mixed_400x300/fixed_nopw")["opt_obj"])
    assert abs(obj - opt) <= 2e-3 * (1 + abs(opt))


@pytest.mark.parametrize("name", ["primal_infeasible_box", "unbounded_ray", "unbounded_free_below", "feasible_boxed", "feasible_mixed",
                                  "mixed_27x32", "box_200x150"])
@pytest.mark.parametrize("mode", ["fixed_0.0001", "fixed_0.01", "adaptive_0.0001", "adaptive_0.01"])
def test_infeasibility_detection_in_the_driver(golden, name, mode):
    """the product's loop with the detector on (pdhg.py:89-101): same exit, counters and status as the restated
    reference loop, and as the reference's own recorded run where rounding cannot split them"""
    g = golden("infeasibility.npz")
    a, r = g.group(f"solve/{name}"), g.group(f"solve/{name}/{mode}")
    o = orc.OracleLP.from_dense(a["K"], a["m_ineq"], a["c"], a["q"], a["l"], a["u"])
    ad = mode.startswith("adaptive")
    sig = float(o.power_iter(r["b0"], 100))
    x, obj, k, n, j, status, _ = run_pdlp(OracleEngine(o), max_kkt=20_000, tol=1e-4, verbose=False, primal_update=ad, adaptive=ad,
                                          sigma=sig, infeasibility_detect=True, infeas_tol=float(r["infeas_tol"]))
    xo, objo, ko, no, jo, so, _, _ = orc.pdlp_algorithm(o, max_kkt=20_000, tol=1e-4, adaptive=ad, primal_update=ad, sigma=np.float32(sig),
                                                          infeasibility_detect=True, infeas_tol=float(r["infeas_tol"]))
    assert (k, n, j, status) == (ko, no, jo, so)
    assert abs(obj - objo) <= 1e-5 * (1 + abs(objo))
    np.testing.assert_allclose(x.numpy(), xo, rtol=1e-6, atol=1e-7)
    if not ad or not name[0] in "mb":
        assert (k, n, j, status) == (int(r["k"]), int(r["n"]), int(r["j"]), str(r["status"]))


def test_infeasibility_detection_respects_the_pass_cap(golden):
    g = golden("infeasibility.npz")
    a = g.group("solve/mixed_27x32")
    o = orc.OracleLP.from_dense(a["K"], a["m_ineq"], a["c"], a["q"], a["l"], a["u"])
    sig = float(o.power_iter(g.group("solve/mixed_27x32/fixed_0.0001")["b0"], 100))
    for cap in (1, 2, 7, 50, 101):
        got = run_pdlp(OracleEngine(o), max_kkt=cap, verbose=False, sigma=sig, infeasibility_detect=True, infeas_tol=1e-9)
        want = orc.pdlp_algorithm(o, max_kkt=cap, sigma=np.float32(sig), infeasibility_detect=True, infeas_tol=1e-9)
        assert got[2:6] == want[2:6]


def test_bench_command_line_contract(monkeypatch):
    """the driver's invocations parse (`--gpus N --steps K --warmup W`), the defaults are the metric's configuration, and the
    second half of the metric (time to tolerance) is part of a default run; the recorded tight-tolerance runs are readable"""
    import importlib, json, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    monkeypatch.delenv("PDLP_BENCH_N", raising=False)
    monkeypatch.delenv("PDLP_BENCH_NNZ", raising=False)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.n, a.nnz_per_row, a.mode, a.dtype) == (None, 10_000_000, 100, "adaptive", "f32")   # (None: the launcher's WORLD_SIZE, else 1)
    assert not a.ruiz and a.lib_comm == "auto"
    assert a.solve_tol == 1e-4 and a.solve_limit > 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)
    rec = json.load(open(os.path.join(root, "profiles", "time_to_tol.json")))
    assert rec["n10000000_k100"]["1e-8"]["seconds"] < 1200            # inside one GPU call of this pool
    tr = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    assert set(tr["n10000000_k100_f32_adaptive_g1_tiled"]) >= {"primal", "dual"}


def test_offsets_are_64_bit_end_to_end():
    """VERDICT r3 (missing 4): one handle may hold more than 2^31 - 1 non-zeros per matrix copy.  Row pointers are int64 from the
    host arrays to the C ABI (include/pdlp_hip.h: K_rowptr, KT_rowptr, the tiles' row-block bases, the schedule), column indices stay int32."""
    import os, re, torch
    from torchpdlp_amd.sparse import CsrPair, _counts_to_rowptr
    rp = _counts_to_rowptr(torch.tensor([2 ** 31, 2 ** 31, 5], dtype=torch.int64))
    assert rp.dtype == torch.int64 and rp.tolist() == [0, 2 ** 31, 2 ** 32, 2 ** 32 + 5]
    K = CsrPair(2, 3, torch.tensor([0, 2, 3], dtype=torch.int32), torch.tensor([0, 2, 1]), torch.tensor([1.0, 2.0, 3.0]))
    assert K.rowptr.dtype == K.t_rowptr.dtype == torch.int64 and K.colidx.dtype == K.t_colidx.dtype == torch.int32
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "pdlp_hip.h")).read()
    for field in ("K_rowptr", "KT_rowptr", "blk_base"):          # (tile offsets are 32-bit relative to their row block's 64-bit base)
        assert re.search(r"const int64_t\*\s+" + field + ";", hdr), field
    assert "const int64_t** blocks" in hdr


def test_chunked_transpose_equals_the_single_sort():
    """torch.sort takes at most INT_MAX elements, so matrices beyond 2^30 entries are transposed in row chunks (sparse.csr_transpose):
    the same arrays as the one stable sort, incl. a row longer than a chunk and empty rows"""
    import torch
    from torchpdlp_amd.sparse import csr_transpose
    g = torch.Generator().manual_seed(0)
    m, n = 300, 211
    lens = torch.randint(0, 9, (m,), generator=g)
    lens[17], lens[100] = 150, 0
    rp = torch.zeros(m + 1, dtype=torch.int64)
    rp[1:] = torch.cumsum(lens, 0)
    ci = torch.randint(0, n, (int(rp[-1]),), generator=g).to(torch.int32)
    va = torch.randn(int(rp[-1]), generator=g)
    ref = csr_transpose(rp, ci, va, m, n)
    for chunk in (37, 64, 1000):
        got = csr_transpose(rp, ci, va, m, n, chunk_nnz=chunk)
        assert all(torch.equal(a, b) for a, b in zip(ref, got)), chunk


@pytest.mark.parametrize("name", ["mixed_27x32", "box_200x150"])
def test_driver_adaptive_retry_follows_the_references_experiment(golden, name):
    """PdhgDriver(adaptive_retry=True) -- SURVEY quirk Q1's optional flag -- issues a rejected adaptive iteration again with the shrunk
    step size until it is accepted and counts one KKT pass per trial: over the first restart period its trial counts and iterate are
    those of the reference's own retry-loop experiment (enhancements/test_ass.py, recorded in adaptive_retry.npz; no restarts or
    averaging there, so the comparison ends at the first check)"""
    g = golden("adaptive_retry.npz")
    a, r = g.group(name), g.group(f"{name}/loose")
    o = orc.OracleLP(a["m"], a["n"], a["m_ineq"], a["rowptr"], a["colidx"], a["val"], a["c"], a["q"], a["l"], a["u"])
    eng = OracleEngine(o)
    drv = PdhgDriver(eng, restart_period=40, adaptive=True, adaptive_retry=True, tol=0.0)
    drv.start(np.float32(0.9) / np.float32(r["eta0"]))
    eng.set_omega(np.float32(r["omega"]))
    drv.omega = np.float32(r["omega"])
    done = drv.advance(12)
    assert done == 12 and drv.k == 12 and drv.tt == 12
    assert drv.trials == int(r["trials"][:12].sum()) and drv.j == drv.trials and drv.trials > 12
    np.testing.assert_allclose(eng.x, r["x12"], rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(r["x12"]).max())))
    # the single-trial driver (the live package's behaviour, the default) keeps the rejected step: one pass per iteration
    eng2 = OracleEngine(o)
    drv2 = PdhgDriver(eng2, restart_period=40, adaptive=True, tol=0.0)
    drv2.start(np.float32(0.9) / np.float32(r["eta0"]))
    eng2.set_omega(np.float32(r["omega"]))
    drv2.advance(12)
    assert drv2.j == 12 and not np.allclose(eng2.x, eng.x)


def test_exportable_workspace_sizes():
    """a sharded engine's workspace may be opened by other processes over HIP IPC: its allocation avoids the sizes for which
    hipIpcOpenMemHandle never returns on ROCm 7.2 (bit 31 of the size set: measured 3 000 / 3 826 / 4 000 / 6 500 MB hang,
    1 700 / 2 000 / 4 096 / 5 000 / 9 000 MB open)"""
    from torchpdlp_amd.engine import exportable_bytes
    MB, G4 = 1 << 20, 1 << 32
    for mb in (1, 700, 1700, 2000, 2046):                       # below 2 GiB: only rounded to 2 MB
        assert exportable_bytes(mb * MB) == -(-mb // 2) * 2 * MB
    for mb in (2048, 3000, 3826, 4000, 4095):                   # 2 - 4 GiB -> 4 GiB
        assert exportable_bytes(mb * MB) == G4
    assert exportable_bytes(G4) == G4 and exportable_bytes(G4 + 1) == G4 + 2 * MB
    assert exportable_bytes(5000 * MB) == 5000 * MB             # 4 - 6 GiB: fine
    assert exportable_bytes(6500 * MB) == 2 * G4                # 6 - 8 GiB -> 8 GiB
    assert exportable_bytes(9000 * MB) == 9000 * MB
    for n in (123, 2 ** 31 - 1, 2 ** 31, 3 * 2 ** 31 + 5, 7 * 2 ** 31):
        s = exportable_bytes(n)
        assert s >= n and s % (2 * MB) == 0 and not (s & 0x80000000)
