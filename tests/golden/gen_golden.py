#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE (SimplySnap/torchPDLP) on CPU.

Runs only in the build container, where the reference is mounted read-only at
/root/reference.  It imports the reference's live package (``PDLP/``), feeds it seeded
inputs and records inputs + outputs as small ``.npz`` fixtures next to this script.
Nothing of the reference's source is stored: fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Groups:
  G1 step_fixed.npz      fixed_one_step_pdhg              (step.py:3-40)
  G2 step_adaptive.npz   adaptive_one_step_pdhg           (step.py:43-115)
  G3 kkt.npz             compute_residuals_and_duality_gap / KKT_error (helpers.py:53-108)
  G4 solve_trace.npz     pdlp_algorithm end-to-end        (primal_dual_hybrid_gradient.py:7-181)
  G5 primal_weight.npz   primal_weight_update             (enhancements.py:73-78)
  G6 ruiz.npz            ruiz_precondition                (enhancements.py:4-71)
  G7 power_iter.npz      spectral_norm_estimate_torch     (helpers.py:41-51)
  G8 mps.npz             mps_to_standard_form on tests/golden/mps/*.mps (util.py:76-268)
  G9 afiro.npz           Netlib afiro: util.mps_to_standard_form + pdlp_algorithm (+ ruiz_precondition)
  G10 infeasibility.npz  detect_infeasibility (enhancements.py:80-161) and pdlp_algorithm(infeasibility_detect=True)
  G11 fishnet.npz        spectral_cast / sample_points / fishnet / init_PDHG_vars / get_best_pts (spectral_casting.py:5-293)
  G13 forced_trace_more.npz  the same for the other three seeded LPs (round 3)
  G12 forced_trace.npz   pdlp_algorithm(adaptive=True): state at the start and end of every 40-iteration block
"""
import contextlib
import io
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("PDLP_GOLDEN_OUT", HERE)      # where the .npz files go (tests regenerate into a scratch folder and compare)
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/PDLP")
sys.path.insert(0, ROOT)

import enhancements as ref_enh            # noqa: E402  (reference)
import helpers as ref_helpers             # noqa: E402  (reference)
import primal_dual_hybrid_gradient as ref_pdhg        # noqa: E402  (reference)
import primal_dual_hybrid_gradient_step as ref_step   # noqa: E402  (reference)
import util as ref_util                   # noqa: E402  (reference)
import spectral_casting as ref_sc         # noqa: E402  (reference)

from torchpdlp_amd.synthetic import csr_to_dense, gen_lp  # noqa: E402

torch.set_num_threads(1)   # deterministic reduction order


def col(v):
    return v.reshape(-1, 1).clone()


def lp_cases():
    """name -> SyntheticLP (float32, CPU)."""
    cases = {}
    cases["mixed_27x32"] = gen_lp(32, 27, 3, seed=11, recipe="mixed", ineq_frac=0.7)
    cases["mixed_400x300"] = gen_lp(300, 400, 5, seed=12, recipe="mixed", ineq_frac=0.8)
    cases["mixed_300x400_alleq"] = gen_lp(400, 300, 5, seed=13, recipe="mixed", ineq_frac=0.0)
    cases["mixed_200x260_allineq"] = gen_lp(260, 200, 4, seed=14, recipe="mixed", ineq_frac=1.0)
    cases["box_200x150"] = gen_lp(150, 200, 5, seed=15, recipe="box_v1", ineq_frac=0.8)      # (the recipe the stored fixtures were drawn with)
    return cases


def lp_arrays(lp):
    return dict(m=lp.m, n=lp.n, m_ineq=lp.m_ineq, rowptr=lp.rowptr.numpy().astype(np.int32), colidx=lp.colidx.numpy(),
                val=lp.val.numpy(), c=lp.c.numpy(), q=lp.q.numpy(), l=lp.l.numpy(), u=lp.u.numpy())


def put(out, prefix, d):
    for k, v in d.items():
        out[f"{prefix}/{k}"] = np.asarray(v)


def masks(l, u):
    is_neg_inf = torch.isinf(l) & (l < 0)
    is_pos_inf = torch.isinf(u) & (u > 0)
    l_dual = l.clone()
    u_dual = u.clone()
    l_dual[is_neg_inf] = 0
    u_dual[is_pos_inf] = 0
    return is_neg_inf, is_pos_inf, l_dual, u_dual


def start_state(lp, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(lp.n, generator=g) * 0.5
    x = torch.minimum(torch.maximum(x, lp.l), lp.u)
    y = torch.randn(lp.m, generator=g) * 0.5
    y[:lp.m_ineq] = y[:lp.m_ineq].clamp(min=0)
    K = csr_to_dense(lp)
    sig = float(torch.linalg.matrix_norm(K.double(), 2))
    eta = torch.tensor(0.9 / sig, dtype=torch.float32)
    cn, qn = torch.linalg.norm(lp.c), torch.linalg.norm(lp.q)
    omega = (cn / qn).to(torch.float32)
    return K, x, y, eta, omega


def g1_step_fixed(cases):
    out = {}
    for name, lp in cases.items():
        K, x, y, eta, omega = start_state(lp, 100)
        put(out, name, lp_arrays(lp))
        put(out, name, dict(x0=x.numpy(), y0=y.numpy(), eta=eta.numpy(), omega=omega.numpy(), theta=1.0))
        xx, yy = col(x), col(y)
        for it in range(1, 41):
            xx, yy, _, _ = ref_step.fixed_one_step_pdhg(xx, yy, col(lp.c), col(lp.q), K, col(lp.l), col(lp.u),
                                                       lp.m_ineq, eta, omega, 1.0)
            if it in (1, 2, 40):
                put(out, name, {f"x{it}": xx.flatten().numpy().copy(), f"y{it}": yy.flatten().numpy().copy()})
    np.savez_compressed(os.path.join(OUT, "step_fixed.npz"), **out)


def g2_step_adaptive(cases):
    out = {}
    for name, lp in cases.items():
        K, x, y, eta, omega = start_state(lp, 200)
        put(out, name, lp_arrays(lp))
        put(out, name, dict(x0=x.numpy(), y0=y.numpy(), omega=omega.numpy(), theta=1.0))
        for tag, scale, k in (("accept", 0.5, 1), ("reject", 25.0, 7), ("late", 1.0, 500)):
            e = eta * scale
            xn, yn, e_used, e_hat, j = ref_step.adaptive_one_step_pdhg(
                col(x), col(y), col(lp.c), col(lp.q), K, col(lp.l), col(lp.u), lp.m_ineq, e, omega, 1.0, k, 3)
            put(out, f"{name}/{tag}", dict(eta_in=e.numpy(), k=k, j_in=3, x1=xn.flatten().numpy(),
                                          y1=yn.flatten().numpy(), eta_used=e_used.numpy(),
                                          eta_hat=e_hat.numpy(), j_out=j))
        # chained: 12 adaptive steps the way the outer loop drives them (pdhg.py:80-112)
        xx, yy, e, j = col(x), col(y), eta * 4.0, 0
        etas, ws = [], []
        for k in range(1, 13):
            xx, yy, e_used, e_hat, j = ref_step.adaptive_one_step_pdhg(
                xx, yy, col(lp.c), col(lp.q), K, col(lp.l), col(lp.u), lp.m_ineq, e, omega, 1.0, k, j)
            ws.append(float(e_used))
            e = e_hat
            etas.append(float(e))
        put(out, f"{name}/chain", dict(eta_in=(eta * 4.0).numpy(), x12=xx.flatten().numpy(), y12=yy.flatten().numpy(),
                                      weights=np.array(ws, dtype=np.float64), etas=np.array(etas, dtype=np.float64), j_out=j))
    # denominator == 0 branch (step.py:99,104-105): x = y = 0 at l = 0 with c >= 0 keeps x at the bound
    n, m = 6, 4
    g = torch.Generator().manual_seed(5)
    K = torch.rand(m, n, generator=g)
    c = torch.rand(n, generator=g) + 0.1
    q = torch.rand(m, generator=g)
    l = torch.zeros(n)
    u = torch.full((n,), float("inf"))
    e = torch.tensor(0.3)
    om = torch.tensor(1.5)
    xn, yn, e_used, e_hat, j = ref_step.adaptive_one_step_pdhg(
        col(torch.zeros(n)), col(torch.zeros(m)), col(c), col(q), K, col(l), col(u), 2, e, om, 1.0, 4, 0)
    put(out, "denzero", dict(K=K.numpy(), c=c.numpy(), q=q.numpy(), l=l.numpy(), u=u.numpy(), m_ineq=2, eta_in=e.numpy(),
                             omega=om.numpy(), k=4, x1=xn.flatten().numpy(), y1=yn.flatten().numpy(),
                             eta_used=e_used.numpy(), eta_hat=e_hat.numpy(), j_out=j))
    np.savez_compressed(os.path.join(OUT, "step_adaptive.npz"), **out)


def g3_kkt(cases):
    out = {}
    for name, lp in cases.items():
        K, x, y, eta, omega = start_state(lp, 300)
        put(out, name, lp_arrays(lp))
        inn, ipn, ld, ud = (col(t) for t in masks(lp.l, lp.u))
        pts = {"rand": (x, y), "zero": (torch.zeros(lp.n), torch.zeros(lp.m)), "feas": (lp.x_feas, y * 0.1)}
        if lp.y_opt is not None:
            pts["opt"] = (lp.x_feas, lp.y_opt)
        for tag, (px, py) in pts.items():
            pr, dr, gap, po, da = ref_helpers.compute_residuals_and_duality_gap(
                col(px), col(py), col(lp.c), col(lp.q), K, lp.m_ineq, inn, ipn, ld, ud)
            kkt = ref_helpers.KKT_error(col(px), col(py), col(lp.c), col(lp.q), K, lp.m_ineq, omega, inn, ipn, ld, ud, "cpu")
            put(out, f"{name}/{tag}", dict(x=px.numpy(), y=py.numpy(), omega=omega.numpy(), pr=pr.numpy(), dr=dr.numpy(),
                                          gap=gap.numpy(), p=po.numpy(), d_adj=da.numpy(), kkt=kkt.numpy()))
    np.savez_compressed(os.path.join(OUT, "kkt.npz"), **out)


_RESTART_RE = re.compile(r"^(Sufficient|Necessary|Artificial) restart at iteration (\d+) using the (Average|Current) iterate")


def g4_solve_trace(cases):
    out = {}
    runs = []
    for name in ("mixed_27x32", "mixed_400x300", "box_200x150", "mixed_300x400_alleq"):
        for adaptive in (False, True):
            for pw in (False, True):
                runs.append((name, adaptive, pw))
    for name, adaptive, pw in runs:
        lp = cases[name]
        K = csr_to_dense(lp)
        seed = 1234
        torch.manual_seed(seed)
        b0 = torch.randn(lp.n, 1)            # what helpers.py:47 will draw after the same seed
        kkts = []
        orig_kkt = ref_pdhg.KKT_error

        def rec_kkt(*a, **kw):
            v = orig_kkt(*a, **kw)
            kkts.append(float(v))
            return v

        omegas = []
        orig_pw = ref_pdhg.primal_weight_update

        def rec_pw(*a, **kw):
            v = orig_pw(*a, **kw)
            omegas.append(float(v))
            return v

        eta0 = []
        orig_sn = ref_pdhg.spectral_norm_estimate_torch

        def rec_sn(*a, **kw):
            v = orig_sn(*a, **kw)
            eta0.append(float(v))
            return v

        ref_pdhg.KKT_error, ref_pdhg.primal_weight_update, ref_pdhg.spectral_norm_estimate_torch = rec_kkt, rec_pw, rec_sn
        buf = io.StringIO()
        try:
            torch.manual_seed(seed)
            with contextlib.redirect_stdout(buf):
                x, obj, k, n, j, status, _ = ref_pdhg.pdlp_algorithm(
                    K, lp.m_ineq, col(lp.c), col(lp.q), col(lp.l), col(lp.u), "cpu", max_kkt=100_000, tol=1e-4,
                    verbose=True, restart_period=40, precondition=False, primal_update=pw, adaptive=adaptive)
        finally:
            ref_pdhg.KKT_error, ref_pdhg.primal_weight_update, ref_pdhg.spectral_norm_estimate_torch = orig_kkt, orig_pw, orig_sn
        restarts = []
        for line in buf.getvalue().splitlines():
            mm = _RESTART_RE.match(line)
            if mm:
                restarts.append(("SNA".index(mm.group(1)[0]), int(mm.group(2)), int(mm.group(3) == "Average")))
        tag = f"{name}/{'adaptive' if adaptive else 'fixed'}_{'pw' if pw else 'nopw'}"
        put(out, name, lp_arrays(lp))
        put(out, tag, dict(b0=b0.flatten().numpy(), sigma=eta0[0], x=x.flatten().numpy(), obj=obj, k=k, n=n, j=j,
                           status=status, kkt_trace=np.array(kkts), omega_trace=np.array(omegas),
                           restarts=np.array(restarts, dtype=np.int64).reshape(-1, 3),
                           opt_obj=(lp.opt_obj if lp.opt_obj is not None else np.nan)))
        print(f"G4 {tag}: k={k} n={n} j={j} {status} obj={obj:.6f} opt={lp.opt_obj}")
    # known-answer for the KKT-pass counter: converges at the first restart
    K = torch.tensor([[1.0, 1.0]])
    c, q, l, u = torch.tensor([1.0, 2.0]), torch.tensor([1.0]), torch.zeros(2), torch.full((2,), float("inf"))
    for adaptive in (False, True):
        torch.manual_seed(7)
        b0 = torch.randn(2, 1)
        torch.manual_seed(7)
        with contextlib.redirect_stdout(io.StringIO()):
            x, obj, k, n, j, status, _ = ref_pdhg.pdlp_algorithm(K, 1, col(c), col(q), col(l), col(u), "cpu", tol=1e-4,
                                                                verbose=False, adaptive=adaptive)
        put(out, f"tiny/{'adaptive' if adaptive else 'fixed'}", dict(b0=b0.flatten().numpy(), x=x.flatten().numpy(), obj=obj,
                                                                    k=k, n=n, j=j, status=status))
        print(f"G4 tiny adaptive={adaptive}: k={k} n={n} j={j} {status} obj={obj}")
    put(out, "tiny", dict(K=K.numpy(), c=c.numpy(), q=q.numpy(), l=l.numpy(), u=u.numpy(), m_ineq=1))
    np.savez_compressed(os.path.join(OUT, "solve_trace.npz"), **out)


def g5_primal_weight():
    out = {}
    g = torch.Generator().manual_seed(9)
    for i, (n, m) in enumerate(((5, 3), (300, 400))):
        xp, x = torch.randn(n, 1, generator=g), torch.randn(n, 1, generator=g)
        yp, y = torch.randn(m, 1, generator=g), torch.randn(m, 1, generator=g)
        om = torch.tensor(0.37)
        put(out, f"case{i}", dict(x_prev=xp.flatten().numpy(), x=x.flatten().numpy(), y_prev=yp.flatten().numpy(),
                                  y=y.flatten().numpy(), omega=om.numpy(), theta=0.5,
                                  omega_new=ref_enh.primal_weight_update(xp, x, yp, y, om, 0.5).numpy()))
        # zero primal movement: omega must be returned unchanged (enhancements.py:76)
        put(out, f"case{i}_zero", dict(x_prev=x.flatten().numpy(), x=x.flatten().numpy(), y_prev=yp.flatten().numpy(),
                                       y=y.flatten().numpy(), omega=om.numpy(), theta=0.5,
                                       omega_new=ref_enh.primal_weight_update(x, x, yp, y, om, 0.5).numpy()))
    np.savez_compressed(os.path.join(OUT, "primal_weight.npz"), **out)


def g6_ruiz(cases):
    out = {}
    todo = {k: cases[k] for k in ("mixed_27x32", "mixed_400x300", "box_200x150")}
    for name, lp in todo.items():
        K = csr_to_dense(lp)
        variants = {"plain": K}
        if name == "mixed_27x32":
            Kz = K.clone()
            Kz[3, :] = 0.0      # all-zero row  -> eps branch (enhancements.py:50)
            Kz[:, 5] = 0.0      # all-zero col  -> eps branch (enhancements.py:55)
            Kz *= torch.logspace(-3, 3, lp.n).view(1, -1)   # badly scaled columns
            variants["zero_rowcol"] = Kz
        for tag, Kv in variants.items():
            for iters in (1, 20):
                Ks, cs, qs, ls, us, (D_col, D_row, *_), _ = ref_enh.ruiz_precondition(
                    col(lp.c), Kv, col(lp.q), col(lp.l), col(lp.u), device="cpu", max_iter=iters)
                put(out, f"{name}/{tag}/it{iters}", dict(K=Kv.numpy(), c=lp.c.numpy(), q=lp.q.numpy(), l=lp.l.numpy(), u=lp.u.numpy(),
                                                        K_s=Ks.numpy(), c_s=cs.flatten().numpy(), q_s=qs.flatten().numpy(),
                                                        l_s=ls.flatten().numpy(), u_s=us.flatten().numpy(),
                                                        D_col=D_col.flatten().numpy(), D_row=D_row.flatten().numpy()))
    np.savez_compressed(os.path.join(OUT, "ruiz.npz"), **out)


def g7_power_iter(cases):
    out = {}
    for name in ("mixed_27x32", "mixed_400x300"):
        lp = cases[name]
        K = csr_to_dense(lp)
        for iters in (10, 100):
            torch.manual_seed(77)
            b0 = torch.randn(lp.n, 1)
            torch.manual_seed(77)
            s = ref_helpers.spectral_norm_estimate_torch(K, num_iters=iters)
            put(out, f"{name}/it{iters}", dict(b0=b0.flatten().numpy(), sigma=s.numpy(),
                                              sigma_exact=float(torch.linalg.matrix_norm(K.double(), 2))))
        put(out, name, lp_arrays(lp))
    np.savez_compressed(os.path.join(OUT, "power_iter.npz"), **out)


def g8_mps():
    out = {}
    d = os.path.join(HERE, "mps")
    for fn in sorted(os.listdir(d)):
        if not fn.endswith(".mps"):
            continue
        try:
            c, K, q, m_ineq, l, u = ref_util.mps_to_standard_form(os.path.join(d, fn), device="cpu")
        except Exception as e:   # record load failures too (e.g. integer MARKER lines)
            out[f"{fn}/error"] = np.asarray(type(e).__name__)
            print(f"G8 {fn}: reference raised {type(e).__name__}: {e}")
            continue
        put(out, fn, dict(c=c.flatten().numpy(), K=K.numpy(), q=q.flatten().numpy(), m_ineq=m_ineq,
                          l=l.flatten().numpy(), u=u.flatten().numpy()))
        print(f"G8 {fn}: K {tuple(K.shape)} m_ineq={m_ineq}")
    np.savez_compressed(os.path.join(OUT, "mps.npz"), **out)


def g9_afiro():
    """Netlib afiro through the reference's own loader and solver (BASELINE.json configs[0])."""
    out = {}
    path = os.path.join(HERE, "mps", "afiro.mps")
    c, K, q, m_ineq, l, u = ref_util.mps_to_standard_form(path, device="cpu")
    for adaptive, pw, precond in ((False, False, False), (True, True, False), (False, True, True), (True, True, True)):
        Kr, cr, qr, lr, ur, dp = K, c, q, l, u, None
        if precond:
            Kr, cr, qr, lr, ur, dp, _ = ref_enh.ruiz_precondition(c, K, q, l, u, device="cpu")
            dp = tuple(t.clone() for t in dp)     # pdhg.py:159-160 clobbers l,u in place
        torch.manual_seed(2024)
        b0 = torch.randn(K.shape[1], 1)
        torch.manual_seed(2024)
        with contextlib.redirect_stdout(io.StringIO()):
            x, obj, k, n, j, status, _ = ref_pdhg.pdlp_algorithm(
                Kr, m_ineq, cr, qr, lr, ur, "cpu", max_kkt=400_000, tol=1e-4, verbose=False, restart_period=40,
                precondition=precond, primal_update=pw, adaptive=adaptive, data_precond=dp)
        tag = f"afiro/{'adaptive' if adaptive else 'fixed'}_{'pw' if pw else 'nopw'}_{'ruiz' if precond else 'noruiz'}"
        put(out, tag, dict(b0=b0.flatten().numpy(), x=x.flatten().numpy(), obj=obj, k=k, n=n, j=j, status=status))
        print(f"G9 {tag}: k={k} n={n} j={j} {status} obj={obj:.6f} (HiGHS -464.753143)")
    np.savez_compressed(os.path.join(OUT, "afiro.npz"), **out)


def tiny_infeas_lps():
    """name -> (K dense, m_ineq, c, q, l, u): hand-made LPs for the detector's branches."""
    inf = float("inf")
    T = lambda *v: torch.tensor(v, dtype=torch.float32)
    lps = {}
    # primal infeasible: x1 + x2 >= 3 with x in [0,1]^2 ... and an equality that is fine
    lps["primal_infeasible_box"] = (torch.tensor([[1.0, 1.0], [1.0, -1.0]]), 1, T(1.0, 1.0), T(3.0, 0.0), T(0.0, 0.0), T(1.0, 1.0))
    # primal infeasible with one-sided bounds: -x1 - x2 >= 1, x >= 0
    lps["primal_infeasible_cone"] = (torch.tensor([[-1.0, -1.0]]), 1, T(1.0, 2.0), T(1.0), T(0.0, 0.0), T(inf, inf))
    # unbounded: min -x1 - x2, x1 - x2 = 0, x >= 0
    lps["unbounded_ray"] = (torch.tensor([[1.0, -1.0]]), 0, T(-1.0, -1.0), T(0.0), T(0.0, 0.0), T(inf, inf))
    # unbounded below through a variable without lower bound: min x1, x1 + x2 >= 1, x1 <= 5 (l = -inf), x2 in [0,1]
    lps["unbounded_free_below"] = (torch.tensor([[1.0, 1.0]]), 1, T(1.0, 0.0), T(1.0), T(-inf, 0.0), T(5.0, 1.0))
    # feasible and bounded, every variable boxed: the detector fires once the iterates stall
    lps["feasible_boxed"] = (torch.tensor([[1.0, 2.0, 0.5], [1.0, -1.0, 1.0]]), 1, T(1.0, -1.0, 0.5), T(1.0, 0.25),
                             T(0.0, 0.0, -1.0), T(2.0, 1.5, 1.0))
    # feasible and bounded, c < 0 on a variable with u = +inf: no branch of the bound test holds for it
    lps["feasible_mixed"] = (torch.tensor([[1.0, 1.0, 1.0], [1.0, -1.0, 0.0]]), 1, T(-1.0, 1.0, 2.0), T(-4.0, 0.0),
                             T(0.0, 0.0, 0.0), T(inf, inf, 3.0))
    lps["feasible_mixed"] = (torch.tensor([[-1.0, -1.0, -1.0], [1.0, -1.0, 0.0]]), 1, T(-1.0, 1.0, 2.0), T(-4.0, 0.0),
                             T(0.0, 0.0, 0.0), T(inf, inf, 3.0))
    return lps


def g10_infeasibility(cases):
    out = {}
    # (a) the operator on consecutive PDHG iterates of the seeded LPs, over a ladder of tolerances so that every
    #     threshold test of the detector flips somewhere along it
    tols = [1e-6, 1e-4, 1e-3, 1e-2, 3e-2, 1e-1, 3e-1, 1.0, 3.0, 10.0, 1e2, 1e3, 1e5]
    for name, lp in cases.items():
        K, x, y, eta, omega = start_state(lp, 400)
        put(out, f"op/{name}", lp_arrays(lp))
        inn, ipn, _, _ = masks(lp.l, lp.u)
        c, q, l, u = col(lp.c), col(lp.q), col(lp.l), col(lp.u)
        xs, ys = [col(x)], [col(y)]
        for it in range(3):
            xn, yn, _, _ = ref_step.fixed_one_step_pdhg(xs[-1].clone(), ys[-1].clone(), c, q, K, l, u, lp.m_ineq, eta, omega, 1.0)
            xs.append(xn)
            ys.append(yn)
        lams = [ref_helpers.project_lambda_box(c - K.T @ yy, col(inn), col(ipn)) for yy in ys]
        for tag, (a, b, lam_prev) in {"s01": (0, 1, lams[0]), "s12": (1, 2, lams[1]), "s23z": (2, 3, torch.zeros_like(lams[0])),
                                      "same": (2, 2, lams[2])}.items():
            st = [ref_enh.detect_infeasibility(xs[b], ys[b], xs[a], ys[a], lams[b], lam_prev, c, q, K, l, u, lp.m_ineq,
                                               "cpu", tol=t) for t in tols]
            put(out, f"op/{name}/{tag}", dict(x=xs[b].flatten().numpy(), y=ys[b].flatten().numpy(),
                                             x_prev=xs[a].flatten().numpy(), y_prev=ys[a].flatten().numpy(),
                                             lam_prev=lam_prev.flatten().numpy(), tols=np.array(tols),
                                             status=np.array([s or "None" for s in st])))
            print(f"G10 op/{name}/{tag}:", [s or "-" for s in st])
    # (b) whole solves with the detector switched on
    solves = dict(tiny_infeas_lps())
    for name in ("mixed_27x32", "box_200x150", "mixed_200x260_allineq"):      # feasible, bounded: what the detector does to them
        lp = cases[name]
        solves[name] = (csr_to_dense(lp), lp.m_ineq, lp.c, lp.q, lp.l, lp.u)
    for name, (K, m_ineq, c, q, l, u) in solves.items():
        put(out, f"solve/{name}", dict(K=K.numpy(), m_ineq=m_ineq, c=c.numpy(), q=q.numpy(), l=l.numpy(), u=u.numpy()))
        for adaptive in (False, True):
            for itol in (1e-4, 1e-2):
                torch.manual_seed(99)
                b0 = torch.randn(K.shape[1], 1)
                torch.manual_seed(99)
                with contextlib.redirect_stdout(io.StringIO()):
                    x, obj, k, n, j, status, _ = ref_pdhg.pdlp_algorithm(
                        K, m_ineq, col(c), col(q), col(l), col(u), "cpu", max_kkt=20_000, tol=1e-4, verbose=False,
                        restart_period=40, primal_update=adaptive, adaptive=adaptive, infeasibility_detect=True, infeas_tol=itol)
                tag = f"solve/{name}/{'adaptive' if adaptive else 'fixed'}_{itol:g}"
                put(out, tag, dict(b0=b0.flatten().numpy(), infeas_tol=itol, x=x.flatten().numpy(), obj=obj, k=k, n=n, j=j,
                                   status=status))
                print(f"G10 {tag}: k={k} n={n} j={j} {status} obj={obj:.6f}")
    np.savez_compressed(os.path.join(OUT, "infeasibility.npz"), **out)


class _TorchProxy:
    """stands in for the ``torch`` module inside the reference's spectral_casting: records what ``rand`` (breeding
    weights, :137) and ``argsort`` (survivor order, :238) return; everything else is torch"""

    def __init__(self):
        self.rands, self.orders, self.gaps = [], [], []

    def __getattr__(self, name):
        return getattr(torch, name)

    def rand(self, *a, **kw):
        v = torch.rand(*a, **kw)
        self.rands.append(v.clone())
        return v

    def argsort(self, t, *a, **kw):
        v = torch.argsort(t, *a, **kw)
        self.gaps.append(t.clone())
        self.orders.append(v.clone())
        return v


def g11_fishnet(cases):
    """The fishnet warm start with the global RNG pinned by torch.manual_seed: the points, the radius, eta / omega of
    init_PDHG_vars, every round's duality gaps and survivor order, the breeding weights and the final (x, y)."""
    out = {}
    for name, i, k, seed in (("mixed_400x300", 3, 32, 4321), ("box_200x150", 4, 16, 4322), ("mixed_27x32", 5, 8, 4323)):
        lp = cases[name]
        K = csr_to_dense(lp)
        c, q, l, u = col(lp.c), col(lp.q), col(lp.l), col(lp.u)
        proxy = _TorchProxy()
        sigmas = []
        orig_sn = ref_sc.spectral_norm_estimate_torch

        def rec_sn(*a, **kw):
            v = orig_sn(*a, **kw)
            sigmas.append(float(v))
            return v

        torch.manual_seed(seed)
        b0_a = torch.randn(lp.n, 1)                      # helpers.py:47 inside sample_points (25 iterations)
        pts_raw = torch.randn(lp.n, 2 ** i)              # :49
        b0_b = torch.randn(lp.n, 1)                      # helpers.py:47 inside init_PDHG_vars (50 iterations)
        ref_sc.torch, ref_sc.spectral_norm_estimate_torch = proxy, rec_sn
        try:
            torch.manual_seed(seed)
            pts, radius = ref_sc.sample_points(K, i, "cpu")
            pts0 = pts.clone()
            x, y = ref_sc.fishnet(pts, K, c, q, l, u, lp.m_ineq, 2, k, "cpu")
            # the same through the top-level entry point: identical by construction
            torch.manual_seed(seed)
            x2, y2 = ref_sc.spectral_cast(K, c, q, l, u, lp.m_ineq, k, 2, i, "cpu")
            assert torch.equal(x, x2) and torch.equal(y, y2)
            eta, omega, ipi, ini = ref_sc.init_PDHG_vars(K, c, q, l, u)    # (draws again: not comparable, shape only)
        finally:
            ref_sc.torch, ref_sc.spectral_norm_estimate_torch = torch, orig_sn
        nrounds = len(proxy.orders) // 2            # (the second half belongs to the spectral_cast replay)
        orders = proxy.orders[:nrounds]
        gaps = proxy.gaps[:nrounds]
        rands = proxy.rands[:len(proxy.rands) // 2]
        put(out, name, lp_arrays(lp))
        put(out, name, dict(i=i, k=k, s=2, seed=seed, b0_radius=b0_a.flatten().numpy(), pts_raw=pts_raw.numpy(),
                            b0_eta=b0_b.flatten().numpy(), radius=float(radius), sigma25=sigmas[0], sigma50=sigmas[1],
                            eta=0.9 / sigmas[1], omega=float(torch.linalg.norm(q, 2) / torch.linalg.norm(c, 2)) ** -1,
                            pts0=pts0.numpy(), x=x.numpy(), y=y.numpy(), nrounds=nrounds, nweights=len(rands)))
        for r in range(nrounds):
            put(out, f"{name}/round{r}", dict(gaps=gaps[r].numpy(), order=orders[r].numpy()))
        for w, v in enumerate(rands):
            out[f"{name}/weights{w}"] = v.numpy()
        sep = min(float((torch.sort(g)[0][1:] - torch.sort(g)[0][:-1]).min() / (g.abs().max() + 1e-30)) for g in gaps if g.numel() > 1)
        print(f"G11 {name}: 2^{i} points, k={k}: rounds={nrounds} weights={len(rands)} radius={float(radius):.5f} "
              f"eta={0.9 / sigmas[1]:.6f} smallest relative gap separation {sep:.2e}")
    np.savez_compressed(os.path.join(OUT, "fishnet.npz"), **out)


def g12_forced_trace(cases, spec=(("mixed_400x300", (0, 3, 9, 20)), ("box_200x150", (0, 2, 7, 12))), fname="forced_trace.npz"):
    """pdlp_algorithm(adaptive=True, primal_update=True): the state (x, y, eta, omega, k) going into EVERY step of a few
    40-iteration blocks of the reference's own run and what the step returned.  A test feeds each recorded state to one
    step and compares with the recorded successor, so the whole real trajectory (accepted and rejected steps, the phases
    where eta has grown past the stable range, iterates after restarts to the average) is covered without inheriting
    the amplification of rounding differences that makes whole blocks incomparable: with the step sizes forced to the
    reference's, a 1e-5 difference still grows 3x per step while eta sits above eta_bar (measured on box_200x150)."""
    out = {}
    for name, blocks in spec:
        lp = cases[name]
        K = csr_to_dense(lp)
        calls = []
        orig = ref_pdhg.adaptive_one_step_pdhg

        def rec(x, y, c, q, Km, l, u, m_ineq, eta, omega, theta, k, j):
            xi, yi = x.flatten().clone(), y.flatten().clone()
            r = orig(x, y, c, q, Km, l, u, m_ineq, eta, omega, theta, k, j)
            calls.append((xi, yi, float(eta), float(omega), int(k), r[0].flatten().clone(), r[1].flatten().clone(),
                          float(r[2]), float(r[3])))
            return r

        ref_pdhg.adaptive_one_step_pdhg = rec
        try:
            torch.manual_seed(1234)
            with contextlib.redirect_stdout(io.StringIO()):
                x, obj, k, n, j, status, _ = ref_pdhg.pdlp_algorithm(
                    K, lp.m_ineq, col(lp.c), col(lp.q), col(lp.l), col(lp.u), "cpu", max_kkt=100_000, tol=1e-4, verbose=False,
                    restart_period=40, precondition=False, primal_update=True, adaptive=True)
        finally:
            ref_pdhg.adaptive_one_step_pdhg = orig
        # restarts happen only at multiples of 40 iterations since the last one, so global call indices 0, 40, 80, ...
        # are exactly the block starts
        blocks = [b for b in blocks if 40 * b + 40 <= len(calls)]
        put(out, name, lp_arrays(lp))
        put(out, name, dict(blocks=np.array(blocks), run_k=k, run_n=n, run_j=j, status=status))
        for b in blocks:
            cs = calls[40 * b:40 * b + 40]
            st = lambda i: np.stack([c_[i].numpy() for c_ in cs]).astype(np.float32)
            sc = lambda i: np.array([c_[i] for c_ in cs], dtype=np.float64)
            put(out, f"{name}/block{b}", dict(x_in=st(0), y_in=st(1), eta_in=sc(2), omega=sc(3), k_in=sc(4).astype(np.int64),
                                             x_out=st(5), y_out=st(6), eta_used=sc(7), eta_hat=sc(8)))
        rej = sum(1 for b in blocks for c_ in calls[40 * b:40 * b + 40] if c_[7] != c_[2])
        print(f"G12 {name}: {len(calls)} adaptive calls, blocks {blocks} recorded step by step ({rej} rejected steps), {status} k={k}")
    np.savez_compressed(os.path.join(OUT, fname), **out)


def g13_forced_trace_more(cases):
    """round 3: the same recording for the other three seeded LPs (all-equality, all-inequality, the afiro-sized one), so that
    every step of real adaptive runs is pinned on all five LP cases (own file: forced_trace.npz stays byte for byte)"""
    g12_forced_trace(cases, (("mixed_27x32", (0, 1, 4)), ("mixed_300x400_alleq", (0, 2, 6)), ("mixed_200x260_allineq", (0, 2, 6))),
                     "forced_trace_more.npz")


def g14_adaptive_retry(cases):
    """round 5 (SURVEY quirk Q1's optional flag): the INTENDED adaptive step -- retry from the saved (x, y) with the shrunk step size
    until a trial is accepted -- as the reference's own experiment runs it: ``pdhg_torch`` of /root/reference/enhancements/test_ass.py
    (:313-363), imported here.  Its step-size estimate is pinned (the exact spectral norm times ``scale``: < 1 makes the first step
    sizes too large, so trials are rejected) and its prints are captured: one three-value print per trial, "eta set" on acceptance.
    Recorded: the trial count and the step size after every iteration, and x after 1, 3, 12 and 40 iterations."""
    import contextlib
    import importlib.util
    import io
    spec = importlib.util.spec_from_file_location("ref_test_ass", "/root/reference/enhancements/test_ass.py")
    ta = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ta)
    out = {}
    for name in ("mixed_27x32", "mixed_400x300", "box_200x150"):
        lp = cases[name]
        K = csr_to_dense(lp)
        sig = float(torch.linalg.matrix_norm(K.double(), 2))
        is_neg_inf, is_pos_inf, l_dual, u_dual = masks(col(lp.l), col(lp.u))
        put(out, name, lp_arrays(lp))
        G, A = K[:lp.m_ineq], K[lp.m_ineq:]
        h, b = col(lp.q[:lp.m_ineq]), col(lp.q[lp.m_ineq:])
        for tag, scale in (("tight", 1.0), ("loose", 0.04)):
            ta.spectral_norm_estimate_torch = lambda A_, num_iters=10, v=sig * scale: torch.tensor(v, dtype=torch.float32)
            log = []
            ta.print = lambda *a, **k: log.append(a)
            rec = {}
            for iters in (1, 3, 12, 40):
                log.clear()
                with contextlib.redirect_stdout(io.StringIO()):
                    x, obj, k = ta.pdhg_torch(col(lp.c), G, h, A, b, col(lp.l), col(lp.u), is_neg_inf, is_pos_inf, l_dual, u_dual, "cpu",
                                              max_iter=iters, tol=0.0, verbose=False)
                assert k == iters
                rec[f"x{iters}"] = x.flatten().numpy().copy()
                rec[f"obj{iters}"] = np.float64(obj)
            trials, etas, cur = [], [], 0
            for a in log:                          # (the log of the 40-iteration run)
                if len(a) == 3:
                    cur += 1
                elif len(a) == 2 and a[0] == "eta set":
                    trials.append(cur)
                    etas.append(float(a[1]))
                    cur = 0
            assert len(trials) == 40, (name, tag, len(trials))
            rec.update(trials=np.asarray(trials, np.int32), eta_after=np.asarray(etas, np.float64),
                       eta0=np.float32(0.9) / np.float32(sig * scale),
                       omega=(torch.linalg.norm(lp.c) / torch.linalg.norm(lp.q)).to(torch.float32).numpy())
            put(out, f"{name}/{tag}", rec)
    np.savez_compressed(os.path.join(OUT, "adaptive_retry.npz"), **out)


if __name__ == "__main__":
    which = set(sys.argv[1:]) or {"g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14"}
    cases = lp_cases()
    if "g1" in which: g1_step_fixed(cases)
    if "g2" in which: g2_step_adaptive(cases)
    if "g3" in which: g3_kkt(cases)
    if "g4" in which: g4_solve_trace(cases)
    if "g5" in which: g5_primal_weight()
    if "g6" in which: g6_ruiz(cases)
    if "g7" in which: g7_power_iter(cases)
    if "g8" in which: g8_mps()
    if "g9" in which: g9_afiro()
    if "g10" in which: g10_infeasibility(cases)
    if "g11" in which: g11_fishnet(cases)
    if "g12" in which: g12_forced_trace(cases)
    if "g13" in which: g13_forced_trace_more(cases)
    if "g14" in which: g14_adaptive_retry(cases)
    print("golden fixtures written to", OUT)

