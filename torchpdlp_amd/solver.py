"""Restarted PDHG outer loop on top of the HIP engine -- the drop-in for the reference's
``pdlp_algorithm`` (``/root/reference/PDLP/primal_dual_hybrid_gradient.py:7-181``).

Same arguments, same result tuple ``(x, prim_obj, k, n, j, status, total_time)``, same status
strings and the same KKT-pass bookkeeping ``j``.  The host only decides restarts: it reads three
KKT errors every ``restart_period`` iterations; everything else stays on the device.
"""
from __future__ import annotations

import time
from typing import Optional

import numpy as np
import torch

from . import _native as N
from .engine import Comm, PdlpEngine
from .sparse import CsrPair, as_vec

STATUS_KKT_LIMIT = "Unsolved (KKT passes limit exceeded)"     # pdhg.py:51
STATUS_TIME_LIMIT = "Unsolved (Time limit exceeded)"          # pdhg.py:71
STATUS_SOLVED = "Solved"                                      # pdhg.py:174


def _np_t(dtype):
    return np.float32 if dtype == torch.float32 else np.float64


def check_termination(primal_residual, dual_residual, duality_gap, prim_obj, adjusted_dual, q_norm, c_norm, tol):
    """check_termination (helpers.py:110-128).  The gap is signed (reference quirk Q2)."""
    cond1 = primal_residual <= tol * (1 + q_norm)
    cond2 = dual_residual <= tol * (1 + c_norm)
    cond3 = duality_gap <= tol * (1 + abs(prim_obj) + abs(adjusted_dual))
    return bool(cond1 and cond2 and cond3)


def kkt_from_residuals(res: dict, omega, t=np.float32):
    """KKT_error (helpers.py:98-108) from already computed residuals, in the working precision."""
    w2 = t(omega) * t(omega)
    pr, dr, gap = t(res["pr"]), t(res["dr"]), t(res["gap"])
    return t(np.sqrt(w2 * (pr * pr) + (dr * dr) / w2 + gap * gap))


def primal_weight_from_distances(dx2: float, dy2: float, omega, smooth_theta=0.5, t=np.float32):
    """primal_weight_update (enhancements.py:73-78) given the two squared restart distances."""
    dxn, dyn = t(np.sqrt(dx2)), t(np.sqrt(dy2))
    if dxn > 0 and dyn > 0:
        return t(np.exp(t(smooth_theta) * t(np.log(dyn / dxn)) + (t(1) - t(smooth_theta)) * t(np.log(t(omega)))))
    return t(omega)


def _global_norm(v_local: torch.Tensor, comm: Optional[Comm]) -> float:
    s = (v_local.double() ** 2).sum().reshape(1)
    if comm is not None:
        comm.all_reduce_sum(s)
    return float(torch.sqrt(s))


def run_pdlp(eng: PdlpEngine, max_kkt=100_000, tol=1e-4, verbose=True, restart_period=40, precondition=False,
             primal_update=False, adaptive=False, time_limit=3600, time_used=0, x_init=None, y_init=None,
             b0=None, sigma=None, power_iters=100, seed=None, trace=None):
    """The outer loop over an existing engine.  Returns (x_local, prim_obj, k, n, j, status, total_time)."""
    t0 = time.time()
    t = _np_t(eng.dtype)
    comm = eng.comm
    q_norm = t(_global_norm(eng.q, comm))                                   # pdhg.py:19-20
    c_norm = t(_global_norm(eng.c, comm))
    if sigma is None:                                                       # pdhg.py:22
        if b0 is None:      # the reference draws an unseeded torch.randn (quirk Q6); same on every rank here
            g = torch.Generator().manual_seed(int(seed) if seed is not None else int(time.time_ns() % (2 ** 31)))
            b0 = torch.randn(eng.n, generator=g, dtype=torch.float32)
            if comm is not None:
                b0 = b0.to(eng.device)
                comm.dist.broadcast(b0, 0, group=comm.group)
        sigma = eng.power_iteration(b0, power_iters)
    eta = t(0.9) / t(sigma)
    omega = (c_norm / q_norm) if (q_norm > 1e-6 and c_norm > 1e-6) else t(1.0)   # pdhg.py:23
    theta = 1.0
    beta = (0.2, 0.8, 0.36)                                                 # pdhg.py:28
    zeros = lambda ln: torch.zeros(ln, dtype=eng.dtype, device=eng.device)
    if x_init is not None and y_init is not None:                           # pdhg.py:31-36
        eng.set_iterate(x_init, y_init)
    else:
        eng.set_iterate(zeros(eng.nl), zeros(eng.ml))
    eng.set_step(eta, omega, theta, 0)
    n = k = j = 0
    KKT_first = t(0)                                                        # pdhg.py:48
    status = STATUS_KKT_LIMIT
    res = None
    while j < max_kkt:                                                      # pdhg.py:54
        tt = 0
        chosen = None
        while j < max_kkt:                                                  # pdhg.py:67
            if time.time() - t0 + time_used >= time_limit:                  # pdhg.py:68-74 (checked per block)
                status = STATUS_TIME_LIMIT
                if verbose:
                    print("Time limit exceeded")
                break
            iters = min(restart_period - tt % restart_period, max_kkt - j)
            eng.iterate(iters, adaptive)                                    # pdhg.py:76-112
            k += iters
            j += iters
            tt += iters
            if tt % restart_period == 0:                                    # pdhg.py:115
                if adaptive:
                    eng.flush_average()
                eng.compute_average()                                       # pdhg.py:118-119
                r_cur = eng.kkt(N.CUR, omega)                               # pdhg.py:122-125
                r_avg = eng.kkt(N.AVG, omega)
                r_prev = eng.kkt(N.PREV, omega)
                k_cur, k_avg, k_prev = t(r_cur["kkt"]), t(r_avg["kkt"]), t(r_prev["kkt"])
                k_min = min(k_cur, k_avg)
                j += 3                                                      # pdhg.py:128
                if trace is not None:
                    trace["kkt"] += [float(k_cur), float(k_avg), float(k_prev)]
                use_avg = bool(k_cur >= k_avg)
                crit = -1
                if k_min <= t(beta[0]) * KKT_first:                         # sufficient, pdhg.py:131
                    crit = 0
                elif k_min <= t(beta[1]) * KKT_first and k_min > k_prev:    # necessary, pdhg.py:135
                    crit = 1
                elif tt >= beta[2] * k:                                     # artificial, pdhg.py:139
                    crit = 2
                if crit >= 0:
                    if verbose:
                        print(f"{('Sufficient', 'Necessary', 'Artificial')[crit]} restart at iteration {tt} using the",
                              "Average iterate." if use_avg else "Current iterate.")
                    if trace is not None:
                        trace["restarts"].append((crit, tt, int(use_avg)))
                    eng.restart(N.AVG if use_avg else N.CUR)
                    chosen = r_avg if use_avg else r_cur
                    break
        if status == STATUS_TIME_LIMIT:
            break
        n += 1
        if chosen is None:       # the KKT-pass cap ended the inner loop: continue from the current iterate
            eng.restart(N.CUR)
        if primal_update:                                                   # pdhg.py:150-151
            dx2, dy2 = eng.restart_distance()
            omega = primal_weight_from_distances(dx2, dy2, omega, 0.5, t)
            eng.set_omega(omega)
            if trace is not None:
                trace["omega"].append(float(omega))
        eng.mark_restart_point()                                            # pdhg.py:63-64 of the next round
        if chosen is None:
            chosen = eng.kkt(N.CUR, omega)
        # KKT_first at the restart point with the (new) omega: the residuals do not depend on omega, so
        # the pass the reference repeats here (pdhg.py:153) is a re-weighting of numbers already known
        KKT_first = kkt_from_residuals(chosen, omega, t)
        j += 1                                                              # pdhg.py:154
        if trace is not None:
            trace["kkt"].append(float(KKT_first))
        res = eng.kkt(N.CUR, omega, unscaled=True) if precondition else chosen   # pdhg.py:157-163
        j += 1                                                              # pdhg.py:165
        if verbose:
            print(f"[{k}] Primal Obj: {res['p']:.4f}, Adjusted Dual Obj: {res['d_adj']:.4f}, "
                  f"Gap: {res['gap'] / (1 + abs(res['p']) + abs(res['d_adj'])):.2e}, "
                  f"Prim Res: {res['pr'] / (1 + q_norm):.2e}, Dual Res: {res['dr'] / (1 + c_norm):.2e}\n")
        if check_termination(t(res["pr"]), t(res["dr"]), t(res["gap"]), t(res["p"]), t(res["d_adj"]), q_norm, c_norm, t(tol)):
            status = STATUS_SOLVED
            if verbose:
                print(f"Converged at iteration {k} restart loop {n}")
            break
    x_local, _ = eng.get_iterate(N.CUR)
    eng.synchronize()                                  # the reference reads its clock without a sync (Q10)
    prim_obj = float(res["p"]) if res is not None else float("nan")
    return x_local, prim_obj, k, n, j, status, time.time() - t0 + time_used


def pdlp_algorithm(K, m_ineq, c, q, l, u, device=None, max_kkt=100_000, tol=1e-4, verbose=True, restart_period=40,
                   precondition=False, primal_update=False, adaptive=False, data_precond=None, infeasibility_detect=False,
                   infeas_tol=1e-4, time_limit=3600, time_used=0, x_init=None, y_init=None, *, b0=None, sigma=None,
                   seed=None, trace=None):
    """Drop-in for the reference's ``pdlp_algorithm`` (primal_dual_hybrid_gradient.py:7) on one MI355X.

    ``K`` may be a dense / COO torch tensor (as the reference takes), a scipy sparse matrix or a
    ``CsrPair``.  With ``precondition=True`` pass the scaled problem and ``data_precond`` as returned by
    ``ruiz_precondition`` (its first two entries ``D_col, D_row`` are what is used).  ``b0`` / ``sigma`` /
    ``seed`` pin the power-iteration start the reference leaves to an unseeded RNG.
    Returns ``(x, prim_obj, k, n, j, status, total_time)``; ``x`` is an (n,1) tensor and, like the
    reference's (quirk Q4), the SCALED iterate when preconditioned.
    """
    if infeasibility_detect:
        raise NotImplementedError("infeasibility detection is outside the accelerated hot path (SURVEY.md 8f row f4)")
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    Kp = CsrPair.from_any(K, device=device)
    dtype = Kp.dtype
    d_col = d_row = None
    if precondition:
        if data_precond is None:
            raise ValueError("precondition=True needs data_precond from ruiz_precondition")
        d_col, d_row = data_precond[0], data_precond[1]
    eng = PdlpEngine.from_full(Kp, c, q, l, u, m_ineq, d_col=d_col, d_row=d_row)
    x, obj, k, n, j, status, total = run_pdlp(eng, max_kkt, tol, verbose, restart_period, precondition, primal_update, adaptive,
                                              time_limit, time_used, x_init, y_init, b0=b0, sigma=sigma, seed=seed, trace=trace)
    return x.view(-1, 1), obj, k, n, j, status, total
