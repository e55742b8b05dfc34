"""``solve_lp``: one call from an MPS file (or arrays) to a solution, the flow of the reference's driver
(``/root/reference/PDLP/main.py:85-137``) and of its older ``pdlp_solver`` wrapper
(``/root/reference/Packages/PDLP_without_presolve_infeasibility.py:748-789``) on the MI355X path."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Union

import torch

from .mps import mps_to_standard_form
from .precondition import ruiz_precondition
from .solver import pdlp_algorithm
from .sparse import CsrPair


@dataclass
class LPResult:
    x: torch.Tensor            # (n, 1) primal solution of the ORIGINAL problem (un-scaled when preconditioned)
    objective: float           # c'x of the original problem
    iterations: int            # k
    restarts: int              # n
    kkt_passes: int            # j
    status: str                # "Solved" | "Unsolved (KKT passes limit exceeded)" | "Unsolved (Time limit exceeded)"
                               # | "DUAL_INFEASIBLE" | "PRIMAL_INFEASIBLE" (only with infeasibility_detect)
    time: float                # seconds, preconditioning included (main.py:107,136)

    def as_tuple(self):
        """the reference's result tuple (pdhg.py:181)"""
        return self.x, self.objective, self.iterations, self.restarts, self.kkt_passes, self.status, self.time


def solve_lp(problem: Union[str, os.PathLike, tuple], device=None, tol: float = 1e-4, precondition: bool = False,
             primal_weight_update: bool = False, adaptive_stepsize: bool = False, max_kkt: int = 100_000,
             time_limit: float = 3600, verbose: bool = False, restart_period: int = 40, dtype=torch.float32,
             seed: Optional[int] = None, compat: bool = True, x_init=None, y_init=None, trace=None,
             fishnet: bool = False, comm=None, infeasibility_detect: bool = False, infeas_tol: float = 1e-4,
             precision: Optional[str] = None, adaptive_retry: bool = False, direct_exchange: bool = False) -> LPResult:
    """Solve ``min c'x, K[:m_ineq]x >= q[:m_ineq], K[m_ineq:]x = q[m_ineq:], l <= x <= u`` on the current HIP device.

    ``problem`` is an MPS path or ``(c, K, q, m_ineq, l, u)`` with ``K`` dense / COO / scipy-sparse / ``CsrPair``.
    Flags carry the reference CLI's names (main.py:11-39); ``adaptive_retry`` (not in the reference's CLI) is ``pdlp_algorithm``'s.  ``dtype=torch.float64`` is the mode for tolerances
    below float32 resolution (the reference is float32 only); ``precision="mixed"`` (the problem is then read in float64) is the
    fast way there: float32 matrix entries under float64 vectors, iterations on the float32 kernels (``pdlp_algorithm``).
    ``infeasibility_detect`` runs the reference's detector
    (enhancements.py:80-161) after every iteration, with its behaviour as it is.  Under ``torchrun`` (one process per GPU, process
    group initialised) pass ``comm=True``: every rank reads the same problem ON THE HOST, puts only its row blocks of K and K'
    on its GPU (the Ruiz sweeps run on the shards), and all return the full solution -- no GPU ever holds the whole LP.
    ``direct_exchange`` (sharded solves, the ranks of ONE node, at most 8): the iterations run without collectives -- every half-step
    stores its block straight into the other ranks' memory over HIP IPC / xGMI (``PdlpEngine.enable_peer_exchange``, DESIGN.md
    section 5); connected and cross-checked against the collective-driven loop first, which stays in charge if anything differs.
    """
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if precision is not None:
        if precision != "mixed":
            raise ValueError(f"unknown precision {precision!r}")
        dtype = torch.float64
    if comm is not None and not fishnet:
        from .engine import Comm
        cm = Comm() if comm is True else comm
        if cm.world > 1:
            return _solve_lp_sharded(problem, cm, device, tol, precondition, primal_weight_update, adaptive_stepsize, max_kkt, time_limit,
                                     verbose, restart_period, dtype, seed, compat, x_init, y_init, trace, infeasibility_detect,
                                     infeas_tol, precision, adaptive_retry, direct_exchange)
    if isinstance(problem, (str, os.PathLike)):
        c, K, q, m_ineq, l, u = mps_to_standard_form(os.fspath(problem), device=device, verbose=verbose, compat=compat, dtype=dtype)
    else:
        c, K, q, m_ineq, l, u = problem
        K = CsrPair.from_any(K, device=device, dtype=dtype)
    time_used, data_precond = 0.0, None
    Ks, cs, qs, ls, us = K, c, q, l, u
    if precondition:                                                    # main.py:106-110
        Ks, cs, qs, ls, us, data_precond, time_used = ruiz_precondition(c, K, q, l, u, device=device)
    if fishnet:                                                         # main.py:114-125 (k=32 points rounds, 2^5 points)
        import time as _time
        from .spectral_casting import spectral_cast
        t0 = _time.time()
        gen = None if seed is None else torch.Generator().manual_seed(int(seed))
        x_init, y_init = spectral_cast(Ks, cs, qs, ls, us, m_ineq, k=32, device=device, generator=gen)
        time_used += _time.time() - t0
    x, obj, k, n, j, status, total = pdlp_algorithm(
        Ks, m_ineq, cs, qs, ls, us, device, max_kkt=max_kkt, tol=tol, verbose=verbose, restart_period=restart_period,
        precondition=precondition, primal_update=primal_weight_update, adaptive=adaptive_stepsize,
        data_precond=data_precond, time_limit=time_limit, time_used=time_used, x_init=x_init, y_init=y_init, seed=seed,
        trace=trace, comm=comm, infeasibility_detect=infeasibility_detect, infeas_tol=infeas_tol, precision=precision,
        adaptive_retry=adaptive_retry)
    if precondition:        # the reference returns the scaled iterate (quirk Q4); solve_lp un-scales: x = D_col x_s (pdhg.py:161)
        x = data_precond[0].view(-1, 1).to(x.dtype) * x
    return LPResult(x, obj, k, n, j, status, total)


def _solve_lp_sharded(problem, comm, device, tol, precondition, primal_weight_update, adaptive_stepsize, max_kkt, time_limit, verbose,
                      restart_period, dtype, seed, compat, x_init, y_init, trace, infeasibility_detect, infeas_tol, precision,
                      adaptive_retry=False, direct_exchange=False) -> LPResult:
    """``solve_lp`` over the ranks of ``comm``: the problem is read (or taken) on the host by every rank, cut into blocks balanced
    by non-zeros, and only this rank's blocks go to its GPU; Ruiz (enhancements.py:4-71) runs on the shards, the solve is
    ``run_pdlp`` on the sharded engine (pdhg.py:7-181), and every rank returns the full un-scaled solution."""
    from .distributed import engine_from_shard, gather_solution, shard_arrays
    from .solver import run_pdlp
    from .sparse import as_vec
    if isinstance(problem, (str, os.PathLike)):
        c, K, q, m_ineq, l, u = mps_to_standard_form(os.fspath(problem), device="cpu", verbose=verbose and comm.rank == 0, compat=compat,
                                                     dtype=dtype)
    else:
        c, K, q, m_ineq, l, u = problem
        K = CsrPair.from_any(K, dtype=dtype)             # (where the caller has it: host or device)
    n, m = K.n, K.m
    sh = shard_arrays(K, c, q, l, u, m_ineq, comm.rank, comm.world, vec_dtype=dtype, balance="nnz")
    part = sh["part"]
    one = lambda t: t.to(device) if isinstance(t, torch.Tensor) else t
    mv = lambda v: tuple(one(t) for t in v) if isinstance(v, tuple) else one(v)
    sh = {k: (v if k == "part" else mv(v)) for k, v in sh.items() if v is not None}
    del K
    eng = engine_from_shard(sh, comm, precision=precision, precondition=precondition)
    time_used = float(getattr(eng, "ruiz_seconds", 0.0))
    if direct_exchange:            # (every rank asks; the ranks agree on every step, and on any failure all stay on the loop)
        on = eng.enable_peer_exchange()
        if verbose and comm.rank == 0:
            print("direct exchange:", "on" if on else "declined", "--", "; ".join(eng.peer_log))
    vdt = eng.dtype
    if x_init is not None and y_init is not None:          # full vectors in (of the scaled problem when preconditioned, like the
        x_init = part.pad_cols(as_vec(x_init, n, device, vdt))[eng.cols[0]:eng.cols[1]]    # one-GPU path and main.py:114-130);
        y_init = part.pad_rows(as_vec(y_init, m, device, vdt))[eng.rows[0]:eng.rows[1]]    # this rank's blocks of the padded layout on
    x, obj, k, nr, j, status, total = run_pdlp(eng, max_kkt, tol, verbose and comm.rank == 0, restart_period, precondition,
                                               primal_weight_update, adaptive_stepsize, time_limit, time_used, x_init, y_init,
                                               seed=0 if seed is None else seed, trace=trace,
                                               infeasibility_detect=infeasibility_detect, infeas_tol=infeas_tol, adaptive_retry=adaptive_retry)
    if precondition:
        x = x * eng.d_col
    return LPResult(gather_solution(eng, x, n).view(-1, 1), obj, k, nr, j, status, total)
