"""The reference's hot-path operators under their own names and signatures, computed by the HIP library.

These mirror what ``pdlp_algorithm`` imports in the reference
(``/root/reference/PDLP/primal_dual_hybrid_gradient.py:3-5``) so that code (and tests) written
against the reference read the same here.  Vectors are ``(len, 1)`` column tensors as in the
reference; ``K`` may be dense, COO, scipy-sparse or a ``CsrPair``.  Each call runs on the device
through a cached ``PdlpEngine`` for ``K`` -- for whole solves use ``pdlp_algorithm`` / ``solve_lp``,
which keep all state on the GPU between iterations.
"""
from __future__ import annotations

import ctypes as C
import time
import weakref

import numpy as np
import torch

from . import _native as N
from .engine import PdlpEngine
from .solver import check_termination, primal_weight_from_distances  # noqa: F401  (re-exported)
from .sparse import CsrPair, as_vec

_pairs = {}      # id(K) -> (weakref | None, CsrPair)
_engines = {}    # key -> PdlpEngine


def _device_of(*tensors):
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    return torch.device("cuda", torch.cuda.current_device())


def _pair(K, device) -> CsrPair:
    if isinstance(K, CsrPair):
        return K.to(device)
    ent = _pairs.get(id(K))
    if ent is not None and (ent[0] is None or ent[0]() is K):
        return ent[1]
    p = CsrPair.from_any(K, device=device)
    try:
        _pairs[id(K)] = (weakref.ref(K), p)
    except TypeError:
        _pairs[id(K)] = (None, p)
    if len(_pairs) > 8:
        _pairs.pop(next(iter(_pairs)))
    return p


def _engine(K, c, q, l, u, m_ineq) -> PdlpEngine:
    dev = _device_of(K if isinstance(K, torch.Tensor) else None, c, q)
    Kp = _pair(K, dev)
    vec = lambda v, ln: as_vec(v, ln, dev, Kp.dtype)
    c, q, l, u = vec(c, Kp.n), vec(q, Kp.m), vec(l, Kp.n), vec(u, Kp.n)
    key = (id(Kp), Kp.val.data_ptr(), Kp.nnz, int(m_ineq), c.data_ptr(), q.data_ptr(), l.data_ptr(), u.data_ptr(), c._version, q._version, l._version, u._version)
    eng = _engines.get(key)
    if eng is None:
        eng = PdlpEngine.from_full(Kp, c, q, l, u, int(m_ineq))
        _engines[key] = eng
        if len(_engines) > 8:
            _engines.pop(next(iter(_engines)))
    return eng


def _col(v):
    return v.view(-1, 1)


def _scalar_like(ref, value, dtype):
    if isinstance(ref, torch.Tensor):
        return torch.tensor(value, dtype=ref.dtype, device=ref.device)
    return torch.tensor(value, dtype=dtype)


def fixed_one_step_pdhg(x, y, c, q, K, l, u, m_ineq, eta, omega, theta):
    """fixed_one_step_pdhg (step.py:3-40): returns ``(x, y, eta, eta)``; ``y`` is updated in place as in the reference."""
    eng = _engine(K, c, q, l, u, m_ineq)
    eng.set_iterate(x, y)
    eng.set_step(float(eta), float(omega), float(theta), 0)
    eng.iterate(1, False)
    xn, yn = eng.get_iterate(N.CUR)
    if isinstance(y, torch.Tensor) and y.is_cuda and y.dtype == yn.dtype:
        y.view(-1).copy_(yn)                 # step.py:34,38 mutate the caller's y
        yn = y.view(-1)
    return _col(xn), _col(yn), eta, eta


def adaptive_one_step_pdhg(x, y, c, q, K, l, u, m_ineq, eta, omega, theta, k, j):
    """adaptive_one_step_pdhg (step.py:43-115): returns ``(x, y, eta_used, eta_hat, j + 1)`` (one trial, quirk Q1)."""
    eng = _engine(K, c, q, l, u, m_ineq)
    eng.set_iterate(x, y)
    eng.set_step(float(eta), float(omega), float(theta), int(k) - 1)
    eng.iterate(1, True)
    xn, yn = eng.get_iterate(N.CUR)
    s = eng.scalars()
    return (_col(xn), _col(yn), _scalar_like(eta, s["w_pending"], xn.dtype), _scalar_like(eta, s["eta"], xn.dtype), j + 1)


def compute_residuals_and_duality_gap(x, y, c, q, K, m_ineq, is_neg_inf=None, is_pos_inf=None, l_dual=None, u_dual=None,
                                      *, l=None, u=None):
    """compute_residuals_and_duality_gap (helpers.py:53-96) -> five (1,) tensors.

    The reference passes the bound masks and the inf->0 bounds; the kernel derives both from ``l, u``.
    Give either ``l, u`` or the reference's four arguments (from which ``l, u`` are rebuilt).
    """
    if l is None or u is None:
        inf = float("inf")
        l = torch.where(is_neg_inf.reshape(-1), torch.full_like(l_dual.reshape(-1), -inf), l_dual.reshape(-1))
        u = torch.where(is_pos_inf.reshape(-1), torch.full_like(u_dual.reshape(-1), inf), u_dual.reshape(-1))
    eng = _engine(K, c, q, l, u, m_ineq)
    eng.set_iterate(x, y)
    r = eng.kkt(N.CUR, 1.0)
    mk = lambda v: torch.tensor([v], dtype=eng.dtype, device=eng.device)
    return mk(r["pr"]), mk(r["dr"]), mk(r["gap"]), mk(r["p"]), mk(r["d_adj"])


def KKT_error(x, y, c, q, K, m_ineq, omega, is_neg_inf=None, is_pos_inf=None, l_dual=None, u_dual=None, device=None,
              *, l=None, u=None):
    """KKT_error (helpers.py:98-108) -> (1,) tensor."""
    if l is None or u is None:
        inf = float("inf")
        l = torch.where(is_neg_inf.reshape(-1), torch.full_like(l_dual.reshape(-1), -inf), l_dual.reshape(-1))
        u = torch.where(is_pos_inf.reshape(-1), torch.full_like(u_dual.reshape(-1), inf), u_dual.reshape(-1))
    eng = _engine(K, c, q, l, u, m_ineq)
    eng.set_iterate(x, y)
    r = eng.kkt(N.CUR, float(omega))
    return torch.tensor([r["kkt"]], dtype=eng.dtype, device=eng.device)


def primal_weight_update(x_prev, x, y_prev, y, omega, smooth_theta):
    """primal_weight_update (enhancements.py:73-78); the two norms come from the device."""
    dev = _device_of(x, y)
    lib = N.load()
    dt = as_vec(x).dtype if as_vec(x).dtype in (torch.float32, torch.float64) else torch.float32
    work = torch.empty(1040, dtype=torch.float64, device=dev)

    def sqdist(a, b):                      # k_sqdiff + k_finalize: the kernels of pdlp_restart_distance_local
        a, b = as_vec(a, device=dev, dtype=dt), as_vec(b, device=dev, dtype=dt)
        if a.numel() != b.numel():
            raise ValueError("primal_weight_update: vectors of different lengths")
        out = C.c_double(0.0)
        N.check(lib.pdlp_vec_sqdist(N.PDLP_F32 if dt == torch.float32 else N.PDLP_F64, a.numel(), a.data_ptr(), b.data_ptr(),
                                    work.data_ptr(), C.byref(out), torch.cuda.current_stream(dev).cuda_stream), "pdlp_vec_sqdist")
        return out.value

    dx2, dy2 = sqdist(x_prev, x), sqdist(y_prev, y)
    t = np.float32 if dt == torch.float32 else np.float64
    return _scalar_like(omega, float(primal_weight_from_distances(dx2, dy2, float(omega), smooth_theta, t)), dt)


def project_lambda_box(grad, is_neg_inf, is_pos_inf):
    """project_lambda_box (helpers.py:3-39): the bound classes come in as the reference's two masks."""
    dev = _device_of(grad)
    g = as_vec(grad, device=dev)
    inf = float("inf")
    zero = torch.zeros_like(g)
    l = torch.where(as_vec(is_neg_inf, device=dev).bool(), torch.full_like(g, -inf), zero)     # (storage only: which class)
    u = torch.where(as_vec(is_pos_inf, device=dev).bool(), torch.full_like(g, inf), zero)
    out = torch.empty_like(g)
    lib = N.load()
    N.check(lib.pdlp_vec_project_lambda(N.PDLP_F32 if g.dtype == torch.float32 else N.PDLP_F64, g.numel(), g.data_ptr(),
                                        l.data_ptr(), u.data_ptr(), out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
            "pdlp_vec_project_lambda")
    return out.view_as(grad) if isinstance(grad, torch.Tensor) else _col(out)


def detect_infeasibility(x, y, x_prev, y_prev, lam, lam_prev, c, q, K, l, u, m_ineq, device=None, tol=1e-2):
    """detect_infeasibility (enhancements.py:80-161) -> "DUAL_INFEASIBLE" | "PRIMAL_INFEASIBLE" | None.
    ``lam`` is what the reference's caller computed as project_lambda_box(c - K.T @ y) (pdhg.py:90); the kernels form it
    from ``y`` themselves, so the argument is accepted and not read."""
    eng = _engine(K, c, q, l, u, m_ineq)
    eng.set_iterate(x, y)
    eng.buffer(N.BUF_X_PREV).copy_(as_vec(x_prev, eng.n, eng.device, eng.dtype))
    eng.buffer(N.BUF_Y_PREV).copy_(as_vec(y_prev, eng.m, eng.device, eng.dtype))
    eng.buffer(N.BUF_LAM_PREV).copy_(as_vec(lam_prev, eng.n, eng.device, eng.dtype))
    return eng.detect_infeasibility(float(tol))


def spectral_norm_estimate_torch(K, num_iters=10, b0=None, seed=None):
    """spectral_norm_estimate_torch (helpers.py:41-51); ``b0``/``seed`` pin the reference's unseeded start vector."""
    dev = _device_of(K if isinstance(K, torch.Tensor) else None)
    Kp = _pair(K, dev)
    z = lambda ln: torch.zeros(ln, dtype=Kp.dtype, device=dev)
    eng = _engine(Kp, z(Kp.n), z(Kp.m), z(Kp.n), z(Kp.n), 0)
    if b0 is None:
        g = torch.Generator().manual_seed(int(seed) if seed is not None else int(time.time_ns() % (2 ** 31)))
        b0 = torch.randn(Kp.n, generator=g, dtype=torch.float32)
    return torch.tensor(eng.power_iteration(b0, num_iters), dtype=Kp.dtype, device=dev)
