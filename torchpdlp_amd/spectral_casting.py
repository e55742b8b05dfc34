""""Fishnet" warm start -- counterpart of ``/root/reference/PDLP/spectral_casting.py`` (opt-in ``--fishnet``,
``main.py:114-125``): cast 2^i points on a sphere of radius ||K||_2, run k fixed-step PDHG iterations on every point,
keep the best 1/s by (signed) duality gap, on odd rounds breed the population back with random convex combinations
plus the midpoint, until one point is left.

The reference advances all points at once as an n x j matrix (``PDHG_step`` :254-293).  Here populations of 8, 16 or
32 points do the same on the device: the multi-vector kernels (``pdlp_mv_steps`` / ``pdlp_mv_gap``, csrc/pdlp_kernel_mv.inc)
read each matrix once per step for all points, an item contributing to all j row sums with one coalesced line of the
population matrix.  Smaller populations (the last rounds) and sharded problems run point by
point through the fused single-vector kernels the solver uses (``PdlpEngine.iterate`` / ``kkt``).
Randomness: the reference draws from the global torch RNG on the device; pass ``generator`` (a CPU generator) to
pin the points and the breeding weights.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _native as N
from .engine import PdlpEngine
from .sparse import CsrPair, as_vec


def sample_points(eng: PdlpEngine, i: int, generator: Optional[torch.Generator] = None, r: Optional[float] = None,
                  b0: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, float]:
    """2^i points on the sphere of radius r = ||K||_2 (25 power iterations) centred at (r/sqrt(n)) 1 -- :32-63"""
    n = eng.n
    # a CPU generator replays the reference's (CPU) random stream draw for draw; without one the points are drawn on the device
    where = "cpu" if generator is not None else eng.device
    if r is None:
        if b0 is None:
            b0 = torch.randn(n, generator=generator, dtype=torch.float32, device=where)
        r = eng.power_iteration(b0, 25)
    pts = torch.randn(n, 2 ** i, generator=generator, dtype=torch.float32, device=where)
    pts = pts * r / torch.norm(pts, dim=0, keepdim=True)
    pts += (r / n ** 0.5) * torch.ones(n, 1, device=where)
    return pts, r


def _gap(res: dict) -> float:
    return res["gap"]            # adjusted_dual - prim_obj, signed (:233)


def fishnet(eng: PdlpEngine, pts: torch.Tensor, s: int = 2, k: int = 32, eta: Optional[float] = None,
            generator: Optional[torch.Generator] = None, b0: Optional[torch.Tensor] = None, multi_vector: bool = True,
            trace: Optional[list] = None):
    """``fishnet`` (:65-159) on an engine; ``pts`` is n x j (moved to the device).  Returns (x, y) flattened on the device.
    ``trace`` (a list) receives every round's (duality gaps, survivor order) of ``get_best_pts`` (:191-252)."""
    dev, dt = eng.device, eng.dtype
    pts = pts.to(device=dev, dtype=dt)                                # the population lives on the device (n x j, m x j)
    j = pts.shape[1]
    # pts_y = K @ pts (:100): one pass over K for all points where the population kernels apply, else point by point
    if multi_vector and eng.comm is None and not eng.mixed and j in eng.MV_WIDTHS:
        pts = pts.contiguous()
        ys = eng.mv_product(pts)
    else:
        ys = torch.stack([eng.spmv(pts[:, p].contiguous(), False) for p in range(j)], dim=1)
    if eta is None:                                                   # init_PDHG_vars :161-189
        if b0 is None:
            b0 = torch.randn(eng.n, generator=generator, dtype=torch.float32, device="cpu" if generator is not None else dev)
        eta = 0.9 / eng.power_iteration(b0, 50)
    q_norm, c_norm = float(torch.linalg.norm(eng.q)), float(torch.linalg.norm(eng.c))
    omega = c_norm / q_norm if (q_norm > 1e-6 and c_norm > 1e-6) else 1.0
    eng.set_step(eta, omega, 1.0, 0)
    i = 0
    while j > 1:                                                      # :105
        if multi_vector and eng.comm is None and j in eng.MV_WIDTHS:  # all points in one pass over the matrices
            pts, ys = pts.contiguous(), ys.contiguous()
            eng.mv_steps(pts, ys, k, eta, omega, 1.0)                 # k PDHG steps on every point (:107-109)
            gaps = eng.mv_gap(pts, ys)                                # get_best_pts :191-252 (duality gap only)
        else:
            gaps = []
            for p in range(j):
                eng.set_iterate(pts[:, p].contiguous(), ys[:, p].contiguous())
                eng.iterate(k, False)
                pts[:, p], ys[:, p] = eng.get_iterate(N.CUR)
                gaps.append(_gap(eng.kkt(N.CUR, omega)))
        old_j = j
        order = torch.argsort(torch.tensor(gaps, dtype=torch.float32)).to(dev)           # ascending (:238)
        if trace is not None:
            trace.append((list(gaps), order.cpu().tolist()))
        keep = max(1, old_j // s)
        pts, ys = pts[:, order][:, :keep].clone(), ys[:, order][:, :keep].clone()
        new_j = keep
        if i % 2 == 1 and new_j > 1:                                  # breed on odd rounds (:117-152)
            mid, mid_y = pts.mean(dim=1, keepdim=True), ys.mean(dim=1, keepdim=True)
            ws = []
            for _ in range(old_j - new_j - 1):
                w = torch.rand(new_j, generator=generator, device="cpu" if generator is not None else dev)
                ws.append((w / w.sum()).to(device=dev, dtype=dt))
            if ws:                                                    # the convex combinations: pts @ W, ys @ W (k_mv_combine)
                new_x, new_y = [], []
                for w0 in range(0, len(ws), 32):
                    W = torch.stack(ws[w0:w0 + 32], dim=1)
                    if new_j <= 32:
                        new_x.append(eng.mv_combine(pts, W))
                        new_y.append(eng.mv_combine(ys, W))
                    else:                                             # (populations beyond the kernels' 32 columns)
                        new_x.append(pts @ W)
                        new_y.append(ys @ W)
                pts = torch.cat([pts] + new_x + [mid], dim=1)
                ys = torch.cat([ys] + new_y + [mid_y], dim=1)
        j = pts.shape[1]
        i += 1
    return pts.flatten(), ys.flatten()


def spectral_cast(K, c, q, l, u, m_ineq, k, s=2, i=5, device=None, generator: Optional[torch.Generator] = None):
    """Drop-in for ``spectral_cast(K,c,q,l,u,m_ineq,k,s=2,i=5,device)`` (:5-29) -> ``(x0, y0)`` for ``pdlp_algorithm``."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    Kp = CsrPair.from_any(K, device=device)
    eng = PdlpEngine.from_full(Kp, as_vec(c), as_vec(q), as_vec(l), as_vec(u), int(m_ineq))
    pts, _ = sample_points(eng, i, generator)
    return fishnet(eng, pts, s, k, generator=generator)
