"""Sharding one LP over the GPUs of a node: one process per GPU, row blocks of K and of K'.

Rank r owns constraints [r*mb, (r+1)*mb) (rows of K, with y and q) and variables [r*nb, (r+1)*nb)
(rows of K', with x, c, l, u).  Each half-step is local except that it gathers from a vector the
other half-step produced, so the exchange is one all-gather of xbar before K xbar and one all-gather
of y before K'y (RCCL over xGMI when the backend is ``nccl``), plus an 8-double all-reduce for the
step-size rule and for each KKT evaluation.  Against the alternative in the north star (replicated
x and an all-reduce of K'y partial sums) this moves the same bytes per iteration
((P-1)/P (n+m) values per GPU) but needs no second summation pass and no redundant primal update.

The LP is padded so both dimensions divide by the world size: padding variables are fixed at 0
(l = u = c = 0, empty column) and padding constraints are empty equality rows with q = 0; neither
changes any iterate, residual or norm.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .engine import Comm, PdlpEngine
from .sparse import CsrPair, as_vec


def padded(v: int, world: int) -> int:
    return (v + world - 1) // world * world


def block(v_padded: int, rank: int, world: int) -> Tuple[int, int]:
    b = v_padded // world
    return rank * b, (rank + 1) * b


def _pad_rowptr(rp: torch.Tensor, rows_have: int, r0: int, r1: int) -> torch.Tensor:
    """row pointers of rows [r0,r1) where rows >= rows_have are empty"""
    hi = min(r1, rows_have)
    if hi <= r0:
        return torch.zeros(r1 - r0 + 1, dtype=torch.int32, device=rp.device)
    base = rp[r0:hi + 1] - rp[r0]
    if r1 > hi:
        base = torch.cat([base, base[-1:].expand(r1 - hi)])
    return base.to(torch.int32).contiguous()


def _pad_vec(v: torch.Tensor, have: int, a: int, b: int, fill: float = 0.0) -> torch.Tensor:
    out = torch.full((b - a,), fill, dtype=v.dtype, device=v.device)
    hi = min(b, have)
    if hi > a:
        out[:hi - a] = v[a:hi]
    return out


def shard_arrays(K: CsrPair, c, q, l, u, m_ineq: int, rank: int, world: int, d_col=None, d_row=None, vec_dtype=None) -> dict:
    """This rank's blocks of a problem held in full: keyword arguments for ``PdlpEngine`` (minus ``comm``)."""
    W, r = world, rank
    m, n = K.m, K.n
    mp, np_ = padded(m, W), padded(n, W)
    r0, r1 = block(mp, r, W)
    c0, c1 = block(np_, r, W)
    dev, dt = K.device, (K.dtype if vec_dtype is None else vec_dtype)
    vec = lambda v, ln: as_vec(v, ln, dev, dt)
    a, b = int(K.rowptr[min(r0, m)]), int(K.rowptr[min(r1, m)])
    K_rows = (_pad_rowptr(K.rowptr, m, r0, r1), K.colidx[a:b].contiguous(), K.val[a:b].contiguous())
    a, b = int(K.t_rowptr[min(c0, n)]), int(K.t_rowptr[min(c1, n)])
    KT_rows = (_pad_rowptr(K.t_rowptr, n, c0, c1), K.t_colidx[a:b].contiguous(), K.t_val[a:b].contiguous())
    opt = lambda v, have, lo, hi, fill: None if v is None else _pad_vec(vec(v, have), have, lo, hi, fill)
    return dict(m=mp, n=np_, m_ineq=m_ineq, K_rows=K_rows, KT_rows=KT_rows,
                c=_pad_vec(vec(c, n), n, c0, c1), q=_pad_vec(vec(q, m), m, r0, r1),
                l=_pad_vec(vec(l, n), n, c0, c1), u=_pad_vec(vec(u, n), n, c0, c1),
                rows=(r0, r1), cols=(c0, c1), d_col=opt(d_col, n, c0, c1, 1.0), d_row=opt(d_row, m, r0, r1, 1.0))


def shard_engine(K: CsrPair, c, q, l, u, m_ineq: int, comm: Optional[Comm], d_col=None, d_row=None, vec_dtype=None) -> PdlpEngine:
    """Engine for this rank's block of a problem every rank holds in full (small/medium problems, tests,
    and the benchmark, where every rank generates the same seeded instance and keeps only its block)."""
    if comm is None or comm.world == 1:
        return PdlpEngine.from_full(K, c, q, l, u, m_ineq, d_col=d_col, d_row=d_row, vec_dtype=vec_dtype)
    return PdlpEngine(comm=comm, vec_dtype=vec_dtype, **shard_arrays(K, c, q, l, u, m_ineq, comm.rank, comm.world, d_col, d_row, vec_dtype))


def gather_solution(eng: PdlpEngine, x_local: torch.Tensor, n_true: int) -> torch.Tensor:
    """the full primal vector on every rank (drops the padding)"""
    if eng.comm is None:
        return x_local
    full = torch.empty(eng.n, dtype=x_local.dtype, device=x_local.device)
    full[eng.cols[0]:eng.cols[1]] = x_local
    eng.comm.all_gather(full)
    return full[:n_true]
