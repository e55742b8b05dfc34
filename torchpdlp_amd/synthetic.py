"""Synthetic sparse LP instances, built directly in CSR on any torch device.

The LP form is the reference's (``/root/reference/PDLP/util.py:76-84``)::

    min c'x   s.t.  K[:m_ineq] x >= q[:m_ineq],  K[m_ineq:] x = q[m_ineq:],  l <= x <= u

Three recipes (``box_v1``: the single-generator form of ``box`` the golden fixtures were drawn with; frozen):

``box``    (every array is seeded per 2^16-row chunk: ``box_rows`` / ``box_vectors`` let one rank of a sharded run generate
           exactly its rows of the same instance -- ``distributed.gen_lp_shard``)
           the distribution of the reference's own generator
           (``/root/reference/Packages/generate_feasible_lp.py:18-41``), restated for a
           row-regular sparse matrix: every row has ``nnz_per_row`` uniformly random
           columns with U[0,1) values, ``x_feas ~ U(-10,10)``, inequality rows get a
           slack ``U(0.1,5)``, every variable is boxed around ``x_feas`` and
           ``c ~ N(0,1)``.  Always feasible and bounded.  This is the bench workload
           (BASELINE.json configs[1] and configs[3]).
``mixed``  all four bound classes of ``project_lambda_box``
           (``/root/reference/PDLP/helpers.py:3-39``) with a primal-dual optimal pair
           built in, so the optimal objective is known exactly.  Used by tests/fixtures.

Nothing here is on the solver hot path; torch is used as the array library.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class SyntheticLP:
    m: int
    n: int
    m_ineq: int
    rowptr: torch.Tensor   # int64 [m+1]
    colidx: torch.Tensor   # int32 [nnz], sorted inside each row (duplicates possible, rare)
    val: torch.Tensor      # dtype [nnz]
    c: torch.Tensor        # [n]
    q: torch.Tensor        # [m]
    l: torch.Tensor        # [n]
    u: torch.Tensor        # [n]
    x_feas: torch.Tensor   # [n] a feasible point (optimal for recipe "mixed")
    y_opt: Optional[torch.Tensor] = None   # [m] dual optimal (recipe "mixed")
    opt_obj: Optional[float] = None

    @property
    def nnz(self) -> int:
        return int(self.colidx.numel())


def _regular_pattern(m, n, k, gen, device, chunk_rows):
    """k uniformly random columns per row, sorted per row; int32 [m*k]."""
    col = torch.empty(m * k, dtype=torch.int32, device=device)
    for r0 in range(0, m, chunk_rows):
        r1 = min(m, r0 + chunk_rows)
        blk = torch.randint(0, n, (r1 - r0, k), generator=gen, device=device, dtype=torch.int32)
        blk, _ = torch.sort(blk, dim=1)
        col[r0 * k:r1 * k] = blk.reshape(-1)
        del blk
    return col


def _regular_matvec(col, val, x, m, k, chunk_rows):
    """(K x) for the row-regular pattern, accumulated in float64."""
    out = torch.empty(m, dtype=torch.float64, device=x.device)
    for r0 in range(0, m, chunk_rows):
        r1 = min(m, r0 + chunk_rows)
        cc = col[r0 * k:r1 * k].long()
        out[r0:r1] = (val[r0 * k:r1 * k].double() * x[cc].double()).view(r1 - r0, k).sum(1)
    return out


def _regular_rmatvec(col, val, y, m, n, k, chunk_rows):
    """(K' y) for the row-regular pattern, accumulated in float64."""
    out = torch.zeros(n, dtype=torch.float64, device=y.device)
    for r0 in range(0, m, chunk_rows):
        r1 = min(m, r0 + chunk_rows)
        cc = col[r0 * k:r1 * k].long()
        w = val[r0 * k:r1 * k].double() * y[r0:r1].double().repeat_interleave(k)
        out.index_add_(0, cc, w)
    return out


CHUNK_ROWS = 1 << 16


def _chunk_gen(seed: int, stream: int, chunk: int, device) -> torch.Generator:
    """the generator of one 2^16-row chunk of one random array of the "box" recipe: every chunk can be produced on its own,
    so a rank of a sharded run generates exactly its rows of the very instance a single process generates"""
    g = torch.Generator(device=device)
    g.manual_seed((int(seed) * 1_000_003 + 7919 * int(stream) + int(chunk)) % (2 ** 63 - 1))
    return g


def box_rows(n: int, k: int, seed: int, r_lo: int, r_hi: int, device, dtype=torch.float32):
    """rows [r_lo, r_hi) of the "box" recipe's matrix: (column indices int32 [rows*k] sorted inside each row, values [rows*k])"""
    device = torch.device(device)
    cols, vals = [], []
    for ch in range(r_lo // CHUNK_ROWS, (max(r_hi, r_lo + 1) - 1) // CHUNK_ROWS + 1):
        c0 = ch * CHUNK_ROWS
        rows = CHUNK_ROWS
        blk = torch.randint(0, n, (rows, k), generator=_chunk_gen(seed, 1, ch, device), device=device, dtype=torch.int32)
        blk, _ = torch.sort(blk, dim=1)
        v = torch.rand(rows * k, generator=_chunk_gen(seed, 2, ch, device), device=device, dtype=torch.float32).view(rows, k)
        a, b = max(r_lo, c0) - c0, min(r_hi, c0 + rows) - c0
        if b > a:
            cols.append(blk[a:b].reshape(-1))
            vals.append(v[a:b].reshape(-1).to(dtype))
        del blk, v
    if not cols:
        return torch.zeros(0, dtype=torch.int32, device=device), torch.zeros(0, dtype=dtype, device=device)
    return torch.cat(cols), torch.cat(vals)


def box_vectors(n: int, m: int, seed: int, device, ineq_frac: float = 0.8) -> dict:
    """the vectors of the "box" recipe in float64 (x_feas, slack of the inequality rows, l, u, c), each from its own seed"""
    device = torch.device(device)
    m_ineq = int(round(ineq_frac * m))

    def U(stream, count, lo, hi):
        return torch.rand(count, generator=_chunk_gen(seed, stream, 0, device), device=device, dtype=torch.float64) * (hi - lo) + lo

    x_feas = U(3, n, -10.0, 10.0)
    return dict(m_ineq=m_ineq, x_feas=x_feas, slack=U(4, m_ineq, 0.1, 5.0),
                l=torch.clamp(x_feas - U(5, n, 1.0, 5.0), min=-1e4), u=torch.clamp(x_feas + U(6, n, 1.0, 5.0), max=1e4),
                c=torch.randn(n, generator=_chunk_gen(seed, 7, 0, device), device=device, dtype=torch.float64))


def gen_lp(n: int, m: int, nnz_per_row: int, seed: int = 0, device="cpu",
           dtype=torch.float32, ineq_frac: float = 0.8, recipe: str = "box",
           chunk_rows: int = 1 << 20) -> SyntheticLP:
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    k = int(nnz_per_row)
    m_ineq = int(round(ineq_frac * m))

    def U(shape, lo, hi):
        return torch.rand(shape, generator=gen, device=device, dtype=torch.float64) * (hi - lo) + lo

    if recipe == "box":
        vec = box_vectors(n, m, seed, device, ineq_frac)
        col, val = box_rows(n, k, seed, 0, m, device, dtype)
        rowptr = torch.arange(0, (m + 1) * k, k, dtype=torch.int64, device=device)
        kx = _regular_matvec(col, val, vec["x_feas"], m, k, chunk_rows)
        q = kx.clone()
        q[:m_ineq] -= vec["slack"]
        return SyntheticLP(m, n, m_ineq, rowptr, col, val, vec["c"].to(dtype), q.to(dtype), vec["l"].to(dtype), vec["u"].to(dtype),
                           vec["x_feas"].to(dtype))

    if recipe not in ("mixed", "box_v1"):
        raise ValueError(f"unknown recipe {recipe!r}")
    col = _regular_pattern(m, n, k, gen, device, chunk_rows)
    rowptr = torch.arange(0, (m + 1) * k, k, dtype=torch.int64, device=device)

    if recipe == "box_v1":
        # The "box" distribution drawn from ONE generator in sequence: the recipe as it stood when tests/golden/*.npz were
        # generated (the box_200x150 case).  Frozen so that tests/golden/gen_golden.py keeps reproducing its fixtures byte for
        # byte; new code uses "box" (chunk-seeded, shardable).
        val = torch.empty(m * k, dtype=dtype, device=device)
        for r0 in range(0, m * k, chunk_rows * 8):
            r1 = min(m * k, r0 + chunk_rows * 8)
            val[r0:r1] = torch.rand(r1 - r0, generator=gen, device=device, dtype=torch.float32).to(dtype)
        x_feas = U(n, -10.0, 10.0)
        kx = _regular_matvec(col, val, x_feas, m, k, chunk_rows)
        q = kx.clone()
        q[:m_ineq] -= U(m_ineq, 0.1, 5.0)
        l = torch.clamp(x_feas - U(n, 1.0, 5.0), min=-1e4)
        u = torch.clamp(x_feas + U(n, 1.0, 5.0), max=1e4)
        c = torch.randn(n, generator=gen, device=device, dtype=torch.float64)
        return SyntheticLP(m, n, m_ineq, rowptr, col, val, c.to(dtype), q.to(dtype), l.to(dtype), u.to(dtype), x_feas.to(dtype))

    val = torch.randn(m * k, generator=gen, device=device, dtype=torch.float64).to(dtype)
    # bound classes: 0 boxed, 1 lower only, 2 upper only, 3 free
    cls = torch.multinomial(torch.tensor([0.4, 0.3, 0.15, 0.15], device=device),
                            n, replacement=True, generator=gen)
    x_feas = U(n, -2.0, 2.0)
    at_bound = torch.rand(n, generator=gen, device=device) < 0.5
    gap_lo = torch.where(at_bound, torch.zeros(n, dtype=torch.float64, device=device), U(n, 0.1, 2.0))
    gap_hi = U(n, 0.1, 2.0)
    inf = torch.full((n,), float("inf"), dtype=torch.float64, device=device)
    l = x_feas - gap_lo
    u = x_feas + gap_hi
    # upper-only variables sit at their upper bound when "at_bound"
    u = torch.where(cls == 2, x_feas + gap_lo, u)
    l = torch.where((cls == 2) | (cls == 3), -inf, l)
    u = torch.where((cls == 1) | (cls == 3), inf, u)
    # reduced costs with the sign complementary slackness allows
    lam_mag = U(n, 0.1, 1.0)
    lam = torch.zeros(n, dtype=torch.float64, device=device)
    lam = torch.where(at_bound & ((cls == 0) | (cls == 1)), lam_mag, lam)     # x at l  -> lam >= 0
    lam = torch.where(at_bound & (cls == 2), -lam_mag, lam)                  # x at u  -> lam <= 0
    kx = _regular_matvec(col, val, x_feas, m, k, chunk_rows)
    active = torch.rand(m, generator=gen, device=device) < 0.5
    slack = torch.where(active, torch.zeros(m, dtype=torch.float64, device=device), U(m, 0.1, 2.0))
    q = kx.clone()
    q[:m_ineq] -= slack[:m_ineq]
    y = torch.randn(m, generator=gen, device=device, dtype=torch.float64)
    y[:m_ineq] = torch.where(active[:m_ineq], y[:m_ineq].abs(), torch.zeros_like(y[:m_ineq]))
    c = _regular_rmatvec(col, val, y, m, n, k, chunk_rows) + lam
    obj = float((c * x_feas).sum())
    return SyntheticLP(m, n, m_ineq, rowptr, col, val, c.to(dtype), q.to(dtype),
                       l.to(dtype), u.to(dtype), x_feas.to(dtype), y.to(dtype), obj)


def csr_to_dense(lp: SyntheticLP) -> torch.Tensor:
    """Dense (m, n) copy of K (duplicates summed), for small fixtures only."""
    rows = torch.repeat_interleave(torch.arange(lp.m, device=lp.val.device),
                                   (lp.rowptr[1:] - lp.rowptr[:-1]).long())
    K = torch.zeros(lp.m, lp.n, dtype=lp.val.dtype, device=lp.val.device)
    K.index_put_((rows, lp.colidx.long()), lp.val, accumulate=True)
    return K
