"""Test / baseline infrastructure -- NOT part of the product path (only tests/ and bench.py's cpu_baseline leg import it).

The reference's own CPU path is eager PyTorch on a sparse-COO ``K`` (``util.py:62`` converts with ``to_sparse()``) with the
transposed product taken as ``K.T @ y`` -- SURVEY.md section 8d calls timing exactly that "flavour (i): reference-faithful".
The reference's files cannot travel to the GPU box, so this is a restatement of the same eager op sequence (one torch op per
arithmetic step, COO ``mm`` for both products, (len, 1) float32 column tensors), pinned against the recorded outputs of the
reference in tests/test_oracle.py.  ``oracle.py`` / ``pdlp_oracle.c`` (CSR + pre-transposed CSR, OpenMP) is flavour (ii).

Each method cites the reference lines it follows (paths under /root/reference/PDLP/).
"""
from __future__ import annotations

import numpy as np
import torch


class TorchCooLP:
    def __init__(self, m, n, m_ineq, rowptr, colidx, val, c, q, l, u):
        rp = torch.as_tensor(np.asarray(rowptr), dtype=torch.int64)
        rows = torch.repeat_interleave(torch.arange(int(m)), rp[1:] - rp[:-1])
        idx = torch.stack([rows, torch.as_tensor(np.asarray(colidx), dtype=torch.int64)])
        self.K = torch.sparse_coo_tensor(idx, torch.as_tensor(np.asarray(val), dtype=torch.float32), (int(m), int(n))).coalesce()
        colv = lambda v: torch.as_tensor(np.asarray(v), dtype=torch.float32).reshape(-1, 1).clone()
        self.c, self.q, self.l, self.u = colv(c), colv(q), colv(l), colv(u)
        self.m, self.n, self.m_ineq = int(m), int(n), int(m_ineq)
        # pdhg.py:11-17
        self.neg_inf = torch.isinf(self.l) & (self.l < 0)
        self.pos_inf = torch.isinf(self.u) & (self.u > 0)
        self.l_dual, self.u_dual = self.l.clone(), self.u.clone()
        self.l_dual[self.neg_inf] = 0
        self.u_dual[self.pos_inf] = 0

    def step_fixed(self, x, y, eta, omega, theta=1.0):
        """fixed_one_step_pdhg, primal_dual_hybrid_gradient_step.py:22-40"""
        K = self.K
        x_prev = x.clone()
        g = self.c - K.T @ y                                              # :25-26
        x = torch.clamp(x - eta / omega * g, min=self.l, max=self.u)      # :27
        xb = x + theta * (x - x_prev)                                     # :30
        y = y + eta * omega * (self.q - K @ xb)                           # :33-34
        if self.m_ineq > 0:
            y[:self.m_ineq] = torch.clamp(y[:self.m_ineq], min=0.0)       # :37-38
        return x, y

    def step_adaptive(self, x, y, eta, omega, theta, k):
        """adaptive_one_step_pdhg, ...step.py:65-115: ONE trial (the reference's loop returns on its first pass, quirk Q1).
        Returns (x+, y+, eta_used, eta_next)."""
        K = self.K
        x0, y0 = x.clone(), y.clone()
        g = self.c - K.T @ y0                                             # :68-69
        x = torch.clamp(x0 - (eta / omega) * g, min=self.l, max=self.u)   # :74-77
        dx = x - x0
        xb = x + theta * dx                                               # :79-80
        y = y0 + (eta * omega) * (self.q - K @ xb)                        # :82-83
        if self.m_ineq > 0:
            y[:self.m_ineq] = torch.clamp(y[:self.m_ineq], min=0.0)       # :85-86
        dy = y - y0
        den = 2 * ((dy.T @ K) @ dx)                                       # :96 -- the reference's third product
        if den != 0:                                                      # :99-102
            eta_bar = (omega * torch.linalg.norm(dx) ** 2 + torch.linalg.norm(dy) ** 2 / omega) / abs(den)
            t1 = (1 - (k + 1) ** (-0.3)) * eta_bar
        else:                                                             # :104-105
            eta_bar = t1 = torch.tensor(float("inf"))
        nxt = torch.min(t1, (1 + (k + 1) ** (-0.6)) * eta)                # :107-108
        if eta <= eta_bar:                                                # :110-111
            return x, y, eta.squeeze(), nxt.squeeze()
        return x, y, nxt.squeeze(), nxt.squeeze()                         # :113-115

    def kkt(self, x, y, omega):
        """compute_residuals_and_duality_gap + KKT_error, helpers.py:75-108 (project_lambda_box :21-37 as mask assignments)"""
        K = self.K
        g = self.c - K.T @ y                                              # :75
        p, d = (self.c.T @ x).flatten(), (self.q.T @ y).flatten()         # :76-77
        lam = g.clone()                                                   # helpers.py:21-37
        free, only_u, only_l = self.neg_inf & self.pos_inf, self.neg_inf & ~self.pos_inf, ~self.neg_inf & self.pos_inf
        lam[free] = 0
        lam[only_u] = torch.clamp(g[only_u], max=0.0)
        lam[only_l] = torch.clamp(g[only_l], min=0.0)
        adj = d + (self.l_dual.T @ torch.clamp(lam, min=0.0)).flatten() + (self.u_dual.T @ torch.clamp(lam, max=0.0)).flatten()   # :81-84
        gap = adj - p                                                     # :85
        r = K @ x - self.q                                                # :88
        r = torch.vstack([torch.clamp(r[:self.m_ineq], max=0.0), r[self.m_ineq:]])    # :89-90
        pr, dr = torch.linalg.norm(r, 2).flatten(), torch.linalg.norm(g - lam, 2).flatten()   # :91,:94
        kkt = torch.sqrt(omega ** 2 * pr ** 2 + dr ** 2 / omega ** 2 + gap ** 2)          # :102-106
        return dict(pr=pr, dr=dr, gap=gap, p=p, d_adj=adj, kkt=kkt)
