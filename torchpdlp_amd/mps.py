"""MPS ingestion straight into CSR -- counterpart of the reference's ``mps_to_standard_form``
(``/root/reference/PDLP/util.py:76-268``), which builds one dense numpy row per constraint (O(m*n) memory,
``util.py:179-183``) and so cannot load the instances the GPU path is for.

Same output convention: ``(c, K, q, m_ineq, l, u)`` for

    min c'x   s.t.  K[:m_ineq] x >= q[:m_ineq],   K[m_ineq:] x = q[m_ineq:],   l <= x <= u

with the inequality block first (ROWS order; a ranged row becomes two consecutive ">=" rows,
``util.py:214-217``), then the equality block; ``L`` rows are negated (``util.py:226-228``); columns in order
of first appearance (``util.py:134-137``).  ``K`` is returned as a ``CsrPair`` (always sparse; the reference's
``support_sparse`` micro-benchmark, ``util.py:29-74,263-267``, is moot), vectors as ``(len, 1)`` float32 tensors.

``compat=True`` (default) reproduces the reference's reading of the format exactly, including its quirks
(SURVEY.md Q5): section keywords count only when they are the whole line, so ``NAME  AFIRO`` and any unknown
section header fall through as data lines of the current section; ``FR`` bounds mean ``[0, +inf)``
(``util.py:162-164``); ``MI``/``PL``/``BV``/``LI``/``UI`` bounds are ignored (``util.py:155-164``); an ``UP`` bound
never moves the lower bound; RHS entries of the objective row are ignored; integer ``MARKER`` lines make the load
fail with ``ValueError`` as in the reference.  ``compat=False`` applies the standard MPS meaning to ``FR``, ``MI``,
``PL``, ``BV`` and negative ``UP`` bounds, and skips ``MARKER`` lines (LP relaxation).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from .sparse import CsrPair

_SECTIONS = ("ROWS", "COLUMNS", "RHS", "RANGES", "BOUNDS")


def parse_mps(mps_file: str, compat: bool = True):
    """Parse into numpy pieces: (c, (rowptr, colidx, val, m, n), q, m_ineq, l, u), all float64 / int64."""
    with open(mps_file, "r") as f:       # util.py:93-94: blank lines and lines starting with '*' are dropped
        lines = [ln.strip() for ln in f if ln.strip() and not ln.startswith("*")]
    section = None
    row_sense, row_order = {}, []
    obj_row = None
    var_index, var_names = {}, []
    ent_var, ent_row, ent_val = [], [], []          # COLUMNS entries in file order
    rhs, ranges = {}, {}
    lo, up = {}, {}
    for line in lines:
        if line == "NAME" or line == "ENDATA":       # util.py:106-107 (only when the whole line)
            continue
        if line in _SECTIONS:                        # util.py:108-122
            section = line
            continue
        tok = line.split()
        if section == "ROWS":                        # util.py:125-131
            sense, name = tok                        # (ValueError on a malformed line, as the reference)
            if name not in row_sense:
                row_order.append(name)
            row_sense[name] = sense
            if sense == "N":
                obj_row = name
        elif section == "COLUMNS":                   # util.py:133-140
            if not compat and len(tok) >= 3 and tok[1] == "'MARKER'":
                continue
            name = tok[0]
            j = var_index.get(name)
            if j is None:
                j = var_index[name] = len(var_names)
                var_names.append(name)
            for i in range(1, len(tok), 2):
                ent_var.append(j)
                ent_row.append(tok[i])
                ent_val.append(float(tok[i + 1]))
        elif section == "RHS":                       # util.py:142-145
            for i in range(1, len(tok), 2):
                rhs[tok[i]] = float(tok[i + 1])
        elif section == "RANGES":                    # util.py:147-150
            for i in range(1, len(tok), 2):
                ranges[tok[i]] = float(tok[i + 1])
        elif section == "BOUNDS":                    # util.py:152-164
            btype, _, name = tok[:3]
            val = float(tok[3]) if len(tok) > 3 else None
            if btype == "LO":
                lo[name] = val
            elif btype == "UP":
                up[name] = val
                if not compat and val is not None and val < 0 and name not in lo:
                    lo[name] = -np.inf
            elif btype == "FX":
                lo[name] = val
                up[name] = val
            elif btype == "FR":
                lo[name] = 0.0 if compat else -np.inf
                up[name] = np.inf
            elif not compat:
                if btype == "MI":
                    lo[name] = -np.inf
                elif btype == "PL":
                    up[name] = np.inf
                elif btype == "BV":
                    lo[name], up[name] = 0.0, 1.0
    n = len(var_names)
    # every entry must name a declared row (the reference raises KeyError at util.py:183)
    row_id = {name: i for i, name in enumerate(row_order)}
    try:
        R = np.fromiter((row_id[r] for r in ent_row), dtype=np.int64, count=len(ent_row))
    except KeyError as e:
        raise KeyError(e.args[0])
    J = np.asarray(ent_var, dtype=np.int64)
    V = np.asarray(ent_val, dtype=np.float64)
    del ent_row, ent_var, ent_val
    # last occurrence of a (row, column) pair wins (dense assignment, util.py:183); vectorised (round 2: the per-entry
    # dictionaries of round 1 made files beyond ~1e7 non-zeros unusable): stable sort by (row, column), keep each group's last
    order = np.argsort(R * max(n, 1) + J, kind="stable")
    ks = (R * max(n, 1) + J)[order]
    keep = np.ones(ks.size, dtype=bool)
    keep[:-1] = ks[1:] != ks[:-1]
    sel = order[keep]
    R, J, V = R[sel], J[sel], V[sel]                  # now sorted by (row, column)
    c = np.zeros(n)
    if obj_row is not None:                           # util.py:172-177
        mo = R == row_id[obj_row]
        c[J[mo]] = V[mo]
    starts = np.zeros(len(row_order) + 1, np.int64)
    starts[1:] = np.cumsum(np.bincount(R, minlength=len(row_order)))
    # constraint rows in ROWS order: inequality block then equality block (util.py:190-231,250-261)
    g_src, g_sign, g_rhs, a_src, a_rhs = [], [], [], [], []
    for name in row_order:
        if name == obj_row:
            continue
        sense = row_sense[name]
        b = rhs.get(name, 0.0)
        rng = ranges.get(name)
        src = row_id[name]
        if rng is not None:                           # util.py:197-217
            if sense == "G":
                lb, ub = b, b + abs(rng)
            elif sense == "L":
                lb, ub = b - abs(rng), b
            elif sense == "E":
                lb, ub = (b, b + rng) if rng > 0 else (b + rng, b)
            else:
                raise ValueError(f"Unsupported ranged sense: {sense}")
            g_src += [src, src]
            g_sign += [1.0, -1.0]
            g_rhs += [lb, -ub]
        elif sense == "E":                            # util.py:220-222
            a_src.append(src)
            a_rhs.append(b)
        elif sense == "G":                            # util.py:223-225
            g_src.append(src)
            g_sign.append(1.0)
            g_rhs.append(b)
        elif sense == "L":                            # util.py:226-228
            g_src.append(src)
            g_sign.append(-1.0)
            g_rhs.append(-b)
    m_ineq, m = len(g_src), len(g_src) + len(a_src)
    if m == 0:
        raise RuntimeError("the model has no constraint rows (the reference fails in torch.vstack, util.py:260)")
    src = np.asarray(g_src + a_src, dtype=np.int64)
    sign = np.asarray(g_sign + [1.0] * len(a_src), dtype=np.float64)
    lens = starts[src + 1] - starts[src]
    rowptr = np.zeros(m + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    pos = np.repeat(starts[src] - rowptr[:-1], lens) + np.arange(int(rowptr[-1]))     # source position of every output entry
    colidx = J[pos]
    vals = np.repeat(sign, lens) * V[pos]
    q = np.array(g_rhs + a_rhs, dtype=np.float64)
    l = np.array([lo.get(v, 0.0) if lo.get(v, 0.0) is not None else 0.0 for v in var_names], dtype=np.float64)   # util.py:234-239
    u = np.array([up.get(v, np.inf) if up.get(v, np.inf) is not None else np.inf for v in var_names], dtype=np.float64)
    return c, (rowptr, colidx.astype(np.int64), vals.astype(np.float64), m, n), q, m_ineq, l, u


def mps_to_standard_form(mps_file, device="cpu", support_sparse=True, verbose=False, *, compat: bool = True,
                         dtype=torch.float32) -> Tuple[torch.Tensor, CsrPair, torch.Tensor, int, torch.Tensor, torch.Tensor]:
    """Drop-in for the reference's ``mps_to_standard_form(mps_file, device, support_sparse, verbose)``."""
    c, (rowptr, colidx, vals, m, n), q, m_ineq, l, u = parse_mps(mps_file, compat=compat)
    dev = torch.device(device)
    t = lambda a: torch.tensor(a, dtype=dtype, device=dev).view(-1, 1)          # util.py:240-246: float32 column vectors
    K = CsrPair(m, n, torch.from_numpy(rowptr).to(torch.int64), torch.from_numpy(colidx).to(torch.int32),
                torch.from_numpy(vals).to(dtype)).to(dev)
    if verbose:
        print(f"Using Sparse operations ({m} x {n}, {K.nnz} non-zeros)")
    return t(c), K, t(q), m_ineq, t(l), t(u)
