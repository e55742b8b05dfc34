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
    for r in ent_row:
        if r not in row_sense:
            raise KeyError(r)
    # last occurrence of a (row, column) pair wins (dense assignment, util.py:183)
    ent = {}
    for j, r, v in zip(ent_var, ent_row, ent_val):
        ent[(r, j)] = v
    c = np.zeros(n)
    by_row = {}
    for (r, j), v in ent.items():
        if r == obj_row:
            c[j] = v                                  # util.py:172-177
        else:
            by_row.setdefault(r, []).append((j, v))
    # constraint rows in ROWS order: inequality block then equality block (util.py:190-231,250-261)
    g_rows, g_rhs, a_rows, a_rhs = [], [], [], []
    for name in row_order:
        if name == obj_row:
            continue
        sense = row_sense[name]
        b = rhs.get(name, 0.0)
        rng = ranges.get(name)
        cols = by_row.get(name, [])
        if rng is not None:                           # util.py:197-217
            if sense == "G":
                lb, ub = b, b + abs(rng)
            elif sense == "L":
                lb, ub = b - abs(rng), b
            elif sense == "E":
                lb, ub = (b, b + rng) if rng > 0 else (b + rng, b)
            else:
                raise ValueError(f"Unsupported ranged sense: {sense}")
            g_rows.append((cols, 1.0))
            g_rhs.append(lb)
            g_rows.append((cols, -1.0))
            g_rhs.append(-ub)
        elif sense == "E":                            # util.py:220-222
            a_rows.append((cols, 1.0))
            a_rhs.append(b)
        elif sense == "G":                            # util.py:223-225
            g_rows.append((cols, 1.0))
            g_rhs.append(b)
        elif sense == "L":                            # util.py:226-228
            g_rows.append((cols, -1.0))
            g_rhs.append(-b)
    rows = g_rows + a_rows
    m_ineq, m = len(g_rows), len(rows)
    if m == 0:
        raise RuntimeError("the model has no constraint rows (the reference fails in torch.vstack, util.py:260)")
    rowptr = np.zeros(m + 1, np.int64)
    colidx, vals = [], []
    for i, (cols, sign) in enumerate(rows):
        cols = sorted(cols)
        colidx.extend(j for j, _ in cols)
        vals.extend(sign * v for _, v in cols)
        rowptr[i + 1] = len(colidx)
    q = np.array(g_rhs + a_rhs, dtype=np.float64)
    l = np.array([lo.get(v, 0.0) if lo.get(v, 0.0) is not None else 0.0 for v in var_names], dtype=np.float64)   # util.py:234-239
    u = np.array([up.get(v, np.inf) if up.get(v, np.inf) is not None else np.inf for v in var_names], dtype=np.float64)
    return c, (rowptr, np.asarray(colidx, np.int64), np.asarray(vals, np.float64), m, n), q, m_ineq, l, u


def mps_to_standard_form(mps_file, device="cpu", support_sparse=True, verbose=False, *, compat: bool = True,
                         dtype=torch.float32) -> Tuple[torch.Tensor, CsrPair, torch.Tensor, int, torch.Tensor, torch.Tensor]:
    """Drop-in for the reference's ``mps_to_standard_form(mps_file, device, support_sparse, verbose)``."""
    c, (rowptr, colidx, vals, m, n), q, m_ineq, l, u = parse_mps(mps_file, compat=compat)
    dev = torch.device(device)
    t = lambda a: torch.tensor(a, dtype=dtype, device=dev).view(-1, 1)          # util.py:240-246: float32 column vectors
    K = CsrPair(m, n, torch.from_numpy(rowptr).to(torch.int32), torch.from_numpy(colidx).to(torch.int32),
                torch.from_numpy(vals).to(dtype)).to(dev)
    if verbose:
        print(f"Using Sparse operations ({m} x {n}, {K.nnz} non-zeros)")
    return t(c), K, t(q), m_ineq, t(l), t(u)
