"""Ruiz equilibration on the two CSR copies of K, on the device.

Drop-in for ``ruiz_precondition`` (``/root/reference/PDLP/enhancements.py:4-71``), which is
dense-only (``torch.linalg.norm(K, ord=inf, dim=...)`` does not take sparse input) and therefore
cannot run at the benchmark sizes.  Each sweep is: row factors of K (sqrt of the row's max |.|,
1 when < eps), divide K's rows and K''s columns by them; row factors of K' (= K's columns),
divide K''s rows and K's columns by them.  The kernels are the ``pdlp_csr_*`` / ``pdlp_vec_*``
entry points of the C ABI.
"""
from __future__ import annotations

import ctypes as C
import time

import torch

from . import _native as N
from .engine import _DT
from .sparse import CsrPair, as_vec


def ruiz_precondition(c, K, q, l, u, device=None, max_iter=20, eps=1e-6):
    """Returns ``(K_s, c_s, q_s, l_s, u_s, (D_col, D_row, K, c, q, l, u), time_used)`` like the reference.

    ``K_s`` is a ``CsrPair`` (both copies scaled consistently).  Reproduces the reference's early-exit
    test, which looks at the ROW factors twice (quirk Q3, enhancements.py:60-61).
    """
    t0 = time.time()
    lib = N.load()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    Kp = CsrPair.from_any(K, device=device)
    dev, dt = Kp.device, Kp.dtype
    if dev.type != "cuda":
        raise N.PdlpError("ruiz_precondition runs on the HIP device (there is no CPU fallback)")
    code = _DT[dt]
    stream = torch.cuda.current_stream(dev).cuda_stream
    Ks = Kp.clone()
    m, n, nnz = Ks.m, Ks.n, int(Ks.val.numel())
    vec = lambda v, ln: as_vec(v, ln, dev, dt).clone()
    c_s, q_s, l_s, u_s = vec(c, n), vec(q, m), vec(l, n), vec(u, n)
    D_row = torch.ones(m, dtype=dt, device=dev)
    D_col = torch.ones(n, dtype=dt, device=dev)
    rn = torch.empty(m, dtype=dt, device=dev)
    cn = torch.empty(n, dtype=dt, device=dev)
    work = torch.zeros(1, dtype=torch.float64, device=dev)
    p = lambda t: t.data_ptr()
    for _ in range(int(max_iter)):
        N.check(lib.pdlp_csr_row_scale_factors(code, m, p(Ks.rowptr), p(Ks.val), float(eps), p(rn), stream), "row factors")   # :49-50
        N.check(lib.pdlp_vec_muldiv(code, m, p(D_row), p(rn), 1, stream), "D_row /= r")                                       # :51
        N.check(lib.pdlp_csr_div_rows(code, m, p(Ks.rowptr), p(Ks.val), p(rn), stream), "K rows /= r")                         # :52
        N.check(lib.pdlp_csr_div_cols(code, nnz, p(Ks.t_colidx), p(Ks.t_val), p(rn), stream), "K' cols /= r")
        N.check(lib.pdlp_csr_row_scale_factors(code, n, p(Ks.t_rowptr), p(Ks.t_val), float(eps), p(cn), stream), "col factors")  # :54-55
        N.check(lib.pdlp_vec_muldiv(code, n, p(D_col), p(cn), 1, stream), "D_col /= c")                                        # :56
        N.check(lib.pdlp_csr_div_rows(code, n, p(Ks.t_rowptr), p(Ks.t_val), p(cn), stream), "K' rows /= c")                    # :57
        N.check(lib.pdlp_csr_div_cols(code, nnz, p(Ks.colidx), p(Ks.val), p(cn), stream), "K cols /= c")
        dev_from_one = C.c_double(0)
        N.check(lib.pdlp_vec_max_dev_from_one(code, m, p(rn), p(work), C.byref(dev_from_one), stream), "max|1-r|")             # :60-61
        if dev_from_one.value < eps:
            break
    N.check(lib.pdlp_vec_muldiv(code, n, p(c_s), p(D_col), 0, stream), "c *= D_col")      # :64
    N.check(lib.pdlp_vec_muldiv(code, m, p(q_s), p(D_row), 0, stream), "q *= D_row")      # :65
    N.check(lib.pdlp_vec_muldiv(code, n, p(l_s), p(D_col), 1, stream), "l /= D_col")      # :66
    N.check(lib.pdlp_vec_muldiv(code, n, p(u_s), p(D_col), 1, stream), "u /= D_col")      # :67
    torch.cuda.current_stream(dev).synchronize()
    col = lambda v: v.view(-1, 1)
    return (Ks, col(c_s), col(q_s), col(l_s), col(u_s), (col(D_col), col(D_row), K, c, q, l, u), time.time() - t0)
