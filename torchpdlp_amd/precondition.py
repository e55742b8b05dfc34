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


def _sweeps(lib, code, stream, rows_K, rows_KT, K_blk, KT_blk, D_row, D_col, max_iter, eps, comm=None, r0=0, c0=0):
    """The Ruiz sweeps (enhancements.py:45-62) on one rank's row block of K (``rows_K`` rows, global column indices) and of K'
    (``rows_KT`` rows, global row indices), in place.  Row norms of K and of K' are rank local; what the other matrix copy needs
    of them -- the factors of ALL constraints resp. ALL variables, for its columns -- is one all-gather per half-sweep, and the
    early-exit test (quirk Q3: the ROW factors, twice) one all-reduce(max).  Every entry is divided by the same factors in the
    same order as in the single-process sweep, so the scaled shards equal the shards of the scaled matrix bit for bit."""
    (rp, ci, va), (t_rp, t_ci, t_va) = K_blk, KT_blk
    dev, dt = va.device, va.dtype
    p = lambda t: t.data_ptr()
    world = 1 if comm is None else comm.world
    m_full, n_full = rows_K * world, rows_KT * world            # (equal, padded blocks)
    rn_full = torch.empty(m_full, dtype=dt, device=dev)
    cn_full = torch.empty(n_full, dtype=dt, device=dev)
    rn, cn = rn_full[r0:r0 + rows_K], cn_full[c0:c0 + rows_KT]
    work = torch.zeros(1, dtype=torch.float64, device=dev)
    nnz_K, nnz_KT = int(va.numel()), int(t_va.numel())
    sweeps = 0
    tstream = torch.cuda.current_stream(dev)
    for _ in range(int(max_iter)):
        with N.trace_range(f"pdlp: Ruiz sweep {sweeps + 1}", tstream):
            sweeps += 1
            N.check(lib.pdlp_csr_row_scale_factors(code, rows_K, p(rp), p(va), float(eps), p(rn), stream), "row factors")            # :49-50
            N.check(lib.pdlp_vec_muldiv(code, rows_K, p(D_row), p(rn), 1, stream), "D_row /= r")                                     # :51
            N.check(lib.pdlp_csr_div_rows(code, rows_K, p(rp), p(va), p(rn), stream), "K rows /= r")                                 # :52
            if comm is not None:
                comm.all_gather(rn_full)
            N.check(lib.pdlp_csr_div_cols(code, nnz_KT, p(t_ci), p(t_va), p(rn_full), stream), "K' cols /= r")
            N.check(lib.pdlp_csr_row_scale_factors(code, rows_KT, p(t_rp), p(t_va), float(eps), p(cn), stream), "col factors")       # :54-55
            N.check(lib.pdlp_vec_muldiv(code, rows_KT, p(D_col), p(cn), 1, stream), "D_col /= c")                                    # :56
            N.check(lib.pdlp_csr_div_rows(code, rows_KT, p(t_rp), p(t_va), p(cn), stream), "K' rows /= c")                           # :57
            if comm is not None:
                comm.all_gather(cn_full)
            N.check(lib.pdlp_csr_div_cols(code, nnz_K, p(ci), p(va), p(cn_full), stream), "K cols /= c")
            dev_from_one = C.c_double(0)
            N.check(lib.pdlp_vec_max_dev_from_one(code, rows_K, p(rn), p(work), C.byref(dev_from_one), stream), "max|1-r|")          # :60-61
            worst = dev_from_one.value
            if comm is not None:
                w = torch.tensor([worst], dtype=torch.float64, device=dev)
                comm.all_reduce_max(w)
                worst = float(w)
            if worst < eps:
                break
    return sweeps


def ruiz_precondition_shard(shard: dict, comm, max_iter=20, eps=1e-6) -> dict:
    """Ruiz on a problem that only exists as shards: ``shard`` = this rank's keyword arguments of ``PdlpEngine`` as
    ``distributed.shard_arrays`` / ``gen_lp_shard_arrays`` build them (row block of K and of K' in the padded layout, local
    ``c, q, l, u``).  Returns a new dict with the scaled blocks and vectors plus ``d_col`` / ``d_row`` (local blocks) -- no rank
    ever holds a full matrix.  ``shard["ruiz_seconds"]`` / ``["ruiz_sweeps"]`` record the cost.  (enhancements.py:4-71)"""
    t0 = time.time()
    lib = N.load()
    (rp, ci, va), (t_rp, t_ci, t_va) = shard["K_rows"], shard["KT_rows"]
    dev, dt = va.device, va.dtype
    if dev.type != "cuda":
        raise N.PdlpError("ruiz_precondition_shard runs on the HIP device (there is no CPU fallback)")
    code = _DT[dt]
    stream = torch.cuda.current_stream(dev).cuda_stream
    (r0, r1), (c0, c1) = shard["rows"], shard["cols"]
    ml, nl = r1 - r0, c1 - c0
    i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()
    i64 = lambda t: t.to(device=dev, dtype=torch.int64).contiguous()
    K_blk = (i64(rp), i32(ci), va.clone())
    KT_blk = (i64(t_rp), i32(t_ci), t_va.to(dev).clone())
    vec = lambda v, ln: as_vec(v, ln, dev, dt).clone()
    c_s, q_s, l_s, u_s = vec(shard["c"], nl), vec(shard["q"], ml), vec(shard["l"], nl), vec(shard["u"], nl)
    D_row = torch.ones(ml, dtype=dt, device=dev)
    D_col = torch.ones(nl, dtype=dt, device=dev)
    world = 1 if comm is None else comm.world
    if ml * world != shard["m"] or nl * world != shard["n"]:
        raise ValueError("sharded Ruiz needs the equal, padded blocks of torchpdlp_amd/distributed.py")
    sweeps = _sweeps(lib, code, stream, ml, nl, K_blk, KT_blk, D_row, D_col, max_iter, eps,
                     comm if world > 1 else None, r0, c0)
    p = lambda t: t.data_ptr()
    N.check(lib.pdlp_vec_muldiv(code, nl, p(c_s), p(D_col), 0, stream), "c *= D_col")      # :64
    N.check(lib.pdlp_vec_muldiv(code, ml, p(q_s), p(D_row), 0, stream), "q *= D_row")      # :65
    N.check(lib.pdlp_vec_muldiv(code, nl, p(l_s), p(D_col), 1, stream), "l /= D_col")      # :66
    N.check(lib.pdlp_vec_muldiv(code, nl, p(u_s), p(D_col), 1, stream), "u /= D_col")      # :67
    torch.cuda.current_stream(dev).synchronize()
    out = dict(shard)
    out.update(K_rows=K_blk, KT_rows=KT_blk, c=c_s, q=q_s, l=l_s, u=u_s, d_col=D_col, d_row=D_row,
               ruiz_seconds=time.time() - t0, ruiz_sweeps=sweeps)
    return out


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
    p = lambda t: t.data_ptr()
    _sweeps(lib, code, stream, m, n, (Ks.rowptr, Ks.colidx, Ks.val), (Ks.t_rowptr, Ks.t_colidx, Ks.t_val), D_row, D_col, max_iter, eps)
    N.check(lib.pdlp_vec_muldiv(code, n, p(c_s), p(D_col), 0, stream), "c *= D_col")      # :64
    N.check(lib.pdlp_vec_muldiv(code, m, p(q_s), p(D_row), 0, stream), "q *= D_row")      # :65
    N.check(lib.pdlp_vec_muldiv(code, n, p(l_s), p(D_col), 1, stream), "l /= D_col")      # :66
    N.check(lib.pdlp_vec_muldiv(code, n, p(u_s), p(D_col), 1, stream), "u /= D_col")      # :67
    torch.cuda.current_stream(dev).synchronize()
    col = lambda v: v.view(-1, 1)
    return (Ks, col(c_s), col(q_s), col(l_s), col(u_s), (col(D_col), col(D_row), K, c, q, l, u), time.time() - t0)
