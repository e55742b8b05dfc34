"""The constraint matrix as the HIP kernels want it: CSR of K and CSR of K' (int32 column indices, int64 row pointers: one
matrix copy may hold more than 2^31 non-zeros).

The reference keeps ``K`` as a dense or COO torch tensor (``/root/reference/PDLP/util.py:240-267``)
and multiplies with ``K @ v`` / ``K.T @ v``.  Here both products are row-parallel SpMVs over
two CSR copies, so ``K.T`` never needs atomics.  Building the transposed copy is a one-time setup
step done with torch's sort (storage/plumbing, not the hot path).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def _counts_to_rowptr(counts: torch.Tensor) -> torch.Tensor:
    rp = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=counts.device)
    torch.cumsum(counts, 0, out=rp[1:])
    return rp


_SORT_LIMIT = 1 << 30       # torch.sort takes at most INT_MAX elements: larger matrices are transposed in row chunks


def csr_transpose(rowptr: torch.Tensor, colidx: torch.Tensor, val: torch.Tensor, m: int, n: int, chunk_nnz: int = _SORT_LIMIT
                  ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CSR of the transpose; rows of the result keep the original row order (stable sort).  Matrices with more than
    ``chunk_nnz`` entries go through in chunks of whole rows: every chunk is sorted by column on its own and scattered to where the
    one big stable sort would have put it (start of the column + what earlier chunks hold of that column + rank inside the chunk)."""
    nnz = int(colidx.numel())
    dev = val.device
    rp = rowptr.long()
    if nnz <= chunk_nnz:
        counts = rp[1:] - rp[:-1]
        rows = torch.repeat_interleave(torch.arange(m, dtype=torch.int32, device=dev), counts)
        _, order = torch.sort(colidx, stable=True)
        t_col = rows[order]
        del rows
        t_val = val[order]
        del order
        t_rowptr = _counts_to_rowptr(torch.bincount(colidx, minlength=n) if nnz else torch.zeros(n, dtype=torch.int64, device=dev))
        return t_rowptr, t_col.contiguous(), t_val.contiguous()
    # chunk boundaries: whole rows, at most chunk_nnz entries each (a single longer row stands alone and is cut by entries)
    cuts = [0]
    rp_h = rp.cpu()
    while cuts[-1] < m:
        r0 = cuts[-1]
        r1 = int(torch.searchsorted(rp_h, rp_h[r0] + chunk_nnz, right=True)) - 1
        cuts.append(min(m, max(r1, r0 + 1)))
    col_counts = torch.zeros(n, dtype=torch.int64, device=dev)
    per_chunk = []
    for r0, r1 in zip(cuts[:-1], cuts[1:]):
        a, b = int(rp_h[r0]), int(rp_h[r1])
        c = torch.zeros(n, dtype=torch.int64, device=dev)
        for p0 in range(a, b, chunk_nnz):              # (pieces only when one row alone exceeds the limit)
            c += torch.bincount(colidx[p0:min(b, p0 + chunk_nnz)], minlength=n)
        per_chunk.append(c)
        col_counts += c
    t_rowptr = _counts_to_rowptr(col_counts)
    t_col = torch.empty(nnz, dtype=torch.int32, device=dev)
    t_val = torch.empty(nnz, dtype=val.dtype, device=dev)
    base = t_rowptr[:-1].clone()                       # where the next chunk's entries of every column go
    for (r0, r1), c in zip(zip(cuts[:-1], cuts[1:]), per_chunk):
        a, b = int(rp_h[r0]), int(rp_h[r1])
        for p0 in range(a, b, chunk_nnz):
            p1 = min(b, p0 + chunk_nnz)
            cc = colidx[p0:p1]
            rows = (torch.searchsorted(rp, torch.arange(p0, p1, device=dev), right=True) - 1).to(torch.int32)
            cs, order = torch.sort(cc, stable=True)
            cpiece = torch.bincount(cc, minlength=n) if (p1 - p0) != (b - a) else c
            excl = torch.cumsum(cpiece, 0) - cpiece     # first position of every column's run in the sorted piece
            csl = cs.long()
            dest = base[csl] + (torch.arange(p1 - p0, device=dev) - excl[csl])
            t_col[dest] = rows[order]
            t_val[dest] = val[p0:p1][order]
            base += cpiece
            del rows, cs, order, csl, dest, excl
    return t_rowptr, t_col, t_val


class CsrPair:
    """K (m x n) as CSR plus K' as CSR, on one device, values in ``dtype``."""

    def __init__(self, m: int, n: int, rowptr, colidx, val, t_rowptr=None, t_colidx=None, t_val=None):
        self.m, self.n = int(m), int(n)
        self.rowptr = rowptr.to(torch.int64).contiguous()
        self.colidx = colidx.to(torch.int32).contiguous()
        self.val = val.contiguous()
        if t_rowptr is None:
            t_rowptr, t_colidx, t_val = csr_transpose(self.rowptr, self.colidx, self.val, self.m, self.n)
        self.t_rowptr = t_rowptr.to(torch.int64).contiguous()
        self.t_colidx = t_colidx.to(torch.int32).contiguous()
        self.t_val = t_val.contiguous()
        if self.rowptr.numel() != self.m + 1 or self.t_rowptr.numel() != self.n + 1:
            raise ValueError("row pointer length does not match the shape")
        if self.colidx.numel() != self.val.numel() or self.t_colidx.numel() != self.t_val.numel():
            raise ValueError("index / value length mismatch")

    # -- the bits of the torch.Tensor interface the reference's callers use on K ------------------
    @property
    def shape(self):
        return (self.m, self.n)

    @property
    def device(self):
        return self.val.device

    @property
    def dtype(self):
        return self.val.dtype

    @property
    def nnz(self) -> int:
        return int(self.val.numel())

    @property
    def is_sparse(self) -> bool:
        return True

    # -- constructors ------------------------------------------------------------------------------
    @classmethod
    def from_dense(cls, K: torch.Tensor) -> "CsrPair":
        m, n = K.shape
        nz = K != 0
        counts = nz.sum(1)
        idx = nz.nonzero()
        return cls(m, n, _counts_to_rowptr(counts), idx[:, 1].to(torch.int32), K[nz])

    @classmethod
    def from_coo(cls, K: torch.Tensor) -> "CsrPair":
        K = K.coalesce()
        m, n = K.shape
        r, c = K.indices()
        return cls(m, n, _counts_to_rowptr(torch.bincount(r, minlength=m)), c.to(torch.int32), K.values())

    @classmethod
    def from_any(cls, K, device=None, dtype=None) -> "CsrPair":
        """dense / COO / CSR torch tensor, scipy sparse matrix or CsrPair -> CsrPair."""
        if isinstance(K, CsrPair):
            out = K
        elif isinstance(K, torch.Tensor):
            if K.layout == torch.sparse_coo:
                out = cls.from_coo(K)
            elif K.layout == torch.sparse_csr:
                out = cls(K.shape[0], K.shape[1], K.crow_indices(), K.col_indices(), K.values())
            else:
                out = cls.from_dense(K)
        else:   # scipy.sparse
            Ks = K.tocsr()
            Ks.sort_indices()
            out = cls(Ks.shape[0], Ks.shape[1], torch.from_numpy(Ks.indptr.copy()), torch.from_numpy(Ks.indices.copy()),
                      torch.from_numpy(Ks.data.copy()))
        return out.to(device=device, dtype=dtype)

    def to(self, device=None, dtype=None) -> "CsrPair":
        device = self.device if device is None else torch.device(device)
        dtype = self.dtype if dtype is None else dtype
        if device == self.device and dtype == self.dtype:
            return self
        mv = lambda t: t.to(device)
        return CsrPair(self.m, self.n, mv(self.rowptr), mv(self.colidx), mv(self.val).to(dtype),
                       mv(self.t_rowptr), mv(self.t_colidx), mv(self.t_val).to(dtype))

    def clone(self) -> "CsrPair":
        return CsrPair(self.m, self.n, self.rowptr, self.colidx, self.val.clone(), self.t_rowptr, self.t_colidx,
                       self.t_val.clone())

    # -- shards (one process per GPU) ---------------------------------------------------------------
    @staticmethod
    def _row_slice(rowptr, colidx, val, r0, r1):
        a, b = int(rowptr[r0]), int(rowptr[r1])
        return (rowptr[r0:r1 + 1] - a).contiguous(), colidx[a:b].contiguous(), val[a:b].contiguous()

    def shard(self, row0: int, row1: int, col0: int, col1: int):
        """(rows [row0,row1) of K, rows [col0,col1) of K') with GLOBAL column indices."""
        return (self._row_slice(self.rowptr, self.colidx, self.val, row0, row1),
                self._row_slice(self.t_rowptr, self.t_colidx, self.t_val, col0, col1))

    # -- small-problem helpers (tests, fixtures) ----------------------------------------------------
    def to_dense(self) -> torch.Tensor:
        counts = (self.rowptr[1:] - self.rowptr[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(self.m, device=self.device), counts)
        K = torch.zeros(self.m, self.n, dtype=self.dtype, device=self.device)
        K.index_put_((rows, self.colidx.long()), self.val, accumulate=True)
        return K


def as_vec(v: torch.Tensor, length: Optional[int] = None, device=None, dtype=None) -> torch.Tensor:
    """(len,) contiguous view/copy of a (len,1) or (len,) tensor."""
    v = v.reshape(-1)
    if device is not None or dtype is not None:
        v = v.to(device=device, dtype=dtype)
    v = v.contiguous()
    if length is not None and v.numel() != length:
        raise ValueError(f"expected a vector of length {length}, got {v.numel()}")
    return v
