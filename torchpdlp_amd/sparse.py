"""The constraint matrix as the HIP kernels want it: CSR of K and CSR of K' (int32 indices).

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
    if int(rp[-1]) >= 2 ** 31:
        raise ValueError("more than 2^31-1 non-zeros in one shard: split the problem across more ranks")
    return rp.to(torch.int32)


def csr_transpose(rowptr: torch.Tensor, colidx: torch.Tensor, val: torch.Tensor, m: int, n: int
                  ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CSR of the transpose; rows of the result keep the original row order (stable sort)."""
    counts = (rowptr[1:] - rowptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(m, dtype=torch.int32, device=val.device), counts)
    _, order = torch.sort(colidx, stable=True)
    t_col = rows[order]
    del rows
    t_val = val[order]
    del order
    t_rowptr = _counts_to_rowptr(torch.bincount(colidx, minlength=n) if colidx.numel() else
                                 torch.zeros(n, dtype=torch.int64, device=val.device))
    return t_rowptr, t_col.contiguous(), t_val.contiguous()


class CsrPair:
    """K (m x n) as CSR plus K' as CSR, on one device, values in ``dtype``."""

    def __init__(self, m: int, n: int, rowptr, colidx, val, t_rowptr=None, t_colidx=None, t_val=None):
        self.m, self.n = int(m), int(n)
        self.rowptr = rowptr.to(torch.int32).contiguous()
        self.colidx = colidx.to(torch.int32).contiguous()
        self.val = val.contiguous()
        if t_rowptr is None:
            t_rowptr, t_colidx, t_val = csr_transpose(self.rowptr, self.colidx, self.val, self.m, self.n)
        self.t_rowptr = t_rowptr.to(torch.int32).contiguous()
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
