"""Panel-tiled storage of a CSR matrix for the fast SpMV kernel (``k_tiled_fused`` in csrc/pdlp_hip.hip).

Why (profiles/r01_microbench_gather.txt): at 10M columns the gathered vector (40 MB)
lives in the Infinity Cache and every 4-byte gather costs a 128-byte line fill -- the plain CSR kernel
moves 15x its algorithmic bytes.  Gathers run at the streaming rate only when (a) the gathered window is
L2 resident and (b) consecutive lanes share cache lines.  So:

* columns are cut into panels of ``W = 2**lw`` entries (512 KB of f32 for lw = 17), rows into blocks of
  ``RB = 2**lrb`` rows (one workgroup each); a *tile* is (row block, panel);
* inside a tile the non-zeros are stored sorted by COLUMN, so a wave's 64 gathers touch a handful of
  lines of an L2-resident panel; all workgroups walk the panels in the same order at the same pace;
* each item carries ``slot``, its rank in ROW order inside the tile: the product is written to
  ``lds[slot]`` (a transposition through LDS, plain stores, no atomics), after which every thread reduces
  the segments of its 16 rows; segment lengths come from one byte per (tile, row).

Item = 4-byte value + 4-byte ``(slot << lw) | (col - panel*W)``.  Tiles are padded to multiples of 4 items
(zero value, unused slot) so the kernel streams them with 16-byte loads.  Bytes per non-zero: 8 + RB*P/nnz_per_block
(the count bytes): 8.8 for 100 non-zeros per row at 10M columns.

A matrix is eligible when every tile holds at most ``cap`` items and no (tile, row) more than 255;
otherwise ``build_tiles`` returns None and the CSR kernel is used.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

LW_DEFAULT = 17      # panel = 131072 columns
LRB = 13             # 8192 rows per block = 512 threads x 16 rows (fixed by the kernel instantiation)
CAP = 12288          # items per tile the kernel's LDS product buffer holds


@dataclass
class Tiles:
    lw: int
    lrb: int
    cap: int
    nblk: int
    npanel: int
    nrows: int
    ncols: int
    idx: torch.Tensor        # int32 [items]  (slot << lw) | local column
    val: torch.Tensor        # dtype [items]
    tile_ptr: torch.Tensor   # int32 [nblk*npanel + 1], item offsets, multiples of 4
    cnt: torch.Tensor        # uint8 [nblk*npanel*RB], non-zeros of each row of each tile

    @property
    def items(self) -> int:
        return int(self.idx.numel())

    def bytes(self) -> int:
        return sum(int(t.numel()) * t.element_size() for t in (self.idx, self.val, self.tile_ptr, self.cnt))


def _wrap_i32(v: torch.Tensor) -> torch.Tensor:
    """int64 values in [0, 2^32) -> the int32 with the same bit pattern"""
    return torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32)


def build_tiles(rowptr: torch.Tensor, colidx: torch.Tensor, val: torch.Tensor, nrows: int, ncols: int,
                lw: int = LW_DEFAULT, lrb: int = LRB, cap: int = CAP, max_chunk_nnz: int = 1 << 26) -> Optional[Tiles]:
    """CSR (any row lengths, columns sorted or not) -> Tiles, or None when not eligible.  Runs on the
    tensors' device with torch sorts (setup cost, done once per matrix)."""
    dev = val.device
    RB, W = 1 << lrb, 1 << lw
    P = max(1, (ncols + W - 1) // W)
    NB = max(1, (nrows + RB - 1) // RB)
    if cap + 4 > (1 << (32 - lw)):
        raise ValueError("cap does not fit the slot field")
    nnz = int(colidx.numel())
    if nnz + 4 * NB * P >= 2 ** 31:
        return None
    rp = rowptr.long()
    row_counts = rp[1:] - rp[:-1]
    out_idx = torch.zeros(nnz + 4 * NB * P, dtype=torch.int32, device=dev)
    out_val = torch.zeros(nnz + 4 * NB * P, dtype=val.dtype, device=dev)
    tile_ptr = torch.zeros(NB * P + 1, dtype=torch.int64, device=dev)
    cnt = torch.zeros(NB * P * RB, dtype=torch.uint8, device=dev)
    base = 0
    # chunks of whole row blocks with about max_chunk_nnz non-zeros
    blk_nnz = rp[torch.clamp(torch.arange(NB + 1, device=dev) * RB, max=nrows)].cpu().tolist()
    b_lo = 0
    while b_lo < NB:
        b_hi = b_lo + 1
        while b_hi < NB and blk_nnz[b_hi + 1] - blk_nnz[b_lo] <= max_chunk_nnz:
            b_hi += 1
        r_lo, r_hi = b_lo * RB, min(b_hi * RB, nrows)
        a, b = blk_nnz[b_lo], blk_nnz[b_hi]
        n = b - a
        ntl = (b_hi - b_lo) * P
        t0 = b_lo * P
        if n > 0:
            cols = colidx[a:b].long()
            v = val[a:b]
            rloc = torch.repeat_interleave(torch.arange(r_hi - r_lo, device=dev), row_counts[r_lo:r_hi])
            tile = (rloc >> lrb) * P + (cols >> lw)
            tsz = torch.bincount(tile, minlength=ntl)
            if int(tsz.max()) > cap:
                return None
            c_tr = torch.bincount(tile * RB + (rloc & (RB - 1)), minlength=ntl * RB)
            if int(c_tr.max()) > 255:
                return None
            cnt[t0 * RB:(t0 + ntl) * RB] = c_tr.to(torch.uint8)
            del c_tr
            tstart = torch.cumsum(tsz, 0) - tsz
            ar = torch.arange(n, device=dev)
            # slot = rank inside the tile in (row, column) order = CSR order restricted to the tile
            order1 = torch.argsort(tile, stable=True)
            slot = torch.empty(n, dtype=torch.int64, device=dev)
            slot[order1] = ar - tstart[tile[order1]]
            del order1
            # storage order: by column inside the tile
            lcol = cols & (W - 1)
            order2 = torch.argsort(tile * W + lcol, stable=True)
            tile_s = tile[order2]
            packed = (slot[order2] << lw) | lcol[order2]
            tsz_pad = (tsz + 3) // 4 * 4
            pstart = torch.cumsum(tsz_pad, 0) - tsz_pad
            total = int(tsz_pad.sum())
            seg_i = out_idx[base:base + total]
            seg_v = out_val[base:base + total]
            # padding items: value 0, slot = first unused slot of the tile, local column 0
            seg_i.copy_(_wrap_i32(torch.repeat_interleave(tsz << lw, tsz_pad)))
            dest = pstart[tile_s] + (ar - tstart[tile_s])
            seg_i[dest] = _wrap_i32(packed)
            seg_v[dest] = v[order2]
            tile_ptr[t0:t0 + ntl] = base + pstart
            base += total
        else:
            tile_ptr[t0:t0 + ntl] = base
        b_lo = b_hi
    tile_ptr[NB * P] = base
    return Tiles(lw, lrb, cap, NB, P, nrows, ncols, out_idx[:base].contiguous(), out_val[:base].contiguous(),
                 tile_ptr.to(torch.int32), cnt)


def emulate_spmv(t: Tiles, x: torch.Tensor) -> torch.Tensor:
    """What the kernel computes, step by step, in torch on any device (tests only: pass 1 writes the
    products to their row-order slots, pass 2 sums each row's segment)."""
    RB, W = 1 << t.lrb, 1 << t.lw
    y = torch.zeros(t.nblk * RB, dtype=torch.float64, device=x.device)
    idx = t.idx.long() & 0xFFFFFFFF
    tp = t.tile_ptr.long().tolist()
    for b in range(t.nblk):
        for p in range(t.npanel):
            tile = b * t.npanel + p
            i0, i1 = tp[tile], tp[tile + 1]
            if i1 == i0:
                continue
            pk = idx[i0:i1]
            prod = torch.zeros(t.cap + 8, dtype=torch.float64, device=x.device)
            xin = x[p * W:(p + 1) * W].double()
            lcol = pk & (W - 1)
            prod[pk >> t.lw] = t.val[i0:i1].double() * xin[lcol]
            c = t.cnt[tile * RB:(tile + 1) * RB].long()
            ends = torch.cumsum(c, 0)
            csum = torch.cat([torch.zeros(1, dtype=torch.float64, device=x.device), torch.cumsum(prod[:int(ends[-1])], 0)])
            y[b * RB:(b + 1) * RB] += csum[ends] - csum[ends - c]
    return y[:t.nrows]
