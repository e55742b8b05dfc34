"""Panel-tiled storage of a CSR matrix for the fast SpMV kernel (``k_tiled_fused`` in csrc/pdlp_hip.hip).

Why (profiles/r01_microbench_gather.txt): at 10M columns the gathered vector (40 MB)
lives in the Infinity Cache and every 4-byte gather costs a 128-byte line fill -- the plain CSR kernel
moves 15x its algorithmic bytes.  Gathers run at the streaming rate only when (a) the gathered window is
L2 resident and (b) consecutive lanes share cache lines.  So:

* columns are cut into panels of ``W = 2**lw`` entries (256 KB of f32 for lw = 16), rows into blocks of
  ``RB = 512*rpt`` rows (one 512-thread workgroup each, ``rpt`` rows per thread); a *tile* is (row block, panel);
* inside a tile the non-zeros are stored sorted by COLUMN, so a wave's 64 gathers touch a handful of
  lines of an L2-resident panel; all workgroups walk the panels in the same order at the same pace;
* each item carries ``slot``, its rank in ROW order inside the tile: the product is written to
  ``lds[slot]`` (a transposition through LDS, plain stores, no atomics), after which every thread reduces
  the segments of its ``rpt`` rows; segment lengths come from 4 bits per (tile, row).  Lane ``l`` of wave ``w``
  owns rows ``w*64*rpt + i*64 + l`` (``i < rpt``) of the row block: for every ``i`` a wave sits on 64 consecutive
  rows, whose segments are one contiguous run of LDS words (conflict-free reads), and a row's first slot is a
  wave scan of the counts away (done four ``i`` at a time in 8-bit fields).

Item = 4-byte value + 4-byte ``(slot << lw) | (col - panel*W)``.  Tiles are padded to multiples of 256 items
(zero value, unused slot) and streamed with 16-byte loads.  Gather cost on gfx950 grows with the number of
distinct 128-byte lines one wave instruction touches (about 3.4 clocks per line, measured), so inside every
group of 256 column-sorted items the storage order is interleaved: position ``4*lane + j`` holds sorted item
``64*j + lane``.  A lane's 16-byte load then yields items j = 0..3 of four different 64-item runs and gather
instruction j covers 64 CONSECUTIVE sorted items (about 25 lines at 100 non-zeros per row) instead of 64 items
spaced four apart (64 lines).  Bytes per non-zero: 8 + 20*512*P/nnz_per_block (the count nibbles): about 8.9 for 100 non-zeros per row at
10M columns.

A matrix is eligible when every tile holds at most ``cap`` items (16384 in float32, 8192 in float64), no
(tile, row) more than 15 and no 64 consecutive rows more than 255 items of one tile; otherwise ``build_tiles`` returns None and the CSR kernel is used.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

LW_DEFAULT = 16      # panel = 65536 columns (256 KB of f32: L2 resident on every XCD)
NT = 512             # threads per workgroup (fixed by the kernel)
RPT_MAX = 40         # float32: most rows per thread; 32-bit words of 4-bit counts per (tile, thread) = RPT_MAX/8
CAP = 16384          # float32: items per tile (64 KB of LDS products per workgroup)
RPT_MAX_F64 = 24     # float64: the same LDS and register budget holds half as many
CAP_F64 = 8192
GROUP = 256          # tiles are padded to whole groups of 4 x 64 items (interleaved, see above)
NCU = 512            # two workgroups per CU at a time: row blocks are sized to fill whole rounds


@dataclass
class Tiles:
    lw: int
    rpt: int                 # rows per thread: a row block is 512*rpt rows
    cap: int
    nblk: int
    npanel: int
    nrows: int
    ncols: int
    idx: torch.Tensor        # int32 [items]  (slot << lw) | local column
    val: torch.Tensor        # float32 [items]
    tile_ptr: torch.Tensor   # int32 [nblk*npanel + 1], item offsets, multiples of 256
    cnt: torch.Tensor        # int32 [nblk*npanel*512*cw]: 8*cw nibbles per (tile, thread); cw = 5 (f32) or 3 (f64)
    groups: int = 1          # workgroups sharing a row block (each walks ceil(npanel/groups) panels)
    rpt_max: int = RPT_MAX   # rows per thread the kernel instantiation for this precision supports

    @property
    def cw(self) -> int:
        return self.rpt_max // 8

    @property
    def items(self) -> int:
        return int(self.idx.numel())

    @property
    def rows_per_block(self) -> int:
        return NT * self.rpt

    def bytes(self) -> int:
        return sum(int(t.numel()) * t.element_size() for t in (self.idx, self.val, self.tile_ptr, self.cnt))


def rowsum_groups(rows: int) -> int:
    """panel groups the library's row-sum scratch has room for on a handle whose longer local side has ``rows`` rows
    (``rowsum_groups`` in csrc/pdlp_hip.hip; ``pdlp_tile_limits`` reports the same number for a live handle): splitting a
    row block's panels over several workgroups only pays while one workgroup per row block cannot fill 2 x 256 CUs"""
    if rows <= NT * RPT_MAX * 128:
        return 16
    return 8 if rows <= NT * RPT_MAX * 512 else 1


def _wrap_i32(v: torch.Tensor) -> torch.Tensor:
    """int64 values in [0, 2^32) -> the int32 with the same bit pattern"""
    return torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32)


def limits(dtype):
    """(most rows per thread, most items per tile) of the kernel instantiation for ``dtype``"""
    return (RPT_MAX, CAP) if dtype == torch.float32 else (RPT_MAX_F64, CAP_F64)


def choose_shape(nrows: int, nnz: int, ncols: int, lw: int, cap: int = CAP, slots: int = NCU, max_groups: int = 8,
                 rpt_max: int = RPT_MAX):
    """(rows per thread, panel groups).  Denser tiles mean fewer cache lines per gather, so take as many rows per
    workgroup as the tile capacity allows; when that leaves too few row blocks to fill the chip (few rows, e.g. one
    rank's shard), let several workgroups share a row block by splitting its panels into groups."""
    W = 1 << lw
    P = max(1, (ncols + W - 1) // W)
    per_row_panel = max(nnz / max(nrows, 1) * min(W, ncols) / max(ncols, 1), 1e-9)   # mean items of a row in a panel
    best, best_score = (1, 1), -1.0
    for rpt in range(1, rpt_max + 1):
        rb = NT * rpt
        mean_tile = rb * per_row_panel
        if mean_tile + 6.0 * mean_tile ** 0.5 > cap and rpt > 1:      # keep 6 sigma below the LDS capacity
            break
        nblk = (nrows + rb - 1) // rb
        density = min(1.0, 0.35 + 0.65 * rpt / rpt_max)
        for groups in range(1, min(max_groups, P) + 1):
            if normalize_groups(groups, P, max_groups) != groups:
                continue
            blocks = nblk * groups
            rounds = (blocks + slots - 1) // slots
            eff = blocks / (rounds * slots)                           # fill of the last round of workgroups
            score = eff * density * (1.0 if groups == 1 else 0.97)   # split tiles pay a small epilogue kernel
            if score > best_score:
                best, best_score = (rpt, groups), score
    return best


def choose_lw(nrows: int, nnz: int, ncols: int) -> int:
    """log2 of the panel width: 64K columns unless the rows are so long that 64 consecutive rows would hold more
    than the 255 items of a tile the kernel's 8-bit scan fields allow -- then narrower panels (down to 4K)"""
    per_row = nnz / max(nrows, 1)
    for lw in range(LW_DEFAULT, 11, -1):
        W = 1 << lw
        if per_row * min(W, ncols) / max(ncols, 1) <= 3.0:          # mean 192 per 64 rows: 4.5 sigma below 255
            return lw
    return 12


def normalize_groups(groups: int, npanel: int, max_groups: int = 8) -> int:
    """the kernel gives every group ceil(npanel/groups) panels: shrink the count until no group is empty"""
    g = max(1, min(int(groups), npanel, max_groups))
    ppg = -(-npanel // g)
    return -(-npanel // ppg)


def choose_rpt(nrows: int, nnz: int, ncols: int, lw: int, cap: int = CAP, ncu: int = NCU) -> int:
    return choose_shape(nrows, nnz, ncols, lw, cap, ncu)[0]


def build_tiles(rowptr: torch.Tensor, colidx: torch.Tensor, val: torch.Tensor, nrows: int, ncols: int,
                lw: Optional[int] = None, rpt: Optional[int] = None, cap: Optional[int] = None, max_chunk_nnz: int = 1 << 26,
                groups: Optional[int] = None, max_groups: int = 8) -> Optional[Tiles]:
    """CSR (any row lengths, columns sorted or not) -> Tiles, or None when not eligible.  Runs on the
    tensors' device with torch sorts (setup cost, done once per matrix)."""
    dev = val.device
    if val.dtype not in (torch.float32, torch.float64):
        return None
    rpt_max, cap_max = limits(val.dtype)
    cap = cap_max if cap is None else cap
    CW = rpt_max // 8
    nnz = int(colidx.numel())
    if lw is None:
        lw = choose_lw(nrows, nnz, ncols)
    W = 1 << lw
    if rpt is None:
        rpt, g_auto = choose_shape(nrows, nnz, ncols, lw, cap, rpt_max=rpt_max, max_groups=max(1, min(8, int(max_groups))))
        groups = g_auto if groups is None else groups
    groups = 1 if groups is None else int(groups)
    if not 1 <= rpt <= rpt_max:
        raise ValueError(f"rpt must be in 1..{rpt_max}")
    RB = NT * rpt
    P = max(1, (ncols + W - 1) // W)
    NB = max(1, (nrows + RB - 1) // RB)
    if cap + 4 > (1 << (32 - lw)) or cap > cap_max:
        raise ValueError("cap does not fit")
    if nnz + GROUP * NB * P >= 2 ** 31:
        return None
    rp = rowptr.long()
    row_counts = rp[1:] - rp[:-1]
    out_idx = torch.zeros(nnz + GROUP * NB * P, dtype=torch.int32, device=dev)
    out_val = torch.zeros(nnz + GROUP * NB * P, dtype=val.dtype, device=dev)
    tile_ptr = torch.zeros(NB * P + 1, dtype=torch.int64, device=dev)
    cnt = torch.zeros(NB * P * NT * CW, dtype=torch.int32, device=dev)
    shifts = (torch.arange(8, device=dev) * 4).view(1, 1, 1, 8)
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
            rblk = rloc // RB
            tile = rblk * P + (cols >> lw)
            tsz = torch.bincount(tile, minlength=ntl)
            if int(tsz.max()) > cap:
                return None
            # items of every (tile, row): 4 bits each.  Lane l of wave w owns rows w*64*rpt + i*64 + l (i < rpt) of
            # the row block -- for every i the 64 lanes of a wave sit on 64 consecutive rows -- and the kernel
            # scans the lanes' counts in 8-bit fields, so 64 consecutive rows may hold at most 255 items of a tile
            c_tr = torch.bincount(tile * RB + (rloc - rblk * RB), minlength=ntl * RB)
            if int(c_tr.max()) > 15:
                return None
            c_w = c_tr.view(ntl, NT // 64, rpt, 64)
            if int(c_w.sum(-1).max()) > 255:
                return None
            nib = torch.zeros(ntl, NT, rpt_max, dtype=torch.int64, device=dev)     # CW words of 8 nibbles per thread
            nib[:, :, :rpt] = c_w.permute(0, 1, 3, 2).reshape(ntl, NT, rpt)
            del c_tr, c_w
            words = (nib.view(ntl, NT, CW, 8) << shifts).sum(-1)
            cnt[t0 * NT * CW:(t0 + ntl) * NT * CW] = _wrap_i32(words.reshape(-1))
            del nib, words
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
            tsz_pad = (tsz + GROUP - 1) // GROUP * GROUP
            pstart = torch.cumsum(tsz_pad, 0) - tsz_pad
            total = int(tsz_pad.sum())
            seg_i = out_idx[base:base + total]
            seg_v = out_val[base:base + total]
            # padding items: value 0, slot = first unused slot of the tile, local column 0
            seg_i.copy_(_wrap_i32(torch.repeat_interleave(tsz << lw, tsz_pad)))
            dest = pstart[tile_s] + (ar - tstart[tile_s])
            seg_i[dest] = _wrap_i32(packed)
            seg_v[dest] = v[order2]
            # interleave every 256-item group: physical 4*lane + j  <-  sorted 64*j + lane
            seg_i.copy_(seg_i.view(-1, 4, 64).transpose(1, 2).reshape(-1))
            seg_v.copy_(seg_v.view(-1, 4, 64).transpose(1, 2).reshape(-1))
            tile_ptr[t0:t0 + ntl] = base + pstart
            base += total
        else:
            tile_ptr[t0:t0 + ntl] = base
        b_lo = b_hi
    tile_ptr[NB * P] = base
    return Tiles(lw, rpt, cap, NB, P, nrows, ncols, out_idx[:base].contiguous(), out_val[:base].contiguous(),
                 tile_ptr.to(torch.int32), cnt, normalize_groups(groups, P, max(1, int(max_groups))), rpt_max)


def tile_row_counts(t: Tiles, tile: int) -> torch.Tensor:
    """items of each of the tile's 1024*rpt rows (unpacks the nibbles)"""
    CW = t.cw
    w = t.cnt[tile * NT * CW:(tile + 1) * NT * CW].long() & 0xFFFFFFFF
    shifts = (torch.arange(8, device=w.device) * 4).view(1, 1, 8)
    nib = (w.view(NT, CW, 1) >> shifts) & 15
    per_thread = nib.reshape(NT // 64, 64, t.rpt_max)[:, :, :t.rpt]           # [wave, lane, i]
    return per_thread.permute(0, 2, 1).reshape(-1)                          # row = wave*64*rpt + i*64 + lane


def emulate_spmv(t: Tiles, x: torch.Tensor) -> torch.Tensor:
    """What the kernel computes, step by step, in torch on any device (tests only: pass 1 writes the
    products to their row-order slots, pass 2 sums each row's segment)."""
    RB, W = t.rows_per_block, 1 << t.lw
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
            c = tile_row_counts(t, tile)
            ends = torch.cumsum(c, 0)
            csum = torch.cat([torch.zeros(1, dtype=torch.float64, device=x.device), torch.cumsum(prod[:int(ends[-1])], 0)])
            y[b * RB:(b + 1) * RB] += csum[ends] - csum[ends - c]
    return y[:t.nrows]
