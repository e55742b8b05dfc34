"""Panel-tiled storage of a CSR matrix for the fast SpMV kernel (``k_tiled_fused`` in csrc/pdlp_kernel_tiled.inc).

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
spaced four apart (64 lines).  Bytes per non-zero: 8 + 20*512*P/nnz_per_block (the count nibbles): about 8.9 for
100 non-zeros per row at 10M columns.

Limits of the tiles proper: at most ``cap`` items per tile (16384 in float32, 8192 in float64), at most 15 items of one row
in one tile (4-bit counts) and at most 255 items of 64 consecutive rows in one tile (the kernel scans the lanes' counts in 8-bit
fields).  What exceeds them -- dense rows or columns of real LPs, local clusters -- goes to the *remainder* (round 2): a
compact CSR over the rows that have such items, cut into segments of at most 512 items; ``k_rem_segments`` / ``k_rem_rows`` add
them up in fixed order into a dense vector (zero for all other rows) that the tiled kernel's epilogue adds to the row sum.  It
is empty for well-spread matrices such as the bench LP, which then run exactly the round-1 kernel.

``build_tiles`` returns None only when tiling makes no sense: more than ``max_rest`` (30 %) of the items would leave the tiles
-- the matrix is clustered (banded, block structured) and the CSR kernel, whose gathers are then cache friendly anyway, is the
better kernel.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Optional

import torch

LW_DEFAULT = 16      # panel = 65536 columns (256 KB of f32: L2 resident on every XCD)
NT = 512             # threads per workgroup of the standard kernel (``Tiles.nt``: what a tile set was built for)
RPT_MAX = 40         # float32: most rows per thread; 32-bit words of 4-bit counts per (tile, thread) = RPT_MAX/8
CAP = 16384          # float32: items per tile (64 KB of LDS products per workgroup)
RPT_MAX_F64 = 24     # float64: the same LDS and register budget holds half as many
CAP_F64 = 8192
GROUP = 256          # tiles are padded to whole groups of 4 x 64 items (interleaved, see above)
NCU = 512            # two workgroups of 512 threads per CU at a time: row blocks are sized to fill whole rounds
LMAX = 15            # items of one row in one tile (4-bit counts)
GMAX = 255           # items of 64 consecutive rows (one i of one wave) in one tile (8-bit scan fields)
SEG = 512            # items per remainder segment (eight lanes: 64 per lane, in batches of eight)


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
    val: torch.Tensor        # float32 / float64 [items]
    tile_ptr: torch.Tensor   # int64 [nblk*npanel + 1], item offsets, multiples of 256 (the library takes them per row block: abi_tile_ptr)
    cnt: torch.Tensor        # int32: 8*cw nibbles per (tile, thread), cw = 5 (f32) or 3 (f64), laid out for coalesced loads:
                             #   [tile][thread][4] (words 0..3, zero padded) then [tile][cw - 4][thread] (the other words)
    groups: int = 1          # workgroups sharing a row block (each walks ceil(npanel/groups) panels)
    rpt_max: int = RPT_MAX   # rows per thread the kernel instantiation for this precision supports
    nt: int = NT             # threads per workgroup of the kernel that reads these tiles (a row block is nt*rpt rows)
    # remainder: segments of <= SEG items of the rows listed in rem_rows
    rem_rows: Optional[torch.Tensor] = None  # int32 [nr]   rows (local to this matrix) with a remainder, ascending
    rem_rptr: Optional[torch.Tensor] = None  # int32 [nr+1] their segment ranges
    rem_sptr: Optional[torch.Tensor] = None  # int32 [ns+1] item ranges of the segments
    rem_col: Optional[torch.Tensor] = None   # int32 [nrem]
    rem_val: Optional[torch.Tensor] = None   # [nrem]
    stats: dict = field(default_factory=dict)

    @property
    def cw(self) -> int:
        return self.rpt_max // 8

    @property
    def items(self) -> int:
        return int(self.idx.numel())

    @property
    def rows_per_block(self) -> int:
        return self.nt * self.rpt

    @property
    def nrem(self) -> int:
        return 0 if self.rem_col is None else int(self.rem_col.numel())

    def abi_tile_ptr(self):
        """(int32 [nblk][npanel + 1] offsets relative to each row block's first item, int64 [nblk] those first items): what
        ``pdlp_attach_tiles`` takes.  A row block never holds 2^31 items, a matrix copy may: the kernel adds the block's 64-bit base to
        its pointers once and indexes with 32 bits."""
        if getattr(self, "_abi_tp", None) is None:
            tp = self.tile_ptr.view(-1)
            nb, P = self.nblk, self.npanel
            base = tp[:nb * P:P].contiguous()                                   # first item of every row block
            body = tp[:nb * P].view(nb, P)
            ends = torch.cat([tp[P:nb * P:P], tp[nb * P:nb * P + 1]]).view(nb, 1)   # end of a block = start of the next (the last: total)
            rel = torch.cat([body, ends], dim=1) - base.view(nb, 1)
            if int(rel.max()) >= 2 ** 31:
                raise ValueError("a single row block holds 2^31 items or more")
            self._abi_tp = (rel.to(torch.int32).contiguous(), base.to(torch.int64).contiguous())
        return self._abi_tp

    def bytes(self) -> int:
        ts = (self.idx, self.val, self.tile_ptr, self.cnt, self.rem_rows, self.rem_rptr, self.rem_sptr, self.rem_col, self.rem_val)
        return sum(int(t.numel()) * t.element_size() for t in ts if t is not None)


def rowsum_groups(rows: int) -> int:
    """panel groups the library's row-sum scratch has room for on a handle whose longer local side has ``rows`` rows
    (``rowsum_groups`` in csrc/pdlp_hip.hip; ``pdlp_tile_limits`` reports the same number for a live handle): splitting a
    row block's panels over several workgroups only pays while one workgroup per row block cannot fill 2 x 256 CUs"""
    if rows <= 512 * 40 * 128:
        return 32               # (small shards: the local panels' groups plus those of every chunk of a chunked exchange)
    return 8 if rows <= 512 * 40 * 512 else 1


def _wrap_i32(v: torch.Tensor) -> torch.Tensor:
    """int64 values in [0, 2^32) -> the int32 with the same bit pattern"""
    return torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32)


def limits(dtype):
    """(most rows per thread, most items per tile) of the kernel instantiation for ``dtype``"""
    return (RPT_MAX, CAP) if dtype == torch.float32 else (RPT_MAX_F64, CAP_F64)


def choose_shape(nrows: int, nnz: int, ncols: int, lw: int, cap: int = CAP, slots: Optional[int] = None, max_groups: int = 8,
                 rpt_max: int = RPT_MAX, nt: int = NT):
    """(rows per thread, panel groups).  Denser tiles mean fewer cache lines per gather, so take as many rows per
    workgroup as the tile capacity allows; when that leaves too few row blocks to fill the chip (few rows, e.g. one
    rank's shard), let several workgroups share a row block by splitting its panels into groups."""
    W = 1 << lw
    slots = NCU * NT // nt if slots is None else slots               # workgroups resident at a time (1024 threads per CU)
    P = max(1, (ncols + W - 1) // W)
    per_row_panel = max(nnz / max(nrows, 1) * min(W, ncols) / max(ncols, 1), 1e-9)   # mean items of a row in a panel
    best, best_score = (1, 1), -1.0
    for rpt in range(1, rpt_max + 1):
        rb = nt * rpt
        mean_tile = rb * min(per_row_panel, float(LMAX))
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


def _rank_in_group(key: torch.Tensor, ngroups: int):
    """for items with integer ``key`` (any order): (rank of the item among the items of its key in input order, counts per key)"""
    n = key.numel()
    counts = torch.bincount(key, minlength=ngroups)
    order = torch.argsort(key, stable=True)
    start = torch.cumsum(counts, 0) - counts
    rank = torch.empty(n, dtype=torch.int64, device=key.device)
    rank[order] = torch.arange(n, device=key.device) - start[key[order]]
    return rank, counts


def _pack(tile: torch.Tensor, row_in_block: torch.Tensor, sortcol: torch.Tensor, low: torch.Tensor, shift: int, val: torch.Tensor,
          ntl: int, RB: int, rpt: int, rpt_max: int, extra: Optional[torch.Tensor] = None, NT: int = NT):
    """Lay the items of ``ntl`` tiles out the way the kernel reads them.  Items arrive in row order inside every tile (CSR
    order); ``sortcol`` orders a tile's items for the gathers; the stored word is ``(slot << shift) | low``.
    Returns (idx int32, val, extra int32 or None, padded tile sizes, count words int32 [ntl*512*CW])."""
    dev = val.device
    n = tile.numel()
    CW = rpt_max // 8
    tsz = torch.bincount(tile, minlength=ntl)
    tsz_pad = (tsz + GROUP - 1) // GROUP * GROUP
    total = int(tsz_pad.sum())
    out_idx = torch.zeros(total, dtype=torch.int32, device=dev)
    out_val = torch.zeros(total, dtype=val.dtype, device=dev)
    out_extra = None if extra is None else torch.zeros(total, dtype=torch.int32, device=dev)
    cnt = torch.zeros(ntl * NT * CW, dtype=torch.int32, device=dev)
    if n == 0:
        return out_idx, out_val, out_extra, tsz_pad, cnt
    # items of every (tile, row): 4 bits each.  Lane l of wave w owns rows w*64*rpt + i*64 + l (i < rpt) of the row block
    c_tr = torch.bincount(tile * RB + row_in_block, minlength=ntl * RB)
    assert int(c_tr.max()) <= LMAX
    c_w = c_tr.view(ntl, NT // 64, rpt, 64)
    assert int(c_w.sum(-1).max()) <= GMAX
    nib = torch.zeros(ntl, NT, rpt_max, dtype=torch.int64, device=dev)
    nib[:, :, :rpt] = c_w.permute(0, 1, 3, 2).reshape(ntl, NT, rpt)
    shifts = (torch.arange(8, device=dev) * 4).view(1, 1, 1, 8)
    cnt.copy_(_wrap_i32((nib.view(ntl, NT, CW, 8) << shifts).sum(-1).reshape(-1)))
    del nib, c_tr, c_w
    tstart = torch.cumsum(tsz, 0) - tsz
    ar = torch.arange(n, device=dev)
    # slot = rank inside the tile in row order = input order restricted to the tile
    order1 = torch.argsort(tile, stable=True)
    slot = torch.empty(n, dtype=torch.int64, device=dev)
    slot[order1] = ar - tstart[tile[order1]]
    del order1
    # storage order: by column inside the tile
    order2 = torch.argsort(tile * (int(sortcol.max()) + 1) + sortcol, stable=True)
    tile_s = tile[order2]
    packed = (slot[order2] << shift) | low[order2]
    pstart = torch.cumsum(tsz_pad, 0) - tsz_pad
    # padding items: value 0, slot = first unused slot of the tile, low bits 0
    out_idx.copy_(_wrap_i32(torch.repeat_interleave(tsz << shift, tsz_pad)))
    dest = pstart[tile_s] + (ar - tstart[tile_s])
    out_idx[dest] = _wrap_i32(packed)
    out_val[dest] = val[order2]
    # interleave every 256-item group: physical 4*lane + j  <-  sorted 64*j + lane
    out_idx.copy_(out_idx.view(-1, 4, 64).transpose(1, 2).reshape(-1))
    out_val.copy_(out_val.view(-1, 4, 64).transpose(1, 2).reshape(-1))
    if extra is not None:
        out_extra[dest] = extra[order2].to(torch.int32)
        out_extra.copy_(out_extra.view(-1, 4, 64).transpose(1, 2).reshape(-1))
    return out_idx, out_val, out_extra, tsz_pad, cnt


def build_tiles(rowptr: torch.Tensor, colidx: torch.Tensor, val: torch.Tensor, nrows: int, ncols: int,
                lw: Optional[int] = None, rpt: Optional[int] = None, cap: Optional[int] = None, max_chunk_nnz: int = 1 << 26,
                groups: Optional[int] = None, max_groups: int = 8, max_rest: float = 0.30,
                kernel_limits: Optional[tuple] = None, nt: Optional[int] = None) -> Optional[Tiles]:
    """CSR (any row lengths, columns sorted or not) -> Tiles, or None when more than ``max_rest`` of the items would not
    fit the tiles proper (a clustered matrix: the CSR kernel is the better one).  Runs on the tensors' device with torch
    sorts (setup cost, done once per matrix)."""
    dev = val.device
    if val.dtype not in (torch.float32, torch.float64):
        return None
    # (rows per thread, items per tile) of the kernel that will read the tiles: the library's build says (pdlp_tile_limits)
    rpt_max, cap_max = limits(val.dtype) if kernel_limits is None else (int(kernel_limits[0]), int(kernel_limits[1]))
    cap = cap_max if cap is None else cap
    NT = int(nt) if nt is not None else (int(kernel_limits[2]) if kernel_limits is not None and len(kernel_limits) > 2 else globals()["NT"])
    CW = rpt_max // 8
    nnz = int(colidx.numel())
    if lw is None:
        lw = choose_lw(nrows, nnz, ncols)
    W = 1 << lw
    if rpt is None:
        rpt, g_auto = choose_shape(nrows, nnz, ncols, lw, cap, rpt_max=rpt_max, max_groups=max(1, min(8, int(max_groups))), nt=NT)
        groups = g_auto if groups is None else groups
    groups = 1 if groups is None else int(groups)
    if not 1 <= rpt <= rpt_max:
        raise ValueError(f"rpt must be in 1..{rpt_max}")
    RB = NT * rpt
    P = max(1, (ncols + W - 1) // W)
    NB = max(1, (nrows + RB - 1) // RB)
    if cap + 4 > (1 << (32 - lw)) or cap > cap_max:
        raise ValueError("cap does not fit")
    # (item positions are 64-bit in the library: no limit on the items of one matrix copy; a remainder above 2^31 - 1 items -- its
    #  ranges are 32-bit -- cannot happen: max_rest stops a matrix that clustered long before)
    rp = rowptr.long()
    row_counts = rp[1:] - rp[:-1]
    parts_idx, parts_val, parts_cnt = [], [], []
    tile_ptr = torch.zeros(NB * P + 1, dtype=torch.int64, device=dev)
    over_row, over_col, over_val = [], [], []          # what the tiles proper do not hold (row, column, value)
    n_over = 0
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
            rib = rloc - rblk * RB                                   # row inside its block
            tile = rblk * P + (cols >> lw)
            # a (tile, row) pair keeps its first `limit` items, limit <= LMAX, lowered where 64 consecutive rows (one i of one
            # wave) would hold more than GMAX items of a tile or a tile more than `cap`; the rest goes to the remainder
            key = tile * RB + rib
            c_tr = torch.bincount(key, minlength=ntl * RB)
            keep = None
            if int(c_tr.max()) > LMAX or int(c_tr.view(ntl, NT // 64, rpt, 64).sum(-1).max()) > GMAX or \
                    int(torch.bincount(tile, minlength=ntl).max()) > cap:
                rank, _ = _rank_in_group(key, ntl * RB)
                lim_g = torch.full((ntl, NT // 64, rpt), LMAX, dtype=torch.int64, device=dev)       # per (tile, wave, i)
                c4 = c_tr.view(ntl, NT // 64, rpt, 64)
                for L in range(LMAX, -1, -1):
                    over = torch.clamp(c4, max=L).sum(-1) > GMAX
                    lim_g = torch.where(over & (lim_g >= L), max(L - 1, 0), lim_g)
                    if not bool(over.any()):
                        break
                lim_t = torch.full((ntl,), LMAX, dtype=torch.int64, device=dev)                     # per tile
                for _ in range(LMAX + 1):
                    eff = torch.minimum(lim_g, lim_t.view(ntl, 1, 1))
                    tsz = torch.minimum(c4, eff.unsqueeze(-1)).sum((1, 2, 3))
                    big = tsz > cap
                    if not bool(big.any()):
                        break
                    lim_t = torch.where(big, lim_t - 1, lim_t)
                eff = torch.minimum(lim_g, lim_t.view(ntl, 1, 1)).unsqueeze(-1).expand(ntl, NT // 64, rpt, 64).reshape(-1)
                keep = rank < eff[key]
                del lim_g, lim_t, eff, c4, rank
            del c_tr, key
            nd = 0 if keep is None else int((~keep).sum())
            if nd:
                drop = ~keep
                n_over += nd
                if n_over > max_rest * nnz or n_over >= 2 ** 31:
                    return None
                over_row.append(rloc[drop] + r_lo)
                over_col.append(cols[drop])
                over_val.append(v[drop])
                tile, rib, cols, v = tile[keep], rib[keep], cols[keep], v[keep]
                del drop
            del keep, rloc, rblk
            lcol = cols & (W - 1)
            p_idx, p_val, _, tsz_pad, p_cnt = _pack(tile, rib, lcol, lcol, lw, v, ntl, RB, rpt, rpt_max, NT=NT)
            parts_idx.append(p_idx)
            parts_val.append(p_val)
            parts_cnt.append(p_cnt)
            tile_ptr[t0:t0 + ntl] = base + torch.cumsum(tsz_pad, 0) - tsz_pad
            base += int(tsz_pad.sum())
        else:
            parts_cnt.append(torch.zeros(ntl * NT * CW, dtype=torch.int32, device=dev))
            tile_ptr[t0:t0 + ntl] = base
        b_lo = b_hi
    tile_ptr[NB * P] = base
    cat = lambda parts, dt: (parts[0] if len(parts) == 1 else torch.cat(parts)) if parts else torch.zeros(0, dtype=dt, device=dev)
    idx_all, val_all = cat(parts_idx, torch.int32), cat(parts_val, val.dtype)
    del parts_idx, parts_val
    t = Tiles(lw, rpt, cap, NB, P, nrows, ncols, idx_all, val_all, tile_ptr, _count_layout(cat(parts_cnt, torch.int32), NB * P, CW, NT),
              normalize_groups(groups, P, max(1, int(max_groups))), rpt_max, NT)
    t.stats = dict(nnz=nnz, tiled=nnz - n_over, remainder=0)
    if n_over == 0:
        return t
    orow, ocol, oval = torch.cat(over_row), torch.cat(over_col), torch.cat(over_val)
    del over_row, over_col, over_val
    # ---- the remainder: segments of <= SEG items, row by row (the items arrive in row order)
    nrem = int(orow.numel())
    t.stats["remainder"] = nrem
    if nrem > 0:
        rrow, rcol, rval = orow, ocol, oval
        rows_u, per_row = torch.unique_consecutive(rrow, return_counts=True)
        segs_per_row = (per_row + SEG - 1) // SEG
        rptr = torch.zeros(rows_u.numel() + 1, dtype=torch.int64, device=dev)
        rptr[1:] = torch.cumsum(segs_per_row, 0)
        ns = int(rptr[-1])
        row_start = torch.cumsum(per_row, 0) - per_row
        seg_row = torch.repeat_interleave(torch.arange(rows_u.numel(), device=dev), segs_per_row)
        seg_j = torch.arange(ns, device=dev) - rptr[:-1][seg_row]
        s0 = row_start[seg_row] + SEG * seg_j
        sptr = torch.cat([s0, torch.tensor([nrem], device=dev)])
        t.rem_rows, t.rem_rptr, t.rem_sptr = rows_u.to(torch.int32), rptr.to(torch.int32), sptr.to(torch.int32)
        t.rem_col, t.rem_val = rcol.to(torch.int32).contiguous(), rval.contiguous()
    return t


def _count_layout(cnt_tm: torch.Tensor, ntiles: int, CW: int, NT: int = NT) -> torch.Tensor:
    """thread-major count words [tile][thread][CW] -> the kernel's layout: [tile][thread][4] (words 0..3 as one 16-byte entry per
    thread, zero padded when CW < 4), then [tile][CW - 4][thread] (k_tiled_fused, load_counts)"""
    c = cnt_tm.view(ntiles, NT, CW)
    a = torch.zeros(ntiles, NT, 4, dtype=torch.int32, device=c.device)
    a[:, :, :min(CW, 4)] = c[:, :, :4]
    if CW <= 4:
        return a.reshape(-1)
    return torch.cat([a.reshape(-1), c[:, :, 4:].permute(0, 2, 1).reshape(-1)])


def _tile_count_words(t: Tiles, tile: int) -> torch.Tensor:
    """[nt][cw] count words of one tile out of the kernel's layout"""
    CW, ntiles, NT = t.cw, t.nblk * t.npanel, t.nt
    a = t.cnt[tile * NT * 4:(tile + 1) * NT * 4].view(NT, 4)[:, :min(CW, 4)]
    if CW <= 4:
        return a
    off = ntiles * NT * 4 + tile * (CW - 4) * NT
    return torch.cat([a, t.cnt[off:off + (CW - 4) * NT].view(CW - 4, NT).t()], dim=1)


def tile_row_counts(t: Tiles, tile: int) -> torch.Tensor:
    """items of each of the tile's nt*rpt rows (unpacks the nibbles)"""
    CW, NT = t.cw, t.nt
    w = _tile_count_words(t, tile).reshape(-1).long() & 0xFFFFFFFF
    shifts = (torch.arange(8, device=w.device) * 4).view(1, 1, 8)
    nib = (w.view(NT, CW, 1) >> shifts) & 15
    per_thread = nib.reshape(NT // 64, 64, t.rpt_max)[:, :, :t.rpt]           # [wave, lane, i]
    return per_thread.permute(0, 2, 1).reshape(-1)                          # row = wave*64*rpt + i*64 + lane


def emulate_spmv(t: Tiles, x: torch.Tensor) -> torch.Tensor:
    """What the kernels compute, step by step, in torch on any device (tests only: pass 1 writes the
    products to their row-order slots, pass 2 sums each row's segment; then the remainder)."""
    RB, W = t.rows_per_block, 1 << t.lw
    y = torch.zeros(t.nblk * RB, dtype=torch.float64, device=x.device)
    xd = x.double()

    def one(slots, val, cols, i0, i1, counts, b):
        if i1 == i0:
            return
        prod = torch.zeros(t.cap + 8, dtype=torch.float64, device=x.device)
        prod[slots] = val[i0:i1].double() * xd[cols]
        ends = torch.cumsum(counts, 0)
        csum = torch.cat([torch.zeros(1, dtype=torch.float64, device=x.device), torch.cumsum(prod[:int(ends[-1])], 0)])
        y[b * RB:(b + 1) * RB] += csum[ends] - csum[ends - counts]

    idx = t.idx.long() & 0xFFFFFFFF
    tp = t.tile_ptr.long().tolist()
    for b in range(t.nblk):
        for p in range(t.npanel):
            tile = b * t.npanel + p
            i0, i1 = tp[tile], tp[tile + 1]
            pk = idx[i0:i1]
            one(pk >> t.lw, t.val, torch.clamp(p * W + (pk & (W - 1)), max=t.ncols - 1), i0, i1, tile_row_counts(t, tile), b)
    if t.nrem:
        sp, rptr = t.rem_sptr.long(), t.rem_rptr.long().tolist()
        prod = t.rem_val.double() * xd[t.rem_col.long()]
        cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=x.device), torch.cumsum(prod, 0)])
        seg = cs[sp[1:]] - cs[sp[:-1]]
        for i, r in enumerate(t.rem_rows.tolist()):
            y[r] += seg[rptr[i]:rptr[i + 1]].sum()
    return y[:t.nrows]
