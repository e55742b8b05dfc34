"""The panel-tiled matrix format (torchpdlp_amd/tiled.py) on CPU: building it from CSR and replaying the
kernel's two passes in torch must reproduce the CSR product; eligibility limits are honoured."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from torchpdlp_amd import tiled as T
from torchpdlp_amd.tiled import NT, build_tiles, choose_rpt, choose_shape, emulate_spmv, normalize_groups, tile_row_counts


def _csr(m, n, lens, seed):
    rng = np.random.default_rng(seed)
    rp = np.zeros(m + 1, np.int64)
    rp[1:] = np.cumsum(lens)
    ci = np.concatenate([np.sort(rng.choice(n, size=int(k), replace=False)) for k in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    va = rng.standard_normal(int(rp[-1])).astype(np.float32)
    return torch.from_numpy(rp.astype(np.int32)), torch.from_numpy(ci), torch.from_numpy(va)


@pytest.mark.parametrize("case", ["regular", "ragged_empty", "single_panel", "tall", "one_row"])
def test_build_and_replay(case):
    rng = np.random.default_rng(1)
    lw, rpt, cap = 6, 4, 1200         # 64-column panels, 2048-row blocks
    if case == "regular":
        m, n, lens = 5000, 1000, np.full(5000, 7)
    elif case == "ragged_empty":
        m, n = 4100, 777
        lens = rng.integers(0, 6, m)
        lens[:40] = 0
        lens[-3:] = 0
    elif case == "single_panel":
        m, n, lens = 100, 50, rng.integers(0, 5, 100)
    elif case == "tall":
        m, n, lens = 7000, 130, rng.integers(0, 2, 7000)
    else:
        m, n, lens = 1, 500, np.array([9])
    rp, ci, va = _csr(m, n, lens, 2)
    t = build_tiles(rp, ci, va, m, n, lw=lw, rpt=rpt, cap=cap, max_chunk_nnz=9000)   # small chunks: several of them
    assert t is not None
    RB, W = NT * rpt, 1 << lw
    assert t.nblk == (m + RB - 1) // RB and t.npanel == (n + W - 1) // W
    tp = t.tile_ptr.numpy()
    assert tp[0] == 0 and np.all(np.diff(tp) >= 0) and np.all(tp % 256 == 0) and tp[-1] == t.items
    assert sum(int(tile_row_counts(t, k).sum()) for k in range(t.nblk * t.npanel)) == int(rp[-1])   # each non-zero once
    # items of a tile are sorted by column and their slots are a permutation of the row-order ranks
    # undo the 256-item interleave (physical 4*lane + j holds sorted 64*j + lane)
    idx = (t.idx.long().numpy() & 0xFFFFFFFF).reshape(-1, 64, 4).transpose(0, 2, 1).reshape(-1)
    val_sorted = t.val.numpy().reshape(-1, 64, 4).transpose(0, 2, 1).reshape(-1)
    for tile in range(t.nblk * t.npanel):
        seg = idx[tp[tile]:tp[tile + 1]]
        real = int(tile_row_counts(t, tile).sum())
        lcol, slot = seg & (W - 1), seg >> lw
        isreal = slot < real
        assert isreal.sum() == real and np.all(np.diff(lcol[isreal]) >= 0)
        assert sorted(slot[isreal].tolist()) == list(range(real))
        assert np.all(val_sorted[tp[tile]:tp[tile + 1]][~isreal] == 0)
    x = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
    ref = sp.csr_matrix((va.numpy().astype(np.float64), ci.numpy(), rp.numpy()), shape=(m, n)) @ x.numpy().astype(np.float64)
    np.testing.assert_allclose(emulate_spmv(t, x).numpy(), ref, rtol=1e-12, atol=1e-12)


def _check(t, rp, ci, va, m, n, seed=5):
    x = torch.from_numpy(np.random.default_rng(seed).standard_normal(n)).to(va.dtype)
    ref = sp.csr_matrix((va.numpy().astype(np.float64), ci.numpy(), rp.numpy()), shape=(m, n)) @ x.numpy().astype(np.float64)
    np.testing.assert_allclose(emulate_spmv(t, x).numpy(), ref, rtol=1e-11, atol=1e-11)
    assert t.stats["tiled"] + t.stats["remainder"] == t.stats["nnz"] == int(rp[-1])
    assert t.nrem == t.stats["remainder"]
    if t.nrem:
        sp_, rptr = t.rem_sptr.numpy(), t.rem_rptr.numpy()
        assert sp_[0] == 0 and sp_[-1] == t.nrem and np.all(np.diff(sp_) >= 1) and np.all(np.diff(sp_) <= 512)
        assert rptr[0] == 0 and rptr[-1] == len(sp_) - 1 and np.all(np.diff(rptr) >= 1)
        assert np.all(np.diff(t.rem_rows.numpy()) > 0)


def test_what_a_tile_cannot_hold_goes_to_the_remainder():
    """round 1 gave the whole matrix back to the CSR kernel when a single tile, row or 64-row group broke a format limit; now the
    surplus items form a small CSR remainder next to the tiles (None only when > 30 % of the items would leave them)"""
    rp, ci, va = _csr(640, 64, np.full(640, 3), 3)
    t = build_tiles(rp, ci, va, 640, 64, lw=6, rpt=2, cap=1000)                    # 640 rows x 3 = 1920 items in one tile of <= 1000
    assert t is None                                                               # ... 48 % would leave: the CSR kernel's case
    t = build_tiles(rp, ci, va, 640, 64, lw=6, rpt=2, cap=1500, max_rest=0.4)      # every row keeps 2 of its 3
    assert t is not None and t.stats["remainder"] == 640 and int(np.diff(t.tile_ptr.numpy()).max()) <= 1536
    _check(t, rp, ci, va, 640, 64)
    t = build_tiles(rp, ci, va, 640, 64, lw=6, rpt=2, cap=8000)
    assert t is not None and t.nrem == 0
    rp, ci, va = _csr(640, 64, np.full(640, 5), 3)                                # 64 consecutive rows x 5 = 320 items of one
    assert build_tiles(rp, ci, va, 640, 64, lw=6, rpt=2, cap=8000) is None         # tile: more than the 8-bit scan fields hold
    t = build_tiles(rp, ci, va, 640, 64, lw=6, rpt=2, cap=8000, max_rest=0.5)      # every row keeps 3 (192 per 64 rows), 2 go
    assert t is not None and t.stats["remainder"] == 1280
    _check(t, rp, ci, va, 640, 64)
    for tile in range(t.nblk * t.npanel):
        c = tile_row_counts(t, tile).view(NT // 64, t.rpt, 64)
        assert int(c.sum(-1).max()) <= 255 and int(c.max()) <= 15
    t2 = build_tiles(rp, ci, va, 640, 64, lw=5, rpt=2, cap=8000)                   # (two panels: 160 per 64 rows)
    assert t2 is not None and t2.nrem == 0
    # a dense row: 16+ entries of one row in one panel; and a row far longer than a remainder segment
    lens = np.array([30, 1, 1, 400] + [2] * 2000)
    rp, ci, va = _csr(2004, 600, lens, 4)
    t = build_tiles(rp, ci, va, 2004, 600, lw=10, rpt=2, cap=2600)
    assert t is not None and t.rem_rows.tolist() == [0, 3] and t.stats["remainder"] == (30 - 15) + (400 - 15)
    assert t.rem_rptr.tolist() == [0, 1, 2]
    _check(t, rp, ci, va, 2004, 600)
    t64 = build_tiles(rp, ci, va.double(), 2004, 600, lw=6, rpt=2)                   # float64: 3 count words, 8192-item tiles
    assert t64 is not None and t64.cw == 3 and t64.cap == 8192 and t64.val.dtype == torch.float64 and t64.rem_val.dtype == torch.float64
    _check(t64, rp, ci, va.double(), 2004, 600)
    # a banded matrix (every row's entries inside one panel): nearly everything would be remainder -> not tiled
    m = n = 4000
    rp = torch.arange(0, (m + 1) * 40, 40, dtype=torch.int32)
    ci = ((torch.arange(m).view(-1, 1) + torch.arange(40).view(1, -1)) % n).to(torch.int32).sort(dim=1)[0].reshape(-1)
    assert build_tiles(rp, ci, torch.ones(m * 40), m, n, lw=12, rpt=2) is None


def test_no_panel_group_is_empty():
    # the kernel gives each group ceil(P/groups) panels; 10 panels in 8 groups would leave three groups empty
    assert normalize_groups(8, 10) == 5 and normalize_groups(3, 10) == 3 and normalize_groups(8, 153) == 8
    for P in range(1, 40):
        for g in range(1, 12):
            gg = normalize_groups(g, P)
            ppg = -(-P // gg)
            assert 1 <= gg <= min(8, P) and (gg - 1) * ppg < P <= gg * ppg


def test_choose_rpt_fills_whole_rounds():
    # the bench matrix: 10M rows, 100 per row, 64K panels -> 0.655 items per (row, panel)
    rpt = choose_rpt(10_000_000, 1_000_000_000, 10_000_000, 16)
    nblk = -(-10_000_000 // (512 * rpt))
    assert 512 * rpt * 0.6554 * 1.06 < 16384
    assert nblk / (512 * -(-nblk // 512)) > 0.9
    assert choose_rpt(1000, 3000, 1000, 16) >= 1
    assert choose_shape(10_000_000, 1_000_000_000, 10_000_000, 16) == (rpt, 1)
    # one rank's shard of the same problem on 8 GPUs: big row blocks, panels split over 8 workgroups
    rpt8, g8 = choose_shape(1_250_000, 125_000_000, 10_000_000, 16)
    assert rpt8 >= 32 and g8 == 8 and 0.9 < -(-1_250_000 // (512 * rpt8)) * g8 / 512 <= 1.0
    assert choose_shape(100, 300, 50, 16) == (choose_rpt(100, 300, 50, 16), 1)          # a single panel cannot be split


def test_panel_width_adapts_to_long_rows():
    from torchpdlp_amd.tiled import choose_lw
    assert choose_lw(10_000_000, 1_000_000_000, 10_000_000) == 16            # the bench matrix: 0.66 items per (row, panel)
    assert choose_lw(1_000_000, 100_000_000, 1_000_000) == 14                # 100 per row over 1M columns: 6.6 at 64K, 1.6 at 16K
    m, n = 2000, 70_000
    rp, ci, va = _csr(m, n, np.full(m, 20), 11)
    assert build_tiles(rp, ci, va, m, n, lw=16) is None                      # 64 rows x 18.7 items per panel > 255: 80 % would leave
    t = build_tiles(rp, ci, va, m, n)                                        # auto: 8K-column panels
    assert t is not None and t.lw == 13 and t.nrem == 0
    x = torch.from_numpy(np.random.default_rng(1).standard_normal(n).astype(np.float32))
    ref = sp.csr_matrix((va.numpy().astype(np.float64), ci.numpy(), rp.numpy()), shape=(m, n)) @ x.numpy().astype(np.float64)
    np.testing.assert_allclose(emulate_spmv(t, x).numpy(), ref, rtol=1e-12, atol=1e-12)


def test_chosen_groups_fit_the_rowsum_scratch():
    """the library sizes its row-sum scratch by ``rowsum_groups(rows)`` (1 group above ~10.5M rows) and rejects
    tiles with more panel groups; the shape chooser must be given, and respect, that limit for every size"""
    for rows in (200_000, 1_250_000, 5_000_000, 10_000_000, 10_485_760, 10_485_761, 10_600_000, 12_000_000, 15_000_000,
                 20_000_000, 30_000_000):
        lim = T.rowsum_groups(rows)
        assert lim == (32 if rows <= 512 * 40 * 128 else 8 if rows <= 512 * 40 * 512 else 1)
        for per_row in (5, 20, 50):
            for rpt_max, cap in ((T.RPT_MAX, T.CAP), (T.RPT_MAX_F64, T.CAP_F64)):
                for cols in (rows, 4 * rows, rows // 4):
                    lw = T.choose_lw(rows, rows * per_row, cols)
                    rpt, groups = T.choose_shape(rows, rows * per_row, cols, lw, cap, max_groups=min(8, lim), rpt_max=rpt_max)
                    assert 1 <= groups <= lim and 1 <= rpt <= rpt_max
                    P = max(1, (cols + (1 << lw) - 1) >> lw)
                    assert T.normalize_groups(groups, P, min(8, lim)) == groups


def test_abi_tile_ptr_is_relative_to_64_bit_row_block_bases():
    """ABI 15: the library takes the tile offsets per row block -- int32 [nblk][npanel + 1] relative to the block's first item plus an
    int64 base per block -- so that a thread indexes with 32 bits while a matrix copy may hold more than 2^31 items"""
    tp = torch.tensor([0, 256, 512, 512, 1024, 1280, 1280, 2048, 2304], dtype=torch.int64)          # 2 row blocks x 4 panels
    t = T.Tiles(6, 1, 16384, 2, 4, 10, 10, torch.zeros(1, dtype=torch.int32), torch.zeros(1), tp, torch.zeros(1, dtype=torch.int32))
    rel, base = t.abi_tile_ptr()
    assert rel.dtype == torch.int32 and base.dtype == torch.int64
    assert rel.tolist() == [[0, 256, 512, 512, 1024], [0, 256, 256, 1024, 1280]] and base.tolist() == [0, 1024]
    assert t.abi_tile_ptr()[0] is rel                                   # built once, kept alive with the tiles
    big = tp + (1 << 33)                                                # positions far beyond 32 bits: only the bases grow
    t2 = T.Tiles(6, 1, 16384, 2, 4, 10, 10, torch.zeros(1, dtype=torch.int32), torch.zeros(1), big, torch.zeros(1, dtype=torch.int32))
    rel2, base2 = t2.abi_tile_ptr()
    assert torch.equal(rel2, rel) and base2.tolist() == [1 << 33, (1 << 33) + 1024]
