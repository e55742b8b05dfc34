"""The panel-tiled matrix format (torchpdlp_amd/tiled.py) on CPU: building it from CSR and replaying the
kernel's two passes in torch must reproduce the CSR product; eligibility limits are honoured."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from torchpdlp_amd.tiled import build_tiles, emulate_spmv


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
    lw, lrb, cap = 6, 5, 200          # 64-column panels, 32-row blocks: many tiles even at test sizes
    if case == "regular":
        m, n, lens = 300, 1000, np.full(300, 7)
    elif case == "ragged_empty":
        m, n = 257, 777
        lens = rng.integers(0, 12, m)
        lens[:40] = 0
        lens[-3:] = 0
    elif case == "single_panel":
        m, n, lens = 100, 50, rng.integers(0, 5, 100)
    elif case == "tall":
        m, n, lens = 2000, 130, rng.integers(1, 4, 2000)
    else:
        m, n, lens = 1, 500, np.array([9])
    rp, ci, va = _csr(m, n, lens, 2)
    t = build_tiles(rp, ci, va, m, n, lw=lw, lrb=lrb, cap=cap, max_chunk_nnz=500)    # small chunks: several of them
    assert t is not None
    RB, W = 1 << lrb, 1 << lw
    assert t.nblk == (m + RB - 1) // RB and t.npanel == (n + W - 1) // W
    tp = t.tile_ptr.numpy()
    assert tp[0] == 0 and np.all(np.diff(tp) >= 0) and np.all(tp % 4 == 0) and tp[-1] == t.items
    assert int(t.cnt.long().sum()) == int(rp[-1])                      # every non-zero counted once
    # items of a tile are sorted by column and their slots are a permutation of the row-order ranks
    idx = t.idx.long().numpy() & 0xFFFFFFFF
    for tile in range(t.nblk * t.npanel):
        seg = idx[tp[tile]:tp[tile + 1]]
        real = int(t.cnt[tile * RB:(tile + 1) * RB].long().sum())
        lcol, slot = seg & (W - 1), seg >> lw
        isreal = slot < real
        assert isreal.sum() == real and np.all(np.diff(lcol[isreal]) >= 0)
        assert sorted(slot[isreal].tolist()) == list(range(real))
        assert np.all(t.val.numpy()[tp[tile]:tp[tile + 1]][~isreal] == 0)
    x = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
    ref = sp.csr_matrix((va.numpy().astype(np.float64), ci.numpy(), rp.numpy()), shape=(m, n)) @ x.numpy().astype(np.float64)
    np.testing.assert_allclose(emulate_spmv(t, x).numpy(), ref, rtol=1e-12, atol=1e-12)


def test_not_eligible_when_a_tile_or_a_row_is_too_full():
    rp, ci, va = _csr(64, 64, np.full(64, 40), 3)
    assert build_tiles(rp, ci, va, 64, 64, lw=6, lrb=5, cap=200) is None         # 32 rows x 40 = 1280 items in one tile
    assert build_tiles(rp, ci, va, 64, 64, lw=6, lrb=5, cap=2000) is not None
    rp, ci, va = _csr(4, 600, np.array([300, 1, 1, 1]), 4)
    assert build_tiles(rp, ci, va, 4, 600, lw=10, lrb=5, cap=2000) is None        # 256+ entries of one row in one panel
