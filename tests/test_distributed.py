"""Multi-rank path: row-block shards of K and K', all-gather of xbar / y between the half-steps.

CPU (gloo, world size 2): the sharding + padding + exchange logic, with the ORACLE's SpMV standing in
for the local kernels (the HIP kernels need a GPU) -- the result must equal the unsharded oracle step.
GPU (marked): two ranks share the one MI355X of the test box, collectives over gloo; whole solves must
match the single-rank engine.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from torchpdlp_amd.distributed import block, make_partition, padded, shard_arrays
from torchpdlp_amd.sparse import CsrPair
from torchpdlp_amd.synthetic import gen_lp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _lp(device="cpu"):
    # sizes that do NOT divide by the world size: exercises the padding
    lp = gen_lp(301, 403, 4, seed=21, recipe="mixed", ineq_frac=0.6, device=device)
    return lp, CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)


def _skewed_lp():
    """a dense column and a dense row on top of the regular pattern: blocks balanced by non-zeros differ a lot in row count"""
    import scipy.sparse as sp
    lp, K = _lp()
    A = sp.csr_matrix((lp.val.numpy(), lp.colidx.numpy(), lp.rowptr.numpy()), shape=(lp.m, lp.n)).tolil()
    rng = np.random.default_rng(5)
    A[rng.choice(lp.m, 250, replace=False), 17] = rng.standard_normal(250).astype(np.float32)
    A[40, rng.choice(lp.n, 200, replace=False)] = rng.standard_normal(200).astype(np.float32)
    A = A.tocsr()
    A.sort_indices()
    K2 = CsrPair(lp.m, lp.n, torch.from_numpy(A.indptr.copy()), torch.from_numpy(A.indices.copy()), torch.from_numpy(A.data.astype(np.float32)))
    return lp, K2


def _cpu_worker(rank, world, port, ret, balance="rows"):
    from oracle import oracle as orc          # checker only
    from torchpdlp_amd.engine import Comm
    _init(rank, world, port)
    try:
        orc.set_threads(1)
        comm = Comm()
        lp, K = _lp() if balance == "rows" else _skewed_lp()
        sh = shard_arrays(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, rank, world, balance=balance)
        part = sh.pop("part")
        (r0, r1), (c0, c1) = sh["rows"], sh["cols"]
        mp_, np_ = sh["m"], sh["n"]
        if balance == "nnz":
            lens = [part.rb[g + 1] - part.rb[g] for g in range(world)]
            assert max(lens) > 1.2 * min(lens)            # really unequal row counts
        z = np.zeros
        # local operators through the oracle's CSR product (global column indices, padded widths)
        Kr = orc.OracleLP(r1 - r0, np_, 0, *(t.numpy() for t in sh["K_rows"]), z(np_), z(r1 - r0), z(np_), z(np_),
                          trans=(z(np_ + 1, np.int32), z(0, np.int32), z(0, np.float32)))
        KTr = orc.OracleLP(c1 - c0, mp_, 0, *(t.numpy() for t in sh["KT_rows"]), z(mp_), z(c1 - c0), z(mp_), z(mp_),
                           trans=(z(mp_ + 1, np.int32), z(0, np.int32), z(0, np.float32)))
        g = torch.Generator().manual_seed(3)
        x0 = torch.minimum(torch.maximum(torch.randn(lp.n, generator=g), lp.l), lp.u)
        y0 = torch.randn(lp.m, generator=g); y0[:lp.m_ineq].clamp_(min=0)
        x, y = part.pad_cols(x0), part.pad_rows(y0)          # the padded layout (identity + trailing padding for balance="rows")
        eta, omega = np.float32(0.07), np.float32(1.3)
        tau, sigma = eta / omega, eta * omega
        c, q, l, u = (sh[k].numpy() for k in ("c", "q", "l", "u"))
        for _ in range(3):     # three fixed steps, sharded (step.py:25-38)
            kty = KTr.spmv(y.numpy())                                   # this rank's block of K'y
            xl = x[c0:c1].numpy()
            xn = np.minimum(np.maximum(xl - tau * (c - kty), l), u)
            xbar = torch.zeros(np_); xbar[c0:c1] = torch.from_numpy(xn + 1.0 * (xn - xl))
            comm.all_gather(xbar)                                       # every rank's block of xbar
            kxb = Kr.spmv(xbar.numpy())
            yl = y[r0:r1].numpy()
            yn = yl + sigma * (q - kxb)
            ineq_end = sh["m_ineq"] - r0                               # this block's inequalities are its first rows
            yn[:ineq_end] = np.maximum(yn[:ineq_end], 0)
            x = torch.zeros(np_); x[c0:c1] = torch.from_numpy(xn); comm.all_gather(x)
            y = torch.zeros(mp_); y[r0:r1] = torch.from_numpy(yn); comm.all_gather(y)
        # reference: the unsharded oracle
        o = orc.OracleLP(lp.m, lp.n, lp.m_ineq, *(t.numpy() for t in (K.rowptr, K.colidx, K.val, lp.c, lp.q, lp.l, lp.u)))
        g = torch.Generator().manual_seed(3)
        xo = torch.minimum(torch.maximum(torch.randn(lp.n, generator=g), lp.l), lp.u).numpy()
        yo = torch.randn(lp.m, generator=g); yo[:lp.m_ineq].clamp_(min=0); yo = yo.numpy()
        for _ in range(3):
            xo, yo = o.step_fixed(xo, yo, eta, omega, 1.0)
        np.testing.assert_allclose(part.unpad_cols(x).numpy(), xo, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(part.unpad_rows(y).numpy(), yo, rtol=1e-5, atol=1e-6)
        padx, pady = torch.ones(np_, dtype=torch.bool), torch.ones(mp_, dtype=torch.bool)
        padx[part.col_map("cpu")] = False
        pady[part.row_map("cpu")] = False
        assert float(x[padx].abs().sum()) == 0 and float(y[pady].abs().sum()) == 0           # padding stays at 0
        # the chunked exchange's transport: pieces = the same slice of every rank's block, asynchronous, in order
        B = 1000
        full = torch.zeros(world * B)
        full[rank * B:(rank + 1) * B] = torch.arange(B, dtype=torch.float32) + 10_000 * (rank + 1)
        works = [comm.all_gather_piece(full, lo, hi) for lo, hi in ((0, 448), (448, 1000))]
        for w in works:
            if w is not None:
                w.wait()
        want = torch.cat([torch.arange(B, dtype=torch.float32) + 10_000 * (q + 1) for q in range(world)])
        assert torch.equal(full, want) and comm.all_gather_piece(full, 5, 5) is None
        # an 8-double all-reduce like the KKT partial sums
        red = torch.tensor([float(rank + 1)] * 8, dtype=torch.float64)
        comm.all_reduce_sum(red)
        assert red.tolist() == [world * (world + 1) / 2.0] * 8
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,balance", [(2, "rows"), (3, "rows"), (2, "nnz"), (3, "nnz")])
def test_sharded_step_equals_unsharded_cpu_gloo(world, balance):
    port = _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_cpu_worker, args=(world, port, ret, balance), nprocs=world, join=True)
        assert dict(ret) == {r: "ok" for r in range(world)}


def _shard_gen_worker(rank, world, port, ret):
    from torchpdlp_amd.distributed import gen_lp_shard_arrays
    from torchpdlp_amd.engine import Comm
    _init(rank, world, port)
    try:
        n, m, k = 70_001, 131_075, 6            # rows span three 2^16-row generator chunks, sizes that do not divide
        got = gen_lp_shard_arrays(n, m, k, 9, Comm(), "cpu")
        lp = gen_lp(n, m, k, seed=9)             # the whole instance, for comparison only
        K = CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
        want = shard_arrays(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, rank, world, balance="rows")
        for key in ("m", "n", "m_ineq", "rows", "cols"):
            assert got[key] == want[key], key
        for key in ("K_rows", "KT_rows"):
            for a, b in zip(got[key], want[key]):
                assert torch.equal(a, b.to(a.dtype)), key
        for key in ("c", "q", "l", "u"):
            assert torch.equal(got[key], want[key]), key
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_generated_in_place_equals_the_shard_of_the_whole_instance(world):
    """per-shard generation + distributed transpose -- no rank materialises the whole instance, yet every rank ends up
    with exactly its blocks of gen_lp's instance (K rows from per-chunk seeds, K' rows from the all-to-all of the entries)"""
    port = _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_shard_gen_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {r: "ok" for r in range(world)}


def test_partition_balanced_by_nonzeros():
    """SURVEY 8e / VERDICT r1: blocks by non-zeros, not by row count.  neos3-shaped K' (6 624 rows, one of them with 207k of the
    1.46M entries): equal row counts would give one rank 4x the mean; by non-zeros no rank holds more than the dense row's share
    plus its fair share, and every block but the dense row's is within 20 % of the mean."""
    rng = np.random.default_rng(0)
    m, n = 6624, 512_209
    lens = rng.integers(50, 350, m)
    lens[100] = 207_217
    rp = np.zeros(m + 1, np.int64)
    rp[1:] = np.cumsum(lens)
    nnz = int(rp[-1])
    KT = CsrPair.__new__(CsrPair)                      # only the row pointers matter for the boundaries
    from torchpdlp_amd.distributed import _bounds_by_nnz, _bounds_by_rows
    for W in (2, 4, 8):
        b = _bounds_by_nnz(torch.from_numpy(rp), m, W)
        assert b[0] == 0 and b[-1] == m and all(b[i] <= b[i + 1] for i in range(W))
        per = [int(rp[b[g + 1]] - rp[b[g]]) for g in range(W)]
        assert sum(per) == nnz
        heavy = max(per)
        assert heavy <= nnz / W + 207_217 + 350
        others = sorted(per)[:-1]
        if W <= 4:
            assert max(others) <= 1.2 * nnz / W
        eq = _bounds_by_rows(m, W)
        per_eq = [int(rp[eq[g + 1]] - rp[eq[g]]) for g in range(W)]
        assert max(per_eq) >= max(per)                   # never worse than equal row counts
    # the padded layout: maps are injective, monotone inside a block, and pad/unpad round-trips
    lp, K = _skewed_lp()
    part = make_partition(K, 3, "nnz")
    rm, cm = part.row_map("cpu"), part.col_map("cpu")
    assert rm.unique().numel() == lp.m and cm.unique().numel() == lp.n and int(rm.max()) < part.mp and int(cm.max()) < part.np_
    assert bool((rm[1:] > rm[:-1]).all()) and bool((cm[1:] > cm[:-1]).all())
    v = torch.randn(lp.n)
    assert torch.equal(part.unpad_cols(part.pad_cols(v)), v)
    parts = [shard_arrays(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, r, 3, balance="nnz") for r in range(3)]
    assert sum(int(p["K_rows"][2].numel()) for p in parts) == K.nnz == sum(int(p["KT_rows"][2].numel()) for p in parts)
    per = [int(p["K_rows"][2].numel()) for p in parts]
    assert max(per) <= 1.25 * K.nnz / 3 + 250
    assert sum(p["m_ineq"] - p["rows"][0] for p in parts) == lp.m_ineq          # every inequality is some block's prefix row


def test_block_partition_and_padding():
    for v, w in ((403, 2), (301, 2), (10_000_000, 8), (7, 4), (8, 8)):
        vp = padded(v, w)
        assert vp % w == 0 and 0 <= vp - v < w
        blocks = [block(vp, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == vp
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
    lp, K = _lp()
    parts = [shard_arrays(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, r, 2) for r in range(2)]
    assert all(p.pop("part").Bm * 2 == p["m"] for p in parts)
    assert sum(int(p["K_rows"][2].numel()) for p in parts) == K.nnz == sum(int(p["KT_rows"][2].numel()) for p in parts)
    assert all(p["K_rows"][0].numel() == p["rows"][1] - p["rows"][0] + 1 for p in parts)
    # the padding variables are fixed at zero and the padding rows are empty equalities
    last = parts[-1]
    npad = last["n"] - lp.n
    assert npad == 1 and float(last["l"][-1]) == 0 == float(last["u"][-1]) == float(last["c"][-1])
    assert int(last["K_rows"][0][-1]) == int(last["K_rows"][0][-2])      # empty last row
    assert all(int(p["K_rows"][1].max()) < lp.n for p in parts)


def _gpu_worker(rank, world, port, ret):
    import torchpdlp_amd as tp
    from torchpdlp_amd.distributed import gather_solution, shard_engine
    from torchpdlp_amd.solver import run_pdlp
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        lp, K = _lp(dev)
        comm = tp.Comm()
        b0 = torch.randn(lp.n, generator=torch.Generator().manual_seed(9)).to(dev)      # (original variable order)
        out = {}
        for adaptive in (False, True):
            eng = shard_engine(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, comm)                  # blocks balanced by non-zeros
            trace = dict(kkt=[], omega=[], restarts=[])
            x, obj, k, n, j, status, _ = run_pdlp(eng, tol=1e-4, verbose=False, primal_update=True, adaptive=adaptive,
                                                  b0=eng.part.pad_cols(b0), trace=trace)
            xf = gather_solution(eng, x, lp.n)
            out[adaptive] = (xf.cpu(), obj, k, n, j, status, trace)
            del eng
        if rank == 0:      # the same solves on one rank
            for adaptive in (False, True):
                eng1 = tp.PdlpEngine.from_full(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq)
                tr1 = dict(kkt=[], omega=[], restarts=[])
                x1, obj1, k1, n1, j1, st1, _ = run_pdlp(eng1, tol=1e-4, verbose=False, primal_update=True, adaptive=adaptive,
                                                        b0=b0, trace=tr1)
                xf, obj, k, n, j, status, trace = out[adaptive]
                assert status == st1 == "Solved"
                assert abs(obj - lp.opt_obj) <= 2e-3 * (1 + abs(lp.opt_obj))
                assert abs(obj - obj1) <= 2e-3 * (1 + abs(obj1))
                nk = 4 if adaptive else 10
                np.testing.assert_allclose(trace["kkt"][:nk], tr1["kkt"][:nk], rtol=5e-2 if adaptive else 1e-3)
                assert trace["restarts"][:1] == tr1["restarts"][:1]
                assert j == k + (len(trace["kkt"]) - n) + 2 * n
        # the detector, sharded: dx, dy all-gathered, eight sums all-reduced -> the single-rank diagnostics
        eng = shard_engine(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, comm)
        eng.set_iterate(torch.zeros(eng.nl, device=dev), torch.zeros(eng.ml, device=dev))
        eng.set_step(0.05, 1.0, 1.0, 0)
        eng.infeas_reset()
        eng.iterate(2, True)
        st2, d2 = eng.detect_infeasibility(1e-2, diagnostics=True)
        eng.iterate(1, True)
        st3, d3 = eng.detect_infeasibility(1e3, diagnostics=True)
        del eng
        if rank == 0:
            e1 = tp.PdlpEngine.from_full(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq)
            e1.set_iterate(torch.zeros(lp.n, device=dev), torch.zeros(lp.m, device=dev))
            e1.set_step(0.05, 1.0, 1.0, 0)
            e1.iterate(2, True)
            s2, f2 = e1.detect_infeasibility(1e-2, diagnostics=True)
            e1.iterate(1, True)
            s3, f3 = e1.detect_infeasibility(1e3, diagnostics=True)
            assert (st2, st3) == (s2, s3)
            np.testing.assert_allclose(d2, f2, rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(d3, f3, rtol=1e-4, atol=1e-5)
        # the user-facing entry point, sharded: every rank passes the whole problem and gets the whole solution
        res = tp.solve_lp((lp.c, K, lp.q, lp.m_ineq, lp.l, lp.u), device=dev, tol=1e-4, precondition=True,
                          primal_weight_update=True, adaptive_stepsize=True, seed=1, comm=True)
        assert res.status == "Solved" and res.x.shape == (lp.n, 1)
        assert abs(res.objective - lp.opt_obj) <= 2e-3 * (1 + abs(lp.opt_obj))
        xs = res.x.view(-1)
        assert float((lp.c.view(-1).to(xs) * xs).sum()) == pytest.approx(res.objective, rel=1e-3, abs=1e-3)
        # mixed precision + delta mode, sharded: the ranks exchange the float32 differences x+ - x and y+ - y instead of xbar and y
        v64 = [t.double() for t in (lp.c, lp.q, lp.l, lp.u)]
        xm, objm, km, nm, jm, stm, _ = tp.pdlp_algorithm(K, lp.m_ineq, *v64, dev, tol=1e-6, verbose=False, adaptive=True, primal_update=True,
                                                         seed=2, precision="mixed", comm=comm, max_kkt=400_000)
        assert stm == "Solved" and xm.dtype == torch.float64 and xm.shape == (lp.n, 1)
        assert abs(objm - lp.opt_obj) <= 3e-5 * (1 + abs(lp.opt_obj))              # (float32 could not certify 1e-6)
        if rank == 0:
            x1, obj1, k1, n1, j1, st1, _ = tp.pdlp_algorithm(K, lp.m_ineq, *v64, dev, tol=1e-6, verbose=False, adaptive=True,
                                                             primal_update=True, seed=2, precision="mixed")
            assert st1 == "Solved" and abs(obj1 - objm) <= 3e-5 * (1 + abs(obj1)) and 0.3 * k1 <= km <= 3 * k1
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_match_single_rank(world):
    port = _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_gpu_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {r: "ok" for r in range(world)}


def _gpu_split_worker(rank, world, port, ret):
    """two ranks, tiled kernels forced, a problem wide enough for several 64K-column panels per rank: the products
    are split (local panels on the side stream before the all-gather, the others after it)"""
    os.environ["PDLP_TILED"] = "1"
    import torchpdlp_amd as tp
    from torchpdlp_amd import _native as N
    from torchpdlp_amd.distributed import gather_solution, shard_engine
    from torchpdlp_amd.synthetic import gen_lp
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        lp = gen_lp(330_000, 300_000, 4, seed=12, device=dev, recipe="mixed")
        K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
        comm = tp.Comm()
        eng = shard_engine(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, comm)
        assert all(t is not None for t in eng.tiles)
        for tr in (0, 1):
            info = eng.split_info(tr)
            assert info["local_groups"] >= 1 and info["other_groups"] >= 1, info
            assert info["local_panels"][1] - info["local_panels"][0] >= 1
        res = {}
        for adaptive in (True, False):
            g = torch.Generator().manual_seed(3)
            x0 = torch.randn(lp.n, generator=g).to(dev)          # original order; the engine's layout is the padded one
            y0 = torch.randn(lp.m, generator=g).to(dev)
            eng.set_iterate(eng.part.pad_cols(x0)[eng.cols[0]:eng.cols[1]], eng.part.pad_rows(y0)[eng.rows[0]:eng.rows[1]])
            eng.set_step(0.02, 1.1, 1.0, 0)
            eng.iterate(9, adaptive)
            eng.iterate(4, adaptive)
            x, y = eng.get_iterate(N.CUR)
            kkt = eng.kkt(N.CUR, 1.0)
            res[adaptive] = (gather_solution(eng, x, lp.n).cpu(), eng.scalars()["eta"], kkt["kkt"], x0.cpu(), y0.cpu())
        del eng
        if rank == 0:
            os.environ["PDLP_TILED"] = "0"
            e1 = tp.PdlpEngine.from_full(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq)
            for adaptive in (True, False):
                xs, eta, kkt, x0, y0 = res[adaptive]
                e1.set_iterate(x0.to(dev), y0.to(dev))
                e1.set_step(0.02, 1.1, 1.0, 0)
                e1.iterate(13, adaptive)
                x1, _ = e1.get_iterate(N.CUR)
                np.testing.assert_allclose(xs.numpy(), x1.cpu().numpy(), rtol=2e-4, atol=2e-5)
                np.testing.assert_allclose(eta, e1.scalars()["eta"], rtol=1e-4)
                np.testing.assert_allclose(kkt, e1.kkt(N.CUR, 1.0)["kkt"], rtol=1e-4)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_split_products_match_single_rank():
    world = 2
    port = _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_gpu_split_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}


def _rccl_world1_worker(rank, world, port, ret):
    """backend nccl (= RCCL) with the one rank this box has: the library's own communicator, its all-gather / all-reduce round
    trip, and pdlp_iterate running the sharded sequence (collectives enqueued between the half-steps) -- bit for bit the
    iterations of a handle without a communicator"""
    import torchpdlp_amd as tp
    from torchpdlp_amd import _native as N
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        # torch.distributed's RCCL first (what the engine's default path uses)
        comm = tp.Comm()
        assert comm.backend == "nccl" and comm.world == 1
        full = torch.arange(12, dtype=torch.float32, device=dev)
        comm.all_gather(full)
        red = torch.ones(8, dtype=torch.float64, device=dev)
        comm.all_reduce_sum(red)
        assert torch.equal(full.cpu(), torch.arange(12, dtype=torch.float32)) and float(red.sum()) == 8.0
        work = comm.all_gather_piece(full, 2, 6)            # a piece of a chunked exchange: asynchronous list all-gather under RCCL
        assert work is not None
        work.wait()
        torch.cuda.current_stream().synchronize()
        assert torch.equal(full.cpu(), torch.arange(12, dtype=torch.float32)) and comm.all_gather_piece(full, 4, 4) is None
        for dtype, kw in ((torch.float32, {}), (torch.float64, {}), (torch.float32, dict(vec_dtype=torch.float64))):
            lp = gen_lp(40_000, 30_000, 6, seed=3, device=dev, recipe="mixed", dtype=torch.float64)
            K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val.to(dtype))
            vd = kw.get("vec_dtype", dtype)
            mk = lambda: tp.PdlpEngine.from_full(K, lp.c.to(vd), lp.q.to(vd), lp.l.to(vd), lp.u.to(vd), lp.m_ineq, **kw)
            e0, e1 = mk(), mk()
            assert e1.enable_library_comm(dist=dist) and e1.lib_comm and not e0.lib_comm
            g = torch.Generator().manual_seed(3)
            x0 = torch.randn(lp.n, generator=g, dtype=torch.float64).to(dev, vd)
            y0 = torch.randn(lp.m, generator=g, dtype=torch.float64).to(dev, vd)
            for adaptive in (True, False):
                outs = []
                for e in (e0, e1):
                    e.set_iterate(x0, y0)
                    e.set_step(0.02, 1.1, 1.0, 0)
                    e.iterate(9, adaptive)
                    e.iterate(4, adaptive)
                    x, y = e.get_iterate(N.CUR)
                    outs.append((x.clone(), y.clone(), e.scalars()["eta"], e.kkt(N.CUR, 1.0)["kkt"]))
                assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
                assert outs[0][2] == outs[1][2] and outs[0][3] == outs[1][3]
            # every collective the first multi-GPU run issues, on REAL RCCL -- the chunked exchange of the library
            # driver too: pieces as grouped in-place ncclBroadcast calls on the library's communication stream, behind the rows they
            # are made of (producer pieces, round 5) or after the half-step (round-4 form), the step-size all-reduce queued behind the
            # pieces on that stream, the consumer's waits on the pieces' events.  With one rank every piece is this rank's own
            # data, so the iterates must be the bits of the handle without a communicator.
            for pieces, producer in ((2, True), (3, True), (2, False)):
                e1.set_exchange_chunks(pieces)
                e1.set_producer_pieces(producer)
                assert len(e1.exchange_plan(0)) == pieces and len(e1.exchange_plan(1)) == pieces
                for adaptive in (True, False):
                    outs = []
                    for e in (e0, e1):
                        e.set_iterate(x0, y0)
                        e.set_step(0.02, 1.1, 1.0, 0)
                        e.iterate(5, adaptive)
                        e.iterate(2, adaptive)
                        x, y = e.get_iterate(N.CUR)
                        outs.append((x.clone(), y.clone(), e.scalars()["eta"], e.kkt(N.CUR, 1.0)["kkt"]))
                    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (pieces, producer, adaptive)
                    assert outs[0][2] == outs[1][2] and outs[0][3] == outs[1][3]
            e1.set_exchange_chunks(1)
            e1.set_producer_pieces(True)
        # per-shard generation + distributed transpose under RCCL (device-side count exchange, all_to_all_single with uneven
        # splits): with one rank the shard is the whole instance, entry for entry the one gen_lp builds
        from torchpdlp_amd.distributed import gen_lp_shard
        es = gen_lp_shard(70_000, 90_000, 7, 5, comm, dev)
        lp = gen_lp(70_000, 90_000, 7, seed=5, device=dev)
        ef = tp.PdlpEngine.from_full(tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val), lp.c, lp.q, lp.l, lp.u, lp.m_ineq)
        outs = []
        for e in (ef, es):
            e.set_step(0.02, 1.1, 1.0, 0)
            e.iterate(7, True)
            x, y = e.get_iterate(N.CUR)
            outs.append((x[:lp.n].clone(), y[:lp.m].clone(), e.kkt(N.CUR, 1.0)["kkt"]))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_world1_library_communicator():
    port = _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_rccl_world1_worker, args=(1, port, ret), nprocs=1, join=True)
        assert dict(ret) == {0: "ok"}


# ---------------------------------------------------------------------------------------------------------------------
# round 3: the library's own exchange (pdlp_iterate on a sharded handle) with 2 and 3 ranks, sharded Ruiz, sharded mixed + Ruiz
# ---------------------------------------------------------------------------------------------------------------------
FAKE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl")


def _fake_rccl():
    """path of the test-only librccl stand-in (tests/fake_rccl: ranks sharing one GPU, staging through shared memory), built on
    first use -- real RCCL refuses two ranks on one device, and this box has one"""
    import subprocess
    so, src = os.path.join(FAKE_DIR, "libfake_rccl.so"), os.path.join(FAKE_DIR, "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["bash", os.path.join(FAKE_DIR, "build.sh")])
    return so


def _ordered_comm():
    """a Comm whose all-reduce adds the ranks' contributions in rank order (as the stand-in library does), so that the
    torch.distributed loop and the library path see the same bits also with three ranks (gloo's ring adds in another order)"""
    from torchpdlp_amd.engine import Comm

    class OrderedComm(Comm):
        def all_reduce_sum(self, t, op=None):
            if op is not None and op != self.dist.ReduceOp.SUM:
                return super().all_reduce_sum(t, op)
            host = t.detach().cpu()
            parts = [torch.zeros_like(host) for _ in range(self.world)]
            self.dist.all_gather(parts, host, group=self.group)
            acc = parts[0].clone()
            for p in parts[1:]:
                acc += p
            t.copy_(acc)
    return OrderedComm()


def _libcomm_worker(rank, world, port, ret, tiled):
    """every sharded engine twice -- driven by the torch.distributed loop (engine.py) and by the library's own exchange
    (pdlp_iterate -> iterate_sharded, csrc/pdlp_hip.hip) over the stand-in librccl -- from the same state: same bits"""
    os.environ["PDLP_TILED"] = "1" if tiled else "0"
    if tiled:
        os.environ["PDLP_TILE_LW"] = "13"       # 8K-column panels: ~40 panels, so that a chunked exchange has panels in every piece
    import torchpdlp_amd as tp
    from torchpdlp_amd import _native as N
    from torchpdlp_amd.distributed import gather_solution, shard_engine
    from torchpdlp_amd.solver import run_pdlp
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        fake = _fake_rccl() if rank == 0 else None
        dist.barrier()
        fake = _fake_rccl()
        comm = _ordered_comm()
        if tiled:        # wide enough for several 64K-column panels per rank: the products are split around the exchange
            lp = gen_lp(330_000, 300_000, 4, seed=12, device=dev, recipe="mixed", dtype=torch.float64)
        else:            # sizes that do not divide by the world size
            lp = gen_lp(301, 403, 4, seed=21, recipe="mixed", ineq_frac=0.6, device=dev, dtype=torch.float64)
        g = torch.Generator().manual_seed(3)
        x0 = torch.randn(lp.n, generator=g, dtype=torch.float64).to(dev)
        y0 = torch.randn(lp.m, generator=g, dtype=torch.float64).to(dev)
        rough = torch.randn(lp.val.numel(), generator=g, dtype=torch.float64).to(dev) * 1e-9       # makes entries that are no float32 numbers
        configs = [("f32", torch.float32, None, False), ("f64", torch.float64, None, False), ("mixed", torch.float32, torch.float64, False)]
        if not tiled:
            configs.append(("mixed+exact", torch.float64, torch.float64, True))
        elif world > 2:          # (ranks sharing one card take turns on it: every synchronisation costs a time slice)
            configs = [configs[0], configs[2]]
        n1, n2 = (9, 4) if not tiled else (3, 2)
        for name, mat_dt, vec_dt, exact in configs:
            vd = vec_dt or mat_dt
            val = lp.val + rough if exact else lp.val
            K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, val.to(mat_dt))
            vecs = [t.to(vd) for t in (lp.c, lp.q, lp.l, lp.u)]

            def mk():
                if exact:
                    return shard_engine(K.to(dtype=torch.float32), *vecs, lp.m_ineq, comm, vec_dtype=vd, exact=K)
                return shard_engine(K, *vecs, lp.m_ineq, comm, vec_dtype=vec_dt)
            eA, eB = mk(), mk()
            assert eB.enable_library_comm(rccl_path=fake, timeout=60.0), (name, eB.lib_comm_log)
            assert eB.lib_comm and not eA.lib_comm and any("bit-identical" in s for s in eB.lib_comm_log)
            if tiled:
                assert all(t is not None for t in eB.tiles) and eB.split_info(0)["local_groups"] >= 1
            whole = {}
            trace = (lambda *a: print(f"[rank {rank}]", *a, flush=True)) if os.environ.get("PDLP_TEST_TRACE") else (lambda *a: None)
            # (CSR kernels never split a product, but the exchange may still travel in pieces: same collectives on every rank)
            # (suite budget: ranks sharing one card take turns on it.  Every precision meets the one-piece exchange on both step rules;
            #  the chunked exchange is covered by f32 on 2 and 3 ranks (3 pieces on 2 ranks) and by mixed precision on 2 ranks)
            if tiled:
                chunk_list = {"f32": (1, 2, 3) if world == 2 else (1, 2), "mixed": (1, 2) if world == 2 else (1,)}.get(name, (1,))
            else:
                chunk_list = (1, 2) if name == "f32" else (1,)
            for chunks in chunk_list:
                trace(name, "chunks", chunks)
                # chunked exchange (tiled products only): the gathered vector travels in `chunks` pieces and the panels a piece
                # completes are multiplied while the next piece is on the wire -- again the same bits on both drivers, and the
                # same numbers as the one-piece exchange up to the grouping of the partial row sums
                if chunks > 1:
                    for e in (eA, eB):
                        e.set_exchange_chunks(chunks)
                        for tr in (0, 1):
                            plan = e.exchange_plan(tr)
                            blk = e.ml if tr else e.nl
                            want = chunks if blk >= 64 * chunks else 1        # (pieces are at least 64 elements: tiny blocks travel whole)
                            assert len(plan) == want and plan[0][0] == 0 and plan[-1][1] == blk, (plan, blk)
                            assert all(plan[c][1] == plan[c + 1][0] and plan[c][0] % 64 == 0 for c in range(want - 1))
                            info = e.split_info(tr)
                            assert (info["local_groups"] >= 1 and info["other_groups"] >= want) if tiled else info["local_groups"] == 0, info
                for adaptive in ((True, False) if chunks == 1 else (True,)):
                    outs = []
                    for e in (eA, eB):
                        trace(name, chunks, "adaptive" if adaptive else "fixed", "library driver" if e is eB else "torch loop")
                        e.set_iterate(e.part.pad_cols(x0.to(vd))[e.cols[0]:e.cols[1]], e.part.pad_rows(y0.to(vd))[e.rows[0]:e.rows[1]])
                        e.set_step(0.02, 1.1, 1.0, 0)
                        e.iterate(n1, adaptive)
                        e.iterate(n2, adaptive)
                        x, y = e.get_iterate(N.CUR)
                        outs.append((x.clone(), y.clone(), e.scalars()["eta"], e.kkt(N.CUR, 1.0)["kkt"]))
                    (xa, ya, ea, ka), (xb, yb, eb, kb) = outs
                    assert torch.equal(xa, xb) and torch.equal(ya, yb), (name, adaptive, chunks)
                    assert ea == eb and ka == kb, (name, adaptive, chunks, ea, eb, ka, kb)
                    assert bool(torch.isfinite(xa).all()) and float(xa.abs().sum()) > 0
                    if chunks == 2 and tiled and name == "f32":
                        # producer side of the chunked exchange (round 5): the result of a split product leaves piece by piece, piece
                        # c's all-gather behind the rows it is made of.  The same products as the half-step that finishes first, up to
                        # the grouping of the partial row sums (a piece's launch covers fewer row blocks and takes more panel groups)
                        assert eA.producer_pieces and eB.producer_pieces
                        res = []
                        for on in (True, False):
                            eA.set_producer_pieces(on)
                            eA.set_iterate(eA.part.pad_cols(x0.to(vd))[eA.cols[0]:eA.cols[1]], eA.part.pad_rows(y0.to(vd))[eA.rows[0]:eA.rows[1]])
                            eA.set_step(0.02, 1.1, 1.0, 0)
                            eA.iterate(n1 + n2, False)
                            res.append([t.clone() for t in eA.get_iterate(N.CUR)])
                        eA.set_producer_pieces(True)
                        for tr in (0, 1):
                            assert eA.split_info(tr)["other_groups"] >= 2
                        np.testing.assert_allclose(res[0][0].cpu().numpy(), res[1][0].cpu().numpy(), rtol=2e-5, atol=2e-5)
                        np.testing.assert_allclose(res[0][1].cpu().numpy(), res[1][1].cpu().numpy(), rtol=2e-5, atol=2e-5)
                        assert not torch.equal(res[0][0], torch.zeros_like(res[0][0]))
                    if chunks == 1:
                        whole[adaptive] = (xa, ya, ka)
                    else:
                        tol = {"f32": 2e-4, "f64": 1e-9}.get(name, 2e-5)       # (mixed: float32 sums of difference products under float64 vectors)
                        np.testing.assert_allclose(xa.cpu().numpy(), whole[adaptive][0].cpu().numpy(), rtol=tol, atol=tol)
                        np.testing.assert_allclose(ya.cpu().numpy(), whole[adaptive][1].cpu().numpy(), rtol=tol, atol=tol)
                        np.testing.assert_allclose(ka, whole[adaptive][2], rtol=10 * tol)
            for e in (eA, eB):
                e.set_exchange_chunks(1)
            if not tiled and name == "f32":      # (the sharded mixed-precision solve to 1e-8 is test_sharded_ruiz_and_config4_mixed_precision's)
                # a whole restarted solve: identical restart decisions, counters and solution on both paths
                sols = []
                for e in (eA, eB):
                    b0 = e.part.pad_cols(torch.randn(lp.n, generator=torch.Generator().manual_seed(9)).to(dev))
                    tr = dict(kkt=[], omega=[], restarts=[])
                    x, obj, k, n, j, st, _ = run_pdlp(e, tol=1e-4 if name == "f32" else 1e-7, verbose=False, primal_update=True,
                                                      adaptive=True, b0=b0, trace=tr, max_kkt=300_000)
                    sols.append((gather_solution(e, x, lp.n).cpu(), obj, k, n, j, st, tr["restarts"]))
                assert sols[0][1:] == sols[1][1:] and torch.equal(sols[0][0], sols[1][0]), name
                assert sols[0][5] == "Solved" and abs(sols[0][1] - lp.opt_obj) <= 2e-3 * (1 + abs(lp.opt_obj))
            del eA, eB
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,tiled", [(2, False), (3, False), (2, True), (3, True)])
def test_library_exchange_equals_the_torch_distributed_loop(world, tiled):
    """iterate_sharded had never run with more than one rank.  2 and 3 ranks share the card; the library's
    communicator is the test-only stand-in (tests/fake_rccl).  f32, f64, mixed/delta (+ a separate exact matrix), adaptive and
    fixed step, plain and split (tiled) products."""
    port = _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_libcomm_worker, args=(world, port, ret, tiled), nprocs=world, join=True)
        assert dict(ret) == {r: "ok" for r in range(world)}


def _peer_worker(rank, world, port, ret, tiled):
    """every sharded engine twice -- driven by the torch.distributed loop and by the direct exchange (pdlp_peer_*: the half-steps
    store their blocks straight into the other ranks' workspaces over HIP IPC, flags in a mailbox, no collective in the iteration)
    -- from the same state: same bits (the loop's all-reduce adds in rank order here, as the direct exchange does)"""
    os.environ["PDLP_TILED"] = "1" if tiled else "0"
    if tiled:
        os.environ["PDLP_TILE_LW"] = "13"
    import torchpdlp_amd as tp
    from torchpdlp_amd import _native as N
    from torchpdlp_amd.distributed import gather_solution, shard_engine
    from torchpdlp_amd.solver import run_pdlp
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        comm = _ordered_comm()
        if tiled:
            lp = gen_lp(330_000, 300_000, 4, seed=12, device=dev, recipe="mixed", dtype=torch.float64)
        else:
            lp = gen_lp(301, 403, 4, seed=21, recipe="mixed", ineq_frac=0.6, device=dev, dtype=torch.float64)
        g = torch.Generator().manual_seed(3)
        x0 = torch.randn(lp.n, generator=g, dtype=torch.float64).to(dev)
        y0 = torch.randn(lp.m, generator=g, dtype=torch.float64).to(dev)
        rough = torch.randn(lp.val.numel(), generator=g, dtype=torch.float64).to(dev) * 1e-9
        configs = [("f32", torch.float32, None, False), ("mixed", torch.float32, torch.float64, False)]
        if not tiled:
            configs += [("f64", torch.float64, None, False), ("mixed+exact", torch.float64, torch.float64, True)]
        elif world > 2:          # (suite budget: ranks sharing one card take turns on it; mixed / delta on tiles runs with 2 ranks)
            configs = configs[:1]
        n1, n2 = (9, 4) if not tiled else (3, 2)
        for name, mat_dt, vec_dt, exact in configs:
            vd = vec_dt or mat_dt
            val = lp.val + rough if exact else lp.val
            K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, val.to(mat_dt))
            vecs = [t.to(vd) for t in (lp.c, lp.q, lp.l, lp.u)]

            def mk():
                if exact:
                    return shard_engine(K.to(dtype=torch.float32), *vecs, lp.m_ineq, comm, vec_dtype=vd, exact=K)
                return shard_engine(K, *vecs, lp.m_ineq, comm, vec_dtype=vec_dt)
            eA, eB = mk(), mk()
            assert eB.enable_peer_exchange(local_first=True), (name, eB.peer_log)
            st = eB.peer_status()
            assert eB.peer_on and st["connected"] and st["enabled"] and st["gave_up_on"] is None and st["exchanges"] > 0, st
            assert any("bit-identical" in s for s in eB.peer_log), eB.peer_log
            assert not eA.peer_on and not eA.peer_status()["connected"]
            if tiled:
                assert all(t is not None for t in eB.tiles) and eB.split_info(0)["local_groups"] >= 1
            loop_result = {}
            for adaptive in (True, False):
                outs = []
                for e in (eA, eB):
                    e.set_iterate(e.part.pad_cols(x0.to(vd))[e.cols[0]:e.cols[1]], e.part.pad_rows(y0.to(vd))[e.rows[0]:e.rows[1]])
                    e.set_step(0.02, 1.1, 1.0, 0)
                    before = e.peer_status()["exchanges"]
                    e.iterate(n1, adaptive)
                    e.iterate(n2, adaptive)
                    x, y = e.get_iterate(N.CUR)
                    outs.append((x.clone(), y.clone(), e.scalars()["eta"], e.kkt(N.CUR, 1.0)["kkt"]))
                    # two exchanges per iteration and one entry handshake per call -- and none on the engine that was not connected
                    assert e.peer_status()["exchanges"] - before == (2 * (n1 + n2) + 2 if e is eB else 0)
                (xa, ya, ea, ka), (xb, yb, eb, kb) = outs
                assert torch.equal(xa, xb) and torch.equal(ya, yb), (name, adaptive)
                assert ea == eb and ka == kb, (name, adaptive, ea, eb, ka, kb)
                assert bool(torch.isfinite(xa).all()) and float(xa.abs().sum()) > 0
                loop_result[adaptive] = (xa, ya, ea)
            # the push form: the epilogues store locally, a copy kernel on the side stream carries the block to the peers beside the
            # own block's panels -- the same split products as the loop's, so the same bits
            eB.set_peer_form(2)
            for adaptive in (True, False):
                eB.set_iterate(eB.part.pad_cols(x0.to(vd))[eB.cols[0]:eB.cols[1]], eB.part.pad_rows(y0.to(vd))[eB.rows[0]:eB.rows[1]])
                eB.set_step(0.02, 1.1, 1.0, 0)
                eB.iterate(n1, adaptive)
                eB.iterate(n2, adaptive)
                xp, yp = eB.get_iterate(N.CUR)
                eB._peer_check()
                xl, yl, el = loop_result[adaptive]
                assert torch.equal(xp, xl) and torch.equal(yp, yl) and eB.scalars()["eta"] == el, (name, adaptive, "push")
            eB.set_peer_form(0)
            # the default form of the direct exchange -- signal, wait, the WHOLE product: unsplit products, so for tiles the partial row
            # sums are grouped differently (CSR: the same bits again); both step rules
            for adaptive in (True, False):
                eB.set_iterate(eB.part.pad_cols(x0.to(vd))[eB.cols[0]:eB.cols[1]], eB.part.pad_rows(y0.to(vd))[eB.rows[0]:eB.rows[1]])
                eB.set_step(0.02, 1.1, 1.0, 0)
                eB.iterate(n1, adaptive)
                eB.iterate(n2, adaptive)
                xw, yw = eB.get_iterate(N.CUR)
                eB._peer_check()
                xl, yl, el = loop_result[adaptive]
                if tiled:
                    tol = 2e-4 if name == "f32" else 2e-5
                    np.testing.assert_allclose(xw.cpu().numpy(), xl.cpu().numpy(), rtol=tol, atol=tol)
                    np.testing.assert_allclose(yw.cpu().numpy(), yl.cpu().numpy(), rtol=tol, atol=tol)
                    np.testing.assert_allclose(eB.scalars()["eta"], el, rtol=1e-4)
                else:
                    assert torch.equal(xw, xl) and torch.equal(yw, yl) and eB.scalars()["eta"] == el, (name, adaptive)
            if not tiled and name == "f32":
                # a whole restarted solve: identical restart decisions, counters and solution on both drivers
                sols = []
                for e in (eA, eB):
                    b0 = e.part.pad_cols(torch.randn(lp.n, generator=torch.Generator().manual_seed(9)).to(dev))
                    tr = dict(kkt=[], omega=[], restarts=[])
                    x, obj, k, n, j, st, _ = run_pdlp(e, tol=1e-4, verbose=False, primal_update=True, adaptive=True, b0=b0, trace=tr,
                                                      max_kkt=300_000)
                    sols.append((gather_solution(e, x, lp.n).cpu(), obj, k, n, j, st, tr["restarts"]))
                assert sols[0][1:] == sols[1][1:] and torch.equal(sols[0][0], sols[1][0]), name
                assert sols[0][5] == "Solved" and abs(sols[0][1] - lp.opt_obj) <= 2e-3 * (1 + abs(lp.opt_obj))
            # switching the connected exchange off puts the engine back on the loop (same bits again), closing it disconnects
            eB.set_peer_exchange(False)
            eB.set_iterate(eB.part.pad_cols(x0.to(vd))[eB.cols[0]:eB.cols[1]], eB.part.pad_rows(y0.to(vd))[eB.rows[0]:eB.rows[1]])
            eB.set_step(0.02, 1.1, 1.0, 0)
            eB.iterate(n1, False)
            eB.iterate(n2, False)
            assert torch.equal(eB.get_iterate(N.CUR)[0], xb)
            dist.barrier()                       # (nobody unmaps while a peer may still be storing)
            eB.disable_peer_exchange()
            assert not eB.peer_status()["connected"]
            dist.barrier()
            del eA, eB
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,tiled", [(2, False), (3, False), (2, True), (3, True)])
def test_direct_exchange_equals_the_torch_distributed_loop(world, tiled):
    """VERDICT r4 item 1c (SURVEY section 8e "hand-rolled P2P copies over xGMI with IPC buffers"): 2 and 3 ranks share the card, each
    opens the others' workspaces over HIP IPC; f32, f64, mixed/delta (+ a separate exact matrix), adaptive and fixed step, plain
    and split (tiled) products, a whole solve"""
    port = _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_peer_worker, args=(world, port, ret, tiled), nprocs=world, join=True)
        assert dict(ret) == {r: "ok" for r in range(world)}


def _peer_timeout_worker(rank, world, port, ret):
    import torchpdlp_amd as tp
    from torchpdlp_amd import _native as N
    from torchpdlp_amd.distributed import shard_engine
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        comm = tp.Comm()
        lp = gen_lp(301, 403, 4, seed=21, recipe="mixed", ineq_frac=0.6, device=dev, dtype=torch.float32)
        K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
        eng = shard_engine(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, comm)
        assert eng.enable_peer_exchange(cross_check=False, timeout_ms=200), eng.peer_log
        eng.set_step(0.02, 1.0, 1.0, 0)
        dist.barrier()
        if rank == 0:                            # the other rank never comes: every wait gives up after 0.2 s, the grid drains
            eng.iterate(1, False)
            torch.cuda.synchronize()
            assert eng.peer_status()["gave_up_on"] == 1
            with pytest.raises(N.PdlpError, match="rank 1 did not signal"):
                eng._peer_check()
            with pytest.raises(N.PdlpError, match="did not signal"):
                eng.iterate(1, False)
        dist.barrier()
        eng.disable_peer_exchange()
        dist.barrier()
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_direct_exchange_wait_gives_up():
    """a peer that never signals: the wait kernel's spin is bounded, the stream drains, and the host refuses to go on"""
    port = _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_peer_timeout_worker, args=(2, port, ret), nprocs=2, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}


def _ruiz_worker(rank, world, port, ret):
    """sharded Ruiz: the scaled shards are the shards of the scaled matrix, bit for bit; then configs[4]'s combination
    (Ruiz + adaptive + primal weight, mixed precision) sharded over the ranks with no full copy anywhere"""
    import torchpdlp_amd as tp
    from torchpdlp_amd import _native as N
    from torchpdlp_amd.distributed import engine_from_shard, gather_solution, gen_lp_shard_arrays, shard_arrays
    from torchpdlp_amd.precondition import ruiz_precondition_shard
    from torchpdlp_amd.solver import run_pdlp
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        comm = tp.Comm()
        gz = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ruiz.npz"))
        cases = sorted({"/".join(k.split("/")[:3]) for k in gz.files})
        t = lambda v: torch.tensor(np.asarray(v), dtype=torch.float32, device=dev)
        for case in cases:        # the reference's own inputs (dense K), incl. an all-zero row / column and the Q3 early exit
            r = {k.split("/")[-1]: gz[k] for k in gz.files if k.startswith(case + "/")}
            iters = int(case.rsplit("it", 1)[1])
            K = tp.CsrPair.from_any(t(r["K"]), device=dev)
            m_ineq = K.m // 2
            full = tp.ruiz_precondition(t(r["c"]), K, t(r["q"]), t(r["l"]), t(r["u"]), device=dev, max_iter=iters)
            Ks, c_s, q_s, l_s, u_s, (D_col, D_row, *_), _ = full
            for balance in ("rows", "nnz"):
                sh = shard_arrays(K, t(r["c"]), t(r["q"]), t(r["l"]), t(r["u"]), m_ineq, rank, world, balance=balance)
                part = sh["part"]
                got = ruiz_precondition_shard({k: v for k, v in sh.items() if k != "part"}, comm, max_iter=iters)
                want = shard_arrays(Ks, c_s, q_s, l_s, u_s, m_ineq, rank, world, d_col=D_col, d_row=D_row, balance=balance, part=part)
                for key in ("K_rows", "KT_rows"):
                    for a, b in zip(got[key], want[key]):
                        assert torch.equal(a, b.to(a.dtype)), (case, balance, key)
                for key in ("c", "q", "l", "u", "d_col", "d_row"):
                    assert torch.equal(got[key], want[key]), (case, balance, key)
        # a generated instance that exists only as shards: (a) against the single-process Ruiz of the whole instance
        n, m, k = 500_003, 1_000_001, 6         # (1M constraints, sizes that divide by no world size: padded blocks)
        sh = gen_lp_shard_arrays(n, m, k, 9, comm, dev, torch.float64)
        got = ruiz_precondition_shard({kk: v for kk, v in sh.items() if kk not in ("part", "nnz_local")}, comm)
        lp = gen_lp(n, m, k, seed=9, device=dev, dtype=torch.float64)
        K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
        Ks, c_s, q_s, l_s, u_s, (D_col, D_row, *_), _ = tp.ruiz_precondition(lp.c, K, lp.q, lp.l, lp.u, device=dev)
        want = shard_arrays(Ks, c_s, q_s, l_s, u_s, lp.m_ineq, rank, world, d_col=D_col, d_row=D_row, balance="rows")
        for key in ("K_rows", "KT_rows"):
            assert torch.equal(got[key][2], want[key][2]) and torch.equal(got[key][1], want[key][1]), key
        for key in ("c", "q", "l", "u", "d_col", "d_row"):
            assert torch.equal(got[key], want[key]), key
        assert got["ruiz_sweeps"] >= 1
        del got, want, Ks, K, lp, sh
        if world > 2:      # (suite budget: the sharded 1e-8 solve, the single-rank comparison and the entry points run on 2 ranks)
            ret[rank] = "ok"
            return
        # (b) the whole configs[4] combination -- Ruiz + adaptive + primal weight, mixed precision, 1e-8 -- on an instance that exists
        # only as shards (small: every iteration of this rehearsal crosses the host three times over gloo).  The scaled entries are
        # no float32 numbers, so every rank iterates on the float32 rounding and refreshes the anchors from its float64 blocks.
        n, m, k = 3000, 5000, 6
        sh = gen_lp_shard_arrays(n, m, k, 9, comm, dev, torch.float64)
        eng = engine_from_shard(sh, comm, precision="mixed", precondition=True)
        assert eng.mixed and eng.delta and eng.exact is not None and eng.d_col is not None and eng.ruiz_sweeps >= 1
        TOL = 1e-8
        x, obj, kk, nn, jj, status, _ = run_pdlp(eng, tol=TOL, verbose=False, precondition=True, primal_update=True, adaptive=True,
                                                 seed=2, max_kkt=2_000_000, time_limit=240)
        assert status == "Solved", status
        xs = gather_solution(eng, x, n)                       # scaled iterate (quirk Q4); un-scale with the gathered D_col
        dfull = torch.empty(eng.n, dtype=torch.float64, device=dev)
        dfull[eng.cols[0]:eng.cols[1]] = eng.d_col
        comm.all_gather(dfull)
        xu = xs * eng.part.unpad_cols(dfull)
        # float64 check against the ORIGINAL problem, independent of the engine: feasibility and objective
        lp = gen_lp(n, m, k, seed=9, device=dev, dtype=torch.float64)
        K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
        rows = torch.repeat_interleave(torch.arange(lp.m, device=dev), (lp.rowptr[1:] - lp.rowptr[:-1]).long())
        kx = torch.zeros(lp.m, dtype=torch.float64, device=dev).index_add_(0, rows, lp.val * xu[lp.colidx.long()])
        res = kx - lp.q
        res[:lp.m_ineq].clamp_(max=0)
        assert float(res.norm()) <= 1.01 * TOL * (1 + float(lp.q.norm()))
        assert float((xu - lp.l).clamp(max=0).abs().max()) <= 1e-9 and float((lp.u - xu).clamp(max=0).abs().max()) <= 1e-9
        assert abs(float((lp.c * xu).sum()) - obj) <= 1e-9 * (1 + abs(obj))
        if rank == 0:      # the same solve on one rank reaches the same optimum
            e1 = engine_from_shard(dict(shard_arrays(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, 0, 1, balance="rows")), None,
                                   precision="mixed", precondition=True)
            _, obj1, *_rest, st1, _ = run_pdlp(e1, tol=TOL, verbose=False, precondition=True, primal_update=True, adaptive=True,
                                               seed=2, max_kkt=2_000_000, time_limit=240)
            assert st1 == "Solved" and abs(obj1 - obj) <= 20 * TOL * (1 + abs(obj))
        # the user-facing entry points with the combination ADVICE r2 reported as refused: mixed + precondition, sharded, incl. a
        # matrix on which Ruiz is the identity (+-1 entries: the scaled matrix IS float32-valued, no separate exact matrix)
        lp2 = gen_lp(301, 403, 4, seed=21, recipe="mixed", ineq_frac=0.6, device=dev, dtype=torch.float64)
        K2 = tp.CsrPair(lp2.m, lp2.n, lp2.rowptr, lp2.colidx, lp2.val)
        res = tp.solve_lp((lp2.c, K2, lp2.q, lp2.m_ineq, lp2.l, lp2.u), device=dev, tol=1e-8, precondition=True, primal_weight_update=True,
                          adaptive_stepsize=True, seed=1, comm=True, precision="mixed", max_kkt=2_000_000)
        assert res.status == "Solved" and abs(res.objective - lp2.opt_obj) <= 1e-6 * (1 + abs(lp2.opt_obj))
        afiro = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mps", "afiro.mps")
        res = tp.solve_lp(afiro, device=dev, tol=1e-8, precondition=True, primal_weight_update=True, adaptive_stepsize=True, seed=3,
                          comm=True, precision="mixed", max_kkt=2_000_000)
        assert res.status == "Solved" and abs(res.objective - (-464.7531428571)) <= 1e-7 * (1 + 2 * 464.7531428571), \
            (res.status, res.objective, res.iterations)      # (the tests allow 1e-8 (1 + |p| + |d|) of gap, signed, plus the residuals' share)
        Kpm = tp.CsrPair(lp2.m, lp2.n, lp2.rowptr, lp2.colidx, torch.sign(lp2.val) + (lp2.val == 0))
        res = tp.solve_lp((lp2.c, Kpm, lp2.q, lp2.m_ineq, lp2.l, lp2.u), device=dev, tol=1e-6, precondition=True, primal_weight_update=True,
                          adaptive_stepsize=True, seed=1, comm=True, precision="mixed", max_kkt=3_000, time_limit=120)
        assert res.status in ("Solved", "Unsolved (KKT passes limit exceeded)") and res.x.shape == (lp2.n, 1)   # (it used to raise)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ruiz_and_config4_mixed_precision(world):
    port = _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_ruiz_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {r: "ok" for r in range(world)}


def _infeas_delta_worker(rank, world, port, ret):
    """the detector on a sharded mixed/delta engine multiplied K' by stale blocks of y (the delta iterations exchange
    only differences).  Now y is gathered first: same diagnostics as one rank."""
    import torchpdlp_amd as tp
    from torchpdlp_amd.distributed import shard_engine
    _init(rank, world, port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        lp = gen_lp(301, 403, 4, seed=21, recipe="mixed", ineq_frac=0.6, device=dev, dtype=torch.float64)
        K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val.float())
        comm = tp.Comm()

        def run(e):
            e.set_iterate(torch.zeros(e.nl, dtype=torch.float64, device=dev), torch.zeros(e.ml, dtype=torch.float64, device=dev))
            e.set_step(0.05, 1.0, 1.0, 0)
            e.infeas_reset()
            out = []
            for tol in (1e-2, 1e-2, 1e3):
                e.iterate(2, True)
                out.append(e.detect_infeasibility(tol, diagnostics=True))
            return out
        es = shard_engine(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, comm, vec_dtype=torch.float64)
        assert es.delta and es.comm is not None
        got = run(es)
        if rank == 0:
            e1 = tp.PdlpEngine.from_full(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, vec_dtype=torch.float64)
            for (s, d), (s1, d1) in zip(got, run(e1)):
                assert s == s1
                np.testing.assert_allclose(d, d1, rtol=2e-5, atol=1e-5)      # (float32 difference products: summation order; stale y is O(1) off)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_delta_mode_infeasibility_detector_sees_the_whole_y():
    port = _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_infeas_delta_worker, args=(2, port, ret), nprocs=2, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}


def _agree_worker(rank, world, port, ret):
    from torchpdlp_amd.distributed import agree_failed
    _init(rank, world, port)
    try:
        got = [agree_failed(dist, rank, world, False, "none", 20.0),                   # nobody failed
               agree_failed(dist, rank, world, True, "all", 20.0),                     # every rank alike (a bad option, a bad instance)
               agree_failed(dist, rank, world, rank == 1, "one", 20.0)]                # rank 1 alone: everybody learns it
        # a rank that fails while its peer sits in a collective: the peer never posts, the failing rank gets -1 after the timeout and
        # must not have issued a collective of its own (rank 0 stands in for the blocked peer by simply not calling)
        if rank == 1:
            got.append(agree_failed(dist, rank, world, True, "stuck_peer", 1.0))
        ret[rank] = got
    finally:
        dist.destroy_process_group()


def test_failure_agreement_is_out_of_band_and_times_out():
    """the CLI's per-instance failure count must not be a collective on the solve's process group"""
    port = _free_port()
    with mp.Manager() as man:
        ret = man.dict()
        mp.spawn(_agree_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret[0] == [0, 2, 1] and ret[1] == [0, 2, 1, -1]
