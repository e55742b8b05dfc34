"""Sharding one LP over the GPUs of a node: one process per GPU, row blocks of K and of K'.

Rank r owns a block of constraints (rows of K, with y and q) and a block of variables (rows of K', with x, c, l, u).
Each half-step is local except that it gathers from a vector the other half-step produced, so the exchange is one
all-gather of xbar before K xbar and one all-gather of y before K'y (RCCL over xGMI when the backend is ``nccl``), plus
an 8-double all-reduce for the step-size rule and for each KKT evaluation.  Against the alternative in the north star
(replicated x and an all-reduce of K'y partial sums) this moves the same bytes per iteration ((P-1)/P (n+m) values per
GPU) but needs no second summation pass and no redundant primal update.

Blocks are chosen by NON-ZEROS (``balance="nnz"``, SURVEY 8e: real instances are skewed) or by row count
(``balance="rows"``).  Either way every rank's block is padded to the same length ``B`` and the index space is
re-laid-out as ``world`` slots of ``B``: original row ``i`` of block ``g`` becomes ``g*B + (i - first row of g)``, and
the column indices of both matrices are remapped the same way at setup.  The collectives then always move equal
shards (one ``all_gather_into_tensor``, no staging copies), and the kernels never see the difference: padding
variables are fixed at 0 (``l = u = c = 0``, empty column) and padding constraints are empty equality rows with
``q = 0``; neither changes any iterate, residual or norm.  With ``balance="rows"`` the re-layout is the identity.

``gen_lp_shard`` + ``sharded_transpose`` build one rank's shard of the synthetic bench LP without any rank ever holding
the whole instance: a rank generates its own rows of K (the generator is seeded per 2^16-row chunk), the entries are
exchanged by column block (all-to-all) and sorted into the rank's rows of K'.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from .engine import Comm, PdlpEngine
from .sparse import CsrPair, _counts_to_rowptr, as_vec


def padded(v: int, world: int) -> int:
    return (v + world - 1) // world * world


def block(v_padded: int, rank: int, world: int) -> Tuple[int, int]:
    b = v_padded // world
    return rank * b, (rank + 1) * b


@dataclass
class Partition:
    """blocks of the constraints (rows of K) and of the variables (rows of K') and the padded index space"""
    world: int
    m: int
    n: int
    rb: List[int]            # [world+1] first constraint of every block (original indices)
    cb: List[int]            # [world+1] first variable of every block
    Bm: int                  # padded block lengths
    Bn: int

    @property
    def mp(self) -> int:
        return self.world * self.Bm

    @property
    def np_(self) -> int:
        return self.world * self.Bn

    def _map(self, bounds, B, total, device):
        idx = torch.arange(total, device=device)
        g = torch.bucketize(idx, torch.tensor(bounds[1:-1], device=device), right=True) if self.world > 1 else torch.zeros_like(idx)
        return g * B + (idx - torch.tensor(bounds, device=device)[g])

    def row_map(self, device) -> torch.Tensor:
        """[m] original constraint -> padded index"""
        return self._map(self.rb, self.Bm, self.m, device)

    def col_map(self, device) -> torch.Tensor:
        return self._map(self.cb, self.Bn, self.n, device)

    def pad_cols(self, v: torch.Tensor, fill: float = 0.0) -> torch.Tensor:
        """a length-n vector in the padded layout (length world*Bn)"""
        out = torch.full((self.np_,), fill, dtype=v.dtype, device=v.device)
        out[self.col_map(v.device)] = v
        return out

    def pad_rows(self, v: torch.Tensor, fill: float = 0.0) -> torch.Tensor:
        out = torch.full((self.mp,), fill, dtype=v.dtype, device=v.device)
        out[self.row_map(v.device)] = v
        return out

    def unpad_cols(self, full: torch.Tensor) -> torch.Tensor:
        return full[self.col_map(full.device)]

    def unpad_rows(self, full: torch.Tensor) -> torch.Tensor:
        return full[self.row_map(full.device)]

    def rows(self, rank: int) -> Tuple[int, int]:
        """this rank's slot in the padded constraint space"""
        return rank * self.Bm, (rank + 1) * self.Bm

    def cols(self, rank: int) -> Tuple[int, int]:
        return rank * self.Bn, (rank + 1) * self.Bn


def _bounds_by_rows(total: int, world: int) -> List[int]:
    b = (total + world - 1) // world
    return [min(g * b, total) for g in range(world + 1)]


def _bounds_by_nnz(rowptr: torch.Tensor, total: int, world: int) -> List[int]:
    """boundaries so that every block holds about nnz/world non-zeros: block g starts at the first row whose prefix count
    reaches g*nnz/world (a single row heavier than that share stays whole, of course)"""
    rp = rowptr.long()
    nnz = int(rp[-1])
    if nnz == 0 or total == 0:
        return _bounds_by_rows(total, world)
    targets = torch.arange(1, world, device=rp.device, dtype=torch.float64) * (nnz / world)
    cuts = torch.searchsorted(rp[:-1].double(), targets, right=False).tolist()
    b = [0] + [min(max(int(c), 0), total) for c in cuts] + [total]
    for g in range(1, world + 1):                      # monotone
        b[g] = max(b[g], b[g - 1])
    return b


def make_partition(K: CsrPair, world: int, balance: str = "nnz") -> Partition:
    if balance == "rows":
        rb, cb = _bounds_by_rows(K.m, world), _bounds_by_rows(K.n, world)
    elif balance == "nnz":
        rb, cb = _bounds_by_nnz(K.rowptr, K.m, world), _bounds_by_nnz(K.t_rowptr, K.n, world)
    else:
        raise ValueError(f"balance must be 'nnz' or 'rows', not {balance!r}")
    Bm = max(1, max(rb[g + 1] - rb[g] for g in range(world)))
    Bn = max(1, max(cb[g + 1] - cb[g] for g in range(world)))
    return Partition(world, K.m, K.n, rb, cb, Bm, Bn)


def _block_rows(rowptr, colidx, val, lo: int, hi: int, B: int, index_map: torch.Tensor):
    """rows [lo, hi) of a CSR matrix as a block of B rows (empty ones appended) with remapped column indices"""
    a, b = int(rowptr[lo]), int(rowptr[hi])
    rp = (rowptr[lo:hi + 1].long() - a)
    if B > hi - lo:
        rp = torch.cat([rp, rp[-1:].expand(B - (hi - lo))])
    ci = index_map[colidx[a:b].long()].to(torch.int32)
    return rp.contiguous(), ci.contiguous(), val[a:b].contiguous()


def _block_vec(v: Optional[torch.Tensor], lo: int, hi: int, B: int, fill: float = 0.0):
    if v is None:
        return None
    out = torch.full((B,), fill, dtype=v.dtype, device=v.device)
    out[:hi - lo] = v[lo:hi]
    return out


def shard_arrays(K: CsrPair, c, q, l, u, m_ineq: int, rank: int, world: int, d_col=None, d_row=None, vec_dtype=None,
                 balance: str = "rows", part: Optional[Partition] = None) -> dict:
    """This rank's blocks of a problem held in full: keyword arguments for ``PdlpEngine`` (minus ``comm``), plus ``part``."""
    part = make_partition(K, world, balance) if part is None else part
    r = rank
    m, n = K.m, K.n
    dev, dt = K.device, (K.dtype if vec_dtype is None else vec_dtype)
    vec = lambda v, ln: None if v is None else as_vec(v, ln, dev, dt)
    rmap, cmap = part.row_map(dev), part.col_map(dev)
    r_lo, r_hi, c_lo, c_hi = part.rb[r], part.rb[r + 1], part.cb[r], part.cb[r + 1]
    K_rows = _block_rows(K.rowptr, K.colidx, K.val, r_lo, r_hi, part.Bm, cmap)
    KT_rows = _block_rows(K.t_rowptr, K.t_colidx, K.t_val, c_lo, c_hi, part.Bn, rmap)
    # the library takes "the first m_ineq constraints are inequalities" against this rank's first padded row: the block keeps
    # the original order, so its inequalities are still a prefix -- express their count relative to the padded first row
    row0 = r * part.Bm
    local_ineq = max(0, min(r_hi, int(m_ineq)) - r_lo)
    return dict(m=part.mp, n=part.np_, m_ineq=row0 + local_ineq, K_rows=K_rows, KT_rows=KT_rows,
                c=_block_vec(vec(c, n), c_lo, c_hi, part.Bn), q=_block_vec(vec(q, m), r_lo, r_hi, part.Bm),
                l=_block_vec(vec(l, n), c_lo, c_hi, part.Bn), u=_block_vec(vec(u, n), c_lo, c_hi, part.Bn),
                rows=part.rows(r), cols=part.cols(r), d_col=_block_vec(vec(d_col, n), c_lo, c_hi, part.Bn, 1.0),
                d_row=_block_vec(vec(d_row, m), r_lo, r_hi, part.Bm, 1.0), part=part)


def shard_engine(K: CsrPair, c, q, l, u, m_ineq: int, comm: Optional[Comm], d_col=None, d_row=None, vec_dtype=None,
                 balance: str = "nnz", exact: Optional[CsrPair] = None) -> PdlpEngine:
    """Engine for this rank's block of a problem every rank holds in full (small/medium problems and tests; the benchmark
    and anything that does not fit one GPU use ``gen_lp_shard`` / per-rank loading instead).  ``eng.part`` maps between
    the original and the padded index space.  ``exact``: the float64 matrix of which ``K`` is the float32 rounding (mixed
    precision on a matrix that is not float32-valued); it is cut into the same blocks."""
    if comm is None or comm.world == 1:
        return PdlpEngine.from_full(K, c, q, l, u, m_ineq, d_col=d_col, d_row=d_row, vec_dtype=vec_dtype, exact=exact)
    args = shard_arrays(K, c, q, l, u, m_ineq, comm.rank, comm.world, d_col, d_row, vec_dtype, balance)
    part = args.pop("part")
    ex = None
    if exact is not None:
        ea = shard_arrays(exact, c, q, l, u, m_ineq, comm.rank, comm.world, None, None, vec_dtype, balance, part=part)
        ex = (ea["K_rows"], ea["KT_rows"])
    eng = PdlpEngine(comm=comm, vec_dtype=vec_dtype, exact=ex, **args)
    eng.part = part
    return eng


def engine_from_shard(shard: dict, comm: Optional[Comm], precision: Optional[str] = None, precondition: bool = False,
                      ruiz_iters: int = 20, ruiz_eps: float = 1e-6) -> PdlpEngine:
    """Engine over one rank's shard of a problem that exists only as shards (``gen_lp_shard_arrays``, ``shard_arrays``): no
    rank ever holds the whole LP, also not for the preconditioning -- ``precondition=True`` runs the sharded Ruiz sweeps
    (``precondition.ruiz_precondition_shard``: row norms are rank local, one all-gather of the factors per half-sweep) and the
    engine then carries ``d_col`` / ``d_row`` for the un-scaled termination test (pdhg.py:157-161).
    ``precision="mixed"``: float64 vectors over float32 matrix values (pass the shard in float64).  If some rank's scaled
    entries are not float32 numbers, every rank iterates on the float32 rounding and evaluates the anchors of delta mode with
    its float64 blocks (``exact``)."""
    from .engine import values_are_float32
    shard = dict(shard)
    extra = {k: shard.pop(k) for k in ("part", "nnz_local", "ruiz_seconds", "ruiz_sweeps") if k in shard}
    if precondition:
        from .precondition import ruiz_precondition_shard
        shard = ruiz_precondition_shard(shard, comm, ruiz_iters, ruiz_eps)
        extra["ruiz_seconds"], extra["ruiz_sweeps"] = shard.pop("ruiz_seconds"), shard.pop("ruiz_sweeps")
    vec_dtype, exact = None, None
    if precision is not None:
        if precision != "mixed":
            raise ValueError(f"unknown precision {precision!r}")
        K_rows, KT_rows = shard["K_rows"], shard["KT_rows"]
        dev = K_rows[2].device
        f32ok = torch.tensor([int(values_are_float32(K_rows[2]) and values_are_float32(KT_rows[2]))], dtype=torch.int32, device=dev)
        if comm is not None and comm.world > 1:
            comm.all_reduce_min(f32ok)                   # every rank must build the same kind of engine
        if not int(f32ok):
            exact = ((K_rows[0], K_rows[1], K_rows[2].double()), (KT_rows[0], KT_rows[1], KT_rows[2].double()))
        shard["K_rows"] = (K_rows[0], K_rows[1], K_rows[2].float())
        shard["KT_rows"] = (KT_rows[0], KT_rows[1], KT_rows[2].float())
        vec_dtype = torch.float64
    eng = PdlpEngine(comm=comm, vec_dtype=vec_dtype, exact=exact, **shard)
    for k, v in extra.items():
        setattr(eng, k, v)
    return eng


def gather_solution(eng: PdlpEngine, x_local: torch.Tensor, n_true: int) -> torch.Tensor:
    """the full primal vector in the ORIGINAL variable order on every rank (drops the padding)"""
    if eng.comm is None:
        return x_local
    full = torch.empty(eng.n, dtype=x_local.dtype, device=x_local.device)
    full[eng.cols[0]:eng.cols[1]] = x_local
    eng.comm.all_gather(full)
    part = getattr(eng, "part", None)
    return full[:n_true] if part is None else part.unpad_cols(full)


# ---------------------------------------------------------------------------------------------------------------------
# building a shard without ever holding the whole matrix
# ---------------------------------------------------------------------------------------------------------------------
def _exchange(buckets: List[torch.Tensor], comm: Comm) -> List[torch.Tensor]:
    """all-to-all of one tensor per destination rank (uneven sizes); returns one tensor per source rank"""
    dist, W = comm.dist, comm.world
    dev, dt = buckets[0].device, buckets[0].dtype
    counts = torch.tensor([int(b.numel()) for b in buckets], dtype=torch.int64)
    allc = [torch.zeros(W, dtype=torch.int64) for _ in range(W)]
    if comm.backend == "nccl":
        cdev = counts.to(dev)
        alld = [torch.zeros(W, dtype=torch.int64, device=dev) for _ in range(W)]
        dist.all_gather(alld, cdev, group=comm.group)
        allc = [t.cpu() for t in alld]
    else:
        dist.all_gather(allc, counts, group=comm.group)
    recv_counts = [int(allc[src][comm.rank]) for src in range(W)]
    if comm.backend == "nccl":
        send = torch.cat(buckets)
        out = torch.empty(sum(recv_counts), dtype=dt, device=dev)
        dist.all_to_all_single(out, send, recv_counts, [int(c) for c in counts], group=comm.group)
        return list(torch.split(out, recv_counts))
    # gloo (CPU rehearsal): pairwise sends through host memory
    outs = [torch.empty(rc, dtype=dt) for rc in recv_counts]
    outs[comm.rank] = buckets[comm.rank].cpu()
    reqs = []
    for shift in range(1, W):
        dst, src = (comm.rank + shift) % W, (comm.rank - shift) % W
        reqs.append(dist.isend(buckets[dst].cpu().contiguous(), dst, group=comm.group))
        reqs.append(dist.irecv(outs[src], src, group=comm.group))
    for rq in reqs:
        rq.wait()
    return [o.to(dev) for o in outs]


def sharded_transpose(rp: torch.Tensor, ci: torch.Tensor, va: torch.Tensor, row_offset: int, part: Partition, comm: Comm):
    """This rank holds rows [row_offset, row_offset + len) of K (``ci`` in PADDED column indices).  Returns this rank's block of
    K' -- rows = its ``Bn`` padded variables, column indices = padded constraint indices -- after an all-to-all of the entries
    by column block.  No rank ever sees more than its own rows and columns."""
    W, Bn = comm.world, part.Bn
    nrows = int(rp.numel()) - 1
    rows = torch.repeat_interleave(torch.arange(nrows, device=ci.device, dtype=torch.int64), (rp[1:] - rp[:-1]).long()) + row_offset
    dest = (ci.long() // Bn)
    order = torch.argsort(dest, stable=True)
    cnt = torch.bincount(dest, minlength=W).tolist()
    rs, cs, vs = rows[order].to(torch.int32), ci[order], va[order]
    del rows, dest, order
    got_r = _exchange(list(torch.split(rs, cnt)), comm)
    got_c = _exchange(list(torch.split(cs, cnt)), comm)
    got_v = _exchange(list(torch.split(vs, cnt)), comm)
    r_all, c_all, v_all = torch.cat(got_r).long(), torch.cat(got_c).long() - comm.rank * Bn, torch.cat(got_v)
    # rows of K' = local variable, sorted by (variable, constraint): sources arrive in rank order = ascending constraint blocks,
    # so a stable sort by variable keeps the constraints ascending inside a row
    order = torch.argsort(c_all, stable=True)
    t_rp = _counts_to_rowptr(torch.bincount(c_all, minlength=Bn))
    return t_rp, r_all[order].to(torch.int32).contiguous(), v_all[order].contiguous()


def gen_lp_shard_arrays(n: int, m: int, nnz_per_row: int, seed: int, comm: Comm, device, dtype=torch.float32, ineq_frac: float = 0.8,
                        vec_dtype=None) -> dict:
    """One rank's shard of ``gen_lp(n, m, nnz_per_row, seed, recipe="box")`` -- the same instance entry for entry -- built
    from this rank's own rows only: rows are generated per 2^16-row chunk from per-chunk seeds, K' comes from the
    distributed transpose.  Row-regular pattern => equal row blocks are also nnz-balanced.  Returns the keyword arguments of
    ``PdlpEngine`` (minus ``comm``) plus ``part``."""
    from .synthetic import box_rows, box_vectors
    W, r = comm.world, comm.rank
    part = Partition(W, m, n, _bounds_by_rows(m, W), _bounds_by_rows(n, W), max(1, -(-m // W)), max(1, -(-n // W)))
    r_lo, r_hi, c_lo, c_hi = part.rb[r], part.rb[r + 1], part.cb[r], part.cb[r + 1]
    col, val = box_rows(n, nnz_per_row, seed, r_lo, r_hi, device, dtype)          # identity column map (balance by rows)
    k = int(nnz_per_row)
    vec = box_vectors(n, m, seed, device, ineq_frac)                               # x_feas, slack, l, u, c: O(n + m), every rank
    m_ineq = vec["m_ineq"]
    kx = (val.double() * vec["x_feas"][col.long()]).view(r_hi - r_lo, k).sum(1)
    q = kx.clone()
    ni = max(0, min(r_hi, m_ineq) - r_lo)
    q[:ni] -= vec["slack"][r_lo:r_lo + ni]
    rp = torch.arange(0, (r_hi - r_lo + 1) * k, k, dtype=torch.int64, device=device)
    t_rp, t_ci, t_va = sharded_transpose(rp, col, val, r * part.Bm, part, comm)
    if part.Bm > r_hi - r_lo:
        rp = torch.cat([rp, rp[-1:].expand(part.Bm - (r_hi - r_lo))])
    vd = dtype if vec_dtype is None else vec_dtype
    bv = lambda v, lo, hi, B: _block_vec(v.to(vd), lo, hi, B)
    return dict(m=part.mp, n=part.np_, m_ineq=r * part.Bm + ni, K_rows=(rp, col, val), KT_rows=(t_rp, t_ci, t_va),
                c=bv(vec["c"], c_lo, c_hi, part.Bn), q=bv(q, 0, r_hi - r_lo, part.Bm), l=bv(vec["l"], c_lo, c_hi, part.Bn),
                u=bv(vec["u"], c_lo, c_hi, part.Bn), rows=part.rows(r), cols=part.cols(r), part=part, nnz_local=int(col.numel()))


def gen_lp_shard(n: int, m: int, nnz_per_row: int, seed: int, comm: Comm, device, dtype=torch.float32, ineq_frac: float = 0.8,
                 vec_dtype=None, precision: Optional[str] = None, precondition: bool = False) -> PdlpEngine:
    """engine over ``gen_lp_shard_arrays``: no rank ever holds the whole instance (``precision`` / ``precondition``:
    ``engine_from_shard``; mixed precision generates the shard in float64)"""
    if precision is not None:
        dtype, vec_dtype = torch.float64, None
    args = gen_lp_shard_arrays(n, m, nnz_per_row, seed, comm, device, dtype, ineq_frac, vec_dtype)
    if precision is None and not precondition:
        part = args.pop("part")
        nnz_local = args.pop("nnz_local")
        eng = PdlpEngine(comm=comm, vec_dtype=vec_dtype, **args)
        eng.part, eng.nnz_local = part, nnz_local
        return eng
    return engine_from_shard(args, comm, precision, precondition)


_agree_calls = 0


def agree_failed(dist, rank: int, world: int, failed: bool, tag: str, timeout_s: float = 60.0, seq: Optional[int] = None) -> int:
    """How many ranks failed at the point ``tag``; -1 if the ranks cannot agree within ``timeout_s``.  Out of band, over the process
    group's key-value store -- NOT a collective: a rank that raises in the middle of a solve has peers blocked inside the solve's own
    all-gathers / all-reduces, and a collective issued from its exception path would pair with one of those (undefined behaviour
    under RCCL: a hang or garbage).  Ranks that fail alike (an unsupported option, a bad instance: before any collective) post
    their flags at once; a rank that fails alone waits ``timeout_s`` for flags that never come and gets -1.
    The keys carry a sequence number (``seq``; default: this process's call count -- every rank calls once per agreement point, in
    the same order) so that a second run over the same instance names in one process group never reads the flags of the first;
    a rank deletes its own key once it has the answer."""
    import datetime
    global _agree_calls
    if seq is None:
        seq = _agree_calls
    _agree_calls += 1
    store = dist.distributed_c10d._get_default_store()
    key = lambda r: f"pdlp/fail/{int(seq)}/{tag}/{r}"
    keys = [key(r) for r in range(world)]
    try:
        store.set(key(rank), "1" if failed else "0")
        store.wait(keys, datetime.timedelta(seconds=timeout_s))
        bad = sum(int(store.get(k) == b"1") for k in keys)
    except Exception:
        return -1
    try:                                   # everybody has posted; every rank read all keys before deleting only after a second wait
        store.set(key(rank) + "/seen", "1")
        store.wait([k + "/seen" for k in keys], datetime.timedelta(seconds=timeout_s))
        store.delete_key(key(rank))
    except Exception:
        pass                               # (a store without delete, or a peer that left: the sequence number keeps later calls apart)
    return bad
