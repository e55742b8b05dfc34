"""PdlpEngine: one LP (or one rank's shard of it) resident on one MI355X, driven through the C ABI.

torch is used for storage (problem arrays, one workspace tensor) and, when the problem is sharded
over several GPUs, for the collectives between the half-steps (``torch.distributed`` backend
``nccl`` = RCCL over xGMI).  All arithmetic of the hot path happens in ``libpdlp_hip.so``.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import time
from typing import Optional, Tuple

import torch

from . import _native as N
from .sparse import CsrPair, as_vec
from . import tiled as _tiled

_DT = {torch.float32: N.PDLP_F32, torch.float64: N.PDLP_F64}


def values_are_float32(val: torch.Tensor) -> bool:
    """every entry of a float64 array is a float32 number (so the matrix can be held in float32 without changing the LP)"""
    return val.dtype == torch.float32 or bool((val.float().to(val.dtype) == val).all())


def exportable_bytes(nbytes: int) -> int:
    """size of a device allocation that can hold ``nbytes`` AND be opened by another process: a multiple of 2 MB (what torch's
    allocator makes of a large request anyway) without bit 31 -- hipIpcOpenMemHandle never returns for 2-4 GiB, 6-8 GiB, ... on
    ROCm 7.2 (tools/ipc_torch_probe.py) -- i.e. such sizes go up to the next multiple of 4 GiB"""
    size = -(-int(nbytes) // (2 << 20)) * (2 << 20)
    if size & 0x80000000:
        size = (size | 0xFFFFFFFF) + 1
    return size


class Comm:
    """One process per GPU.  Vectors are sharded in equal blocks (the LP is padded so the sizes divide)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    def all_gather(self, full: torch.Tensor):
        """full = concat over ranks of equal shards; this rank's shard is already in place."""
        shard = full.numel() // self.world
        mine = full[self.rank * shard:(self.rank + 1) * shard]
        if self.backend == "gloo" and full.is_cuda:     # test path: gloo has no device all-gather
            host = torch.empty(full.numel(), dtype=full.dtype)
            self.dist.all_gather_into_tensor(host, mine.cpu(), group=self.group)
            full.copy_(host)
        else:
            self.dist.all_gather_into_tensor(full, mine, group=self.group)

    def all_gather_async(self, full: torch.Tensor):
        """the same all-gather, not waited for: returns a handle whose ``wait()`` makes the current stream wait (None: already done)"""
        if self.backend == "gloo" and full.is_cuda:
            self.all_gather(full)
            return None
        shard = full.numel() // self.world
        return self.dist.all_gather_into_tensor(full, full[self.rank * shard:(self.rank + 1) * shard], group=self.group, async_op=True)

    def all_gather_piece(self, full: torch.Tensor, lo: int, hi: int):
        """elements [lo, hi) of EVERY rank's block of ``full`` (this rank's are in place); returns a handle whose ``wait()`` makes the
        current stream wait for the piece (None: already there) -- the pieces of a chunked exchange queue up behind each other while
        the products of the earlier ones run"""
        B = full.numel() // self.world
        views = [full[q * B + lo:q * B + hi] for q in range(self.world)]
        if hi <= lo:
            return None
        if self.backend == "gloo" and full.is_cuda:     # test path: through the host, synchronous
            parts = [torch.empty(hi - lo, dtype=full.dtype) for _ in range(self.world)]
            self.dist.all_gather(parts, views[self.rank].cpu(), group=self.group)
            for q, part in enumerate(parts):
                if q != self.rank:
                    views[q].copy_(part)
            return None
        # (the input is a copy of this rank's piece: an output list that aliases the input is not something to try for the first
        # time inside a timed run; the copy is 1/world of a piece)
        return self.dist.all_gather(views, views[self.rank].clone(), group=self.group, async_op=True)

    def all_reduce_sum(self, t: torch.Tensor, op=None):
        op = self.dist.ReduceOp.SUM if op is None else op
        if self.backend == "gloo" and t.is_cuda:
            host = t.cpu()
            self.dist.all_reduce(host, op=op, group=self.group)
            t.copy_(host)
        else:
            self.dist.all_reduce(t, op=op, group=self.group)

    def all_reduce_sum_async(self, t: torch.Tensor):
        """the same reduction, not waited for: returns a handle whose ``wait()`` makes the current stream wait (None: already done)"""
        if self.backend == "gloo" and t.is_cuda:
            self.all_reduce_sum(t)
            return None
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_reduce_max(self, t: torch.Tensor):
        self.all_reduce_sum(t, self.dist.ReduceOp.MAX)

    def all_reduce_min(self, t: torch.Tensor):
        self.all_reduce_sum(t, self.dist.ReduceOp.MIN)


class PdlpEngine:
    """Device-resident restarted-PDHG state for one LP shard.

    Parameters are this rank's blocks: ``K_rows`` = CSR of rows [row0,row1) of K, ``KT_rows`` = CSR of
    rows [col0,col1) of K' (global indices), ``c,l,u`` of length col1-col0, ``q`` of length row1-row0.
    """

    def __init__(self, m: int, n: int, m_ineq: int, K_rows, KT_rows, c, q, l, u, rows: Tuple[int, int] = None,
                 cols: Tuple[int, int] = None, d_col=None, d_row=None, comm: Optional[Comm] = None, vec_dtype=None,
                 delta: Optional[bool] = None, exact=None, tiles: bool = True):
        """``vec_dtype=torch.float64`` over float32 matrix values selects the mixed precision (``PDLP_MIXED``): float64 vectors,
        products and sums on a float32 matrix (12 -> 8 bytes per non-zero); ``delta`` (default: ``PDLP_DELTA`` in the environment,
        else on) then runs the iterations on the float32 kernels over float32 difference vectors added to float64 anchor
        products (``pdlp_set_delta`` in include/pdlp_hip.h).  ``exact`` = ``(K_rows, KT_rows)`` in float64: the TRUE matrix when the
        float32 one handed in as ``K_rows`` / ``KT_rows`` is only its rounding (any float64 matrix, a Ruiz-scaled one): the anchors
        of delta mode are then evaluated with it (a second, float64 CSR handle over the same row blocks; ``pdlp_set_anchors``).
        A scaled matrix whose entries ARE float32 numbers (Ruiz on a +-1 matrix is the identity) needs no ``exact``."""
        self.lib = N.load()
        N.trace_range.enabled()                # (PDLP_ROCTX=1|2: roctx ranges for rocprofv3 --marker-trace; once per process)
        rows = (0, m) if rows is None else rows
        cols = (0, n) if cols is None else cols
        self.m, self.n, self.m_ineq = int(m), int(n), int(m_ineq)
        self.rows, self.cols = (int(rows[0]), int(rows[1])), (int(cols[0]), int(cols[1]))
        self.ml, self.nl = self.rows[1] - self.rows[0], self.cols[1] - self.cols[0]
        self.comm = comm if (comm is not None and comm.world > 1) else None
        if self.comm is None and (self.ml != self.m or self.nl != self.n):
            raise ValueError("a sharded problem needs a communicator")
        val = K_rows[2]
        self.device, self.mat_dtype = val.device, val.dtype
        self.dtype = self.mat_dtype if vec_dtype is None else vec_dtype       # the working precision: vectors, sums, scalars
        if self.device.type != "cuda":
            raise N.PdlpError("PdlpEngine needs the problem on a HIP device (there is no CPU fallback)")
        if self.mat_dtype not in _DT or self.dtype not in _DT:
            raise ValueError(f"unsupported dtype {self.mat_dtype} / {self.dtype}")
        self.mixed = self.dtype != self.mat_dtype
        if self.mixed and (self.mat_dtype, self.dtype) != (torch.float32, torch.float64):
            raise ValueError("mixed precision means float32 matrix values under float64 vectors")
        if exact is not None and not self.mixed:
            raise ValueError("`exact` (the float64 matrix behind a float32 rounding) belongs to mixed precision")
        i32 = lambda t: t.to(device=self.device, dtype=torch.int32).contiguous()
        i64 = lambda t: t.to(device=self.device, dtype=torch.int64).contiguous()        # row pointers: 64-bit in the ABI
        fv = lambda t, ln: None if t is None else as_vec(t, ln, self.device, self.dtype)
        # keep every tensor the library points into alive
        self.K = (i64(K_rows[0]), i32(K_rows[1]), K_rows[2].to(self.mat_dtype).contiguous())
        self.KT = (i64(KT_rows[0]), i32(KT_rows[1]), KT_rows[2].to(self.device, self.mat_dtype).contiguous())
        self.c, self.l, self.u = fv(c, self.nl), fv(l, self.nl), fv(u, self.nl)
        self.q = fv(q, self.ml)
        self.d_col, self.d_row = fv(d_col, self.nl), fv(d_row, self.ml)
        self.stream = torch.cuda.current_stream(self.device)
        ptr = lambda t: None if t is None else t.data_ptr()
        self.prob = N.PdlpProblem(N.PDLP_MIXED if self.mixed else _DT[self.dtype], self.device.index or 0, self.m, self.n, self.m_ineq,
                                  self.rows[0], self.rows[1], self.cols[0], self.cols[1],
                                  ptr(self.K[0]), ptr(self.K[1]), ptr(self.K[2]),
                                  ptr(self.KT[0]), ptr(self.KT[1]), ptr(self.KT[2]),
                                  ptr(self.c), ptr(self.l), ptr(self.u), ptr(self.q), ptr(self.d_col), ptr(self.d_row),
                                  self.stream.cuda_stream)
        nbytes = C.c_int64(0)
        N.check(self.lib.pdlp_workspace_bytes(C.byref(self.prob), C.byref(nbytes)), "pdlp_workspace_bytes")
        self.workspace = self._alloc_workspace(int(nbytes.value) + 256, exportable=comm is not None)
        self._ws_off = (-self.workspace.data_ptr()) % 256
        self.h = N._H()
        N.check(self.lib.pdlp_create(C.byref(self.h), C.byref(self.prob), self.workspace.data_ptr() + self._ws_off,
                                     nbytes.value), "pdlp_create")
        self._views = {}
        self._want_tiles = bool(tiles)
        # test / tool knobs of the handle (the library reads no environment: pdlp_set_option)
        if os.environ.get("PDLP_RUNNING_KKT", "1")[:1] == "0":
            self.set_option(N.OPT_RUNNING_KKT, 0)
        if os.environ.get("PDLP_NO_KTY_REUSE") is not None:
            self.set_option(N.OPT_KTY_REUSE, 0)
        if self.comm is not None and os.environ.get("PDLP_BEGIN_INLINE", "1")[:1] != "0":
            # the torch.distributed loop issues every exchange asynchronously BEFORE it starts the next product on the own block
            # (iterate): the local panels then go onto the handle's stream, no side stream / events (pdlp_hip.h, PDLP_OPT_BEGIN_INLINE)
            self.set_option(N.OPT_BEGIN_INLINE, 1)
        self.producer_pieces = os.environ.get("PDLP_PRODUCER_PIECES", "1")[:1] != "0"
        if not self.producer_pieces:
            self.set_option(N.OPT_PRODUCER_PIECES, 0)
        if os.environ.get("PDLP_GRAPH") is not None and self.comm is None:
            self.set_option(N.OPT_GRAPH, 1)
        self.exact = None
        if exact is not None:          # the true float64 matrix, CSR kernels only: two products per restart
            self.exact = PdlpEngine(m, n, m_ineq, exact[0], exact[1], c, q, l, u, rows=rows, cols=cols, comm=comm, tiles=False)
        self._sorted = [None, None]
        self._mv_work = {}
        self.xchunks, self._plans = 1, {}
        self.tiles = [None, None]
        self.kernels = ["csr", "csr"]
        self._maybe_attach_tiles()
        if self.comm is not None and tiles and int(os.environ.get("PDLP_EXCHANGE_CHUNKS", "1")) > 1:
            self.set_exchange_chunks(int(os.environ["PDLP_EXCHANGE_CHUNKS"]))
        self.delta = False
        if self.mixed and (delta if delta is not None else os.environ.get("PDLP_DELTA", "1") != "0"):
            self.set_delta(True)
        # the exchange inside the library (one C call per restart period) is opt-in: PDLP_LIB_COMM=1 here, or
        # enable_library_comm() by the caller (bench.py does); the default is the torch.distributed loop of iterate()
        self.lib_comm, self.lib_comm_log = False, []
        self.peer_on, self.peer_log, self.peer_local_first = False, [], False     # direct exchange over HIP IPC (enable_peer_exchange)
        self.peer_push, self.peer_form = False, 0
        if self.comm is not None and self.comm.backend == "nccl" and tiles and os.environ.get("PDLP_LIB_COMM", "0") == "1":
            self.enable_library_comm()

    def set_producer_pieces(self, on: bool):
        """results of split products leave piece by piece with a chunked exchange (default) or only when the half-step has finished
        (round-4 behaviour; A/B timing and tests).  Every rank must choose the same."""
        self.set_option(N.OPT_PRODUCER_PIECES, int(bool(on)))
        self.producer_pieces = bool(on)

    def set_option(self, option: int, value: int):
        """``pdlp_set_option``: the handle's test / tool switches (``N.OPT_*``)"""
        N.check(self.lib.pdlp_set_option(self.h, int(option), int(value)), "pdlp_set_option")

    # ---- panel-tiled matrix copies (fast path for wide gathered vectors) -------------------------------
    def tile_limits(self) -> dict:
        """what ``pdlp_attach_tiles`` accepts on this handle (the row-sum scratch and the partial-sum slots are sized at creation)"""
        out = (C.c_int32 * 6)()
        N.check(self.lib.pdlp_tile_limits(self.h, out), "pdlp_tile_limits")
        return dict(max_groups=out[0], max_blocks=out[1], rpt_max=out[2], cap=out[3], nt=out[4])

    def _maybe_attach_tiles(self):
        """Which kernel multiplies each matrix -- decided from the shape alone, so a run (and every rank of a sharded
        one) always takes the same kernel and therefore the same summation order:
        ``PDLP_TILED=0`` never tiles, ``=1`` tiles whenever the matrix is eligible, ``auto`` (default) tiles matrices with
        >= 2^20 non-zeros, >= 10 per row on average and a gathered vector of >= 2^16 entries (measured: 1M x 1M with 100
        per row 3.3x faster tiled, 500k x 500k with 20 per row 1.35x; with 5 per row the CSR kernel is ahead),
        ``=time`` builds the tiles for every candidate, times both kernels on this device and keeps the faster one
        (not reproducible run to run; tuning only).  ``self.kernels`` records the choice per matrix."""
        mode = os.environ.get("PDLP_TILED", "auto") if self._want_tiles else "0"
        self.kernels = ["csr", "csr"]
        if os.environ.get("PDLP_SORTED") == "1" and self._want_tiles:            # tests / tuning: sorted row blocks for every matrix the CSR kernel keeps
            for transpose in (0, 1):
                self.attach_sorted(transpose)
        if mode == "0":
            return
        lim = self.tile_limits()
        for transpose, (rp, ci, va), rows, cols in ((0, self.K, self.ml, self.n), (1, self.KT, self.nl, self.m)):
            nnz = int(va.numel())
            if rows == 0 or (mode != "1" and (cols < (1 << 16) or nnz < (1 << 20))):
                continue
            if mode == "auto" and nnz < 10 * rows:
                continue
            knob = lambda name: int(os.environ[name]) if os.environ.get(name) else None      # tuning experiments
            with N.trace_range("pdlp: tile build (K')" if transpose else "pdlp: tile build (K)", self.stream):
                t = _tiled.build_tiles(rp, ci, va, rows, cols, lw=knob("PDLP_TILE_LW"), rpt=knob("PDLP_TILE_RPT"),
                                       groups=knob("PDLP_TILE_GROUPS"), max_groups=lim["max_groups"],
                                       kernel_limits=(lim["rpt_max"], lim["cap"], lim["nt"]))
            if t is None or t.nblk > lim["max_blocks"]:
                # clustered (banded, block structured): the CSR kernel, with every row block's items sorted by column
                if t is None and os.environ.get("PDLP_SORTED", "auto") != "0":
                    self.attach_sorted(transpose, force=False)        # (only if the blocks' columns do cluster)
                continue
            if mode != "time":
                self.attach_tiles(transpose, t)
                continue
            g = torch.Generator(device=self.device).manual_seed(1)
            vin = torch.randn(cols, dtype=self.dtype, device=self.device, generator=g)
            out = torch.empty(rows, dtype=self.dtype, device=self.device)
            t_csr = self._time_spmv(transpose, vin, out)
            self.attach_tiles(transpose, t)
            if self._time_spmv(transpose, vin, out) >= t_csr:
                self.attach_tiles(transpose, None)

    def _time_spmv(self, transpose: int, vin: torch.Tensor, out: torch.Tensor, reps: int = 3) -> float:
        call = lambda: N.check(self.lib.pdlp_spmv(self.h, int(transpose), vin.data_ptr(), out.data_ptr()), "pdlp_spmv")
        call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(self.stream)
        for _ in range(reps):
            call()
        b.record(self.stream)
        b.synchronize()
        return a.elapsed_time(b) / reps

    def attach_sorted(self, transpose: int, on: bool = True, force: bool = True):
        """Column-sorted copy of every row block's items for the CSR kernel (``pdlp_attach_sorted``): for matrices whose entries
        cluster (banded, block structured) a wave's gathers then touch a few cache lines instead of one per lane.  Same sums."""
        transpose = int(transpose)
        if not on:
            N.check(self.lib.pdlp_attach_sorted(self.h, transpose, None, None, None), "pdlp_attach_sorted")
            self._sorted[transpose] = None
            self.kernels[transpose] = "csr"
            return
        rp, ci, va = self.KT if transpose else self.K
        nnz = int(va.numel())
        if nnz == 0:
            return
        nb, bp = C.c_int32(0), C.c_void_p()
        N.check(self.lib.pdlp_schedule_info(self.h, transpose, C.byref(nb), C.byref(bp)), "pdlp_schedule_info")
        off = bp.value - self.workspace.data_ptr()
        blk = self.workspace[off:off + (nb.value + 1) * 16].view(torch.int64).view(-1, 2)
        first = blk[:, 1].long()                                   # first non-zero of every block (and the end)
        lens = first[1:] - first[:-1]
        dev = self.device
        bid = torch.repeat_interleave(torch.arange(nb.value, device=dev), lens)
        cl = ci.long()
        big = torch.iinfo(torch.int64).max
        cmin = torch.full((nb.value,), big, dtype=torch.int64, device=dev).scatter_reduce_(0, bid, cl, "amin")
        cmax = torch.full((nb.value,), -1, dtype=torch.int64, device=dev).scatter_reduce_(0, bid, cl, "amax")
        ok = (lens > 0) & (lens <= 2048) & (cmax - cmin < (1 << 21))
        if not force:
            # Sorting pays when neighbouring sorted items share 128-byte lines.  A block whose columns are spread so thinly that sorted
            # neighbours are >= 32 columns apart on average touches one line per item either way, and the sorted form then only costs:
            # the extra dependent load of the block's base column and scattered LDS stores (neos3-shaped K': 24.5k -> 24.9k it/s
            # in plain CSR order, profiles/r05_small_lp/).  Keep CSR order unless most items sit in blocks that do cluster.
            gap = (cmax - cmin).double() / (lens - 1).clamp(min=1).double()
            dense = ok & (gap < 32.0)
            if float(lens[dense].sum()) < 0.5 * nnz:
                return
        order = torch.argsort((bid << 32) | cl, stable=True)        # by (block, column); blocks stay in place
        slot = (torch.arange(nnz, device=dev) - first[bid])[order]
        sidx = (slot << 21) | (cl[order] - cmin[bid]).clamp_(0, (1 << 21) - 1)
        sidx = torch.where(sidx >= 2 ** 31, sidx - 2 ** 32, sidx).to(torch.int32)
        sval = va[order].contiguous()
        cbase = torch.where(ok, cmin, torch.full_like(cmin, -1)).to(torch.int32)
        N.check(self.lib.pdlp_attach_sorted(self.h, transpose, sidx.data_ptr(), sval.data_ptr(), cbase.data_ptr()), "pdlp_attach_sorted")
        self._sorted[transpose] = (sidx, sval, cbase)              # keep the arrays alive
        self.kernels[transpose] = f"csr, sorted row blocks ({int(ok.sum())} of {nb.value})"

    def attach_tiles(self, transpose: int, t: Optional["_tiled.Tiles"]):
        if t is None:
            N.check(self.lib.pdlp_attach_tiles(self.h, int(transpose), None), "pdlp_attach_tiles")
            self._plans = {}
            self.tiles[int(transpose)] = None
            self.kernels[int(transpose)] = "csr"
            return
        rem = [0, 0] + [None] * 8
        if t.nrem:
            rows = self.nl if transpose else self.ml
            t._work = torch.empty(int(t.rem_sptr.numel()) - 1, dtype=torch.float64, device=self.device)
            t._extra = torch.zeros(rows, dtype=self.dtype, device=self.device)
            t._extra32 = torch.zeros(rows, dtype=torch.float32, device=self.device) if self.mixed else None
            rem = [int(t.rem_rows.numel()), int(t.rem_sptr.numel()) - 1, t.rem_rows.data_ptr(), t.rem_rptr.data_ptr(), t.rem_sptr.data_ptr(),
                   t.rem_col.data_ptr(), t.rem_val.data_ptr(), t._work.data_ptr(), t._extra.data_ptr(),
                   None if t._extra32 is None else t._extra32.data_ptr()]
        rel, base = t.abi_tile_ptr()          # (int32 offsets relative to each row block's first item + the 64-bit bases; kept alive on t)
        desc = N.PdlpTiles(t.lw, t.rpt, t.cap, t.nblk, t.npanel, t.groups, t.idx.data_ptr(), t.val.data_ptr(), rel.data_ptr(), base.data_ptr(),
                           t.cnt.data_ptr(), *rem)
        N.check(self.lib.pdlp_attach_tiles(self.h, int(transpose), C.byref(desc)), "pdlp_attach_tiles")
        self._plans = {}
        self.tiles[int(transpose)] = t       # keep the arrays alive
        self.kernels[int(transpose)] = ("tiled" if t.groups == 1 else f"tiled/{t.groups} groups") + (f" + remainder {t.nrem}" if t.nrem else "")

    # ---- the exchange inside the library (RCCL) --------------------------------------------------------------
    @staticmethod
    def _loaded_rccl():
        """path of the RCCL library this process already has mapped (PyTorch's), so that the library joins the same one"""
        try:
            for line in open("/proc/self/maps"):
                if "librccl" in line:
                    return line.split()[-1]
        except OSError:
            pass
        return None

    def enable_library_comm(self, dist=None, group=None, rccl_path: Optional[str] = None, timeout: float = 120.0,
                            cross_check: bool = True) -> bool:
        """Give the handle its own RCCL communicator (``pdlp_comm_init``): ``iterate`` then is ONE library call per restart
        period -- half-steps, all-gathers and the step-size all-reduce enqueued back to back on the stream -- instead of six
        ctypes calls and three torch collectives per iteration.  Every step is agreed on by ALL ranks over the existing process
        group before the next one: (1) the library loads (``pdlp_comm_load``, rank local), (2) the id travels from rank 0,
        (3) ``pdlp_comm_init`` in a helper thread with ``timeout`` seconds -- a hang becomes a fallback --, (4) a round trip of
        both collectives against known values, (5) ``cross_check``: two adaptive iterations from one synthetic state through the
        torch.distributed loop and through the library path must agree bit for bit.  Any failure anywhere leaves all ranks on
        the torch.distributed loop.  Call before the iterate is set (the cross-check overwrites it and resets it to zero).
        ``rccl_path``: the library to dlopen (default: the librccl this process has mapped -- PyTorch's).  ``self.lib_comm_log``
        records what happened.  Returns whether the library path is on."""
        import threading
        if dist is None:
            if self.comm is None:
                return False
            dist, group = self.comm.dist, self.comm.group
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        host_coll = dist.get_backend(group) == "gloo"           # rehearsal on a shared card: collectives through host tensors
        log = self.lib_comm_log = []

        def agree(ok: int) -> int:                               # MIN over the ranks
            flag = torch.tensor([int(ok)], dtype=torch.int32, device="cpu" if host_coll else self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag)

        path = rccl_path if rccl_path is not None else self._loaded_rccl()
        cpath = None if path is None else path.encode()
        if not agree(self.lib.pdlp_comm_load(cpath) == 0):
            log.append("load failed on some rank")
            return False
        idbuf = (C.c_char * 128)()
        ok = 1
        if rank == 0 and self.lib.pdlp_comm_unique_id(cpath, idbuf) != 0:
            ok = 0
        t = torch.tensor(list(idbuf.raw) + [ok], dtype=torch.uint8, device="cpu" if host_coll else self.device)
        dist.broadcast(t, 0, group=group)
        raw = bytes(t.cpu().tolist())
        if not raw[128]:
            log.append("unique id failed")
            return False
        C.memmove(idbuf, raw[:128], 128)
        res = {}

        def init():
            res["rc"] = self.lib.pdlp_comm_init(self.h, cpath, idbuf, rank, world)
        th = threading.Thread(target=init, daemon=True)
        th.start()
        th.join(timeout)
        if th.is_alive():
            self._comm_init_thread = th          # poisoned: the handle is never destroyed while that thread lives (__del__)
        if not agree((not th.is_alive()) and res.get("rc") == 0):
            log.append(f"init failed or timed out (this rank: alive={th.is_alive()}, rc={res.get('rc')})"
                       + ("; the handle is poisoned (a thread is still inside pdlp_comm_init): end this process rather than reuse it" if th.is_alive() else ""))
            return False
        # round trip: all-gather of a full-length vector and the 8-double all-reduce (rank-local errors are caught, so that
        # every rank reaches the agreement below)
        ok = 1
        try:
            dx, red = self.buffer(N.BUF_DX), self.buffer(N.BUF_RED)
            dx.zero_()
            dx[self.cols[0]:self.cols[1]] = rank + 1
            red.fill_(rank + 1)
            N.check(self.lib.pdlp_comm_all_gather(self.h, N.BUF_DX), "pdlp_comm_all_gather")
            N.check(self.lib.pdlp_comm_all_reduce_red(self.h), "pdlp_comm_all_reduce_red")
            self.stream.synchronize()
            want = torch.arange(1, world + 1, device=self.device, dtype=dx.dtype).repeat_interleave(self.nl)
            ok = int(torch.equal(dx, want) and bool((red == world * (world + 1) / 2).all()))
            dx.zero_()
            red.zero_()
        except N.PdlpError:
            ok = 0
        if not agree(ok):
            log.append("round trip wrong")
            return False
        if cross_check:
            try:
                same = int(self._cross_check_paths())
            except N.PdlpError:
                same = 0
            if not agree(same):
                self.lib_comm = False
                log.append("cross-check against the torch.distributed loop differs")
                return False
            log.append("cross-check: 2 adaptive iterations bit-identical on both paths")
        self.lib_comm = True
        return True

    # ---- direct exchange (pdlp_peer_*): iterations without collectives ------------------------------------------------------
    def enable_peer_exchange(self, cross_check: bool = True, timeout_ms: Optional[int] = None, local_first: bool = False) -> bool:
        """Connect the ranks' handles over HIP IPC (``pdlp_peer_export`` / ``pdlp_peer_connect``, include/pdlp_hip.h): ``iterate`` then
        is ONE library call per restart period with NO collective in it -- every half-step stores its block of the exchanged vector
        straight into the other ranks' memory (xGMI between the GPUs of a node) and a flag follows; the step-size rule's sums travel
        with the flag.  At most 8 ranks, all on one node.  Every step is agreed on by all ranks over the process group: export,
        connect, and (``cross_check``) two fixed-step iterations that must equal the torch.distributed loop bit for bit plus two
        adaptive ones that must agree to 1e-5 (the ranks' three sums are added in rank order here, in the collective's order there;
        identical for two ranks).  Any failure leaves all ranks where they were.  Call before the iterate is set (the cross-check
        overwrites it and resets it to zero).  ``self.peer_log`` records what happened.  Returns whether the direct exchange is on.
        ``local_first``: split products with the own block's panels between signal and wait (``PDLP_OPT_PEER_LOCAL_FIRST``; the
        cross-check always runs in that form -- it is the one whose partial sums are grouped like the loop's)."""
        if self.comm is None or not hasattr(self.comm, "dist"):
            return False
        dist, group = self.comm.dist, self.comm.group
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        log = self.peer_log = []
        host_coll = dist.get_backend(group) == "gloo"
        cdev = "cpu" if host_coll else self.device
        t_start = time.time()

        def trace(what):                                         # (PDLP_PEER_TRACE=1: where a first multi-GPU run spends its time)
            if os.environ.get("PDLP_PEER_TRACE"):
                sys.stderr.write(f"[peer exchange, rank {rank}, {time.time() - t_start:7.2f} s] {what}\n")
                sys.stderr.flush()

        def agree(ok) -> int:                                    # MIN over the ranks
            flag = torch.tensor([int(bool(ok))], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag)

        if not agree(2 <= world <= 8):
            log.append(f"{world} ranks: the direct exchange connects 2 to 8")
            return False
        nb = N.PEER_INFO_BYTES
        info = (C.c_char * nb)()
        rc = self.lib.pdlp_peer_export(self.h, info)
        trace(f"exported (rc {rc}); the workspace ({self.workspace.numel() >> 20} MB) lies in an allocation of "
              f"{int.from_bytes(info.raw[232:240], 'little') >> 20} MB")
        mine = torch.tensor(list(info.raw) + [int(rc == 0)], dtype=torch.uint8, device=cdev)
        every = torch.empty(world * (nb + 1), dtype=torch.uint8, device=cdev)
        dist.all_gather_into_tensor(every, mine, group=group)
        raw = bytes(every.cpu().tolist())
        if not all(raw[q * (nb + 1) + nb] for q in range(world)):
            log.append(f"export failed on some rank (this rank: rc={rc})")
            return False
        infos = b"".join(raw[q * (nb + 1):q * (nb + 1) + nb] for q in range(world))
        trace("infos gathered")
        # one rank at a time (a few milliseconds each): N processes mapping each other's memory at the same moment is a first on any
        # machine this runs on, and a rank that never returns from hipIpcOpenMemHandle is then easy to tell apart in PDLP_PEER_TRACE
        # (the hang that did occur -- allocations with bit 31 of their size set -- is kept out by pdlp_peer_export / _alloc_workspace)
        rc = 0
        for r in range(world):
            if r == rank:
                rc = self.lib.pdlp_peer_connect(self.h, rank, world, infos, 0)
                trace(f"connected (rc {rc})")
            agree(1)
        if not agree(rc == 0):
            log.append(f"connect failed on some rank (this rank: rc={rc})")
            if rc == 0:
                self.lib.pdlp_peer_close(self.h)
            return False
        if timeout_ms is not None:
            self.set_option(N.OPT_PEER_TIMEOUT_MS, int(timeout_ms))
        self.peer_on = True
        self.set_option(N.OPT_PEER_EXCHANGE, 1)
        if cross_check:
            try:
                same = int(self._cross_check_peer(trace=trace))
            except N.PdlpError as e:
                log.append(f"cross-check raised: {e}")
                same = 0
            trace(f"cross-check done on this rank: {same}")
            if not agree(same):
                log.append("cross-check against the torch.distributed loop differs")
                self.disable_peer_exchange()
                return False
            log.append("cross-check: 2 fixed-step iterations bit-identical, 2 adaptive ones within 1e-5 of the torch.distributed loop")
        self.set_peer_local_first(local_first)
        return True

    def set_peer_local_first(self, on: bool):
        self.peer_local_first = bool(on)
        self.set_option(N.OPT_PEER_LOCAL_FIRST, int(bool(on)))

    def set_peer_push(self, on: bool):
        """``PDLP_OPT_PEER_PUSH``: the block leaves by a copy kernel on a side stream, beside the own block's panels of the next product
        (the form for 2 and 4 ranks, where those panels are long enough to hide the links)"""
        self.peer_push = bool(on)
        self.set_option(N.OPT_PEER_PUSH, int(bool(on)))

    PEER_FORMS = ("whole product after the wait", "own-block panels between signal and wait", "push beside the own-block panels")

    def set_peer_form(self, form: int):
        """0: signal, wait, whole product; 1: own-block panels between signal and wait; 2: push kernel beside the own-block panels"""
        self.set_peer_local_first(form == 1)
        self.set_peer_push(form == 2)
        self.peer_form = int(form)

    def disable_peer_exchange(self):
        self.peer_on = False
        N.check(self.lib.pdlp_peer_close(self.h), "pdlp_peer_close")

    def set_peer_exchange(self, on: bool):
        """use (or not) a connected direct exchange for the iterations"""
        st = self.peer_status()
        self.peer_on = bool(on) and st["connected"]
        self.set_option(N.OPT_PEER_EXCHANGE, int(self.peer_on))

    def peer_status(self) -> dict:
        out = (C.c_int32 * 4)()
        N.check(self.lib.pdlp_peer_status(self.h, out), "pdlp_peer_status")
        return dict(connected=bool(out[0]), enabled=bool(out[1]), gave_up_on=(out[2] - 1 if out[2] else None), exchanges=out[3])

    def _peer_check(self):
        """a wait of the direct exchange that gave up leaves incomplete vectors behind: never compute on"""
        if self.peer_on:
            st = self.peer_status()
            if st["gave_up_on"] is not None:
                raise N.PdlpError(f"direct exchange: rank {st['gave_up_on']} did not signal within the timeout "
                                  f"(exchange {st['exchanges']}); the iterate of this rank is incomplete")

    def _cross_check_peer(self, iters: int = 2, eta: float = 1e-2, trace=lambda what: None) -> bool:
        zeros = lambda ln: torch.zeros(ln, dtype=self.dtype, device=self.device)
        ok = True
        saved_lib = self.lib_comm
        self.lib_comm = False
        self.set_peer_local_first(True)
        for adaptive in (False, True):
            out = []
            for peer in (False, True):
                self.set_peer_exchange(peer)
                self.set_iterate(zeros(self.nl), zeros(self.ml))
                self.set_step(eta, 1.0, 1.0, 0)
                trace(f"cross-check: {'adaptive' if adaptive else 'fixed'}, {'direct exchange' if peer else 'loop'}: state set")
                self.iterate(iters, adaptive)
                trace("  iterations issued")
                x, y = self.get_iterate(N.CUR)         # (synchronises)
                trace("  synchronised")
                self._peer_check()
                out.append((x, y, self.scalars()["eta"]))
            (x0, y0, e0), (x1, y1, e1) = out
            fin = bool(torch.isfinite(x1).all()) and bool(torch.isfinite(y1).all())
            if adaptive and self.comm.world > 2:
                close = lambda a, b: bool(((a - b).abs() <= 1e-5 * (1 + b.abs())).all())
                ok = ok and fin and close(x1, x0) and close(y1, y0) and abs(e1 - e0) <= 1e-5 * abs(e0)
            else:
                ok = ok and fin and torch.equal(x0, x1) and torch.equal(y0, y1) and e0 == e1
        self.lib_comm = saved_lib
        self.set_peer_exchange(True)
        self.set_iterate(zeros(self.nl), zeros(self.ml))
        self.set_step(0.0, 1.0, 1.0, 0)
        return bool(ok)

    def _cross_check_paths(self, iters: int = 2, eta: float = 1e-2) -> bool:
        """`iters` adaptive iterations from x = y = 0 through the torch.distributed loop and through the library's own
        exchange: the same bits on this rank?  Leaves the engine at x = y = 0, eta = 0 (a fresh handle's state)."""
        zeros = lambda ln: torch.zeros(ln, dtype=self.dtype, device=self.device)
        out = []
        saved_peer = self.peer_on
        if saved_peer:
            self.set_peer_exchange(False)
        for lib in (False, True):
            self.lib_comm = lib
            self.set_iterate(zeros(self.nl), zeros(self.ml))
            self.set_step(eta, 1.0, 1.0, 0)
            self.iterate(iters, True)
            x, y = self.get_iterate(N.CUR)
            out.append((x, y, self.scalars()["eta"]))
        self.lib_comm = False
        if saved_peer:
            self.set_peer_exchange(True)
        self.set_iterate(zeros(self.nl), zeros(self.ml))
        self.set_step(0.0, 1.0, 1.0, 0)
        (x0, y0, e0), (x1, y1, e1) = out
        return bool(torch.equal(x0, x1) and torch.equal(y0, y1) and e0 == e1 and bool(torch.isfinite(x0).all()))

    def set_delta(self, on: bool):
        """delta mode of a mixed-precision engine (include/pdlp_hip.h, pdlp_set_delta)"""
        N.check(self.lib.pdlp_set_delta(self.h, int(bool(on))), "pdlp_set_delta")
        self.delta = bool(on)

    def delta_state(self) -> dict:
        out = (C.c_int32 * 3)()
        N.check(self.lib.pdlp_delta_state(self.h, out), "pdlp_delta_state")
        return dict(delta=bool(out[0]), anchors_valid=bool(out[1]), dy_folded=bool(out[2]))

    def refresh_products(self):
        """recompute K x and K'y of the current iterate exactly (float64 accumulation): the anchors of delta mode"""
        if self.exact is not None:     # the handle's matrix is the float32 rounding of the true one: anchors from the true one
            if self.comm is not None:  # (sharded: the products need the complete iterate)
                self._gather(N.BUF_X_CUR)
                self._gather(N.BUF_Y_CUR)
                x, y = self.buffer(N.BUF_X_CUR), self.buffer(N.BUF_Y_CUR)
            else:
                x, y = self.get_iterate(N.CUR)
            kx, kty = self.exact.spmv(x, False), self.exact.spmv(y, True)
            N.check(self.lib.pdlp_set_anchors(self.h, kx.data_ptr(), kty.data_ptr()), "pdlp_set_anchors")
            return
        if self.comm is not None:
            self._gather(N.BUF_X_CUR)
            self._gather(N.BUF_Y_CUR)
        N.check(self.lib.pdlp_refresh_products(self.h), "pdlp_refresh_products")

    def set_exchange_chunks(self, chunks: int):
        """Sharded, tiled products: move the gathered vector in ``chunks`` pieces (piece c = a slice of EVERY rank's block) and
        multiply the panels a piece completes while the next piece is on the wire (include/pdlp_hip.h, pdlp_set_exchange_chunks).
        1 = one all-gather per product.  Every rank must choose the same number."""
        N.check(self.lib.pdlp_set_exchange_chunks(self.h, int(chunks)), "pdlp_set_exchange_chunks")
        self.xchunks = int(chunks)
        self._plans = {}

    def tune_exchange_chunks(self, reps: int = 4) -> dict:
        """Choose the number of pieces from what THIS machine does: time the all-gather of one gathered vector and this rank's
        product with K (max over the ranks).  Measured with stand-ins on one GPU (profiles/r03_split_chunks.log): one piece wins
        while the all-gather takes less than about half a product (the extra launches and partial-sum slots of a chunked exchange
        cost more than the overlap gains), two pieces win beyond that (-7 % at 8 ranks with a 0.2 ms all-gather, -12 % at 4 ranks
        with 0.3 ms).  Collective: every rank calls it (whatever kernel its own shard uses); all end up with the same choice.
        Call it before the iterate is set or between restart periods: ``pdlp_set_exchange_chunks`` refuses while a product is pending."""
        out = dict(chunks=1, all_gather_ms=None, product_ms=None)
        if self.comm is None:
            return out
        # Every decision below is GLOBAL: a rank whose shard is not tiled or not split (CSR fallback, too many row blocks) must not
        # leave before the collectives the others are about to issue, and all ranks must end up with the same number of pieces.
        mine = int(self.tiles[0] is not None and self.split_info(0)["local_groups"] > 0)
        flag = torch.tensor([-float(mine)], dtype=torch.float64, device=self.device)
        self.comm.all_reduce_max(flag)                      # max of the negated flags = -(min of the flags)
        if float(flag[0]) != -1.0:
            return out
        live = self.buffer(N.BUF_GDX if self.delta else N.BUF_XBAR)
        full = torch.zeros_like(live)                       # a scratch vector of the exchange's size: the live buffer is not touched
        vin = torch.zeros(self.n, dtype=self.dtype, device=self.device)
        res = torch.empty(self.ml, dtype=self.dtype, device=self.device)
        ev = lambda: torch.cuda.Event(enable_timing=True)

        def timed(fn):
            fn()
            self.stream.synchronize()
            a, b = ev(), ev()
            a.record(self.stream)
            for _ in range(reps):
                fn()
            b.record(self.stream)
            b.synchronize()
            return a.elapsed_time(b) / reps
        self.comm.dist.barrier(group=self.comm.group)
        ag = timed(lambda: self.comm.all_gather(full))
        prod = timed(lambda: N.check(self.lib.pdlp_spmv(self.h, 0, vin.data_ptr(), res.data_ptr()), "pdlp_spmv"))
        t = torch.tensor([ag, prod], dtype=torch.float64, device=self.device)
        self.comm.all_reduce_max(t)
        ag, prod = float(t[0]), float(t[1])
        chunks = 2 if ag > 0.5 * prod else 1
        self.set_exchange_chunks(chunks)
        # too few panels for pieces on some matrix of some rank: everybody stays with one all-gather (the plan is a function of the
        # block length alone, so this test gives the same answer everywhere; the reduction makes that a guarantee, not a hope)
        ok = torch.tensor([-float(all(len(self.exchange_plan(tr)) == chunks for tr in (0, 1)))], dtype=torch.float64, device=self.device)
        self.comm.all_reduce_max(ok)
        if chunks > 1 and float(ok[0]) != -1.0:
            chunks = 1
            self.set_exchange_chunks(1)
        out.update(chunks=chunks, all_gather_ms=round(ag, 4), product_ms=round(prod, 4))
        return out

    def exchange_plan(self, transpose: int) -> list:
        """[(lo, hi), ...]: the element ranges (inside one rank's block) of the pieces in which the input of K xbar (0) / K'y (1) travels"""
        plan = self._plans.get(int(transpose))
        if plan is None:
            nc, b = C.c_int32(0), (C.c_int64 * 5)()
            N.check(self.lib.pdlp_exchange_plan(self.h, int(transpose), C.byref(nc), b), "pdlp_exchange_plan")
            plan = self._plans[int(transpose)] = [(int(b[c]), int(b[c + 1])) for c in range(nc.value)]
        return plan

    def _half_in_pieces(self, dual: bool, a: int, full: torch.Tensor):
        """One half-step.  If the exchange that follows it is chunked its result leaves in the pieces of that plan: the rows of
        piece c (``pdlp_*_half_piece``), then piece c's all-gather -- issued behind those rows, it runs while piece c + 1's rows
        are multiplied.  Returns the pieces' handles, or None when the half-step went out whole (the caller exchanges afterwards).
        Every rank takes the same branch: the plan is a function of the block length and the piece count alone."""
        plan = self.exchange_plan(1 if dual else 0)        # the plan of the exchange that FOLLOWS: y after the dual, xbar after the primal
        lib, h = self.lib, self.h
        if len(plan) == 1 or not self.producer_pieces:
            N.check((lib.pdlp_dual_half if dual else lib.pdlp_primal_half)(h, a), "pdlp_dual_half" if dual else "pdlp_primal_half")
            return None
        piece = lib.pdlp_dual_half_piece if dual else lib.pdlp_primal_half_piece
        works = []
        for c, (lo, hi) in enumerate(plan):
            N.check(piece(h, a, c, len(plan)), "pdlp_dual_half_piece" if dual else "pdlp_primal_half_piece")
            works.append(self.comm.all_gather_piece(full, lo, hi))
        return works

    def _start_exchange(self, transpose: int, full: torch.Tensor, works):
        """issue the exchange of ``full`` now, asynchronously, unless its pieces are on their way already; returns the handles
        (a one-element list for a one-piece plan)"""
        if works is not None:
            return works
        plan = self.exchange_plan(transpose)
        if len(plan) == 1:
            return [self.comm.all_gather_async(full)]
        return [self.comm.all_gather_piece(full, lo, hi) for lo, hi in plan]

    def _exchange(self, transpose: int, full: torch.Tensor, works=None):
        """the input of the next product to every rank, in the pieces of its plan; the panels a piece completes are multiplied as
        soon as it is there (all but the last piece's: those belong to the half-step that follows).  ``works``: the pieces are
        on their way already (``_half_in_pieces``)"""
        plan = self.exchange_plan(transpose)
        if works is None and len(plan) == 1:
            self.comm.all_gather(full)
            return
        if works is None:
            works = [self.comm.all_gather_piece(full, lo, hi) for lo, hi in plan]
        if len(plan) == 1:                       # (issued by _start_exchange)
            if works[0] is not None:
                works[0].wait()
            return
        for c, w in enumerate(works):
            if w is not None:
                w.wait()
            if c + 1 < len(works):
                N.check(self.lib.pdlp_half_chunk(self.h, int(transpose), c), "pdlp_half_chunk")

    def split_info(self, transpose: int) -> dict:
        """how a sharded product is split so that its local panels overlap the all-gather (zeros: not split)"""
        out = (C.c_int32 * 4)()
        N.check(self.lib.pdlp_split_info(self.h, int(transpose), out), "pdlp_split_info")
        return dict(local_panels=(out[0], out[1]), local_groups=out[2], other_groups=out[3])

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        th = getattr(self, "_comm_init_thread", None)
        if h and th is not None and th.is_alive():
            # a pdlp_comm_init that timed out is still inside ncclCommInitRank with this handle: destroying it now would race with
            # that thread (it may yet write the communicator, the stream and the events).  The handle is leaked on purpose; the
            # owner should end the process (bench.py does) rather than reuse it.
            return
        if h:
            self.lib.pdlp_destroy(h)
        if getattr(self, "_ws_pool", None) is not None:        # (the block goes back before its private pool does)
            self._views = {}
            self.workspace = None
            self._ws_pool = None

    @classmethod
    def from_full(cls, K: CsrPair, c, q, l, u, m_ineq: int, d_col=None, d_row=None, vec_dtype=None, delta=None,
                  exact: Optional[CsrPair] = None) -> "PdlpEngine":
        """single-GPU engine over a whole problem (``exact``: the float64 matrix of which ``K`` is the float32 rounding)"""
        ex = None if exact is None else ((exact.rowptr, exact.colidx, exact.val), (exact.t_rowptr, exact.t_colidx, exact.t_val))
        return cls(K.m, K.n, m_ineq, (K.rowptr, K.colidx, K.val), (K.t_rowptr, K.t_colidx, K.t_val), c, q, l, u,
                   d_col=d_col, d_row=d_row, vec_dtype=vec_dtype, delta=delta, exact=ex)

    # ---- buffers ------------------------------------------------------------------------------------
    def buffer(self, which: int) -> torch.Tensor:
        """torch view of one of the library's device buffers (roles move: query again after a step)"""
        p = C.c_void_p()
        N.check(self.lib.pdlp_buffer_ptr(self.h, which, C.byref(p)), "pdlp_buffer_ptr")
        key = (p.value, which in (N.BUF_RED, N.BUF_SCALARS))
        v = self._views.get(key)
        if v is None:
            off = p.value - self.workspace.data_ptr()
            if which in (N.BUF_RED, N.BUF_SCALARS):
                cnt = N.NRED if which == N.BUF_RED else N.NSCAL
                v = self.workspace[off:off + cnt * 8].view(torch.float64)
            elif which in (N.BUF_GDX, N.BUF_GDY):
                v = self.workspace[off:off + (self.n if which == N.BUF_GDX else self.m) * 4].view(torch.float32)
            else:
                ln = {N.BUF_X_SUM: self.nl, N.BUF_Y_SUM: self.ml, N.BUF_DX: self.n, N.BUF_DY: self.m, N.BUF_LAM_PREV: self.nl}.get(
                    which, self.n if which <= N.BUF_X_AVG else self.m)
                v = self.workspace[off:off + ln * self.dtype.itemsize].view(self.dtype)
            self._views[key] = v
        return v

    def _alloc_workspace(self, nbytes: int, exportable: bool) -> torch.Tensor:
        """the handle's workspace.  Of a sharded engine it may be exported to the other ranks over HIP IPC (the direct exchange): it
        then gets an allocation OF ITS OWN (a private pool of torch's allocator: a block carved out of a cached segment drags the
        whole segment along) whose size avoids a bug of hipIpcOpenMemHandle on ROCm 7.2 -- opening an allocation whose size has bit 31
        set (2-4 GiB, 6-8 GiB, ...) never returns (tools/ipc_torch_probe.py; pdlp_peer_export refuses such a workspace)."""
        if not (exportable and self.device.type == "cuda" and hasattr(torch.cuda, "MemPool")):
            return torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._ws_pool = torch.cuda.MemPool()
        with torch.cuda.use_mem_pool(self._ws_pool, device=self.device):
            return torch.empty(exportable_bytes(nbytes), dtype=torch.uint8, device=self.device)

    def _gather(self, which: int):
        if self.comm is not None:
            self.comm.all_gather(self.buffer(which))

    # ---- state --------------------------------------------------------------------------------------
    def set_iterate(self, x_local: torch.Tensor, y_local: torch.Tensor):
        x = as_vec(x_local, self.nl, self.device, self.dtype)
        y = as_vec(y_local, self.ml, self.device, self.dtype)
        N.check(self.lib.pdlp_set_iterate(self.h, x.data_ptr(), y.data_ptr()), "pdlp_set_iterate")
        self._gather(N.BUF_X_CUR)
        self._gather(N.BUF_Y_CUR)

    def get_iterate(self, which: int = N.CUR) -> Tuple[torch.Tensor, torch.Tensor]:
        x = torch.empty(self.nl, dtype=self.dtype, device=self.device)
        y = torch.empty(self.ml, dtype=self.dtype, device=self.device)
        N.check(self.lib.pdlp_get_iterate(self.h, which, x.data_ptr(), y.data_ptr()), "pdlp_get_iterate")
        return x, y

    def set_step(self, eta: float, omega: float, theta: float = 1.0, iteration: int = 0):
        N.check(self.lib.pdlp_set_step(self.h, float(eta), float(omega), float(theta), int(iteration)), "pdlp_set_step")

    def set_omega(self, omega: float):
        N.check(self.lib.pdlp_set_omega(self.h, float(omega)), "pdlp_set_omega")

    def scalars(self) -> dict:
        out = (C.c_double * N.NSCAL)()
        N.check(self.lib.pdlp_get_scalars(self.h, out), "pdlp_get_scalars")
        names = ("eta", "omega", "theta", "tau", "sigma", "w_pending", "eta_sum", "k", "inv1pt", "accepted", "eta_bar",
                 "denominator")
        return {k: out[i] for i, k in enumerate(names)}

    # ---- iterations ---------------------------------------------------------------------------------
    def iterate(self, iters: int, adaptive: bool):
        """`iters` PDHG iterations, no host synchronisation (pdhg.py:76-112)."""
        if self.exact is not None and self.delta and int(iters) > 0 and not self.delta_state()["anchors_valid"]:
            self.refresh_products()
        if self.peer_on:                            # direct exchange: one call, no collective (the anchors of delta mode need gathers)
            self._peer_check()
            if self.delta and int(iters) > 0 and not self.delta_state()["anchors_valid"]:
                self.refresh_products()
            N.check(self.lib.pdlp_iterate(self.h, int(iters), int(adaptive)), "pdlp_iterate")
            return
        if self.comm is None or self.lib_comm:      # single GPU, or the exchange runs inside the library (RCCL)
            N.check(self.lib.pdlp_iterate(self.h, int(iters), int(adaptive)), "pdlp_iterate")
            return
        a = int(adaptive)
        lib, h, comm = self.lib, self.h, self.comm
        # what the other ranks need of a half-step's result: xbar and y -- or, in delta mode, the float32 differences
        # x+ - x and y+ - y (half the bytes on the wire); all four live at fixed addresses
        red = self.buffer(N.BUF_RED)
        xbar = self.buffer(N.BUF_GDX if self.delta else N.BUF_XBAR)
        gdy = self.buffer(N.BUF_GDY) if self.delta else None
        iters = int(iters)
        if self.delta and iters > 0 and not self.delta_state()["anchors_valid"]:
            self.refresh_products()
        for it in range(iters):
            # With a chunked exchange the result of a half-step leaves piece by piece (_half_in_pieces): piece c's all-gather is
            # issued behind the rows it is made of and runs while the rows of piece c + 1 are still being multiplied.
            wx = self._half_in_pieces(False, a, xbar)
            # the panels of K that meet this rank's own block of xbar are multiplied while the other blocks are still on the wire
            # (the exchange is issued first, asynchronously, so the panels go onto the handle's own stream: no fork / join); the
            # same for K' and y below
            wx = self._start_exchange(0, xbar, wx)
            N.check(lib.pdlp_dual_half_begin(h, a), "pdlp_dual_half_begin")
            self._exchange(0, xbar, wx)                    # K xbar needs every rank's block of xbar
            ynew = gdy if self.delta else self.buffer(N.BUF_Y_PREV)     # where the dual half-step writes (the buffers alternate)
            wy = self._half_in_pieces(True, a, ynew)
            ar = None
            if adaptive:
                # the rank's three sums of the step-size rule need only this iteration's partial sums: the kernel that adds them up
                # runs while y is on the wire.  With pieces their all-reduce queues up behind the pieces and is reduced while the
                # panels those pieces complete are multiplied (same sums: only the order in which independent work is issued differs)
                N.check(lib.pdlp_adaptive_reduce(h), "pdlp_adaptive_reduce")
                if wy is not None:
                    ar = comm.all_reduce_sum_async(red)
            wy_sent = wy is not None
            wy = self._start_exchange(1, ynew, wy)
            if it + 1 < iters:                             # (the new y is final: a rejected adaptive step is kept, quirk Q1)
                N.check(lib.pdlp_primal_half_begin(h), "pdlp_primal_half_begin")
            self._exchange(1, ynew, wy)                    # (no product under way after the last iteration: the pieces just arrive)
            if adaptive:
                if not wy_sent:
                    comm.all_reduce_sum(red)
                elif ar is not None:
                    ar.wait()
                N.check(lib.pdlp_adaptive_update(h), "pdlp_adaptive_update")
        if not adaptive and iters > 0:
            N.check(self.lib.pdlp_fixed_advance(self.h, int(iters)), "pdlp_fixed_advance")

    def adaptive_retry(self):
        """discard the adaptive iteration just taken (its trial was rejected: ``scalars()["accepted"] == 0``) so that it can be
        issued again with the shrunk step size -- ``pdlp_adaptive_retry`` (include/pdlp_hip.h), SURVEY quirk Q1's optional flag"""
        N.check(self.lib.pdlp_adaptive_retry(self.h), "pdlp_adaptive_retry")

    # ---- restart machinery --------------------------------------------------------------------------
    def flush_average(self, adaptive: bool = True):
        """close the averaging period before a restart check (after ``kkt(CUR)``: include/pdlp_hip.h, pdlp_flush_average)"""
        N.check(self.lib.pdlp_flush_average(self.h, int(bool(adaptive))), "pdlp_flush_average")

    def compute_average(self):
        N.check(self.lib.pdlp_compute_average(self.h), "pdlp_compute_average")

    def kkt(self, which: int, omega: float, unscaled: bool = False) -> dict:
        """compute_residuals_and_duality_gap + KKT_error at CUR / AVG / PREV (helpers.py:53-108)."""
        if self.exact is not None and self.delta and not self.delta_state()["anchors_valid"]:
            self.refresh_products()
        if self.comm is not None and self.delta:
            # the current iterate is evaluated from the anchors (its pending dy was gathered by the iteration); a candidate
            # from the float32 difference candidate - current, which every rank forms in full
            if not self.delta_state()["anchors_valid"]:
                self.refresh_products()
            if which != N.CUR:
                for b in (N.BUF_X_CUR, N.BUF_Y_CUR, {N.AVG: N.BUF_X_AVG, N.PREV: N.BUF_X_PREV}[which],
                          {N.AVG: N.BUF_Y_AVG, N.PREV: N.BUF_Y_PREV}[which]):
                    self._gather(b)
        elif self.comm is not None:
            self._gather({N.CUR: N.BUF_X_CUR, N.AVG: N.BUF_X_AVG, N.PREV: N.BUF_X_PREV}[which])
            self._gather({N.CUR: N.BUF_Y_CUR, N.AVG: N.BUF_Y_AVG, N.PREV: N.BUF_Y_PREV}[which])
        N.check(self.lib.pdlp_kkt_local(self.h, which, int(unscaled)), "pdlp_kkt_local")
        if self.comm is not None:
            self.comm.all_reduce_sum(self.buffer(N.BUF_RED))
        out = (C.c_double * 6)()
        N.check(self.lib.pdlp_kkt_finish(self.h, float(omega), out), "pdlp_kkt_finish")
        self._peer_check()                       # (the stream has been synchronised: a wait that gave up shows now)
        return dict(pr=out[0], dr=out[1], gap=out[2], p=out[3], d_adj=out[4], kkt=out[5])

    def restart(self, which: int):
        N.check(self.lib.pdlp_restart(self.h, which), "pdlp_restart")

    def restart_distance(self) -> Tuple[float, float]:
        """(||x - x_last_restart||^2, ||y - y_last_restart||^2) over all ranks (enhancements.py:74-75)"""
        N.check(self.lib.pdlp_restart_distance_local(self.h), "pdlp_restart_distance_local")
        if self.comm is not None:
            self.comm.all_reduce_sum(self.buffer(N.BUF_RED))
        out = (C.c_double * N.NRED)()
        N.check(self.lib.pdlp_read_red(self.h, out), "pdlp_read_red")
        return out[0], out[1]

    def mark_restart_point(self):
        N.check(self.lib.pdlp_mark_restart_point(self.h), "pdlp_mark_restart_point")

    # ---- infeasibility detection (opt-in) ---------------------------------------------------------------
    INFEAS_STATUS = (None, "DUAL_INFEASIBLE", "PRIMAL_INFEASIBLE")          # enhancements.py:142,159,161

    def infeas_reset(self):
        """lam_prev = 0 (pdhg.py:39-40)"""
        N.check(self.lib.pdlp_infeas_reset(self.h), "pdlp_infeas_reset")

    def detect_infeasibility(self, tol: float, diagnostics: bool = False):
        """detect_infeasibility (enhancements.py:80-161) for the step just taken, including the lambda of
        pdhg.py:90 and the lam_prev update of pdhg.py:101.  Returns the reference's status string or None."""
        N.check(self.lib.pdlp_infeas_begin(self.h), "pdlp_infeas_begin")
        if self.comm is not None:
            self.comm.all_gather(self.buffer(N.BUF_DX))
            self.comm.all_gather(self.buffer(N.BUF_DY))
            if self.delta:      # lambda = proj(c - K'y) multiplies the COMPLETE y; delta iterations exchange only the differences
                self._gather(N.BUF_Y_CUR)
        N.check(self.lib.pdlp_infeas_local(self.h, float(tol)), "pdlp_infeas_local")
        if self.comm is not None:
            self.comm.all_reduce_sum(self.buffer(N.BUF_RED))
        st, diag = C.c_int32(0), (C.c_double * 8)()
        N.check(self.lib.pdlp_infeas_finish(self.h, float(tol), C.byref(st), diag), "pdlp_infeas_finish")
        status = self.INFEAS_STATUS[st.value]
        return (status, list(diag)) if diagnostics else status

    # ---- populations of points (fishnet) ------------------------------------------------------------------
    MV_WIDTHS = (8, 16, 32)

    def mv_steps(self, X: torch.Tensor, Y: torch.Tensor, steps: int, eta: float, omega: float, theta: float = 1.0):
        """``steps`` fixed-step PDHG iterations on every column of X (n x j) / Y (m x j), in place -- PDHG_step
        (spectral_casting.py:254-293); j in MV_WIDTHS, one pass over each matrix per step for all j points."""
        j = self._mv_check(X, Y)
        work = self._mv_work.get(j)         # kept per width: stream-ordered reuse, no allocation or sync per call
        if work is None:
            work = self._mv_work[j] = torch.empty((2 * self.n + self.m) * j, dtype=self.dtype, device=self.device)
        N.check(self.lib.pdlp_mv_steps(self.h, j, int(steps), float(eta), float(omega), float(theta), X.data_ptr(), Y.data_ptr(),
                                       work.data_ptr()), "pdlp_mv_steps")

    def mv_gap(self, X: torch.Tensor, Y: torch.Tensor) -> list:
        """signed duality gap (adjusted dual - primal objective) of every column -- get_best_pts (spectral_casting.py:215-234)"""
        j = self._mv_check(X, Y)
        work = torch.empty(256 * j * 4 + j * 4, dtype=torch.float64, device=self.device)
        out = (C.c_double * j)()
        N.check(self.lib.pdlp_mv_gap(self.h, j, X.data_ptr(), Y.data_ptr(), work.data_ptr(), out), "pdlp_mv_gap")
        return list(out)

    def mv_product(self, X: torch.Tensor) -> torch.Tensor:
        """K X for a population X (n x j, j in MV_WIDTHS) in one pass over K -- ``pts_y = K @ pts`` (spectral_casting.py:100)"""
        Y = torch.empty(self.m, int(X.shape[1]), dtype=self.dtype, device=self.device)
        j = self._mv_check(X, Y)
        N.check(self.lib.pdlp_mv_product(self.h, j, X.data_ptr(), Y.data_ptr()), "pdlp_mv_product")
        return Y

    def mv_combine(self, V: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
        """V @ W for a population V (rows x j, j <= 32) and weights W (j x nw, nw <= 32) -- the convex combinations of the breeding
        rounds (spectral_casting.py:133-141)"""
        V, W = V.contiguous(), W.to(device=self.device, dtype=self.dtype).contiguous()
        rows, j = V.shape
        nw = int(W.shape[1])
        if W.shape[0] != j or not (1 <= j <= 32 and 1 <= nw <= 32) or V.dtype != self.dtype or V.device != self.device:
            raise ValueError(f"mv_combine: {tuple(V.shape)} @ {tuple(W.shape)}")
        out = torch.empty(rows, nw, dtype=self.dtype, device=self.device)
        N.check(self.lib.pdlp_mv_combine(N.PDLP_F32 if self.dtype == torch.float32 else N.PDLP_F64, rows, j, V.data_ptr(), W.data_ptr(), nw,
                                         out.data_ptr(), self.stream.cuda_stream), "pdlp_mv_combine")
        return out

    def _mv_check(self, X, Y) -> int:
        if self.comm is not None:
            raise N.PdlpError("the population kernels run on one GPU")
        j = int(X.shape[1])
        ok = (X.dim() == 2 and Y.dim() == 2 and X.shape == (self.n, j) and Y.shape == (self.m, j) and j in self.MV_WIDTHS
              and X.is_contiguous() and Y.is_contiguous() and X.dtype == self.dtype == Y.dtype and X.device == self.device == Y.device)
        if not ok:
            raise ValueError(f"population of {tuple(X.shape)} / {tuple(Y.shape)}: need contiguous n x j and m x j, j in {self.MV_WIDTHS}")
        return j

    # ---- products -----------------------------------------------------------------------------------
    def spmv(self, v_full: torch.Tensor, transpose: bool = False) -> torch.Tensor:
        """K v (or K' v) for this rank's rows; ``v_full`` must be complete."""
        v = as_vec(v_full, self.m if transpose else self.n, self.device, self.dtype)
        out = torch.empty(self.nl if transpose else self.ml, dtype=self.dtype, device=self.device)
        N.check(self.lib.pdlp_spmv(self.h, int(transpose), v.data_ptr(), out.data_ptr()), "pdlp_spmv")
        return out

    def power_iteration(self, b0: torch.Tensor, iters: int = 100) -> float:
        """spectral_norm_estimate_torch (helpers.py:41-51) with the start vector given."""
        b0 = as_vec(b0, self.n, self.device, self.dtype)
        if self.comm is None:
            wn = torch.empty(self.n, dtype=self.dtype, device=self.device)
            wm = torch.empty(self.m, dtype=self.dtype, device=self.device)
            s = C.c_double(0)
            N.check(self.lib.pdlp_power_iteration(self.h, b0.data_ptr(), int(iters), wn.data_ptr(), wm.data_ptr(), C.byref(s)),
                    "pdlp_power_iteration")
            return s.value
        # sharded: the same recurrence with the exchange between the two products (setup, runs once)
        b = b0.clone()
        t = torch.zeros(self.m, dtype=self.dtype, device=self.device)
        r0, r1 = self.rows
        c0, c1 = self.cols
        for _ in range(int(iters)):
            t[r0:r1] = self.spmv(b, False)
            self.comm.all_gather(t)
            b[c0:c1] = self.spmv(t, True)
            self.comm.all_gather(b)
            b /= torch.linalg.norm(b)
        t[r0:r1] = self.spmv(b, False)
        self.comm.all_gather(t)
        return float(torch.linalg.norm(t))

    def synchronize(self):
        self.stream.synchronize()
