#!/usr/bin/env python3
"""The column-block (partial-sum) form of K'y on one rank's shard, timed on ONE GPU.

Today a rank multiplies its row block of K' (n/W rows x m columns: few rows, all 153 panels -> panel groups + k_rowsum_epilogue)
with the all-gathered y.  The alternative named by BASELINE's north star: the rank multiplies the TRANSPOSE OF ITS ROW BLOCK OF K
(n rows x m/W columns: 501 row blocks x 19 panels at W = 8, i.e. the full-occupancy fused kernel) with its OWN block of y -- no
all-gather of y -- into an n-vector of partial sums, which a reduce-scatter then sums over the ranks, and a vector pass applies the
primal update to the rank's n/W entries.  This script times the three compute parts for W ranks' worth of shape:
  (a) today's shard-shaped product + fused update (the dual half-step of tools/split_timing.py measures the same shape: 0.27 ms at W=8)
  (b) the partial product: fused tiled kernel, StoreEpi, n outputs
  (c) the vector pass over n/W reduced entries
env: WORLD (8), N (10M), K (100)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PDLP_TILED"] = "1"
import torch
import torchpdlp_amd as tp
from torchpdlp_amd import _native as N_
from torchpdlp_amd.sparse import csr_transpose

W = int(os.environ.get("WORLD", 8))
n = int(os.environ.get("N", 10_000_000)); k = int(os.environ.get("K", 100))
mr = n // W                                  # this rank's rows of K (= its constraints)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
col = torch.empty(mr * k, dtype=torch.int32, device=dev)
ch = 1 << 20
for r0 in range(0, mr, ch):
    r1 = min(mr, r0 + ch)
    blk, _ = torch.sort(torch.randint(0, n, (r1 - r0, k), generator=g, device=dev, dtype=torch.int32), dim=1)
    col[r0 * k:r1 * k] = blk.reshape(-1)
val = torch.rand(mr * k, device=dev, generator=g)
rp = torch.arange(0, (mr + 1) * k, k, dtype=torch.int64, device=dev)
# K_r' : n rows x mr columns
t_rp, t_ci, t_va = csr_transpose(rp, col, val, mr, n)
Kt = tp.CsrPair(n, mr, t_rp, t_ci, t_va, rp, col, val)
z = lambda ln: torch.zeros(ln, device=dev)
eng = tp.PdlpEngine.from_full(Kt, z(mr), z(n), z(mr), z(mr), 0)
t0 = eng.tiles[0]
print(f"WORLD={W}: K_r' is {n} x {mr}, {Kt.nnz} non-zeros; tiles: rpt {t0.rpt}, {t0.nblk} row blocks x {t0.npanel} panels, groups {t0.groups}, "
      f"remainder {t0.stats['remainder']}", flush=True)
y = torch.randn(mr, device=dev, generator=g)
out = torch.empty(n, device=dev)


def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[0], ts[len(ts) // 2]


p = timed(lambda: N_.check(eng.lib.pdlp_spmv(eng.h, 0, y.data_ptr(), out.data_ptr())))
print(f"   (b) partial product K_r' y_r -> n partial sums (fused tiled kernel, StoreEpi): min {p[0]:.3f} med {p[1]:.3f} ms", flush=True)
# value check on sampled rows (float64)
rows = torch.randint(0, n, (2000,), generator=torch.Generator().manual_seed(1)).to(dev)
a_, lens = t_rp[rows], t_rp[rows + 1] - t_rp[rows]
seg = torch.repeat_interleave(torch.arange(2000, device=dev), lens)
pos = a_[seg] + (torch.arange(int(lens.sum()), device=dev) - torch.repeat_interleave(lens.cumsum(0) - lens, lens))
want = torch.zeros(2000, dtype=torch.float64, device=dev).index_add_(0, seg, t_va[pos].double() * y.double()[t_ci[pos].long()])
err = float((out[rows].double() - want).abs().max() / want.abs().max())
print(f"       sampled rows against float64: max relative error {err:.2e}", flush=True)
# (c) the vector pass: the primal update over n/W entries from a finished K'y -- a stand-in engine of n/W variables
nl = n // W
small = tp.PdlpEngine.from_full(tp.CsrPair(8, nl, torch.arange(0, 9, dtype=torch.int64, device=dev), torch.arange(8, dtype=torch.int32, device=dev),
                                           torch.ones(8, device=dev)), z(nl), z(8), z(nl) - 1, z(nl) + 1, 0)
small.set_iterate(z(nl), z(8))
small.set_step(0.01, 1.0, 1.0, 0)
N_.check(small.lib.pdlp_kkt_local(small.h, N_.CUR, 0))          # leaves K'y of the current iterate behind: the next primal half-step is a vector pass
v = timed(lambda: (N_.check(small.lib.pdlp_kkt_local(small.h, N_.CUR, 0)), N_.check(small.lib.pdlp_primal_half(small.h, 1))))
v0 = timed(lambda: N_.check(small.lib.pdlp_kkt_local(small.h, N_.CUR, 0)))
print(f"   (c) vector pass over {nl} entries (PrimalEpi from a finished K'y): ~{max(0.0, v[1] - v0[1]):.3f} ms", flush=True)
rs_bytes = n * 4 * (W - 1) / W
print(f"   reduce-scatter of the partial sums: {rs_bytes / 1e6:.0f} MB leave and enter every rank (the same bytes as the all-gather of y it replaces)", flush=True)
