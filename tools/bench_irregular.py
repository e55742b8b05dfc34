#!/usr/bin/env python3
"""Product time (ms, HIP events) on 10M-wide matrices that are NOT uniformly random:
  uniform      the bench matrix (10M x 10M, 100 per row)                              -> tiled kernel
  dense        the same + a dense row and a dense column of 2M entries each           -> tiles + remainder (round 1: CSR, 8x slower)
  banded       100 per row inside a band of 4096 columns around the diagonal          -> not tiled (clustered): CSR kernel, whose
                                                                                         gathers are cache friendly here
usage: python tools/bench_irregular.py [n] [per_row]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torchpdlp_amd as tp
from torchpdlp_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)


def timed(eng, reps=10):
    x = torch.randn(eng.n, device=dev)
    y = torch.randn(eng.m, device=dev)
    out = []
    for tr, v in ((0, x), (1, y)):
        eng.spmv(v, bool(tr))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(eng.stream)
        for _ in range(reps):
            eng.spmv(v, bool(tr))
        b.record(eng.stream)
        b.synchronize()
        out.append(a.elapsed_time(b) / reps)
    return out


def engine(K, mode):
    os.environ["PDLP_TILED"] = mode
    z = lambda ln: torch.zeros(ln, device=dev)
    return tp.PdlpEngine.from_full(K, z(K.n), z(K.m), z(K.n), z(K.n), 0)


def report(tag, K):
    for mode in ("auto", "0"):
        t0 = time.time()
        e = engine(K, mode)
        ms = timed(e)
        print(f"{tag:8s} PDLP_TILED={mode:4s} kernels={e.kernels}  K x {ms[0]:.3f} ms   K'y {ms[1]:.3f} ms   (setup {time.time() - t0:.1f}s)", flush=True)
        if mode == "auto" and all(kk == "csr" for kk in e.kernels):
            break
        del e
        torch.cuda.empty_cache()


lp = tp.gen_lp(n, n, k, seed=0, device=dev)
K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
report("uniform", K)
# + a dense row and a dense column
g = torch.Generator(device=dev).manual_seed(5)
nd = min(2_000_000, n // 5)
rows = torch.cat([torch.repeat_interleave(torch.arange(n, device=dev), k), torch.full((nd,), 12345, device=dev),
                  torch.randperm(n, device=dev, generator=g)[:nd]])
cols = torch.cat([lp.colidx.long(), torch.randperm(n, device=dev, generator=g)[:nd], torch.full((nd,), 777, device=dev)])
vals = torch.cat([lp.val, torch.rand(2 * nd, device=dev, generator=g)])
del K
order = torch.argsort(rows * n + cols)
rows, cols, vals = rows[order], cols[order], vals[order]
rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
rp[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
K = tp.CsrPair(n, n, rp.to(torch.int32), cols.to(torch.int32), vals)
del rows, cols, vals, order
report("dense", K)
del K
torch.cuda.empty_cache()
# banded
off = torch.sort(torch.randint(-2048, 2048, (n, k), device=dev, generator=g), dim=1)[0]
cols = ((torch.arange(n, device=dev).view(-1, 1) + off) % n).to(torch.int32)
cols = torch.sort(cols, dim=1)[0].reshape(-1)
K = tp.CsrPair(n, n, lp.rowptr, cols, lp.val)
report("banded", K)
