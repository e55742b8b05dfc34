#!/usr/bin/env python3
"""Time the bare tiled SpMV (K x, StoreEpi) of the library named by PDLP_LIB on the bench matrix.
Used with the timing-only ablation builds (tools/ablate_tiled.sh); results are wrong by design there."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PDLP_TILED"] = "0"
import torchpdlp_amd as tp
from torchpdlp_amd.tiled import build_tiles

n = int(os.environ.get("N", 10_000_000)); k = int(os.environ.get("K", 100)); lw = int(os.environ.get("LW", 16))
m = int(os.environ.get("M", n))      # rows (n = columns): M=1250000 is one rank's shard of the bench problem on 8 GPUs
groups = os.environ.get("NGROUPS")
rpt = os.environ.get("RPT")
dt = torch.float64 if os.environ.get("DTYPE") == "f64" else torch.float32
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
col = torch.empty(m * k, dtype=torch.int32, device=dev)
ch = 1 << 20
for r0 in range(0, m, ch):
    r1 = min(m, r0 + ch)
    blk, _ = torch.sort(torch.randint(0, n, (r1 - r0, k), generator=g, device=dev, dtype=torch.int32), dim=1)
    col[r0 * k:r1 * k] = blk.reshape(-1)
val = torch.rand(m * k, device=dev, generator=g).to(dt)
rp = torch.arange(0, (m + 1) * k, k, dtype=torch.int64, device=dev).to(torch.int32)
z = torch.zeros(n, device=dev, dtype=dt)
zm = torch.zeros(m, device=dev, dtype=dt)
one = torch.zeros(2, dtype=torch.int32, device=dev)
eng = tp.PdlpEngine(m, n, 0, (rp, col, val), (torch.zeros(n + 1, dtype=torch.int32, device=dev), one[:0], z[:0]), z, zm, z, z)
t_csr = eng._time_spmv(0, z, torch.empty(m, device=dev, dtype=dt))
t = build_tiles(rp, col, val, m, n, lw=lw, rpt=None if rpt is None else int(rpt), groups=None if groups is None else int(groups))
eng.attach_tiles(0, t)
x = torch.randn(n, device=dev, generator=g).to(dt)
out = eng.spmv(x)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
for a, b in evs:
    a.record(); eng.spmv(x); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in evs)
print(f"{os.environ.get('PDLP_LIB','default').split('/')[-1]:22s} {m}x{n} lw={lw} rpt={t.rpt} groups={t.groups} nblk={t.nblk} csr {t_csr:.3f} ms | tiled_bytes={t.bytes()/1e9:.2f}GB  "
      f"spmv min {ms[0]:.3f} ms  median {ms[len(ms)//2]:.3f} ms  -> {t.bytes()/ms[0]/1e6:.0f} GB/s of format bytes", flush=True)
