#!/usr/bin/env python3
"""Does the tiled kernel's time depend on WHERE its arrays sit?  One LP, one handle; the K' tiles' idx / val arrays are re-allocated
several times (fresh allocations, and shifted copies inside one big buffer) and the primal half-step is timed each time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, copy
import torchpdlp_amd as tp
from torchpdlp_amd import _native as N

dev = torch.device("cuda", 0)
n = int(os.environ.get("N", 10_000_000))
lp = tp.gen_lp(n, n, 100, seed=0, device=dev)
K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
eng = tp.PdlpEngine.from_full(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq)
eng.set_step(0.01, 1.0, 1.0, 0)
eng.iterate(2, True)


def t_primal(reps=40):
    lib, h, s = eng.lib, eng.h, eng.stream
    ev = lambda: torch.cuda.Event(enable_timing=True)
    evs = [(ev(), ev()) for _ in range(reps)]
    for _ in range(3):
        N.check(lib.pdlp_primal_half(h, 1)); N.check(lib.pdlp_dual_half(h, 1))
    s.synchronize()
    for a, b in evs:
        a.record(s); N.check(lib.pdlp_primal_half(h, 1)); b.record(s); N.check(lib.pdlp_dual_half(h, 1))
    s.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


t0 = eng.tiles[1]
print("as built:", f"{t_primal():.4f} ms", hex(t0.idx.data_ptr()), hex(t0.val.data_ptr()), flush=True)
items = t0.items
for trial in range(4):                      # fresh allocations of both arrays
    t = copy.copy(t0)
    junk = torch.empty((trial + 1) * 37_000_001, dtype=torch.uint8, device=dev)       # moves the allocator's cursor
    t.idx, t.val = t0.idx.clone(), t0.val.clone()
    eng.attach_tiles(1, t)
    print(f"fresh {trial}:", f"{t_primal():.4f} ms", hex(t.idx.data_ptr()), hex(t.val.data_ptr()), flush=True)
    del junk
big_i = torch.empty(items + (1 << 22), dtype=torch.int32, device=dev)
big_v = torch.empty(items + (1 << 22), dtype=torch.float32, device=dev)
for off_i, off_v in ((0, 0), (0, 64), (0, 1024), (0, 16384), (0, 1 << 18), (0, 1 << 20), (1 << 20, 0), (4096, 1 << 19), (0, (1 << 20) + 4096)):
    t = copy.copy(t0)
    t.idx = big_i[off_i:off_i + items]; t.idx.copy_(t0.idx)
    t.val = big_v[off_v:off_v + items]; t.val.copy_(t0.val)
    eng.attach_tiles(1, t)
    print(f"shift idx +{off_i * 4} B, val +{off_v * 4} B:", f"{t_primal():.4f} ms", flush=True)
