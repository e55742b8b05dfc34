#!/usr/bin/env python3
"""Setup cost of one rank's shard on ONE GPU (no process group): generate the instance, cut rank 0's blocks for a
world of W, build the engine (tiles + kernel choice).  python tools/shard_setup_timing.py [n] [nnz_per_row] [W ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torchpdlp_amd as tp
from torchpdlp_amd.distributed import shard_arrays


class FakeComm:                      # the engine only needs rank/world here; nothing is exchanged
    def __init__(self, world):
        self.rank, self.world, self.backend, self.group = 0, world, "fake", None

    def all_gather(self, full):
        pass

    def all_reduce_sum(self, t):
        pass


n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
worlds = [int(v) for v in sys.argv[3:]] or [1, 2, 4, 8]
dev = torch.device("cuda", 0)
sync = torch.cuda.synchronize
for W in worlds:
    t0 = time.time()
    lp = tp.gen_lp(n, n, k, seed=0, device=dev)
    sync(); t1 = time.time()
    K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
    sync(); t2 = time.time()
    if W == 1:
        eng = tp.PdlpEngine.from_full(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq)
        t3 = t2
    else:
        parts = shard_arrays(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq, 0, W)
        parts.pop("part")
        sync(); t3 = time.time()
        eng = tp.PdlpEngine(comm=FakeComm(W), **parts)
    sync(); t4 = time.time()
    print(f"W={W}: gen {t1-t0:.2f}s  transpose {t2-t1:.2f}s  shard {t3-t2:.2f}s  engine (tiles + timing) {t4-t3:.2f}s  "
          f"tiles={[None if t is None else (t.lw, t.rpt, t.groups) for t in eng.tiles]}", flush=True)
    del eng, K, lp
    torch.cuda.empty_cache()
