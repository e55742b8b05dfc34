#!/usr/bin/env python3
"""BASELINE.json configs[2]: "Mittelmann LP benchmark instance (e.g. neos3) on single MI355X vs CPU baseline".
neos3 itself is not available offline (no network, not in the reference): this times a synthetic of the same
shape and skew (512 209 x 6 624, 1.54M non-zeros, heavy-tailed column counts -- the generator of
tests/test_gpu_parity.py::test_neos3_shaped_instance_matches_oracle) on the GPU engine and on the CPU oracle.
Give an .mps path as argv[1] to time a real instance instead (parsed by torchpdlp_amd.mps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torchpdlp_amd as tp
from torchpdlp_amd.solver import PdhgDriver, estimate_sigma
from oracle import oracle as orc          # CPU baseline leg only

dev = torch.device("cuda", 0)
if len(sys.argv) > 1:
    c, K, q, m_ineq, l, u = tp.mps_to_standard_form(sys.argv[1], device=dev)
    name = os.path.basename(sys.argv[1])
else:
    rng = np.random.default_rng(8)
    m, n, nnz = 512_209, 6_624, 1_542_816
    w = rng.pareto(1.1, n) + 0.05
    cols = rng.choice(n, size=nnz, p=w / w.sum())
    rows = np.sort(rng.integers(0, m, size=nnz))
    key = np.unique(rows.astype(np.int64) * n + cols)
    rows, cols = (key // n).astype(np.int64), (key % n).astype(np.int32)
    rp = np.zeros(m + 1, np.int32)
    rp[1:] = np.cumsum(np.bincount(rows, minlength=m))
    va = rng.standard_normal(len(cols)).astype(np.float32)
    m_ineq = m // 2
    c = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(dev)
    q = torch.from_numpy(rng.standard_normal(m).astype(np.float32) * 0.1).to(dev)
    l = torch.zeros(n, device=dev)
    u = torch.full((n,), float("inf"), device=dev)
    u[::3] = 5.0
    K = tp.CsrPair(m, n, torch.from_numpy(rp).to(dev), torch.from_numpy(cols).to(dev), torch.from_numpy(va).to(dev))
    name = "neos3-shaped synthetic"
eng = tp.PdlpEngine.from_full(K, c, q, l, u, m_ineq)
print(f"{name}: {K.m} x {K.n}, {K.nnz} non-zeros, longest row of K' {int((K.t_rowptr[1:] - K.t_rowptr[:-1]).max())}, "
      f"tiles={[t is not None for t in eng.tiles]}", flush=True)
for adaptive in (True, False):
    drv = PdhgDriver(eng, restart_period=40, primal_update=True, adaptive=adaptive, tol=1e-30)
    drv.start(estimate_sigma(eng, power_iters=20, seed=0))
    done = 0
    while done < 200:
        done += drv.advance(200 - done)
    torch.cuda.synchronize()
    t0, done, steps = time.time(), 0, 4000
    while done < steps:
        done += drv.advance(steps - done)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"GPU {'adaptive' if adaptive else 'fixed'}: {steps / dt:.0f} iterations/s ({dt / steps * 1e6:.1f} us/iteration incl. restart checks)", flush=True)
h = lambda t: t.reshape(-1).cpu().numpy()
o = orc.OracleLP(K.m, K.n, m_ineq, h(K.rowptr), h(K.colidx), h(K.val), h(c), h(q), h(l), h(u),
                 trans=(h(K.t_rowptr), h(K.t_colidx), h(K.t_val)))
cores = orc.set_threads(int(os.environ.get("PDLP_CPU_THREADS", "16")))
x, y = np.zeros(K.n, np.float32), np.zeros(K.m, np.float32)
eta, om = np.float32(0.01), np.float32(1.0)
for k in range(3):
    x, y, _, eta, _ = o.step_adaptive(x, y, eta, om, 1.0, k + 1)
t0, it = time.time(), 0
while time.time() - t0 < 5.0:
    x, y, _, eta, _ = o.step_adaptive(x, y, eta, om, 1.0, it + 4)
    it += 1
step = (time.time() - t0) / it
t1 = time.time(); o.kkt(x, y, om); kkt = time.time() - t1
print(f"CPU oracle (C + OpenMP, {cores} threads) adaptive: {1.0 / (step + 3 * kkt / 40):.0f} iterations/s "
      f"({step * 1e3:.2f} ms/step, KKT pass {kkt * 1e3:.2f} ms x3/40)", flush=True)
