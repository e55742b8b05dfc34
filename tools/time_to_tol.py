#!/usr/bin/env python3
"""Time-to-tolerance on the bench LP (BASELINE.json metric, second half): full restarted solve, prints the
restart log every check.  python tools/time_to_tol.py [n] [nnz_per_row] [tol] [ruiz 0/1] [max_kkt] [f32|f64|mixed]
(mixed: the generator's matrix entries are float32 numbers, so holding K in float32 under float64 vectors is the same LP
as the f64 run; PDLP_DELTA=0 switches its delta mode off)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torchpdlp_amd as tp
from torchpdlp_amd.solver import run_pdlp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
ruiz = len(sys.argv) > 4 and sys.argv[4] == "1"
max_kkt = int(sys.argv[5]) if len(sys.argv) > 5 else 40000
prec = sys.argv[6] if len(sys.argv) > 6 else "f32"        # f32 | f64 | mixed (float32 matrix, float64 vectors, delta mode)
dt = torch.float32 if prec == "f32" else torch.float64
dev = torch.device("cuda", 0)
time_limit = float(os.environ.get("TIME_LIMIT", "900"))


def _heartbeat():           # long restart periods print nothing for minutes; the GPU runner wants to see progress
    import threading
    def beat():
        while True:
            time.sleep(60)
            print(f"[heartbeat] {time.time() - t0:.0f}s", flush=True)
    threading.Thread(target=beat, daemon=True).start()


t0 = time.time()
_heartbeat()
lp = tp.gen_lp(n, n, k, seed=0, device=dev, dtype=dt)
K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val.float() if (prec == "mixed" and not ruiz) else lp.val)
if prec == "mixed" and not ruiz:
    assert bool((K.val.double() == lp.val).all())
    lp.val = None
c, q, l, u = lp.c, lp.q, lp.l, lp.u
dcol = drow = None
t_ruiz = 0.0
if ruiz:
    K, c, q, l, u, dp, t_ruiz = tp.ruiz_precondition(c, K, q, l, u, device=dev)
    dcol, drow = dp[0], dp[1]
if prec == "mixed" and ruiz:      # the scaled matrix is not float32-valued: iterate on its rounding, anchors from the float64 one
    Kh = K.to(dtype=torch.float32)
    eng = tp.PdlpEngine.from_full(Kh, c, q, l, u, lp.m_ineq, d_col=dcol, d_row=drow, vec_dtype=torch.float64, exact=K)
else:
    eng = tp.PdlpEngine.from_full(K, c, q, l, u, lp.m_ineq, d_col=dcol, d_row=drow, vec_dtype=torch.float64 if prec == "mixed" else None)
torch.cuda.synchronize()
print(f"setup {time.time()-t0:.1f}s (ruiz {t_ruiz:.2f}s) kernels={eng.kernels} precision={prec} delta={eng.delta}", flush=True)
trace = dict(kkt=[], omega=[], restarts=[])
x, obj, it, nr, j, status, secs = run_pdlp(eng, max_kkt=max_kkt, tol=tol, verbose=True, precondition=ruiz, primal_update=True,
                                           adaptive=True, time_limit=time_limit, seed=0, power_iters=100, trace=trace)
print(f"RESULT n={n} k={k} tol={tol} ruiz={ruiz} precision={prec}: status={status} obj={obj:.6f} iterations={it} restarts={nr} kkt_passes={j} "
      f"solve_time={secs:.2f}s  ({it/secs:.1f} it/s incl. power iteration and checks)", flush=True)

# ---- independent float64 check of the exit point (at 10M the only evaluator of the 1e-8 claim was the library's own
# refresh).  Plain torch ops on the ORIGINAL generator arrays (row-regular pattern: k entries per row), no library call:
# the three tests of check_termination (/root/reference/PDLP/helpers.py:110-128) on the un-scaled iterate.
from torchpdlp_amd import _native as N
from torchpdlp_amd.synthetic import _regular_matvec, _regular_rmatvec
xs, ys = eng.get_iterate(N.CUR)
xs, ys = xs.double(), ys.double()
if ruiz:                                     # x = D_col x_s, y = D_row y_s (pdhg.py:161)
    xs, ys = xs * dcol.view(-1).double(), ys * drow.view(-1).double()
del eng
torch.cuda.empty_cache()
val0 = lp.val if lp.val is not None else K.val            # (the generator's entries; float32 numbers either way)
c0, q0, l0, u0 = (t.double().view(-1) for t in (lp.c, lp.q, lp.l, lp.u))
kx = _regular_matvec(lp.colidx, val0, xs, lp.m, k, 1 << 20)
kty = _regular_rmatvec(lp.colidx, val0, ys, lp.m, lp.n, k, 1 << 20)
r = kx - q0
r[:lp.m_ineq].clamp_(max=0)
g = c0 - kty
ninf, pinf = torch.isinf(l0) & (l0 < 0), torch.isinf(u0) & (u0 > 0)
lam = torch.where(ninf & pinf, torch.zeros_like(g), torch.where(ninf, g.clamp(max=0), torch.where(pinf, g.clamp(min=0), g)))
p_obj = float((c0 * xs).sum())
d_adj = float((q0 * ys).sum() + (torch.where(ninf, torch.zeros_like(l0), l0) * lam.clamp(min=0)).sum()
              + (torch.where(pinf, torch.zeros_like(u0), u0) * lam.clamp(max=0)).sum())
rel = (float(r.norm()) / (1 + float(q0.norm())), float((g - lam).norm()) / (1 + float(c0.norm())),
       (d_adj - p_obj) / (1 + abs(p_obj) + abs(d_adj)))
box = max(float((l0 - xs).clamp(min=0).max()), float((xs - u0).clamp(min=0).max()))
ok = rel[0] <= tol and rel[1] <= tol and rel[2] <= tol
print(f"INDEPENDENT float64 check (torch ops on the generator's arrays, un-scaled iterate): primal {rel[0]:.3e}  dual {rel[1]:.3e}  "
      f"gap {rel[2]:+.3e} (signed, Q2)  |gap| {abs(rel[2]):.3e}  bound violation {box:.1e}  objective {p_obj:.6f}  "
      f"-> {'meets' if ok else 'DOES NOT meet'} tol {tol:g}" + ("" if status == "Solved" else f"  (solver status: {status})"), flush=True)
