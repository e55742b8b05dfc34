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
