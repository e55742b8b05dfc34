#!/usr/bin/env python3
"""PDHG iterations/sec on a synthetic sparse LP (BASELINE.json metric), 1..8 MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one PDHG iteration of the production driver (torchpdlp_amd.solver.PdhgDriver): the two
fused SpMV half-steps (+ the step-size rule when adaptive) at the reference's restart cadence -- a restart
check (three KKT evaluations) every 40 iterations and the restart work when one fires.  The timed region is
EXACTLY --steps iterations, starting right after a restart check (the warm-up is extended, untimed, to the next
multiple of 40); it therefore holds floor(steps/40) checks.  So that the number does not depend on --steps, the
cost of the (steps/40 - checks held) missing or surplus checks is added at the measured cost of one check
(timed on the period that follows the timed region): value = steps / (elapsed + (steps/40 - checks) * check_s).
For steps a multiple of 40 that is steps / elapsed exactly.  All three raw numbers are in "timing".
The LP is resident in HBM before the clock starts.  Default workload = the one the metric is quoted
on: 10M x 10M, density 1e-5 (100 non-zeros per row), float32 like the reference; it fits one GPU.
With N > 1 the same instance is sharded (row blocks of K and K'; every rank generates its own shard), so scaling is strong.

Also reports, in the same JSON line:
  roofline     algorithmic HBM bytes of the dominant kernel per launch / its mean duration (HIP events); `traffic` (PMC bytes per
               launch) and `measured_read_ceiling` (a streaming-read probe of this GPU) are recorded runs from profiles/
  cpu_baseline the CPU oracle (C port, OpenMP) on a bounded sample of the same workload, rank 0, N=1 only
  time_to_tol  the second half of the metric: after the timed region the same LP is solved from zero to --solve-tol (default
               1e-4, the reference's default; N=1 only); "recorded" carries the separately measured 1e-8 runs of profiles/
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torchpdlp_amd as tp                                     # noqa: E402
from torchpdlp_amd import _native as N                         # noqa: E402
from torchpdlp_amd.distributed import gen_lp_shard             # noqa: E402
from torchpdlp_amd.solver import PdhgDriver, estimate_sigma    # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=40)
    p.add_argument("--n", type=int, default=int(os.environ.get("PDLP_BENCH_N", 10_000_000)))
    p.add_argument("--m", type=int, default=None)
    p.add_argument("--nnz-per-row", type=int, default=int(os.environ.get("PDLP_BENCH_NNZ", 100)))
    p.add_argument("--mode", choices=["adaptive", "fixed"], default="adaptive")
    p.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--solve-tol", type=float, default=1e-4,
                   help="after the timed region: a full restarted solve of the same LP to this relative KKT tolerance (the second half "
                        "of BASELINE's metric, reported as time_to_tol; N=1 only; 0 = skip)")
    p.add_argument("--solve-limit", type=float, default=120.0, help="time limit of that solve, seconds")
    p.add_argument("--cpu-sample-rows", type=int, default=500_000)
    p.add_argument("--kernel-reps", type=int, default=20)
    return p.parse_args()


def algorithmic_bytes(n, m, nnz, sv, si, adaptive):
    """SURVEY.md section 8d, per launch of each half-step kernel (each array counted once)."""
    primal = nnz * (sv + si) + (n + 1) * si + m * sv + 8 * n * sv          # K' stream, gather y, x c l u | x+ xbar | sum RMW
    dual = nnz * (sv + si) + (m + 1) * si + n * sv + 5 * m * sv            # K stream, gather xbar, y q | y+ | sum RMW
    if adaptive:
        dual += 2 * m * sv                                                 # cached K x read + write
    return primal, dual


def time_half_steps(eng, adaptive, reps):
    """mean duration (ms) of the two fused half-step kernels, launched in the order an iteration launches them (primal,
    dual, primal, ...: each starts with the caches in the state the other one left, as inside the timed region), HIP
    events on the stream the library launches on"""
    stream, a = eng.stream, int(adaptive)
    primal = lambda: N.check(eng.lib.pdlp_primal_half(eng.h, a))
    dual = lambda: N.check(eng.lib.pdlp_dual_half(eng.h, a))       # (flips the iterate buffers each launch: harmless here)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    evs = [(ev(), ev(), ev()) for _ in range(reps)]
    primal(); dual()
    stream.synchronize()
    for e0, e1, e2 in evs:
        e0.record(stream)
        primal()
        e1.record(stream)
        dual()
        e2.record(stream)
    stream.synchronize()
    return (sum(e0.elapsed_time(e1) for e0, e1, _ in evs) / reps, sum(e1.elapsed_time(e2) for _, e1, e2 in evs) / reps)


def host_cores() -> int:
    """CPU cores this process may really use: cgroup quota, then affinity, then the machine's count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("PDLP_CPU_THREADS", "16"))))   # a 1-GPU box shares its host: 16 cores


def cpu_baseline(args, sv_dtype):
    """The CPU oracle (oracle/, a C restatement of the reference pinned by tests/golden) timed on the host
    cores of this box on a bounded sample: the same generator with fewer rows/columns and the same
    non-zeros per row; the rate is scaled by non-zero count to the full workload (SpMV dominated)."""
    from oracle import oracle as orc       # checker/baseline only -- never on the product path
    import numpy as np
    rows = min(args.cpu_sample_rows, args.n)
    lp = tp.gen_lp(rows, rows, args.nnz_per_row, seed=0, device="cuda", dtype=sv_dtype)
    K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
    h = lambda t: t.cpu().numpy()
    o = orc.OracleLP(lp.m, lp.n, lp.m_ineq, h(K.rowptr), h(K.colidx), h(K.val), h(lp.c), h(lp.q), h(lp.l), h(lp.u),
                     dtype=np.float32 if sv_dtype == torch.float32 else np.float64,
                     trans=(h(K.t_rowptr), h(K.t_colidx), h(K.t_val)))
    cores = orc.set_threads(host_cores())
    x, y = np.zeros(lp.n, o.dtype), np.zeros(lp.m, o.dtype)
    eta, om = o.dtype.type(0.01), o.dtype.type(1.0)
    adaptive = args.mode == "adaptive"

    def one(k, x, y, eta):
        if adaptive:
            x, y, _, eta, _ = o.step_adaptive(x, y, eta, om, 1.0, k)
        else:
            x, y = o.step_fixed(x, y, eta, om, 1.0)
        return x, y, eta
    x, y, eta = one(1, x, y, eta)
    t0, iters = time.time(), 0
    while iters < 3 or (time.time() - t0 < 10.0 and iters < 200):
        x, y, eta = one(iters + 2, x, y, eta)
        iters += 1
    step_s = (time.time() - t0) / iters
    t1 = time.time()
    o.kkt(x, y, om)
    kkt_s = time.time() - t1
    per_iter = step_s + 3.0 * kkt_s / 40.0            # the reference's restart cadence
    scale = (lp.nnz / float(args.nnz_per_row * (args.m or args.n)))
    out = dict(value=round(scale / per_iter, 4), unit="iterations/s", cores=cores, kind="port",
               sample=f"oracle (C, OpenMP, CSR + pre-transposed CSR) on gen_lp({rows}x{rows}, {args.nnz_per_row} nnz/row, seed 0): "
                      f"{iters} {args.mode} iterations at {step_s * 1e3:.1f} ms + KKT pass {kkt_s * 1e3:.1f} ms x3/40; "
                      f"rate scaled by nnz ratio {scale:.4g} to the full workload")
    # flavour (i) of SURVEY 8d: what the reference itself does on a CPU -- eager torch ops on a sparse-COO K with K.T @ y
    try:
        from oracle.torch_coo import TorchCooLP
        torch.set_num_threads(cores)
        rows2 = min(rows, 200_000)
        lp2 = tp.gen_lp(rows2, rows2, args.nnz_per_row, seed=0, device="cpu", dtype=torch.float32)
        t = TorchCooLP(lp2.m, lp2.n, lp2.m_ineq, lp2.rowptr, lp2.colidx, lp2.val, lp2.c, lp2.q, lp2.l, lp2.u)
        xx, yy = torch.zeros(lp2.n, 1), torch.zeros(lp2.m, 1)
        e, w = torch.tensor(0.01), torch.tensor(1.0)
        stepf = (lambda k, xx, yy, e: t.step_adaptive(xx, yy, e, w, 1.0, k)[:2] + (e,)) if adaptive else \
                (lambda k, xx, yy, e: t.step_fixed(xx, yy, e, w, 1.0) + (e,))
        xx, yy, e = stepf(1, xx, yy, e)
        t0, it2 = time.time(), 0
        while it2 < 2 or (time.time() - t0 < 8.0 and it2 < 50):
            xx, yy, e = stepf(it2 + 2, xx, yy, e)
            it2 += 1
        s2 = (time.time() - t0) / it2
        t1 = time.time()
        t.kkt(xx, yy, w)
        k2 = time.time() - t1
        sc2 = lp2.nnz / float(args.nnz_per_row * (args.m or args.n))
        out["torch_coo"] = dict(value=round(sc2 / (s2 + 3.0 * k2 / 40.0), 4), unit="iterations/s", cores=cores, kind="port",
                                sample=f"eager torch ops on sparse-COO K (K.T @ y, three products per adaptive step like the reference) on "
                                       f"gen_lp({rows2}x{rows2}, {args.nnz_per_row} nnz/row): {it2} iterations at {s2 * 1e3:.0f} ms + KKT pass "
                                       f"{k2 * 1e3:.0f} ms x3/40; rate scaled by nnz ratio {sc2:.4g}")
    except Exception as ex:
        out["torch_coo"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
    return out


def measure_roofline(args, eng, dt, adaptive, world):
    """roofline of the dominant kernel (per launch, this rank's shard): algorithmic bytes / mean HIP-event duration"""
    sv, si = (4 if dt == torch.float32 else 8), 4
    nl, ml = eng.nl, eng.ml
    nnz_k = int(eng.K[2].numel())
    nnz_kt = int(eng.KT[2].numel())
    b_primal, _ = algorithmic_bytes(nl, eng.m, nnz_kt, sv, si, adaptive)
    _, b_dual = algorithmic_bytes(eng.n, ml, nnz_k, sv, si, adaptive)
    N.check(eng.lib.pdlp_primal_half(eng.h, int(adaptive)))   # (leaves no K'y behind from a restart check: the primal half-step
    N.check(eng.lib.pdlp_dual_half(eng.h, int(adaptive)))     #  timed below is the full kernel)
    ms_primal, ms_dual = time_half_steps(eng, adaptive, args.kernel_reps)
    kfam = lambda t: "k_tiled_fused" if eng.tiles[t] is not None else "k_csr_fused"

    def kdesc(t, epi):       # what actually runs for one half-step of this matrix
        tl = eng.tiles[t]
        if tl is None:
            return f"k_csr_fused<{epi}>" + (" over column-sorted row blocks" if "sorted" in eng.kernels[t] else "")
        if tl.groups > 1:
            return f"k_tiled_fused x {tl.groups} panel groups + k_rowsum_epilogue<{epi}>"
        return f"k_tiled_fused<{epi}>" + (" + remainder kernels" if tl.nrem else "")
    if ms_primal >= ms_dual:
        kname, kms, kbytes = f"{kdesc(1, 'PrimalEpi')} (K'y + primal update)", ms_primal, b_primal
    else:
        kname, kms, kbytes = f"{kdesc(0, 'DualEpi')} (K xbar + dual update)", ms_dual, b_dual
    achieved = kbytes / (kms * 1e-3) / 1e9
    traffic = None
    ceiling = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")       # rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, see profiles/README.md
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            key = f"n{args.n}_k{args.nnz_per_row}_{args.dtype}_{args.mode}_g{world}_{kfam(1).split('_')[1]}"
            traffic = tj.get(key, {}).get("primal" if ms_primal >= ms_dual else "dual")
            ceiling = tj.get("read_ceiling")
        except Exception:
            traffic = None
    roofline = dict(bound="hbm", kernel=kname, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, launch_ms=round(kms, 4),
                    algorithmic_bytes=int(kbytes), other_kernel_ms=round(min(ms_primal, ms_dual), 4))
    if ceiling:      # the device's measured streaming-read ceiling beside the nominal peak (a recorded probe run, not this run)
        roofline["measured_read_ceiling"] = {"value": ceiling["GBs"], "unit": "GB/s", "frac": round(achieved / ceiling["GBs"], 4),
                                             "source": ceiling["source"]}
    return roofline


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PDLP_DIST_BACKEND", "nccl")       # "gloo" + PDLP_BENCH_SHARE_GPU=1: rehearsal on a 1-GPU box
    if os.environ.get("PDLP_BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        comm = tp.Comm()
    m = args.m or args.n
    dt = torch.float32 if args.dtype == "f32" else torch.float64
    adaptive = args.mode == "adaptive"

    t_setup = time.time()
    if comm is None:
        lp = tp.gen_lp(args.n, m, args.nnz_per_row, seed=0, device=dev, dtype=dt)
        K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
        nnz = K.nnz
        eng = tp.PdlpEngine.from_full(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq)
        del K, lp
    else:
        # the same seeded instance, but every rank generates only its own rows of K and receives its rows of K' through the
        # distributed transpose: no rank ever holds the whole LP (SURVEY 8d cfg 4)
        eng = gen_lp_shard(args.n, m, args.nnz_per_row, 0, comm, dev, dt)
        tn = torch.tensor([eng.nnz_local], dtype=torch.int64, device=dev)
        comm.all_reduce_sum(tn)
        nnz = int(tn)
    torch.cuda.empty_cache()
    drv = PdhgDriver(eng, restart_period=40, primal_update=True, adaptive=adaptive, tol=1e-4)
    sigma = estimate_sigma(eng, power_iters=20, seed=0)
    drv.start(sigma)
    setup_s = time.time() - t_setup

    def run(iters):
        done = 0
        while done < iters:
            done += drv.advance(iters - done)

    def fence():
        if comm is not None:
            comm.dist.barrier()
        torch.cuda.synchronize()

    period = drv.period
    run(args.warmup)
    extra = (-drv.tt) % period                 # untimed: up to (and including) the next restart check
    run(extra)
    assert drv.tt % period == 0
    fence()
    checks0, restarts0 = drv.checks, drv.n
    t0 = time.time()
    run(args.steps)
    fence()
    elapsed = time.time() - t0
    checks_in, restarts_in = drv.checks - checks0, drv.n - restarts0
    if comm is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        comm.dist.all_reduce(tt, op=comm.dist.ReduceOp.MAX)
        elapsed = float(tt)
    # the cost of one restart check, measured (wall clock, synchronised) on the next whole period(s)
    run((-drv.tt) % period)
    drv.check_seconds, c1 = 0.0, drv.checks
    run(2 * period)
    check_s = drv.check_seconds / max(1, drv.checks - c1)
    drv.check_seconds = None
    if comm is not None:
        tt = torch.tensor([check_s], dtype=torch.float64, device=dev)
        comm.dist.all_reduce(tt, op=comm.dist.ReduceOp.MAX)
        check_s = float(tt)
    norm_elapsed = elapsed + (args.steps / period - checks_in) * check_s

    try:
        roofline = measure_roofline(args, eng, dt, adaptive, world)
    except Exception as e:          # the headline number must still be reported
        roofline = {"bound": "hbm", "error": f"{type(e).__name__}: {e}"[:200]}

    out = {
        "metric": "PDHG iterations/sec", "value": round(args.steps / norm_elapsed, 3), "unit": "iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(norm_elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"synthetic random feasible LP {args.n}x{m}, {args.nnz_per_row} nnz/row "
                               f"(density {args.nnz_per_row / args.n:.1e}), gen_lp recipe 'box' seed 0",
                   "nnz": nnz, "step": f"{args.mode} PDHG iteration at the reference's restart cadence (one check = 3 KKT evaluations per "
                                       f"{period} iterations, + the restart work when one fires), normalised to steps/{period} checks",
                   "primal_weight_update": True, "kernels": {"K": eng.kernels[0], "K'": eng.kernels[1]},
                   "parallelism": "single GPU" if world == 1 else
                   f"row-block shards of K and K' over {world} GPUs, all-gather(xbar), all-gather(y) per iteration "
                   f"({'RCCL' if comm.backend == 'nccl' else comm.backend + ' (rehearsal, not RCCL)'})"},
        "timing": {"elapsed_s": round(elapsed, 6), "raw_value": round(args.steps / elapsed, 3), "checks_in_timed_region": checks_in,
                   "restarts_in_timed_region": restarts_in, "check_ms": round(check_s * 1e3, 3),
                   "warmup_done": args.warmup + extra, "normalised_elapsed_s": round(norm_elapsed, 6)},
        "roofline": roofline,
        "setup_s": round(setup_s, 1),
    }
    if world == 1 and args.solve_tol > 0:
        # time to tolerance (untimed part of the run, the engine of the timed region re-started from zero): the reference's default
        # tolerance fits a bench run; tighter ones are separate runs (tools/time_to_tol.py), recorded in profiles/time_to_tol.json
        try:
            from torchpdlp_amd.solver import run_pdlp
            _, obj, it, nr, jj, status, secs = run_pdlp(eng, tol=args.solve_tol, verbose=False, primal_update=True, adaptive=adaptive,
                                                        time_limit=args.solve_limit, seed=0, power_iters=100)
            ttt = {"tol": args.solve_tol, "seconds": round(secs, 2), "iterations": int(it), "restarts": int(nr), "kkt_passes": int(jj),
                   "status": status, "objective": float(obj), "includes": "power iteration (100 steps), all restart checks"}
            rec = os.path.join(ROOT, "profiles", "time_to_tol.json")
            if os.path.exists(rec):
                ttt["recorded"] = json.load(open(rec)).get(f"n{args.n}_k{args.nnz_per_row}", None)
            out["time_to_tol"] = ttt
        except Exception as e:
            out["time_to_tol"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        del drv, eng
        torch.cuda.empty_cache()
        try:
            out["cpu_baseline"] = cpu_baseline(args, dt)
        except Exception as e:
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.dist.barrier()
        comm.dist.destroy_process_group()


if __name__ == "__main__":
    main()
