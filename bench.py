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
Started as a plain ``python bench.py --gpus N`` (no WORLD_SIZE in the environment) with N > 1, the script starts the N rank
processes itself (``spawn_ranks``: N child processes with the launcher's environment -- before this process has touched a GPU,
never an exec), relays rank 0's JSON line and exits non-zero if any rank fails.  ``--ruiz`` benchmarks BASELINE configs[4]: the sharded
Ruiz sweeps first (no rank holds the whole LP), then the same timed region on the scaled problem.

Also reports, in the same JSON line:
  roofline     algorithmic HBM bytes of the dominant kernel per launch / its mean duration (HIP events); `traffic` (PMC bytes per
               launch) and `measured_read_ceiling` (a streaming-read probe of this GPU) are recorded runs from profiles/
  cpu_baseline the CPU oracle (C port, OpenMP) on a bounded sample of the same workload, rank 0, N=1 only
  time_to_tol  the second half of the metric: after the timed region the same LP is solved from zero to --solve-tol (default
               1e-4, the reference's default; every N); "recorded" carries the separately measured 1e-8 runs of profiles/
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
INFINITY_CACHE_BYTES = 256 << 20   # MI355X_MICROARCH.md: 256 MB of memory-side cache in front of HBM


def spawn_ranks(gpus: int, argv, script: str = None, env: dict = None, launch_timeout: float = None, grace: float = None) -> int:
    """Start `gpus` rank processes of `script` (this file) as CHILD processes -- one per GPU, each with the launcher environment
    torch.distributed expects (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT) --, pass rank 0's JSON line
    through and return the exit code.  The caller has not touched the GPU yet, and nothing is exec'ed: the parent stays a plain
    Python process that waits.  (Not torch.distributed.run: its argument parser claims abbreviations such as --n or --m of the
    script's own flags.)

    * The rendezvous store lives in THIS process (a TCPStore on a port the OS picks while the socket stays open; the ranks join it
      as clients, TORCHELASTIC_USE_AGENT_STORE like torch's own launcher): no window in which another process can take the port.
    * If a rank fails the others are stopped and the job fails -- unless rank 0's result line has already arrived: a measured
      line is never thrown away because a later, optional phase (or the teardown of a communicator) went wrong.
    * Watchdog: once the result line is there the ranks get `grace` seconds (PDLP_BENCH_GRACE, 120) to leave by themselves, then
      they are terminated and the line is relayed, exit 0.  Without a line the job is stopped after `launch_timeout` seconds
      (PDLP_BENCH_LAUNCH_TIMEOUT, 3300) and fails.  The wait loop can therefore never hold a pool slot forever."""
    launch_timeout = float(os.environ.get("PDLP_BENCH_LAUNCH_TIMEOUT", "3300")) if launch_timeout is None else float(launch_timeout)
    grace = float(os.environ.get("PDLP_BENCH_GRACE", "120")) if grace is None else float(grace)
    base = dict(os.environ if env is None else env)
    store = None
    try:                                                         # (importing torch.distributed initialises no GPU)
        from torch.distributed import TCPStore
        store = TCPStore("127.0.0.1", 0, int(gpus), True, wait_for_workers=False)
        port = store.port
        base["TORCHELASTIC_USE_AGENT_STORE"] = "True"
    except Exception as ex:                                      # no torch here (tests of the launcher alone): a free port, the old way
        print(f"bench.py: launcher-hosted store unavailable ({type(ex).__name__}); falling back to a probed port", file=sys.stderr)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC (RCCL between processes on this host driver)
    base.setdefault("OMP_NUM_THREADS", "4")
    base.update(WORLD_SIZE=str(int(gpus)), LOCAL_WORLD_SIZE=str(int(gpus)), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(int(gpus)):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        # rank 0's stdout carries the result line; the other ranks' output goes to this process's stderr
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True, env=e))
    lines = []
    import threading
    is_result = lambda ln: ln.startswith("{") and '"metric"' in ln
    got_line = [None]                                            # time the (latest) result line arrived

    def pump():
        for line in procs[0].stdout:
            lines.append(line.rstrip("\n"))
            if is_result(lines[-1]):
                got_line[0] = time.time()
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    t_start = time.time()
    rc, pending, why = 0, set(range(len(procs))), None
    while pending and rc == 0 and why is None:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    rc = code if 0 < code < 256 else 1
                    print(f"bench.py: rank {r} of {gpus} failed (exit code {code})", file=sys.stderr)
        now = time.time()
        if got_line[0] is not None and now - got_line[0] > grace:
            why = f"ranks still running {grace:.0f} s after the result line"
        elif got_line[0] is None and now - t_start > launch_timeout:
            why = f"no result line after {launch_timeout:.0f} s"
        if pending and rc == 0 and why is None:
            time.sleep(0.05)
    if why is not None:
        print(f"bench.py: {why}: stopping the ranks", file=sys.stderr)
    for r in sorted(pending):                                    # stop exactly the processes started here
        procs[r].terminate()
    for r in sorted(pending):
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    th.join(timeout=10)
    del store
    result = [ln for ln in lines if is_result(ln)]
    for ln in lines:
        if ln not in result:
            print(ln, file=sys.stderr)
    if result:
        line = result[-1]
        if rc != 0 or why is not None:
            # a salvaged line is still the measurement (exit code 0: the driver must not lose it), but it says so itself
            print("bench.py: relaying the result line that was measured before the failure", file=sys.stderr)
            reason = why if why is not None else f"a rank exited with code {rc} after the result line"
            try:
                obj = json.loads(line)
                obj["degraded"] = True
                obj["degraded_reason"] = (obj.get("degraded_reason", "") + "; " if obj.get("degraded_reason") else "") + reason
                line = json.dumps(obj)
            except Exception:
                pass
        print(line, flush=True)
        return 0
    if rc != 0:
        return rc
    print("bench.py: " + (why or "the ranks finished without a result line"), file=sys.stderr)
    return 1


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=None, help="ranks (= GPUs); default: WORLD_SIZE of the launcher, else 1")
    p.add_argument("--ruiz", action="store_true", help="BASELINE configs[4]: Ruiz-precondition the LP first (sharded sweeps when N > 1)")
    p.add_argument("--exchange-chunks", default="auto",
                   help="N > 1: pieces in which a gathered vector travels (1..4), or auto: time this machine's all-gather against a "
                        "rank's product and use 2 pieces when the all-gather is more than half a product long")
    p.add_argument("--lib-comm", choices=["auto", "off"], default="auto",
                   help="N > 1 under RCCL: try the exchange inside the library (cross-checked against the torch.distributed loop, "
                        "falls back on any difference) or stay on the torch.distributed loop")
    p.add_argument("--direct-exchange", choices=["auto", "off"], default="auto",
                   help="N > 1 (2..8 ranks of one node): try the direct exchange -- the half-steps store their blocks into the other "
                        "ranks' memory over HIP IPC / xGMI, no collective in the iteration (pdlp_peer_*); cross-checked against the "
                        "torch.distributed loop, adopted only if faster")
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=40)
    # (under `python -m torch.distributed.run ... bench.py` use the long spellings or the environment: that launcher's parser
    # takes --n and --m for abbreviations of its own options)
    p.add_argument("--n", "--variables", dest="n", type=int, default=int(os.environ.get("PDLP_BENCH_N", 10_000_000)))
    p.add_argument("--m", "--constraints", dest="m", type=int, default=None)
    p.add_argument("--nnz-per-row", type=int, default=int(os.environ.get("PDLP_BENCH_NNZ", 100)))
    p.add_argument("--mode", choices=["adaptive", "fixed"], default="adaptive")
    p.add_argument("--dtype", choices=["f32", "f64", "mixed"], default="f32",
                   help="mixed: float32 matrix under float64 vectors, delta mode (the path to 1e-8)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--solve-tol", type=float, default=1e-4,
                   help="after the timed region: a full restarted solve of the same LP to this relative KKT tolerance (the second half "
                        "of BASELINE's metric, reported as time_to_tol; N=1 only; 0 = skip)")
    p.add_argument("--solve-limit", type=float, default=120.0, help="time limit of that solve, seconds")
    p.add_argument("--solve-max-kkt", type=int, default=100_000,
                   help="KKT-pass cap of that solve (the reference's max_kkt; the 10M LP needs ~152k passes to reach 1e-8)")
    p.add_argument("--cpu-sample-rows", type=int, default=4_000_000,
                   help="most rows (= columns) of the CPU baseline's sample; the host's free memory may lower it")
    p.add_argument("--kernel-reps", type=int, default=20)
    p.add_argument("--check-periods", type=int, default=5, help="restart periods the cost of one check is averaged over (after the timed region)")
    return p.parse_args(argv)


def algorithmic_bytes(n, m, nnz, sv, si, adaptive, sm=None):
    """SURVEY.md section 8d, per launch of each half-step kernel (each array counted once); sv = bytes of a vector entry,
    sm = bytes of a matrix value (they differ in mixed precision)"""
    sm = sv if sm is None else sm
    primal = nnz * (sm + si) + (n + 1) * si + m * sv + 8 * n * sv          # K' stream, gather y, x c l u | x+ xbar | sum RMW
    dual = nnz * (sm + si) + (m + 1) * si + n * sv + 5 * m * sv            # K stream, gather xbar, y q | y+ | sum RMW
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


def available_cores() -> int:
    """CPU cores this process may really use: its affinity mask, cut by a cgroup quota if there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def host_cores() -> int:
    """threads of the CPU baseline: every core available to this process ("all physical cores, count stated");
    PDLP_CPU_THREADS caps it"""
    cap = os.environ.get("PDLP_CPU_THREADS")
    return max(1, min(available_cores(), int(cap))) if cap else available_cores()


def host_memory_bytes() -> int:
    """memory the CPU baseline's sample may plan with: MemAvailable, cut by a cgroup limit if there is one"""
    avail = 16 << 30
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except Exception:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            used = int(open("/sys/fs/cgroup/memory.current").read())
            avail = min(avail, max(0, int(lim) - used))
    except Exception:
        pass
    return avail


def cpu_baseline(args, sv_dtype):
    """The CPU oracle (oracle/, a C restatement of the reference pinned by tests/golden) timed on the host
    cores of this box on a bounded sample: the same generator with fewer rows/columns and the same
    non-zeros per row; the rate is scaled by non-zero count to the full workload (SpMV dominated)."""
    from oracle import oracle as orc       # checker/baseline only -- never on the product path
    import numpy as np
    # the largest sample the host's memory allows (both CSR copies as numpy arrays + the torch tensors they are copied from: about
    # 40 bytes per non-zero while the oracle is being set up), at most --cpu-sample-rows (default 4M rows: 3.2 GB per copy at 100 per
    # row -- far beyond the caches of any host, and a bench run still ends within minutes)
    mem_rows = int(0.5 * host_memory_bytes() / (40.0 * args.nnz_per_row))
    rows = max(1000, min(args.cpu_sample_rows, args.n, mem_rows))
    lp = tp.gen_lp(rows, rows, args.nnz_per_row, seed=0, device="cuda", dtype=sv_dtype)
    K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
    h = lambda t: t.cpu().numpy()
    o = orc.OracleLP(lp.m, lp.n, lp.m_ineq, h(K.rowptr), h(K.colidx), h(K.val), h(lp.c), h(lp.q), h(lp.l), h(lp.u),
                     dtype=np.float32 if sv_dtype == torch.float32 else np.float64,
                     trans=(h(K.t_rowptr), h(K.t_colidx), h(K.t_val)))
    adaptive = args.mode == "adaptive"
    scale = (lp.nnz / float(args.nnz_per_row * (args.m or args.n)))
    avail = available_cores()

    def measure(threads, budget_s):
        """>= 3 iterations (and up to `budget_s` seconds) of the oracle with `threads` OpenMP threads + one KKT pass"""
        cores = orc.set_threads(threads)
        x, y = np.zeros(lp.n, o.dtype), np.zeros(lp.m, o.dtype)
        eta, om = o.dtype.type(0.01), o.dtype.type(1.0)

        def one(k, x, y, eta):
            if adaptive:
                x, y, _, eta, _ = o.step_adaptive(x, y, eta, om, 1.0, k)
            else:
                x, y = o.step_fixed(x, y, eta, om, 1.0)
            return x, y, eta
        x, y, eta = one(1, x, y, eta)
        t0, iters = time.time(), 0
        while iters < 3 or (time.time() - t0 < budget_s and iters < 200):
            x, y, eta = one(iters + 2, x, y, eta)
            iters += 1
        step_s = (time.time() - t0) / iters
        o.kkt(x, y, om)
        t1 = time.time()
        o.kkt(x, y, om)
        kkt_s = time.time() - t1
        per_iter = step_s + 3.0 * kkt_s / 40.0            # the reference's restart cadence
        return dict(value=round(scale / per_iter, 4), cores=cores, iterations=iters, step_ms=round(step_s * 1e3, 2), kkt_ms=round(kkt_s * 1e3, 2))
    orc.set_threads(host_cores())
    if host_cores() > 16:
        o.spread()                           # (NUMA: pages of the matrix arrays go to the threads that stream them)
    top = measure(host_cores(), 10.0)
    cores = top["cores"]
    out = dict(value=top["value"], unit="iterations/s", cores=cores, kind="port",
               machine_cores=os.cpu_count(), cores_available_to_this_process=avail,
               sample=f"oracle (C, OpenMP, CSR + pre-transposed CSR) on gen_lp({rows}x{rows}, {args.nnz_per_row} nnz/row, seed 0), {cores} threads: "
                      f"{top['iterations']} {args.mode} iterations at {top['step_ms']:.1f} ms + KKT pass {top['kkt_ms']:.1f} ms x3/40; "
                      f"rate scaled by nnz ratio {scale:.4g} to the full workload")
    if cores > 16:     # the figure of rounds 1-4 (a 1-GPU box's CPU share) beside it
        try:
            sub = measure(16, 6.0)
            out["sixteen_cores"] = dict(value=sub["value"], unit="iterations/s", cores=sub["cores"],
                                        sample=f"same sample, 16 threads: {sub['iterations']} iterations at {sub['step_ms']:.1f} ms + KKT pass {sub['kkt_ms']:.1f} ms x3/40")
        except Exception as ex:
            out["sixteen_cores"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
        orc.set_threads(cores)
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
    sv, sm, si = eng.dtype.itemsize, eng.mat_dtype.itemsize, 4
    nl, ml = eng.nl, eng.ml
    nnz_k = int(eng.K[2].numel())
    nnz_kt = int(eng.KT[2].numel())
    b_primal, _ = algorithmic_bytes(nl, eng.m, nnz_kt, sv, si, adaptive, sm)
    _, b_dual = algorithmic_bytes(eng.n, ml, nnz_k, sv, si, adaptive, sm)
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
    if traffic is not None:       # (PMC counters need their own rocprofv3 passes: this is the recorded value of that profile run)
        roofline["traffic_source"] = "recorded: profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this configuration), not measured in this run"
    # the device's streaming-read ceiling beside the nominal peak -- measured in THIS run (pdlp_probe_stream_read: the tiled
    # kernel's access pattern over a 2 GiB zero buffer, 5 launches); the recorded probe of profiles/ only if that fails
    try:
        import ctypes as C
        buf = torch.zeros(1 << 29, dtype=torch.int32, device=eng.device)
        gbs = C.c_double(0)
        N.check(eng.lib.pdlp_probe_stream_read(buf.data_ptr(), buf.numel() * 4, 5, eng.stream.cuda_stream, C.byref(gbs)), "pdlp_probe_stream_read")
        del buf
        roofline["measured_read_ceiling"] = {"value": round(gbs.value, 1), "unit": "GB/s", "frac": round(achieved / gbs.value, 4),
                                             "source": "this run: pdlp_probe_stream_read, 512 workgroups x 512 threads, each its own slice, "
                                                       "four 16-byte non-temporal loads in flight per thread, 2 GiB x 5"}
    except Exception as ex:
        if ceiling:
            roofline["measured_read_ceiling"] = {"value": ceiling["GBs"], "unit": "GB/s", "frac": round(achieved / ceiling["GBs"], 4),
                                                 "source": ceiling["source"] + f" (recorded; live probe failed: {type(ex).__name__})"}
    # The right roof for an LP that lives in the Infinity Cache (256 MB): a CSR product whose gathers share no cache lines is bound by
    # the rate at which the vector memory path serves scattered 4-byte gathers, not by HBM.  Probe that rate in THIS run -- the CSR
    # kernel's launch shape, a table as long as the gathered vector, as many items as the matrix has -- and report both fractions.
    try:
        working_set = (nnz_k + nnz_kt) * (sm + si) + 12 * (eng.n + eng.m) * sv
        dom_t = 1 if ms_primal >= ms_dual else 0
        if working_set <= INFINITY_CACHE_BYTES and eng.tiles[dom_t] is None and eng.dtype == torch.float32:
            import ctypes as C
            entries = eng.m if dom_t else eng.n                      # K'y gathers y, K xbar gathers xbar
            items = max(nnz_kt if dom_t else nnz_k, 1 << 22)
            scratch = torch.empty(4 * entries + (1 << 20) + 8 * items + 8 * 4096, dtype=torch.uint8, device=eng.device)
            off = (-scratch.data_ptr()) % 256
            g = C.c_double(0)
            N.check(eng.lib.pdlp_probe_gather(scratch.data_ptr() + off, scratch.numel() - off, entries, 10, eng.stream.cuda_stream, C.byref(g)),
                    "pdlp_probe_gather")
            del scratch
            rate = (nnz_kt if dom_t else nnz_k) / (kms * 1e-3) / 1e9
            roofline["bound"] = "gather"
            roofline["gather_ceiling"] = {"value": round(g.value, 1), "unit": "G items/s",
                                          "source": f"this run: pdlp_probe_gather, {items} items of 8 bytes + one uniform random 4-byte gather each over "
                                                    f"{entries} floats, k_csr_fused's launch shape, 10 launches"}
            roofline["achieved_items"] = {"value": round(rate, 1), "unit": "G items/s"}
            roofline["frac_of_gather_ceiling"] = round(rate / g.value, 4)
            roofline["note"] = ("working set %.0f MB <= the 256 MB Infinity Cache: the kernel is bound by scattered gathers (one 128-byte line per "
                                "4-byte gather), `frac` (of the HBM peak) is kept for comparison only" % (working_set / 1e6))
    except Exception as ex:
        roofline["gather_ceiling"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
    return roofline


def exchange_phases(eng, comm, roofline, ms_per_iteration, reps: int = 10) -> dict:
    """What an iteration of a sharded run is made of, each part timed ALONE on this machine (max over ranks), so that a scaling curve
    explains itself: the two fused products of this rank's shard, the two all-gathers (x-bar, y), the 3-double all-reduce of the
    step-size rule -- and what the exchange adds on the critical path (iteration - products)."""
    dev, st = eng.device, eng.stream
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def timed(fn):
        fn()
        st.synchronize()
        a, b = ev(), ev()
        a.record(st)
        for _ in range(reps):
            fn()
        b.record(st)
        b.synchronize()
        return a.elapsed_time(b) / reps
    fx = torch.zeros_like(eng.buffer(N.BUF_XBAR))
    fy = torch.zeros_like(eng.buffer(N.BUF_Y_CUR))
    red = torch.zeros(3, dtype=torch.float64, device=dev)
    comm.dist.barrier(group=comm.group)
    t = torch.tensor([timed(lambda: comm.all_gather(fx)), timed(lambda: comm.all_gather(fy)), timed(lambda: comm.all_reduce_sum(red))],
                     dtype=torch.float64, device=dev)
    comm.all_reduce_max(t)
    product = float(roofline.get("launch_ms", 0.0)) + float(roofline.get("other_kernel_ms", 0.0))
    tp_ = torch.tensor([product], dtype=torch.float64, device=dev)
    comm.all_reduce_max(tp_)
    product = float(tp_)
    return dict(product_ms=round(product, 4), all_gather_xbar_ms=round(float(t[0]), 4), all_gather_y_ms=round(float(t[1]), 4),
                allreduce_ms=round(float(t[2]), 4), iteration_ms=round(ms_per_iteration, 4),
                exchange_exposed_ms=round(ms_per_iteration - product, 4),
                note="each part timed alone (max over ranks); exposed = iteration - the two fused products")


class Deadman:
    """A phase of a multi-rank run that may hang (first contact with a second RCCL communicator, a solve over it, a teardown) must not
    cost the line that was measured before it: ``arm(seconds, line_fn, why)`` starts a timer that -- unless ``disarm()`` comes first --
    prints ``line_fn()`` from rank 0 and ends this rank's process with exit code 0.  A process with a thread stuck inside RCCL is
    poisoned; nothing of it is reused or destroyed."""

    def __init__(self, rank: int):
        import threading
        self.rank, self._threading, self._ev = rank, threading, None

    def arm(self, seconds: float, line_fn, why: str):
        self.disarm()
        ev = self._ev = self._threading.Event()
        rank = self.rank

        def run():
            if not ev.wait(seconds):
                try:
                    line = line_fn()
                    obj = json.loads(line)
                    obj["degraded"] = True
                    obj["degraded_reason"] = f"{why}: no progress for {seconds:.0f} s, rank {rank} left with the line already measured"
                    line = json.dumps(obj)
                except Exception:
                    line = None
                if rank == 0 and line:
                    print(line, flush=True)
                sys.stderr.write(f"bench.py: rank {rank}: {why} (no progress for {seconds:.0f} s); leaving with the line already measured\n")
                sys.stderr.flush()
                os._exit(0)
        self._threading.Thread(target=run, daemon=True).start()

    def disarm(self):
        if self._ev is not None:
            self._ev.set()
            self._ev = None


def library_phase(out: dict, exchange: dict, eng, steps: int, timed_region, norm_elapsed: float, rank: int, limit: float = 90.0,
                  slack: float = 120.0, first_region_s: float = 0.0):
    """N > 1 under RCCL, AFTER the headline was measured on the torch.distributed loop: try the library's own communicator (one C call
    per restart period), cross-checked bit for bit against that loop, and repeat the timed region on it.  Whatever happens in here --
    a hang inside ncclCommInitRank included -- the line already measured is what gets printed.  Every decision uses numbers that are
    maxima over the ranks, so all ranks take the same branch."""
    # the deadline covers the communicator's set-up (2 x limit) AND a second timed region: three times the wall time of the first
    deadline = 2 * limit + slack + 3.0 * float(first_region_s)
    snapshot = json.dumps(dict(out, config=dict(out["config"], exchange=dict(
        exchange, path=f"torch.distributed loop (library path abandoned by the watchdog after {deadline:.0f} s)"))))
    deadman = Deadman(rank)
    deadman.arm(deadline, lambda: snapshot, "the library communicator phase hung")
    try:
        on = eng.enable_library_comm(timeout=limit, cross_check=True)
        exchange["log"] = list(getattr(eng, "lib_comm_log", []))
        if on:
            reg2 = timed_region()
            v1, v2 = steps / norm_elapsed, steps / reg2["norm_elapsed"]
            exchange["torch_loop_value"], exchange["library_value"] = round(v1, 3), round(v2, 3)
            if v2 >= v1:
                exchange["path"] = "library RCCL communicator (pdlp_iterate)"
                out["value"], out["ms_per_step"] = round(v2, 3), round(reg2["norm_elapsed"] / steps * 1e3, 4)
                out["timing"].update(elapsed_s=round(reg2["elapsed"], 6), raw_value=round(steps / reg2["elapsed"], 3),
                                     checks_in_timed_region=reg2["checks_in"], restarts_in_timed_region=reg2["restarts_in"],
                                     check_ms=round(reg2["check_s"] * 1e3, 3), normalised_elapsed_s=round(reg2["norm_elapsed"], 6))
            else:
                exchange["path"] = "torch.distributed loop (library path works and is bit-identical, but was slower here)"
                eng.lib_comm = False
        else:
            exchange["path"] = "torch.distributed loop (library path declined)"
    except Exception as e:
        exchange["path"] = f"torch.distributed loop (library path failed: {type(e).__name__})"
        eng.lib_comm = False
    deadman.disarm()


def peer_phase(out: dict, exchange: dict, eng, steps: int, timed_region, norm_elapsed: float, rank: int, first_region_s: float = 0.0) -> bool:
    """N > 1, AFTER the headline was measured on the torch.distributed loop: connect the ranks' handles over HIP IPC (the direct
    exchange, pdlp_peer_*: the half-steps store their blocks straight into the other ranks' memory, no collective in the iteration),
    cross-check it against that loop (two fixed-step iterations bit for bit, two adaptive ones to 1e-5) and repeat the timed region on
    it.  Nothing in it can block for long -- IPC calls fail rather than hang, every wait kernel gives up after its timeout -- but it
    runs under the watchdog all the same.  Returns whether the direct exchange became the headline."""
    deadline = float(os.environ.get("PDLP_PEER_PHASE_DEADLINE", 240.0 + 9.0 * float(first_region_s)))
    snapshot = json.dumps(dict(out, config=dict(out["config"], exchange=dict(
        exchange, path=f"torch.distributed loop (direct exchange abandoned by the watchdog after {deadline:.0f} s)"))))
    deadman = Deadman(rank)
    deadman.arm(deadline, lambda: snapshot, "the direct-exchange phase hung")
    adopted = False
    try:
        on = eng.enable_peer_exchange(cross_check=True, timeout_ms=int(os.environ.get("PDLP_PEER_TIMEOUT_MS", "20000")))
        exchange["direct_log"] = list(getattr(eng, "peer_log", []))
        if on:
            reg2 = timed_region()
            eng._peer_check()
            v1, v2 = steps / norm_elapsed, steps / reg2["norm_elapsed"]
            exchange["torch_loop_value"], exchange["direct_exchange_value"] = round(v1, 3), round(v2, 3)
            if v2 >= v1:
                # its other forms: the own block's panels between signal and wait (hides ranks that finish at different times), and
                # the push kernel on a side stream beside those panels (hides the links where the panels are long: 2, 4 ranks) --
                # what each is worth only a real run can show
                best = 0
                for form, key in ((1, "direct_exchange_value_own_block_first"), (2, "direct_exchange_value_push")):
                    eng.set_peer_form(form)
                    reg3 = timed_region()
                    eng._peer_check()
                    v3 = steps / reg3["norm_elapsed"]
                    exchange[key] = round(v3, 3)
                    if v3 > 1.01 * v2:
                        v2, reg2, best = v3, reg3, form
                eng.set_peer_form(best)
                adopted = True
                exchange["path"] = ("direct exchange (pdlp_peer_*: blocks stored into the peers over HIP IPC, no collective in the iteration; "
                                    + eng.PEER_FORMS[best] + ")")
                out["value"], out["ms_per_step"] = round(v2, 3), round(reg2["norm_elapsed"] / steps * 1e3, 4)
                out["timing"].update(elapsed_s=round(reg2["elapsed"], 6), raw_value=round(steps / reg2["elapsed"], 3),
                                     checks_in_timed_region=reg2["checks_in"], restarts_in_timed_region=reg2["restarts_in"],
                                     check_ms=round(reg2["check_s"] * 1e3, 3), normalised_elapsed_s=round(reg2["norm_elapsed"], 6))
                ph = out["timing"].get("phases")
                if isinstance(ph, dict) and isinstance(ph.get("product_ms"), (int, float)):
                    # what the direct exchange leaves on the critical path of an iteration: everything but the two products
                    ph["direct_exchange_exposed_ms"] = round(reg2["elapsed"] / steps * 1e3 - ph["product_ms"], 4)
            else:
                exchange["direct_exchange"] = "works and passes the cross-check, but was slower here"
                eng.set_peer_exchange(False)
        else:
            exchange["direct_exchange"] = "declined"
    except Exception as e:
        exchange["direct_exchange"] = f"failed: {type(e).__name__}: {e}"[:200]
        try:
            eng.set_peer_exchange(False)
        except Exception:
            pass
    deadman.disarm()
    return adopted


def _imports():
    """torch and the package, only in a process that is going to compute (after the decision to spawn ranks)"""
    global torch, tp, N, PdhgDriver, estimate_sigma
    import torch
    import torchpdlp_amd as tp
    from torchpdlp_amd import _native as N
    from torchpdlp_amd.solver import PdhgDriver, estimate_sigma


def build_engine(args, comm, dev, dt):
    """the LP resident in HBM: (engine, nnz of the whole problem, set-up notes)"""
    m = args.m or args.n
    notes = {}
    precision = "mixed" if args.dtype == "mixed" else None
    if comm is None:
        lp = tp.gen_lp(args.n, m, args.nnz_per_row, seed=0, device=dev, dtype=torch.float64 if precision else dt)
        K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)
        nnz = K.nnz
        c, q, l, u, d_col, d_row, exact = lp.c, lp.q, lp.l, lp.u, None, None, None
        if args.ruiz:
            K, c, q, l, u, data, secs = tp.ruiz_precondition(lp.c, K, lp.q, lp.l, lp.u, device=dev)
            d_col, d_row = data[0], data[1]
            notes["ruiz_seconds"] = round(secs, 2)
        vec = None
        if precision:
            from torchpdlp_amd.engine import values_are_float32
            if not (values_are_float32(K.val) and values_are_float32(K.t_val)):
                exact = K
            K, vec = K.to(dtype=torch.float32), torch.float64
        eng = tp.PdlpEngine.from_full(K, c, q, l, u, lp.m_ineq, d_col=d_col, d_row=d_row, vec_dtype=vec, exact=exact)
        del K, lp
    else:
        # the same seeded instance, but every rank generates only its own rows of K and receives its rows of K' through the
        # distributed transpose; Ruiz runs on the shards: no rank ever holds the whole LP (SURVEY 8d cfg 4 / 5)
        from torchpdlp_amd.distributed import gen_lp_shard
        eng = gen_lp_shard(args.n, m, args.nnz_per_row, 0, comm, dev, dt, precision=precision, precondition=args.ruiz)
        tn = torch.tensor([eng.nnz_local], dtype=torch.int64, device=dev)
        comm.all_reduce_sum(tn)
        nnz = int(tn)
        if args.ruiz:
            notes["ruiz_seconds"], notes["ruiz_sweeps"] = round(eng.ruiz_seconds, 2), eng.ruiz_sweeps
    return eng, nnz, notes


def main(argv=None):
    args = parse(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus is None:
        args.gpus = int(env_world) if env_world else 1
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if env_world is None and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it has not initialised any GPU)
        return spawn_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv))
    world = int(env_world or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        return 2
    _imports()
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
        assert comm.world == world == args.gpus
    m = args.m or args.n
    dt = torch.float32 if args.dtype == "f32" else torch.float64
    adaptive = args.mode == "adaptive"

    t_setup = time.time()
    eng, nnz, notes = build_engine(args, comm, dev, dt)
    torch.cuda.empty_cache()
    exchange = None
    if comm is not None:
        exchange = {"path": "torch.distributed loop", "backend": comm.backend, "ranks": comm.world}
        if args.exchange_chunks == "auto":
            exchange["pieces"] = eng.tune_exchange_chunks()
        else:
            eng.set_exchange_chunks(int(args.exchange_chunks))
            exchange["pieces"] = {"chunks": int(args.exchange_chunks)}
        exchange["pieces"]["producer_pieces"] = bool(eng.producer_pieces and eng.xchunks > 1)

    def fence():
        if comm is not None:
            comm.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if comm is None:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=dev)
        comm.all_reduce_max(tt)
        return float(tt)

    def timed_region():
        """warm-up, EXACTLY --steps timed iterations, then the cost of a restart check: one measurement of the headline number on
        whatever drives the exchange right now (a fresh driver from x = y = 0 every time)"""
        drv = PdhgDriver(eng, restart_period=40, primal_update=True, adaptive=adaptive, precondition=args.ruiz, tol=1e-4)
        sigma = estimate_sigma(eng, power_iters=20, seed=0)
        drv.start(sigma)

        def run(iters):
            done = 0
            while done < iters:
                done += drv.advance(iters - done)
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
        elapsed = max_over_ranks(time.time() - t0)
        checks_in, restarts_in = drv.checks - checks0, drv.n - restarts0
        # the cost of one restart check INCLUDING the restart work at the rate restarts fire (wall clock, synchronised), measured
        # over the next whole periods: a check that evaluates the previous iterate too costs 2 products more than one that does
        # not, and a restart adds its vector passes -- several periods average that out
        run((-drv.tt) % period)
        drv.check_seconds, c1, r1 = 0.0, drv.checks, drv.n
        run(args.check_periods * period)
        checks_m, restarts_m = drv.checks - c1, drv.n - r1
        check_s = max_over_ranks(drv.check_seconds / max(1, checks_m))
        drv.check_seconds = None
        norm_elapsed = elapsed + (args.steps / period - checks_in) * check_s
        return dict(period=period, extra=extra, elapsed=elapsed, checks_in=checks_in, restarts_in=restarts_in, checks_m=checks_m,
                    restarts_m=restarts_m, check_s=check_s, norm_elapsed=norm_elapsed)

    # N > 1: the headline is measured on the torch.distributed loop FIRST; the library's own communicator (first contact with a
    # second RCCL communicator) is tried only afterwards, under a watchdog that prints the line already measured if it hangs
    setup_s = time.time() - t_setup
    t_region = time.time()
    reg = timed_region()
    first_region_s = time.time() - t_region
    period, extra, elapsed, norm_elapsed = reg["period"], reg["extra"], reg["elapsed"], reg["norm_elapsed"]
    checks_in, restarts_in, checks_m, restarts_m, check_s = reg["checks_in"], reg["restarts_in"], reg["checks_m"], reg["restarts_m"], reg["check_s"]

    try:
        roofline = measure_roofline(args, eng, dt, adaptive, world)
    except Exception as e:          # the headline number must still be reported
        roofline = {"bound": "hbm", "error": f"{type(e).__name__}: {e}"[:200]}

    out = {
        "metric": "PDHG iterations/sec", "value": round(args.steps / norm_elapsed, 3), "unit": "iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(norm_elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64 vectors / f32 matrix" if args.dtype == "mixed" else args.dtype,
        "data": "synthetic",
        "config": {"workload": f"synthetic random feasible LP {args.n}x{m}, {args.nnz_per_row} nnz/row "
                               f"(density {args.nnz_per_row / args.n:.1e}), gen_lp recipe 'box' seed 0"
                               + (", Ruiz-preconditioned (20 sweeps, eps 1e-6)" if args.ruiz else ""),
                   "nnz": nnz, "step": f"{args.mode} PDHG iteration at the reference's restart cadence (one check = 3 KKT evaluations per "
                                       f"{period} iterations, + the restart work when one fires), normalised to steps/{period} checks",
                   "primal_weight_update": True, "kernels": {"K": eng.kernels[0], "K'": eng.kernels[1]},
                   "parallelism": "single GPU" if world == 1 else
                   f"{world} ranks: row-block shards of K and K', all-gather(xbar), all-gather(y) per iteration "
                   f"({'RCCL' if comm.backend == 'nccl' else comm.backend + ' (rehearsal, not RCCL)'})"},
        "timing": {"elapsed_s": round(elapsed, 6), "raw_value": round(args.steps / elapsed, 3), "checks_in_timed_region": checks_in,
                   "restarts_in_timed_region": restarts_in, "check_ms": round(check_s * 1e3, 3),
                   "check_ms_measured_over": {"checks": checks_m, "restarts": restarts_m},
                   "warmup_done": args.warmup + extra, "normalised_elapsed_s": round(norm_elapsed, 6)},
        "roofline": roofline,
        "setup_s": round(setup_s, 1),
    }
    out["config"].update(notes)
    deadman = Deadman(rank)
    if exchange is not None:
        exchange["env"] = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "RCCL_MSCCL_ENABLE", "HSA_FORCE_FINE_GRAIN_PCIE")}
        out["config"]["exchange"] = exchange
        # the headline is measured: from here on every collective runs under a watchdog that prints it (a rank that raises before a
        # collective leaves its peers blocked inside it, and the try/except below would swallow the only trace)
        snap0 = json.dumps(out)
        deadman.arm(300.0 + 3.0 * first_region_s, lambda: snap0, "the per-phase timing of the exchange hung")
        try:
            out["timing"]["phases"] = exchange_phases(eng, comm, roofline, norm_elapsed / args.steps * 1e3)
        except Exception as e:
            out["timing"]["phases"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        deadman.disarm()
        direct = False
        # (also in the one-GPU rehearsal -- PDLP_BENCH_SHARE_GPU=1, gloo through the host: HIP IPC between processes on one card is
        # the same code path)
        if args.direct_exchange == "auto" and world <= 8 and (comm.backend == "nccl" or os.environ.get("PDLP_BENCH_SHARE_GPU") == "1"):
            direct = peer_phase(out, exchange, eng, args.steps, timed_region, norm_elapsed, rank, first_region_s=first_region_s)
        if direct:
            exchange["library"] = "not tried: the direct exchange is the headline"
        elif args.lib_comm == "auto" and comm.backend == "nccl":
            # (chunked exchanges too since round 5: the library's form of them -- grouped in-place broadcasts on its communication
            # stream, the pieces behind the rows they are made of -- runs on real RCCL in tests/test_distributed.py::
            # test_rccl_world1_library_communicator; the cross-check below compares it bit for bit with the torch loop first)
            library_phase(out, exchange, eng, args.steps, timed_region, norm_elapsed, rank,
                          limit=float(os.environ.get("PDLP_COMM_TIMEOUT", "90")), first_region_s=first_region_s)
    # from here on (time to tolerance, teardown of the communicators) a hang must not cost the line either
    if comm is not None:
        deadman.arm(args.solve_limit + 600.0, lambda: json.dumps(dict(out, time_to_tol={"error": "abandoned by the watchdog"})),
                    "the phases after the timed region hung")
    if args.solve_tol > 0:
        # time to tolerance (untimed part of the run, the engine of the timed region re-started from zero): the reference's default
        # tolerance fits a bench run; tighter ones are separate runs (tools/time_to_tol.py), recorded in profiles/time_to_tol.json
        # (a solve of several minutes says nothing by itself: a line on stderr every minute shows a supervisor that it is alive)
        import threading
        beat_stop = threading.Event()

        def beat():
            t_beat = time.time()
            while not beat_stop.wait(60.0):
                sys.stderr.write(f"bench.py: rank {rank}: time-to-tolerance solve (tol {args.solve_tol:g}) running, {time.time() - t_beat:.0f} s\n")
                sys.stderr.flush()
        if rank == 0:
            threading.Thread(target=beat, daemon=True).start()
        try:
            from torchpdlp_amd.solver import run_pdlp
            _, obj, it, nr, jj, status, secs = run_pdlp(eng, max_kkt=args.solve_max_kkt, tol=args.solve_tol, verbose=False, primal_update=True, adaptive=adaptive,
                                                        precondition=args.ruiz, time_limit=args.solve_limit, seed=0, power_iters=100)
            ttt = {"tol": args.solve_tol, "seconds": round(max_over_ranks(secs), 2), "iterations": int(it), "restarts": int(nr),
                   "kkt_passes": int(jj), "status": status, "objective": float(obj), "n_gpus": world,
                   "includes": "power iteration (100 steps), all restart checks" + ("; Ruiz set-up reported separately" if args.ruiz else "")}
            rec = os.path.join(ROOT, "profiles", "time_to_tol.json")
            if os.path.exists(rec):
                ttt["recorded"] = json.load(open(rec)).get(f"n{args.n}_k{args.nnz_per_row}", None)
                if isinstance(ttt["recorded"], dict):      # separate builder-run solves (tools/time_to_tol.py), not timed by this run
                    ttt["recorded"] = dict(ttt["recorded"], builder_run=True)
            out["time_to_tol"] = ttt
        except Exception as e:
            out["time_to_tol"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        beat_stop.set()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        del eng
        torch.cuda.empty_cache()
        try:
            out["cpu_baseline"] = cpu_baseline(args, torch.float64 if args.dtype != "f32" else dt)
        except Exception as e:
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    deadman.disarm()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm is not None:
        # (teardown: the line is out; a rank stuck in here is the launcher's business -- spawn_ranks relays the line and stops the
        # ranks after its grace period, torch.distributed.run has its own timeouts)
        comm.dist.barrier()
        comm.dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
