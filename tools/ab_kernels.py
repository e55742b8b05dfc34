#!/usr/bin/env python3
"""A/B timing of the two fused half-step kernels for several builds of the library in ONE process on ONE LP (run-to-run and
box-to-box spread of a bench line is +-3 %, more than most kernel changes): python tools/ab_kernels.py lib1.so lib2.so ...
env: N (10M), K (100), REPS (60), ROUNDS (3), PDLP_RUNNING_KKT etc. apply to all."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torchpdlp_amd as tp
from torchpdlp_amd import _native as N

libs = sys.argv[1:] or [N.LIB_PATH]
n, k = int(os.environ.get("N", 10_000_000)), int(os.environ.get("K", 100))
reps, rounds = int(os.environ.get("REPS", 60)), int(os.environ.get("ROUNDS", 3))
dev = torch.device("cuda", 0)
lp = tp.gen_lp(n, n, k, seed=0, device=dev)
K = tp.CsrPair(lp.m, lp.n, lp.rowptr, lp.colidx, lp.val)


def load(path):
    N._lib = None
    N.LIB_PATH = os.path.abspath(path)
    return N.load()


def measure(eng, adaptive=1):
    lib, h, stream = eng.lib, eng.h, eng.stream
    ev = lambda: torch.cuda.Event(enable_timing=True)
    evs = [(ev(), ev(), ev()) for _ in range(reps)]
    for _ in range(3):
        N.check(lib.pdlp_primal_half(h, adaptive)); N.check(lib.pdlp_dual_half(h, adaptive))
    stream.synchronize()
    for e0, e1, e2 in evs:
        e0.record(stream); N.check(lib.pdlp_primal_half(h, adaptive)); e1.record(stream); N.check(lib.pdlp_dual_half(h, adaptive)); e2.record(stream)
    stream.synchronize()
    p = sorted(e0.elapsed_time(e1) for e0, e1, _ in evs)
    d = sorted(e1.elapsed_time(e2) for _, e1, e2 in evs)
    return p[len(p) // 2], d[len(d) // 2], sum(p) / len(p), sum(d) / len(d)


engines = []
for spec in libs:                      # "lib.so" or "lib.so:ENV=VALUE" (the variable is set while that engine is built)
    path, _, env = spec.partition(":")
    envs = [e.split("=") for e in env.split(",")] if env else []      # "lib.so:A=1,B=2"
    for kk, vv in envs:
        os.environ[kk] = vv
    load(path)
    eng = tp.PdlpEngine.from_full(K, lp.c, lp.q, lp.l, lp.u, lp.m_ineq)
    eng.set_step(0.01, 1.0, 1.0, 0)
    eng.iterate(2, True)
    engines.append((spec, eng))
    for kk, _ in envs:
        os.environ.pop(kk, None)
    print(spec, eng.kernels, flush=True)
    for tr in (0, 1):
        t = eng.tiles[tr]
        if t is not None:
            a, b, c = t.idx.data_ptr(), t.val.data_ptr(), t.cnt.data_ptr()
            print(f"   tiles[{tr}] idx {a:#x} val {b:#x} cnt {c:#x}  val-idx {(b - a) / 2**20:.3f} MiB  mod 2MiB: {a % 2**21:#x} {b % 2**21:#x} {c % 2**21:#x}", flush=True)
for rnd in range(rounds):
    for path, eng in engines:
        pm, dm, pa, da = measure(eng)
        print(f"round {rnd} {os.path.basename(path):44s} primal median {pm:.4f} mean {pa:.4f} | dual median {dm:.4f} mean {da:.4f} ms", flush=True)
