#!/usr/bin/env python3
"""One rank's shard of the bench problem on ONE GPU: how long a dual half-step takes when its product is split into
"local panels first (side stream), the others after the all-gather" versus unsplit, with a spin kernel of the
all-gather's duration standing in for the collective.  Used to choose the panel-group counts of the split.
env: WORLD (8), AG_MS (0.1), SLOTS ("a,b" overrides the library's choice: pdlp_set_option(PDLP_OPT_SPLIT_SLOTS)), N, K,
CHUNKS ("1,2,4": also time the chunked exchange -- the stand-in then runs as that many spin kernels on a second stream, each a
1/chunks of AG_MS, and the panels a piece completes are launched when its spin kernel is done)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PDLP_TILED"] = "1"
import torch
import torchpdlp_amd as tp
from torchpdlp_amd import _native as N_

W = int(os.environ.get("WORLD", 8)); ag_ms = float(os.environ.get("AG_MS", 0.1))
n = int(os.environ.get("N", 10_000_000)); k = int(os.environ.get("K", 100))
m = n // W
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
col = torch.empty(m * k, dtype=torch.int32, device=dev)
ch = 1 << 20
for r0 in range(0, m, ch):
    r1 = min(m, r0 + ch)
    blk, _ = torch.sort(torch.randint(0, n, (r1 - r0, k), generator=g, device=dev, dtype=torch.int32), dim=1)
    col[r0 * k:r1 * k] = blk.reshape(-1)
val = torch.rand(m * k, device=dev, generator=g)
rp = torch.arange(0, (m + 1) * k, k, dtype=torch.int64, device=dev).to(torch.int32)


class FakeComm:                      # rank 0 of W ranks; the collectives are replaced by a spin kernel below
    rank, world, backend, group = 0, W, "fake", None
    def all_gather(self, full): pass
    def all_reduce_sum(self, t): pass


zn, zm = torch.zeros(n // W, device=dev), torch.zeros(m, device=dev)
nk = n // W            # K' block: one entry per row, only there to make a valid engine (the dual half-step does not touch it)
empty_kt = (torch.arange(0, nk + 1, dtype=torch.int32, device=dev), torch.randint(0, n, (nk,), generator=g, device=dev, dtype=torch.int32),
            torch.rand(nk, device=dev, generator=g))
eng = tp.PdlpEngine(n, n, 0, (rp, col, val), empty_kt, zn, zm, zn, zn, rows=(0, m), cols=(0, n // W), comm=FakeComm())
if os.environ.get("SLOTS"):
    sa, sb = (int(v) for v in os.environ["SLOTS"].split(","))
    eng.set_option(N_.OPT_SPLIT_SLOTS, sa | (sb << 16))
print("tiles", [t is not None and (t.rpt, t.groups, t.nblk) for t in eng.tiles], "split", eng.split_info(0), flush=True)
eng.set_step(0.01, 1.0, 1.0, 0)
lib, h = eng.lib, eng.h
# calibrate the spin kernel
torch.cuda._sleep(1000); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); torch.cuda._sleep(10_000_000); b.record(); torch.cuda.synchronize()
cyc_per_ms = 10_000_000 / a.elapsed_time(b)
spin = int(ag_ms * cyc_per_ms)


def run(split, reps=20):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if split:
            N_.check(lib.pdlp_dual_half_begin(h, 0))
        torch.cuda._sleep(spin)                      # the all-gather of xbar
        N_.check(lib.pdlp_dual_half(h, 0))
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def run_chunked(chunks, reps=20):
    """the exchange as `chunks` spin kernels on a communication stream; products of the completed pieces on the main stream"""
    N_.check(lib.pdlp_set_exchange_chunks(h, chunks))
    main, comm = torch.cuda.current_stream(), torch.cuda.Stream()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ready = torch.cuda.Event(); ready.record(main)
        N_.check(lib.pdlp_dual_half_begin(h, 0))
        evs = []
        with torch.cuda.stream(comm):
            comm.wait_event(ready)
            for c in range(chunks):
                torch.cuda._sleep(spin // chunks)
                e = torch.cuda.Event(); e.record(comm); evs.append(e)
        for c in range(chunks - 1):
            main.wait_event(evs[c])
            N_.check(lib.pdlp_half_chunk(h, 0, c))
        main.wait_event(evs[-1])
        N_.check(lib.pdlp_dual_half(h, 0))
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    N_.check(lib.pdlp_set_exchange_chunks(h, 1))
    return ts[0], ts[len(ts) // 2]


run(False, 3); run(True, 3)
u, s = run(False), run(True)
print(f"WORLD={W} shard {m}x{n}  all-gather stand-in {ag_ms} ms:  unsplit min {u[0]:.3f} med {u[1]:.3f} ms | split min {s[0]:.3f} med {s[1]:.3f} ms", flush=True)
for ch in [int(v) for v in os.environ.get("CHUNKS", "").split(",") if v]:
    if ch > 1:
        run_chunked(ch, 3)
        c = run_chunked(ch)
        N_.check(lib.pdlp_set_exchange_chunks(h, ch)); info = eng.split_info(0); N_.check(lib.pdlp_set_exchange_chunks(h, 1))
        print(f"   exchange in {ch} pieces (slots {info['local_groups']} + {info['other_groups']}): min {c[0]:.3f} med {c[1]:.3f} ms", flush=True)
