#!/usr/bin/env python3
"""One rank's shard of the bench problem on ONE GPU, WHOLE iterations: both matrices shard-shaped (K rows block and K' rows block of
rank 0 of WORLD ranks, uniform random, the bench density), the real ``PdlpEngine.iterate`` loop, and stand-ins for the collectives
with torch.distributed's stream semantics -- a collective runs on a communication stream that first waits for everything the
caller's stream has enqueued so far, takes ``AG_MS`` (an all-gather of a whole vector; a piece takes its share) or ``AR_MS`` (the
3-double all-reduce) of spinning, and the caller's stream waits for it where the real loop would (``wait()`` of an asynchronous
piece, at once for a blocking collective).  What it measures: the critical path of an iteration on one rank, i.e. how much of the
exchange the products hide -- the source of DESIGN.md section 5's ESTIMATE.  NOT a scaling measurement: no second GPU, no RCCL kernels
competing for CUs and HBM.
env: WORLD (8), AG_MS (0.1) or AG_LIST ("0.1,0.2"), AR_MS (0.02), CHUNKS ("1,2"), N (10M), K (100), ITERS (40), ADAPTIVE (1),
DTYPE (f32 | mixed), PDLP_PRODUCER_PIECES (0: a half-step finishes before its block is exchanged, the round-4 behaviour),
PEER (1: also the direct exchange in loopback -- pdlp_peer_connect(PDLP_PEER_LOOPBACK): the epilogues store the block into WORLD - 1
scratch blocks of this GPU instead of into peers over xGMI, the flags land in the own mailbox so the waits pass at once: what is
priced is the stores, the two one-wave launches per exchange and the single-stream structure -- not the skew between real ranks)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PDLP_TILED"] = "1"
import torch
import torchpdlp_amd as tp
from torchpdlp_amd import _native as N_

W = int(os.environ.get("WORLD", 8)); ag_ms = float(os.environ.get("AG_MS", 0.1)); ar_ms = float(os.environ.get("AR_MS", 0.02))
n = int(os.environ.get("N", 10_000_000)); k = int(os.environ.get("K", 100))
iters = int(os.environ.get("ITERS", 40)); adaptive = os.environ.get("ADAPTIVE", "1") != "0"
mixed = os.environ.get("DTYPE", "f32") == "mixed"
blk = n // W
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)


def shard():
    col = torch.empty(blk * k, dtype=torch.int32, device=dev)
    ch = 1 << 20
    for r0 in range(0, blk, ch):
        r1 = min(blk, r0 + ch)
        b, _ = torch.sort(torch.randint(0, n, (r1 - r0, k), generator=g, device=dev, dtype=torch.int32), dim=1)
        col[r0 * k:r1 * k] = b.reshape(-1)
    val = torch.rand(blk * k, device=dev, generator=g)
    rp = torch.arange(0, (blk + 1) * k, k, dtype=torch.int64, device=dev)
    return rp, col, val


torch.cuda._sleep(1000); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); torch.cuda._sleep(10_000_000); b.record(); torch.cuda.synchronize()
cyc_per_ms = 10_000_000 / a.elapsed_time(b)


class Work:
    def __init__(self, ev): self.ev = ev
    def wait(self): torch.cuda.current_stream().wait_event(self.ev)


class FakeComm:
    """rank 0 of W ranks; every collective = a spin kernel on the communication stream (torch.distributed's stream semantics)"""
    rank, world, backend, group = 0, W, "fake", None

    def __init__(self):
        self.cs = torch.cuda.Stream()

    def _run(self, ms):
        cur = torch.cuda.current_stream()
        e = torch.cuda.Event(); e.record(cur)
        self.cs.wait_event(e)                                  # the collective sees what the caller has enqueued so far
        with torch.cuda.stream(self.cs):
            torch.cuda._sleep(max(1, int(ms * cyc_per_ms)))
            done = torch.cuda.Event(); done.record(self.cs)
        return Work(done)

    def all_gather(self, full): self._run(ag_ms).wait()
    def all_gather_async(self, full): return self._run(ag_ms)
    def all_gather_piece(self, full, lo, hi):
        B = full.numel() // W
        return self._run(ag_ms * (hi - lo) / B) if hi > lo else None
    def all_reduce_sum(self, t, op=None): self._run(ar_ms).wait()
    def all_reduce_sum_async(self, t): return self._run(ar_ms)
    def all_reduce_max(self, t): self._run(ar_ms).wait()
    def all_reduce_min(self, t): self._run(ar_ms).wait()


z = lambda ln: torch.zeros(ln, device=dev)
comm = FakeComm()
kw = dict(vec_dtype=torch.float64) if mixed else {}
eng = tp.PdlpEngine(n, n, blk // 2, shard(), shard(), torch.randn(blk, device=dev, generator=g), torch.randn(blk, device=dev, generator=g),
                    z(blk), torch.full((blk,), 5.0, device=dev), rows=(0, blk), cols=(0, blk), comm=comm, **kw)
print(f"WORLD={W} shard {blk}x{n} x2, {k} per row, {'mixed/delta' if mixed else 'f32'}, {'adaptive' if adaptive else 'fixed'}; "
      f"tiles (rpt, groups, row blocks): {[t is not None and (t.rpt, t.groups, t.nblk) for t in eng.tiles]}", flush=True)
eng.set_iterate(torch.rand(blk, device=dev, generator=g).to(eng.dtype), torch.rand(blk, device=dev, generator=g).to(eng.dtype))
eng.set_step(1e-3, 1.0, 1.0, 0)


enqueue_ms = [0.0]


def timed(n_it):
    eng.iterate(4, adaptive)
    torch.cuda.synchronize()
    t0 = time.time()
    eng.iterate(n_it, adaptive)
    t1 = time.time()
    torch.cuda.synchronize()
    enqueue_ms[0] = (t1 - t0) / n_it * 1e3          # host time to ISSUE an iteration: if it is close to the total, the host is the limit
    return (time.time() - t0) / n_it * 1e3


def products_alone():
    """the two fused half-steps without any exchange: what no overlap can go below"""
    lib, h = eng.lib, eng.h
    a_ = int(adaptive)
    for _ in range(3):
        N_.check(lib.pdlp_primal_half(h, a_)); N_.check(lib.pdlp_dual_half(h, a_))
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        N_.check(lib.pdlp_primal_half(h, a_)); N_.check(lib.pdlp_dual_half(h, a_))
    torch.cuda.synchronize()
    return (time.time() - t0) / iters * 1e3


print(f"   products alone (2 half-steps, no exchange, unsplit): {products_alone():.3f} ms per iteration", flush=True)
if os.environ.get("PEER", "0") != "0":
    # PEER_HOST=1: one of the stand-in peers lives in pinned host memory -- its 1/WORLD block crosses PCIe (~55 GB/s), about the time
    # the WORLD - 1 blocks of a real exchange take over xGMI together: shows how much of a slow drain of the stores a schedule hides
    slow = os.environ.get("PEER_HOST", "0") != "0"
    N_.check(eng.lib.pdlp_peer_connect(eng.h, 0, W, None, N_.PEER_LOOPBACK | (N_.PEER_LOOPBACK_HOST if slow else 0)), "pdlp_peer_connect")
    where = f"{W - 2} scratch blocks + 1 block in pinned host memory" if slow else f"{W - 1} scratch blocks"
    # (results of the split form in 2 / 4 row-block pieces, each with its own epilogue launch on the same stream, were tried here:
    #  0.796 / 1.120 ms per iteration against 0.668 in one piece -- in-order launches hide nothing of a store drain;
    #  profiles/r05_multi_gpu/direct_exchange_slow_link_stand_in.log)
    variants = [(True, 0, 1), (True, 1, 1), (True, 2, 1), (False, 0, 1)]
    for on, form, pieces in variants:
        eng.set_peer_exchange(on)
        eng.set_peer_form(form)
        eng.set_exchange_chunks(pieces)
        if not on:
            ag_ms = float(os.environ.get("AG_LIST", str(ag_ms)).split(",")[0])
        best = min(timed(iters) for _ in range(3))
        what = f"direct exchange, loopback (stores into {where}, signal + wait kernels; {eng.PEER_FORMS[form]})" if on else \
               f"the same handle on the loop, all-gather {ag_ms:.2f} ms, 1 piece"
        print(f"   {what}: {best:.3f} ms per iteration = {best / 2:.3f} per half-step -> {1e3 / best:.0f} it/s   "
              f"(host issue time {enqueue_ms[0]:.3f} ms per iteration)", flush=True)
    assert eng.peer_status()["gave_up_on"] is None
    N_.check(eng.lib.pdlp_peer_close(eng.h), "pdlp_peer_close")
    eng.peer_on = False
for ags in [float(v) for v in os.environ.get("AG_LIST", str(ag_ms)).split(",")]:
    ag_ms = ags
    for ch in [int(v) for v in os.environ.get("CHUNKS", "1,2").split(",") if v]:
        eng.set_exchange_chunks(ch)
        info = [eng.split_info(t) for t in (0, 1)]
        best = min(timed(iters) for _ in range(3))
        print(f"   all-gather {ag_ms:.2f} ms, all-reduce {ar_ms:.2f} ms, {ch} piece(s) (slots K {info[0]['local_groups']}+{info[0]['other_groups']}, "
              f"K' {info[1]['local_groups']}+{info[1]['other_groups']}): {best:.3f} ms per iteration = {best / 2:.3f} per half-step "
              f"-> {1e3 / best:.0f} it/s   (host issue time {enqueue_ms[0]:.3f} ms per iteration)", flush=True)
