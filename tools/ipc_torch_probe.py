#!/usr/bin/env python3
"""Does hipIpcOpenMemHandle open a block that torch's caching allocator handed out?  Two processes on device 0; each exports the
allocation (hipMemGetAddressRange -> base) that holds a `MB`-sized uint8 tensor and opens the other's, one after the other.
Cases: fresh (the tensor is its own allocation), sub (a sub-block of a larger cached segment, the rest free), sub_used (the rest in use),
pool (allocated in a private torch.cuda.MemPool after the cache was filled).   python tools/ipc_torch_probe.py [MB=1700] [case ...]
Every open runs under a watchdog thread that reports a hang after 20 s and ends the process (a hung open never returns)."""
import ctypes as C, os, sys, threading, time
import torch
import torch.multiprocessing as mp

MB = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
CASES = sys.argv[2:] or ["fresh", "sub", "sub_used", "pool"]


def worker(rank, q_out, q_in, case):
    hip = C.CDLL("libamdhip64.so")
    torch.cuda.set_device(0)
    keep = []
    if case in ("sub", "sub_used", "pool"):
        big = torch.empty((MB * 9 // 4) << 20, dtype=torch.uint8, device="cuda")       # a 2.25x segment, back into the cache
        del big
    if case == "pool":
        pool = torch.cuda.MemPool()
        with torch.cuda.use_mem_pool(pool):
            t = torch.empty(MB << 20, dtype=torch.uint8, device="cuda")
        keep.append(pool)
    else:
        t = torch.empty(MB << 20, dtype=torch.uint8, device="cuda")
    if case == "sub_used":
        keep.append(torch.empty((MB // 2) << 20, dtype=torch.uint8, device="cuda"))  # the rest of the segment is in use
    t.fill_(rank + 1)
    torch.cuda.synchronize()
    base, size = C.c_void_p(), C.c_size_t()
    assert hip.hipMemGetAddressRange(C.byref(base), C.byref(size), C.c_void_p(t.data_ptr())) == 0
    handle = (C.c_char * 64)()
    rc = hip.hipIpcGetMemHandle(handle, base)
    off = t.data_ptr() - base.value
    print(f"[{case} rank {rank}] tensor {MB} MB at +{off >> 20} MB of an allocation of {size.value >> 20} MB; export rc {rc}", flush=True)
    q_out.put((bytes(handle.raw), off))
    peer_handle, peer_off = q_in.get()
    if rank == 1:
        time.sleep(1.0)                                                              # (one after the other)
    done = threading.Event()

    def watchdog():
        if not done.wait(20.0):
            print(f"[{case} rank {rank}] hipIpcOpenMemHandle HUNG (20 s)", flush=True)
            os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    ptr = C.c_void_p()
    t0 = time.time()
    class Handle(C.Structure):                  # (hipIpcMemHandle_t travels BY VALUE)
        _fields_ = [("reserved", C.c_char * 64)]
    hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
    rc = hip.hipIpcOpenMemHandle(C.byref(ptr), Handle.from_buffer_copy(peer_handle), 1)
    done.set()
    print(f"[{case} rank {rank}] open rc {rc} in {time.time() - t0:.3f} s", flush=True)
    if rc == 0:
        got = (C.c_ubyte * 1)()
        hip.hipMemcpy(got, C.c_void_p(ptr.value + peer_off + 12345), 1, 2)
        print(f"[{case} rank {rank}] a byte of the peer's tensor: {got[0]} (expected {2 - rank})", flush=True)
    q_out.put("done")
    q_in.get()
    if rc == 0:
        hip.hipIpcCloseMemHandle(ptr)


if __name__ == "__main__":
    mp.set_start_method("spawn")
    for case in CASES:
        a, b = mp.Queue(), mp.Queue()
        ps = [mp.Process(target=worker, args=(0, a, b, case)), mp.Process(target=worker, args=(1, b, a, case))]
        for p in ps:
            p.start()
        for p in ps:
            p.join(90)
        print(f"== {case}: exit codes {[p.exitcode for p in ps]}", flush=True)
        for p in ps:
            if p.is_alive():
                p.kill()
