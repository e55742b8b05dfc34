// ipc_probe.hip -- round 5: can two PROCESSES exchange a vector block by storing straight into each other's device memory (HIP IPC),
// with flags in a fine-grained mailbox, bounded-spin wait kernels and no collective?  What the direct exchange of pdlp_peer_* relies on:
//   1. hipIpcGetMemHandle on an INTERIOR pointer of an allocation (a torch workspace is carved by torch's allocator) opens at the same
//      interior address in the other process;
//   2. a fine-grained allocation (hipExtMallocWithFlags, hipDeviceMallocFinegrained) can be exported too;
//   3. data stored with system-scope stores by one kernel + a flag stored by the NEXT kernel of the same stream is seen complete by a
//      kernel the peer launches after its wait kernel has seen the flag;
//   4. the round trip signal -> wait -> signal -> wait between the two processes (the latency an exchange adds to a half-step).
// Two processes (fork before any HIP call), both on device 0 -- the only multi-process layout a one-GPU box offers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/src/ipc_probe.hip -o tools/_bin/ipc_probe && tools/_bin/ipc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "[%d] %s:%d %s -> %s\n", g_rank, __FILE__, __LINE__, #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
static int g_rank = -1;

__global__ void k_fill_remote(float* remote, float* local, size_t n, float v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float w = v + (float)(i & 1023);
        local[i] = w;
        __hip_atomic_store(remote + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void k_signal(unsigned* remote_flag, unsigned seq)
{
    __hip_atomic_store(remote_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bounded spin: gives up after `limit` ticks of the 100 MHz constant clock and reports through *err
__global__ void k_wait(const unsigned* flag, unsigned seq, long long limit, int* err)
{
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
        if (wall_clock64() - t0 > limit) { *err = 1; return; }
        __builtin_amdgcn_s_sleep(8);
    }
}

__global__ void k_check(const float* buf, size_t n, float v, unsigned long long* bad)
{
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b += buf[i] != v + (float)(i & 1023);
    if (b) atomicAdd(bad, b);
}

struct Msg { hipIpcMemHandle_t big, box; unsigned long long interior, offset; };

int main(int argc, char** argv)
{
    // argv[1]: size of the exported allocation in MB (default 64) -- the block that is written stays 16 MB at its start
    int ab[2], ba[2];
    if (pipe(ab) || pipe(ba)) return 1;
    const pid_t pid = fork();
    g_rank = pid == 0 ? 1 : 0;
    const int rd = g_rank == 0 ? ba[0] : ab[0], wr = g_rank == 0 ? ab[1] : ba[1];
    CK(hipSetDevice(0));
    const size_t N = 4u << 20, OFF = (1u << 20) + 256;            // a 16 MB block at an odd interior offset of a 64 MB allocation
    char* big = nullptr;
    const size_t big_bytes = (size_t)(argc > 1 ? std::atol(argv[1]) : 64) << 20;
    CK(hipMalloc(&big, big_bytes));
    float* mine = (float*)(big + OFF);
    unsigned* box = nullptr;
    CK(hipExtMallocWithFlags((void**)&box, 4096, hipDeviceMallocFinegrained));
    CK(hipMemset(box, 0, 4096));
    CK(hipMemset(big, 0, 64u << 20));
    const auto t_open0 = std::chrono::steady_clock::now();
    CK(hipDeviceSynchronize());
    Msg out{}, in{};
    // (first try: hipIpcGetMemHandle(mine) -- the peer's hipIpcOpenMemHandle returns the BASE of the allocation, the interior offset is
    //  lost: every element landed 0x100100 bytes too low.  So: the allocation's base from hipMemGetAddressRange, the offset beside it)
    hipDeviceptr_t base = nullptr;
    size_t range = 0;
    CK(hipMemGetAddressRange(&base, &range, (hipDeviceptr_t)mine));
    CK(hipIpcGetMemHandle(&out.big, (void*)base));
    CK(hipIpcGetMemHandle(&out.box, box));
    out.interior = (unsigned long long)mine;
    out.offset = (unsigned long long)((char*)mine - (char*)base);
    std::printf("[%d] allocation base %p (+%#llx), %zu bytes\n", g_rank, (void*)base, out.offset, range);
    if (write(wr, &out, sizeof out) != (ssize_t)sizeof out || read(rd, &in, sizeof in) != (ssize_t)sizeof in) return 3;
    float* theirs = nullptr;
    unsigned* their_box = nullptr;
    char* their_base = nullptr;
    CK(hipIpcOpenMemHandle((void**)&their_base, in.big, hipIpcMemLazyEnablePeerAccess));
    theirs = (float*)(their_base + in.offset);
    CK(hipIpcOpenMemHandle((void**)&their_box, in.box, hipIpcMemLazyEnablePeerAccess));
    std::printf("[%d] own block %p, peer's block opened at %p (peer's own address %#llx); export + open of a %zu MB allocation: %.3f s\n", g_rank,
                (void*)mine, (void*)theirs, in.interior, big_bytes >> 20,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open0).count());
    std::fflush(stdout);
    int* err = nullptr;
    CK(hipHostMalloc((void**)&err, 64, hipHostMallocMapped));
    *err = 0;
    unsigned long long* bad = nullptr;
    CK(hipMalloc(&bad, 8));
    CK(hipMemset(bad, 0, 8));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const long long LIMIT = 300000000LL;                          // 3 s of the 100 MHz clock
    // does the opened pointer address the INTERIOR block?  each rank writes a pattern into the other's block, half a block further in,
    // and the owner checks that half at its own address
    unsigned seq = 0;
    for (int round = 0; round < 3; ++round) {
        const float v = 1000.f * (round + 1) + 17.f * (1 - g_rank);          // what the peer expects from me
        const float vme = 1000.f * (round + 1) + 17.f * g_rank;              // what I expect from the peer
        ++seq;
        hipLaunchKernelGGL(k_fill_remote, dim3(1024), dim3(256), 0, s, theirs + N / 2, mine, N / 2, v);
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s, their_box + g_rank, seq);
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, s, (const unsigned*)box + (1 - g_rank), seq, LIMIT, err);
        hipLaunchKernelGGL(k_check, dim3(1024), dim3(256), 0, s, (const float*)mine + N / 2, N / 2, vme, bad);
        // (the peer may only overwrite my block again once I have checked it: a second handshake)
        ++seq;
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s, their_box + g_rank, seq);
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, s, (const unsigned*)box + (1 - g_rank), seq, LIMIT, err);
    }
    CK(hipStreamSynchronize(s));
    unsigned long long nbad = 0;
    CK(hipMemcpy(&nbad, bad, 8, hipMemcpyDeviceToHost));
    std::printf("[%d] 3 rounds of 8 MB stored into the peer + flag: %llu wrong elements, wait timed out: %d\n", g_rank, nbad, *err);
    // latency: ping-pong of signal / wait pairs, all enqueued up front (as an iteration block would be)
    if (!*err) {
        const int R = 2000;
        CK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < R; ++i) {
            ++seq;
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s, their_box + g_rank, seq);
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, s, (const unsigned*)box + (1 - g_rank), seq, LIMIT, err);
        }
        CK(hipStreamSynchronize(s));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
        std::printf("[%d] signal + wait per exchange, both processes in lock step: %.2f us (timed out: %d)\n", g_rank, us, *err);
        // the same two launches without a peer (flag already there): the launches' own cost
        const auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < R; ++i) {
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s, box + 8, seq);
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, s, (const unsigned*)box + 8, seq, LIMIT, err);
        }
        CK(hipStreamSynchronize(s));
        const double us1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / R;
        std::printf("[%d] the two launches alone (own flag): %.2f us\n", g_rank, us1);
    }
    // both sides done with the peer's memory before anyone frees
    char c = 1;
    if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) return 4;
    CK(hipIpcCloseMemHandle(their_base));
    CK(hipIpcCloseMemHandle(their_box));
    if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) return 4;
    CK(hipFree(big));
    CK(hipFree(box));
    const int rc = (nbad || *err) ? 5 : 0;
    if (g_rank == 0) {
        int st = 0;
        waitpid(pid, &st, 0);
        return rc ? rc : (WIFEXITED(st) ? WEXITSTATUS(st) : 6);
    }
    return rc;
}
