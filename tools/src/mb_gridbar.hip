// mb_gridbar.hip -- what does a grid-wide barrier cost on MI355X (8 XCDs, L2s not coherent with each other)?
// Decides whether a persistent cooperative PDHG kernel (primal rows, barrier, dual rows, barrier, step-size rule, barrier) can beat
// three dependent launches per iteration on small LPs.
//   flat:          every workgroup adds to one counter (agent-scope release), the last arrival bumps a generation word, all spin on it
//   hierarchical:  groups of G workgroups share a counter; the last arrival of a group adds to the master counter
// Between barriers every workgroup writes one word and, after the barrier, reads its neighbour's (checks visibility across XCDs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/src/mb_gridbar.hip -o tools/_bin/mb_gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Bar {
    unsigned* master;      // [0] count, [32] generation (separate lines)
    unsigned* group;       // one counter per group, 32 words apart
    int ngroups, gsize;
};

__device__ __forceinline__ void grid_barrier(const Bar& b, unsigned& gen, int G)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned want = gen + 1;
        bool last;
        if (G <= 1) {
            last = __hip_atomic_fetch_add(b.master, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)gridDim.x * want - 1;
        } else {
            const int g = blockIdx.x / G;
            const int members = (g == b.ngroups - 1) ? (int)gridDim.x - g * G : G;
            last = false;
            if (__hip_atomic_fetch_add(b.group + 32 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)members * want - 1)
                last = __hip_atomic_fetch_add(b.master, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)b.ngroups * want - 1;
        }
        if (last) __hip_atomic_store(b.master + 32, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            while (__hip_atomic_load(b.master + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        gen = want;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_bars(Bar b, int G, int nbar, unsigned* data, unsigned* errors, int work)
{
    unsigned gen = 0;       // (meaningful in thread 0)
    unsigned bad = 0;
    for (int i = 0; i < nbar; ++i) {
        if (threadIdx.x < work) data[(size_t)blockIdx.x * 64 + threadIdx.x] = (unsigned)(i * 1000003 + blockIdx.x + threadIdx.x);
        grid_barrier(b, gen, G);
        if (threadIdx.x < work) {
            const unsigned nb = (blockIdx.x + 1 + (i % 7)) % gridDim.x;
            bad += data[(size_t)nb * 64 + threadIdx.x] != (unsigned)(i * 1000003 + nb + threadIdx.x);
        }
        grid_barrier(b, gen, G);
    }
    if (bad) atomicAdd(errors, bad);
}

int main()
{
    int dev = 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    int perCU = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_bars, 256, 0));
    printf("CUs %d, co-resident workgroups of 256 threads per CU: %d, cooperative launch supported: %d\n", prop.multiProcessorCount, perCU, prop.cooperativeLaunch);
    unsigned *mem, *data, *errors;
    CHECK(hipMalloc(&mem, 1 << 20));
    CHECK(hipMalloc(&data, (size_t)4096 * 64 * 4));
    CHECK(hipMalloc(&errors, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int nbar = 500;
    for (int P : {256, 512, 1024, 2048})
        for (int G : {1, 8, 16, 32}) {
            if (P > perCU * prop.multiProcessorCount) continue;
            CHECK(hipMemset(mem, 0, 1 << 20));
            CHECK(hipMemset(errors, 0, 4));
            Bar b{mem, mem + 1024, G > 1 ? (P + G - 1) / G : 1, G};
            int work = 64;
            void* args[] = {&b, &G, (void*)&nbar, &data, &errors, &work};
            CHECK(hipEventRecord(e0));
            CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_bars), dim3(P), dim3(256), args, 0, 0));
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned he;
            CHECK(hipMemcpy(&he, errors, 4, hipMemcpyDeviceToHost));
            printf("workgroups %4d  group size %2d: %.2f us per barrier (%d barriers, incl. a 256-byte write + neighbour read per pair)  stale reads %u\n", P, G,
                   ms * 1e3 / (2 * nbar), 2 * nbar, he);
            fflush(stdout);
        }
    // for comparison: the launch gap of an (almost) empty kernel, 1000 dependent launches
    {
        CHECK(hipMemset(mem, 0, 1 << 20));
        Bar b{mem, mem + 1024, 1, 1};
        int G = 1, zero = 0, work = 64;
        for (int P : {256, 2048}) {
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_bars, dim3(P), dim3(256), 0, 0, b, G, zero, data, errors, work);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("empty kernel, %4d workgroups, 1000 dependent launches: %.2f us per launch\n", P, ms);
        }
    }
    return 0;
}
