// Read-bandwidth probe (tools only): how fast can 512-thread workgroups stream HBM with 16-byte non-temporal loads,
// (a) every workgroup reading its own contiguous slice (the tiled kernel's pattern: ~1000 separate sequential streams),
// (b) grid-stride (all workgroups advance through the array together).  hipcc --offload-arch=gfx950 -O3 bw_probe.hip -o bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int UNROLL>
__global__ __launch_bounds__(512, 4) void k_read(const u32x4* __restrict__ a, size_t n16, unsigned* out, int nstreams)
{
    const size_t per = n16 / gridDim.x;
    unsigned acc = 0;
    if (MODE == 0) {                       // own slice, optionally split in nstreams sub-streams read in lockstep
        const size_t sub = per / nstreams;
        const u32x4* p = a + (size_t)blockIdx.x * per;
        for (size_t i = threadIdx.x; i + (UNROLL - 1) * 512 < sub; i += 512 * UNROLL) {
            for (int s = 0; s < nstreams; ++s) {
                u32x4 v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + s * sub + i + u * 512);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
            }
        }
    } else {                               // grid-stride
        for (size_t i = (size_t)blockIdx.x * 512 * UNROLL + threadIdx.x; i + (UNROLL - 1) * 512 < n16; i += (size_t)gridDim.x * 512 * UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(a + i + u * 512);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int MODE, int UNROLL> void run(const char* name, const u32x4* a, size_t n16, unsigned* out, int grid, int nstreams)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k_read<MODE, UNROLL><<<grid, 512>>>(a, n16, out, nstreams);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) k_read<MODE, UNROLL><<<grid, 512>>>(a, n16, out, nstreams);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-44s grid %5d unroll %d streams %d: %.3f ms  %.2f TB/s\n", name, grid, UNROLL, nstreams, ms, n16 * 16.0 / ms / 1e9);
    fflush(stdout);
}
int main(int argc, char** argv)
{
    const size_t bytes = (argc > 1 ? atof(argv[1]) : 8.0) * (1ull << 30);
    const size_t n16 = bytes / 16;
    u32x4* a; unsigned* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, bytes);
    hipDeviceSynchronize();
    run<1, 4>("grid-stride", a, n16, out, 2048, 1);
    run<1, 4>("grid-stride", a, n16, out, 1024, 1);
    run<1, 8>("grid-stride", a, n16, out, 1024, 1);
    run<0, 4>("own slice", a, n16, out, 512, 1);
    run<0, 4>("own slice", a, n16, out, 1024, 1);
    run<0, 2>("own slice", a, n16, out, 1024, 1);
    run<0, 4>("own slice, 2 sub-streams (idx + val)", a, n16, out, 512, 2);
    run<0, 4>("own slice, 2 sub-streams", a, n16, out, 1024, 2);
    run<0, 2>("own slice, 2 sub-streams", a, n16, out, 1024, 2);
    run<0, 8>("own slice", a, n16, out, 512, 1);
    return 0;
}
