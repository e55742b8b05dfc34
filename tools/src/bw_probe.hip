// Read-bandwidth and stream+gather probe (tools only): how fast can 512-thread workgroups stream HBM with 16-byte non-temporal loads,
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
// (c) stream + gather: every item = (index word, value word); the index selects a column of a 256 KB table (L2 resident) with the
// tiled kernel's pattern -- inside a 256-item group, component j of lane L is sorted item 64j + L, neighbours SPACING columns apart
// -- and the product is accumulated in a register.  No LDS, no reduction: the ceiling of "8 bytes streamed + one gather per item".
__global__ void k_fill_idx(unsigned* idx, size_t n, int spacing)
{
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const size_t grp = p >> 8;
        const unsigned L = (unsigned)(p & 255) >> 2, j = (unsigned)p & 3;
        idx[p] = (unsigned)((grp * 256 + 64 * j + L) * (size_t)spacing) & 0xFFFFu;
    }
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, int ROUND>
__global__ __launch_bounds__(512, 4) void k_stream_gather(const u32x4* __restrict__ idx, const f32x4* __restrict__ val,
                                                           const float* __restrict__ table, size_t n16, float* out)
{
    const size_t per = n16 / gridDim.x;
    const u32x4* pi = idx + (size_t)blockIdx.x * per;
    const f32x4* pv = val + (size_t)blockIdx.x * per;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * 512 < per; i += 512 * UNROLL) {
        u32x4 k[UNROLL];
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { k[u] = __builtin_nontemporal_load(pi + i + u * 512); v[u] = __builtin_nontemporal_load(pv + i + u * 512); }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const float g0 = table[k[u].x], g1 = table[k[u].y], g2 = table[k[u].z], g3 = table[k[u].w];
            if (ROUND) __builtin_amdgcn_s_waitcnt(0x0F70);        // rounds of four gathers per lane (the tiled kernel's schedule)
            acc += v[u].x * g0 + v[u].y * g1 + v[u].z * g2 + v[u].w * g3;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
template <int UNROLL, int ROUND> void run_sg(const char* name, const u32x4* idx, const f32x4* val, const float* table, size_t n16, float* out, int grid)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k_stream_gather<UNROLL, ROUND><<<grid, 512>>>(idx, val, table, n16, out);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) k_stream_gather<UNROLL, ROUND><<<grid, 512>>>(idx, val, table, n16, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-44s grid %5d unroll %d rounds %d: %.3f ms  %.1f G items/s  stream %.2f TB/s\n", name, grid, UNROLL, ROUND, ms,
           n16 * 4.0 / ms / 1e6, n16 * 32.0 / ms / 1e9);
    fflush(stdout);
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
    {   // stream + gather: half of the buffer as index words, half as values (0x01010101 bit patterns: tiny floats)
        const size_t items = bytes / 8, h16 = items / 4;
        float* table; hipMalloc(&table, 65536 * 4); hipMemset(table, 0, 65536 * 4);
        unsigned* idx = (unsigned*)a;
        const f32x4* val = (const f32x4*)((char*)a + items * 4);
        for (int spacing : {5, 3, 2, 1, 40}) {
            k_fill_idx<<<4096, 256>>>(idx, items, spacing);
            hipDeviceSynchronize();
            char nm[64]; snprintf(nm, sizeof nm, "stream + gather, columns %d apart", spacing);
            run_sg<2, 0>(nm, (const u32x4*)idx, val, table, h16, (float*)out, 512);
            run_sg<4, 0>(nm, (const u32x4*)idx, val, table, h16, (float*)out, 512);
            run_sg<4, 1>(nm, (const u32x4*)idx, val, table, h16, (float*)out, 512);
            run_sg<4, 1>(nm, (const u32x4*)idx, val, table, h16, (float*)out, 1024);
        }
    }
    return 0;
}
