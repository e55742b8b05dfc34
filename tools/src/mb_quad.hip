// mb_quad.hip -- round 5 microbenchmark: what does ONE gather instruction cost on gfx950 as a function of its width (4, 8, 16 bytes
// per lane), of the lines it touches and of the workgroup shape?  Same skeleton as bw_probe.hip's "stream + gather" (8 bytes
// of item stream per item, 16-byte non-temporal loads, the tile format's 256-item interleave, a 256 KB table = one L2-resident
// panel), plus:
//   GW   dwords per gathered lane (1: the shipped kernel; 2, 4: the "deduplicated quad gather" of VERDICT r4 item 3 -- a lane fetches
//        the aligned 8 / 16 bytes that hold its column and takes the component it needs),
//   NG   gather instructions per 4 streamed items and lane (4: one per item; 3: the BEST case of a deduplicated quad gather at the
//        bench tiles' density, where 13 100 sorted items of a tile fall into ~9 000 distinct quads -- no compaction / routing cost
//        is charged, so this is an upper bound of what deduplication can reach),
//   NTHR threads per workgroup (512 x 2 per CU, or 1024 x 1 per CU),
//   spacing of neighbouring sorted items in columns as a rational number (5 = the bench tiles, 5/2 = twice the rows per workgroup).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/src/mb_quad.hip -o tools/_bin/mb_quad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void k_fill_idx(unsigned* idx, size_t n, int num, int den)
{
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const size_t grp = p >> 8;
        const unsigned L = (unsigned)(p & 255) >> 2, j = (unsigned)p & 3;
        idx[p] = (unsigned)(((grp * 256 + 64 * j + L) * (size_t)num) / (size_t)den) & 0xFFFFu;
    }
}

template <int GW> __device__ __forceinline__ float gather(const float* __restrict__ table, unsigned col)
{
    if (GW == 1) return table[col];
    if (GW == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(table + (col & ~1u));
        return (col & 1u) ? v.y : v.x;
    }
    const f32x4 v = *reinterpret_cast<const f32x4*>(table + (col & ~3u));
    const unsigned c = col & 3u;
    return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w;
}

template <int NTHR, int UNROLL, int GW, int NG>
__global__ __launch_bounds__(NTHR, 4) void k_sg(const u32x4* __restrict__ idx, const f32x4* __restrict__ val,
                                                const float* __restrict__ table, size_t n16, float* out)
{
    const size_t per = n16 / gridDim.x;
    const u32x4* pi = idx + (size_t)blockIdx.x * per;
    const f32x4* pv = val + (size_t)blockIdx.x * per;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * NTHR < per; i += NTHR * UNROLL) {
        u32x4 k[UNROLL];
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { k[u] = __builtin_nontemporal_load(pi + i + u * NTHR); v[u] = __builtin_nontemporal_load(pv + i + u * NTHR); }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const float g0 = gather<GW>(table, k[u].x), g1 = gather<GW>(table, k[u].y), g2 = NG > 2 ? gather<GW>(table, k[u].z) : g0;
            const float g3 = NG > 3 ? gather<GW>(table, k[u].w) : g1;
            __builtin_amdgcn_s_waitcnt(0x0F70);        // rounds of four gathers per lane (the tiled kernel's schedule)
            acc += v[u].x * g0 + v[u].y * g1 + v[u].z * g2 + v[u].w * g3;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

// the same loop with the gathers as per-lane LDS-DMA (global_load_lds_dword: the dword of lane L lands at lds[base + 4 L], no VGPR
// return path) followed by a conflict-free ds_read of the lane's own slot: does the direct-to-LDS path handle scattered lines faster?
template <int NTHR, int UNROLL>
__global__ __launch_bounds__(NTHR, 4) void k_sg_ldsdma(const u32x4* __restrict__ idx, const f32x4* __restrict__ val,
                                                       const float* __restrict__ table, size_t n16, float* out)
{
    __shared__ float stage[NTHR / 64][4][64];
    const size_t per = n16 / gridDim.x;
    const u32x4* pi = idx + (size_t)blockIdx.x * per;
    const f32x4* pv = val + (size_t)blockIdx.x * per;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * NTHR < per; i += NTHR * UNROLL) {
        u32x4 k[UNROLL];
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { k[u] = __builtin_nontemporal_load(pi + i + u * NTHR); v[u] = __builtin_nontemporal_load(pv + i + u * NTHR); }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const unsigned c[4] = {k[u].x, k[u].y, k[u].z, k[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(table + c[j]),
                                                 (__attribute__((address_space(3))) void*)&stage[wv][j][0], 4, 0, 0);
            __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): the four LDS-DMA gathers have landed
            const float g0 = stage[wv][0][lane], g1 = stage[wv][1][lane], g2 = stage[wv][2][lane], g3 = stage[wv][3][lane];
            acc += v[u].x * g0 + v[u].y * g1 + v[u].z * g2 + v[u].w * g3;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int NTHR> void run_ldsdma(const char* name, const u32x4* idx, const f32x4* val, const float* table, size_t n16, float* out)
{
    const int grid = 512 * 512 / NTHR;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k_sg_ldsdma<NTHR, 4><<<grid, NTHR>>>(idx, val, table, n16, out);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) k_sg_ldsdma<NTHR, 4><<<grid, NTHR>>>(idx, val, table, n16, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-34s wg %4d x %3d  4-byte gathers by LDS-DMA + ds_read, 4 per 4 items: %.3f ms  %6.1f G items/s\n", name, NTHR, grid, ms,
           n16 * 4.0 / ms / 1e6);
    fflush(stdout);
}

template <int NTHR, int GW, int NG> void run(const char* name, const u32x4* idx, const f32x4* val, const float* table, size_t n16, float* out)
{
    const int grid = 512 * 512 / NTHR;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k_sg<NTHR, 4, GW, NG><<<grid, NTHR>>>(idx, val, table, n16, out);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) k_sg<NTHR, 4, GW, NG><<<grid, NTHR>>>(idx, val, table, n16, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-34s wg %4d x %3d  %d-byte gathers, %d per 4 items: %.3f ms  %6.1f G items/s  (%5.1f G gather instr. lanes/s)\n", name, NTHR, grid,
           4 * GW, NG, ms, n16 * 4.0 / ms / 1e6, n16 * 1.0 * NG / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const size_t bytes = (argc > 1 ? atof(argv[1]) : 8.0) * (1ull << 30);
    char* a; float* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, bytes);
    const size_t items = bytes / 8, h16 = items / 4;
    float* table; hipMalloc(&table, 65536 * 4 + 64); hipMemset(table, 0, 65536 * 4 + 64);
    unsigned* idx = (unsigned*)a;
    const f32x4* val = (const f32x4*)(a + items * 4);
    const int sp[][2] = {{5, 1}, {5, 2}, {40, 1}, {997, 1}};
    for (auto& s : sp) {
        k_fill_idx<<<4096, 256>>>(idx, items, s[0], s[1]);
        hipDeviceSynchronize();
        char nm[64]; snprintf(nm, sizeof nm, "columns %d/%d apart", s[0], s[1]);
        run<512, 1, 4>(nm, (const u32x4*)idx, val, table, h16, out);
        run<512, 2, 4>(nm, (const u32x4*)idx, val, table, h16, out);
        run<512, 4, 4>(nm, (const u32x4*)idx, val, table, h16, out);
        run<512, 4, 3>(nm, (const u32x4*)idx, val, table, h16, out);
        run<512, 1, 3>(nm, (const u32x4*)idx, val, table, h16, out);
        run<1024, 1, 4>(nm, (const u32x4*)idx, val, table, h16, out);
        run<1024, 4, 3>(nm, (const u32x4*)idx, val, table, h16, out);
        run<512, 4, 2>(nm, (const u32x4*)idx, val, table, h16, out);
        run_ldsdma<512>(nm, (const u32x4*)idx, val, table, h16, out);
    }
    return 0;
}
