// mb_window.hip -- pass-1 ceiling of a "window" gather: instead of one TA gather per 64 consecutive column-sorted items (11 cache lines
// at the bench tiles' density), a wave copies the column range those 64 items span (1.3 KB) into its own LDS buffer with 1 KiB
// LDS-DMA pieces and gathers from LDS.  Same item stream and index pattern as tools/src/bw_probe.hip's "stream + gather" (which runs
// 550 G items/s at 5 columns apart with TA gathers).  RIF = gather rows (of 64 items) whose windows are in flight together per wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/src/mb_window.hip -o tools/_bin/mb_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_fill_idx(unsigned* idx, size_t n, int spacing10)
{
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const size_t grp = p >> 8;
        const unsigned L = (unsigned)(p & 255) >> 2, j = (unsigned)p & 3;
        idx[p] = (unsigned)(((grp * 256 + 64 * j + L) * (size_t)spacing10) / 10) & 0xFFFFu;
    }
}

template <int MODE /*0 TA gather rounds of 4, 1 window*/, int RIF, int WINB>
__global__ __launch_bounds__(512, 4) void k_sg(const u32x4* __restrict__ idx, const f32x4* __restrict__ val, const float* __restrict__ table,
                                              size_t n16, float* out)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t per = n16 / gridDim.x;
    const u32x4* pi = idx + (size_t)blockIdx.x * per;
    const f32x4* pv = val + (size_t)blockIdx.x * per;
    char* wbuf = lds + wv * (RIF * WINB);
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < per; i += 512) {
        const u32x4 k = __builtin_nontemporal_load(pi + i);
        const f32x4 v = __builtin_nontemporal_load(pv + i);
        const unsigned kk[4] = {k.x, k.y, k.z, k.w};
        const float vf[4] = {v.x, v.y, v.z, v.w};
        if (MODE == 0) {
            const float g0 = table[kk[0]], g1 = table[kk[1]], g2 = table[kk[2]], g3 = table[kk[3]];
            acc += vf[0] * g0 + vf[1] * g1 + vf[2] * g2 + vf[3] * g3;
        } else {
#pragma unroll
            for (int j0 = 0; j0 < 4; j0 += RIF) {
                unsigned base[RIF];
                bool wrap[RIF];
#pragma unroll
                for (int r = 0; r < RIF; ++r) {
                    const unsigned cmin = (unsigned)__builtin_amdgcn_readlane((int)kk[j0 + r], 0) & ~3u;
                    const unsigned cmax = (unsigned)__builtin_amdgcn_readlane((int)kk[j0 + r], 63);
                    base[r] = cmin;
                    wrap[r] = cmax < cmin || cmax - cmin >= WINB / 4;          // (crosses the table's end, or wider than the buffer: TA gather)
                    if (!wrap[r]) {
                        const int npc = (int)((cmax - cmin) >> 8) + 1;
                        for (int p = 0; p < npc; ++p) {
                            unsigned c = cmin + p * 256 + lane * 4;
                            c = c < 65532u ? c : 65532u;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(table + c),
                                                             (__attribute__((address_space(3))) void*)(wbuf + r * WINB + p * 1024), 16, 0, 0);
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int r = 0; r < RIF; ++r) {
                    const float g = wrap[r] ? table[kk[j0 + r]] : *reinterpret_cast<const float*>(wbuf + r * WINB + (kk[j0 + r] - base[r]) * 4);
                    acc += vf[j0 + r] * g;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the buffer is reused by the next rows
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int RIF, int WINB> void run(const char* name, const u32x4* idx, const f32x4* val, const float* table, size_t n16, float* out, int grid, int lds_pad = 0)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ldsb = lds_pad ? lds_pad : (MODE ? 8 * RIF * WINB : 0);       // (lds_pad: 100 KB forces one workgroup per CU)
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sg<MODE, RIF, WINB>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb > 0 ? ldsb : 16));
    for (int w = 0; w < 2; ++w) k_sg<MODE, RIF, WINB><<<grid, 512, ldsb>>>(idx, val, table, n16, out);
    CHECK(hipGetLastError());
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) k_sg<MODE, RIF, WINB><<<grid, 512, ldsb>>>(idx, val, table, n16, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-34s rows in flight %d, window %4d B, LDS %3d KB/WG, grid %4d: %.3f ms  %.1f G items/s\n", name, RIF, WINB, ldsb / 1024, grid, ms, n16 * 4.0 / ms / 1e6);
    fflush(stdout);
}

int main()
{
    const size_t items = (size_t)1 << 30, n16 = items / 4;
    unsigned* idx; float *val, *table, *out;
    CHECK(hipMalloc(&idx, items * 4)); CHECK(hipMalloc(&val, items * 4)); CHECK(hipMalloc(&table, 65536 * 4 + 4096)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(val, 1, items * 4)); CHECK(hipMemset(table, 0, 65536 * 4 + 4096));
    for (int sp10 : {50, 25}) {
        k_fill_idx<<<4096, 256>>>(idx, items, sp10);
        CHECK(hipDeviceSynchronize());
        printf("--- sorted neighbours %.1f columns apart (%s)\n", sp10 / 10.0, sp10 == 50 ? "the bench tiles: 19 968 rows per workgroup" : "39 936 rows per workgroup");
        run<0, 4, 2048>("TA gathers, ONE workgroup (8 waves) per CU", (const u32x4*)idx, (const f32x4*)val, table, n16, out, 256, 100 * 1024);
        run<0, 4, 2048>("TA gathers, ONE workgroup per CU, 2 rounds", (const u32x4*)idx, (const f32x4*)val, table, n16, out, 512, 100 * 1024);
        for (int grid : {512, 1024}) {
            run<0, 4, 2048>("TA gathers, 4 per lane in flight", (const u32x4*)idx, (const f32x4*)val, table, n16, out, grid);
            run<1, 1, 2048>("windows in LDS", (const u32x4*)idx, (const f32x4*)val, table, n16, out, grid);
            run<1, 2, 2048>("windows in LDS", (const u32x4*)idx, (const f32x4*)val, table, n16, out, grid);
            run<1, 4, 2048>("windows in LDS", (const u32x4*)idx, (const f32x4*)val, table, n16, out, grid);
            run<1, 4, 1024>("windows in LDS", (const u32x4*)idx, (const f32x4*)val, table, n16, out, grid);
        }
    }
    return 0;
}
