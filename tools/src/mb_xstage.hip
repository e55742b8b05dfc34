// mb_xstage.hip -- how fast can every CU pull the SAME sequence of slices of one vector (40 MB, L2 / MALL resident after the
// first touch) into its LDS?  One workgroup per CU, NL loader waves, LDS-DMA pieces of 1 KiB (global_load_lds_dwordx4), DEPTH slices
// in flight (ring of DEPTH + 1 buffers), a barrier per slice like the consumer kernel would have.
//   mode bits: 1 = rotate the piece order by workgroup (spread L2 channels), 2 = register loads instead of LDS-DMA (no LDS write),
//              4 = no barrier
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/src/mb_xstage.hip -o tools/_bin/mb_xstage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NL, int WPIECES, int DEPTH, int MODE>
__global__ __launch_bounds__(NL * 64, 1) void k_stage(const float* __restrict__ x, int n, int nslices, float* out)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NBUF = DEPTH + 1;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rot = (MODE & 1) ? (blockIdx.x * 7) % WPIECES : 0;
    float acc = 0.f;
    auto issue = [&](int t) {
        const int base = (t % NBUF) * WPIECES * 1024;
        for (int pc = wv; pc < WPIECES; pc += NL) {
            int pr = pc + rot;
            pr = pr >= WPIECES ? pr - WPIECES : pr;
            int col = (t * WPIECES + pr) * 256 + lane * 4;
            col = col < n - 4 ? col : n - 4;
            if (MODE & 2) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + col);
                acc += v.x + v.y + v.z + v.w;
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x + col),
                                                 (__attribute__((address_space(3))) void*)(lds + base + pr * 1024), 16, 0, 0);
            }
        }
    };
    constexpr int PPW = (WPIECES + NL - 1) / NL;          // pieces per wave and slice (upper bound)
    for (int t = 0; t < DEPTH; ++t) issue(t);
    for (int t = 0; t < nslices; ++t) {
        issue(t + DEPTH);                                  // (past the end: clamped addresses, harmless)
        if (!(MODE & 2)) {
            // wait until slice t has landed: at most DEPTH slices' pieces of this wave remain in flight
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * 1 > 63 ? 63 : PPW * 1) : "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * 2 > 63 ? 63 : PPW * 2) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * 3 > 63 ? 63 : PPW * 3) : "memory");
        }
        if (!(MODE & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!(MODE & 2)) acc = *reinterpret_cast<const float*>(lds + 4 * threadIdx.x);
    if (acc == 123.456f) out[0] = acc;
}

template <int NL, int WPIECES, int DEPTH, int MODE>
static void run(const float* dx, int n, float* dout, int nblk)
{
    const int nslices = (n + WPIECES * 256 - 1) / (WPIECES * 256);
    const int ldsb = (DEPTH + 1) * WPIECES * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stage<NL, WPIECES, DEPTH, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) k_stage<NL, WPIECES, DEPTH, MODE><<<nblk, NL * 64, ldsb>>>(dx, n, nslices, dout);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) k_stage<NL, WPIECES, DEPTH, MODE><<<nblk, NL * 64, ldsb>>>(dx, n, nslices, dout);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)nblk * nslices * WPIECES * 1024.0;
    printf("loaders %2d  slice %3d KiB  depth %d  mode %d  blocks %d: %.3f ms  %.2f TB/s  %.1f GB/s per CU (256)\n", NL, WPIECES, DEPTH, MODE, nblk, ms,
           bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10000000;
    float *dx, *dout;
    CHECK(hipMalloc(&dx, (size_t)n * 4 + 4096));
    CHECK(hipMalloc(&dout, 4096));
    CHECK(hipMemset(dx, 0, (size_t)n * 4 + 4096));
    for (int nblk : {256, 512}) {
        run<2, 38, 1, 0>(dx, n, dout, nblk);
        run<4, 38, 1, 0>(dx, n, dout, nblk);
        run<8, 38, 1, 0>(dx, n, dout, nblk);
        run<16, 38, 1, 0>(dx, n, dout, nblk);
        run<4, 38, 1, 1>(dx, n, dout, nblk);
        run<8, 38, 1, 1>(dx, n, dout, nblk);
        run<4, 38, 2, 0>(dx, n, dout, nblk);
        run<8, 38, 2, 1>(dx, n, dout, nblk);
        run<8, 38, 3, 1>(dx, n, dout, nblk);
        run<8, 16, 3, 1>(dx, n, dout, nblk);
        run<8, 16, 1, 1>(dx, n, dout, nblk);
        run<8, 38, 1, 4>(dx, n, dout, nblk);
        run<8, 38, 1, 5>(dx, n, dout, nblk);
        run<8, 38, 1, 2 | 4>(dx, n, dout, nblk);
        run<8, 38, 1, 1 | 2 | 4>(dx, n, dout, nblk);
        run<16, 38, 1, 1 | 2 | 4>(dx, n, dout, nblk);
    }
    return 0;
}
