// microbench_gather.hip -- what does a 4-byte gather cost on MI355X, by where the table lives?
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_gather.hip -o tools/_bin/mb_gather
// Informs the SpMV format (DESIGN.md "gather cost"): index stream is read coalesced (4 B/lane), each lane
// gathers table[idx]; variants: table size (L1/L2/Infinity Cache/HBM), line sharing (idx sorted in windows),
// gather from LDS instead, and LDS float atomic adds (scatter side).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#include <functional>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_stream(const int* __restrict__ idx, const float* __restrict__ val, size_t n, float* out)
{
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += val[i] * (float)idx[i];
    if (s == 123.456f) out[0] = s;
}

__global__ __launch_bounds__(256) void k_gather(const int* __restrict__ idx, const float* __restrict__ val,
                                                const float* __restrict__ table, size_t n, float* out)
{
    float s = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const int i0 = idx[i], i1 = idx[i + stride], i2 = idx[i + 2 * stride], i3 = idx[i + 3 * stride];
        const float v0 = val[i], v1 = val[i + stride], v2 = val[i + 2 * stride], v3 = val[i + 3 * stride];
        s += v0 * table[i0] + v1 * table[i1] + v2 * table[i2] + v3 * table[i3];
    }
    for (; i < n; i += stride) s += val[i] * table[idx[i]];
    if (s == 123.456f) out[0] = s;
}

// the same gathers issued as buffer loads (SGPR resource descriptor + 32-bit VGPR byte offset) instead of flat global loads
__global__ __launch_bounds__(256) void k_gather_buf(const int* __restrict__ idx, const float* __restrict__ val,
                                                    const float* __restrict__ table, unsigned table_bytes, size_t n, float* out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)table_bytes, 0x00020000);
    float s = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const int i0 = idx[i], i1 = idx[i + stride], i2 = idx[i + 2 * stride], i3 = idx[i + 3 * stride];
        const float v0 = val[i], v1 = val[i + stride], v2 = val[i + 2 * stride], v3 = val[i + 3 * stride];
        const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i0 * 4, 0, 0));
        const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i1 * 4, 0, 0));
        const float t2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i2 * 4, 0, 0));
        const float t3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i3 * 4, 0, 0));
        s += v0 * t0 + v1 * t1 + v2 * t2 + v3 * t3;
    }
    for (; i < n; i += stride) s += val[i] * table[idx[i]];
    if (s == 123.456f) out[0] = s;
}

// table window copied to LDS first (window = wsize floats), each block loops over its share of the index stream
__global__ __launch_bounds__(256) void k_gather_lds(const int* __restrict__ idx, const float* __restrict__ val,
                                                    const float* __restrict__ table, int wsize, size_t n, float* out)
{
    extern __shared__ float win[];
    for (int i = threadIdx.x; i < wsize; i += 256) win[i] = table[i];
    __syncthreads();
    float s = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const int i0 = idx[i], i1 = idx[i + stride], i2 = idx[i + 2 * stride], i3 = idx[i + 3 * stride];
        const float v0 = val[i], v1 = val[i + stride], v2 = val[i + 2 * stride], v3 = val[i + 3 * stride];
        s += v0 * win[i0] + v1 * win[i1] + v2 * win[i2] + v3 * win[i3];
    }
    for (; i < n; i += stride) s += val[i] * win[idx[i]];
    if (s == 123.456f) out[0] = s;
}

// scatter side: acc[idx] += val in LDS (float atomic), acc size asize floats
__global__ __launch_bounds__(256) void k_lds_atomic(const int* __restrict__ idx, const float* __restrict__ val, int asize,
                                                    size_t n, float* out)
{
    extern __shared__ float acc[];
    for (int i = threadIdx.x; i < asize; i += 256) acc[i] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const int i0 = idx[i], i1 = idx[i + stride], i2 = idx[i + 2 * stride], i3 = idx[i + 3 * stride];
        const float v0 = val[i], v1 = val[i + stride], v2 = val[i + 2 * stride], v3 = val[i + 3 * stride];
        atomicAdd(&acc[i0], v0); atomicAdd(&acc[i1], v1); atomicAdd(&acc[i2], v2); atomicAdd(&acc[i3], v3);
    }
    for (; i < n; i += stride) atomicAdd(&acc[idx[i]], val[i]);
    __syncthreads();
    float s = 0;
    for (int i = threadIdx.x; i < asize; i += 256) s += acc[i];
    if (s == 123.456f) out[0] = s;
}

// the candidate SpMV inner loop: packed index = (local row << 17) | local col; gather x from an L2-resident
// panel (items sorted by column), scatter-add the product into LDS row accumulators
__global__ __launch_bounds__(512) void k_gather_scatter(const unsigned* __restrict__ pidx, const float* __restrict__ val,
                                                        const float* __restrict__ table, int asize, size_t n, float* out)
{
    extern __shared__ float acc[];
    for (int i = threadIdx.x; i < asize; i += 512) acc[i] = 0;
    __syncthreads();
    const size_t per = n / gridDim.x;              // contiguous chunk per block (sorted batches stay together)
    const size_t b0 = per * blockIdx.x, b1 = b0 + per;
    size_t i = b0 + threadIdx.x;
    for (; i + 3 * 512 < b1; i += 4 * 512) {
        const unsigned p0 = pidx[i], p1 = pidx[i + 512], p2 = pidx[i + 1024], p3 = pidx[i + 1536];
        const float v0 = val[i], v1 = val[i + 512], v2 = val[i + 1024], v3 = val[i + 1536];
        const float x0 = table[p0 & 0x1ffff], x1 = table[p1 & 0x1ffff], x2 = table[p2 & 0x1ffff], x3 = table[p3 & 0x1ffff];
        atomicAdd(&acc[p0 >> 17], v0 * x0); atomicAdd(&acc[p1 >> 17], v1 * x1);
        atomicAdd(&acc[p2 >> 17], v2 * x2); atomicAdd(&acc[p3 >> 17], v3 * x3);
    }
    for (; i < b1; i += 512) { const unsigned p = pidx[i]; atomicAdd(&acc[p >> 17], val[i] * table[p & 0x1ffff]); }
    __syncthreads();
    float s = 0;
    for (int i = threadIdx.x; i < asize; i += 512) s += acc[i];
    if (s == 123.456f) out[0] = s;
}

static float run(const char* name, size_t n, double bytes_per_item, std::function<void()> fn)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn(); CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) fn();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("%-46s %8.3f ms  %8.1f Gitem/s  stream %7.1f GB/s\n", name, ms, n / ms * 1e-6, n * bytes_per_item / ms * 1e-6);
    return ms;
}

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 26;     // 67M items = 512 MiB of (idx,val)
    int *d_idx; float *d_val, *d_table, *d_out;
    const size_t tmax = (size_t)1 << 26;                                       // 256 MiB table max
    CK(hipMalloc(&d_idx, n * 4)); CK(hipMalloc(&d_val, n * 4)); CK(hipMalloc(&d_table, tmax * 4)); CK(hipMalloc(&d_out, 4));
    std::vector<float> hv(1 << 20, 1.0f);
    for (size_t o = 0; o < n; o += hv.size()) CK(hipMemcpy(d_val + o, hv.data(), std::min(hv.size(), n - o) * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_table, 0, tmax * 4));
    std::vector<int> hi(n);
    std::mt19937_64 rng(1);
    const int grid = 2048;
    auto fill = [&](size_t tsize, size_t sort_window) {
        for (size_t i = 0; i < n; ++i) hi[i] = (int)(rng() % tsize);
        if (sort_window > 1) for (size_t o = 0; o + sort_window <= n; o += sort_window) std::sort(hi.begin() + o, hi.begin() + o + sort_window);
        CK(hipMemcpy(d_idx, hi.data(), n * 4, hipMemcpyHostToDevice));
    };
    const bool quick = argc > 2;
    run("stream idx+val only", n, 8, [&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n, d_out); });
    char name[128];
    if (!quick)
    for (size_t tsize : {(size_t)8 << 10, (size_t)64 << 10, (size_t)256 << 10, (size_t)512 << 10, (size_t)1 << 20, (size_t)2 << 20,
                         (size_t)10 << 20, (size_t)64 << 20}) {
        fill(tsize, 1);
        snprintf(name, sizeof name, "gather table %6zu KiB random", tsize * 4 >> 10);
        run(name, n, 8, [&] { hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, d_idx, d_val, d_table, n, d_out); });
    }
    // buffer loads vs flat global loads for the same gathers
    for (size_t tsize : {(size_t)64 << 10, (size_t)1 << 20}) {
        fill(tsize, 1);
        snprintf(name, sizeof name, "gather table %6zu KiB random, global", tsize * 4 >> 10);
        run(name, n, 8, [&] { hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, d_idx, d_val, d_table, n, d_out); });
        snprintf(name, sizeof name, "gather table %6zu KiB random, buffer", tsize * 4 >> 10);
        run(name, n, 8, [&] { hipLaunchKernelGGL(k_gather_buf, dim3(grid), dim3(256), 0, 0, d_idx, d_val, d_table, (unsigned)(tsize * 4), n, d_out); });
    }
    {
        fill((size_t)64 << 10, 32768);
        run("gather 256 KiB, sorted windows 32768, global", n, 8, [&] { hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, d_idx, d_val, d_table, n, d_out); });
        run("gather 256 KiB, sorted windows 32768, buffer", n, 8, [&] { hipLaunchKernelGGL(k_gather_buf, dim3(grid), dim3(256), 0, 0, d_idx, d_val, d_table, (unsigned)((64 << 10) * 4), n, d_out); });
    }
    // line sharing: consecutive items sorted inside windows (the wave's 64 lanes then touch few lines)
    if (!quick)
    for (size_t win : {(size_t)4096, (size_t)32768, (size_t)262144}) {
        fill((size_t)256 << 10, win);
        snprintf(name, sizeof name, "gather table 1024 KiB, sorted windows of %zu", win);
        run(name, n, 8, [&] { hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, d_idx, d_val, d_table, n, d_out); });
    }
    if (!quick)
    for (size_t win : {(size_t)32768, (size_t)262144}) {
        fill((size_t)10 << 20, win);
        snprintf(name, sizeof name, "gather table 40 MiB, sorted windows of %zu", win);
        run(name, n, 8, [&] { hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, d_idx, d_val, d_table, n, d_out); });
    }
    if (!quick)
    for (int wsize : {4096, 8192, 16384}) {
        fill(wsize, 1);
        snprintf(name, sizeof name, "gather from LDS window %d floats", wsize);
        const int g = wsize <= 8192 ? 2048 : 1024;
        run(name, n, 8, [&] { hipLaunchKernelGGL(k_gather_lds, dim3(g), dim3(256), wsize * 4, 0, d_idx, d_val, d_table, wsize, n, d_out); });
    }
    for (int asize : {2048, 8192, 16384}) {
        fill(asize, 1);
        snprintf(name, sizeof name, "LDS atomicAdd random, acc %d floats", asize);
        const int g = asize <= 8192 ? 2048 : 1024;
        run(name, n, 8, [&] { hipLaunchKernelGGL(k_lds_atomic, dim3(g), dim3(256), asize * 4, 0, d_idx, d_val, asize, n, d_out); });
        fill(asize, (size_t)asize * 3);      // runs of ~3 equal addresses in consecutive lanes (items sorted by row)
        snprintf(name, sizeof name, "LDS atomicAdd sorted runs~3, acc %d", asize);
        run(name, n, 8, [&] { hipLaunchKernelGGL(k_lds_atomic, dim3(g), dim3(256), asize * 4, 0, d_idx, d_val, asize, n, d_out); });
    }
    // candidate: 512 blocks, each a contiguous chunk made of batches of 25600 items sorted by column over a
    // 131072-column panel (density 1/5), rows random in [0, 19532)
    {
        const int rows = 19532, batch = 25600;
        for (size_t i = 0; i < n; ++i) hi[i] = (int)(((unsigned)(rng() % rows) << 17) | (unsigned)(rng() % 131072));
        for (size_t o = 0; o + batch <= n; o += batch)
            std::sort(hi.begin() + o, hi.begin() + o + batch, [](int a, int b) { return (a & 0x1ffff) < (b & 0x1ffff); });
        CK(hipMemcpy(d_idx, hi.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipFuncSetAttribute((const void*)k_gather_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, rows * 4));
        run("candidate: L2 panel gather + LDS scatter", n, 8, [&] {
            hipLaunchKernelGGL(k_gather_scatter, dim3(512), dim3(512), rows * 4, 0, (const unsigned*)d_idx, d_val, d_table, rows, n, d_out); });
    }
    return 0;
}
