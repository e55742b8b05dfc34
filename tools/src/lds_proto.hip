// lds_proto.hip -- prototype / microbenchmark of the "LDS-resident" product kernel (round 4).
// One workgroup per CU; the row block's accumulators (R rows) AND the current slice of the gathered vector (W columns,
// double buffered, filled by LDS-DMA from L2 by dedicated loader waves) live in LDS.  Items of a (row block, slice) tile are
// sorted by row; a compute wave takes a chunk of 256 items (4 per lane: one 16-byte index load, one 16-byte value load),
// reads x[col] from LDS, and adds the products into acc[row] in LDS (read + add + write; same-row neighbours are combined in
// registers first: in-lane scan + one DPP step, runs <= 4, no row crosses a chunk).
// Template flags switch parts off for ablation (timing only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/src/lds_proto.hip -o tools/_bin/lds_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#include <cmath>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NCW = 8;                   // compute waves
constexpr int W = 9728;                  // columns per slice (38 LDS-DMA pieces of 256 floats)
constexpr int R = 19968;                 // rows per block
constexpr int NPIECE = W / 256;
constexpr int XB0 = 0, XB1 = 4 * W, ACC0 = 8 * W;          // byte offsets in LDS
constexpr int LDS_BYTES = ACC0 + 4 * (R + 64);
constexpr int CHUNK = 256;
#ifndef PROTO_D
#define PROTO_D 4
#endif
constexpr int D = PROTO_D;           // tiles of stream prefetch in registers

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// flags: bit0 x DMA, bit1 stream loads, bit2 x gather from LDS, bit3 RMW (read+write), bit4 atomic add instead, bit5 dedup
template <int NLW, int FLAGS>
__global__ __launch_bounds__((NCW + NLW) * 64, 1) void k_lds(const uint32_t* __restrict__ tidx, const float* __restrict__ tval,
                                                            const int32_t* __restrict__ tile_ptr, int64_t items_per_block, int ntiles,
                                                            const float* __restrict__ x, int n, float* __restrict__ y)
{
    constexpr bool XDMA = FLAGS & 1, STREAM = FLAGS & 2, GATHER = FLAGS & 4, RMW = FLAGS & 8, ATOM = FLAGS & 16, DEDUP = FLAGS & 32;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    // zero the accumulators
    for (int i = tid; i < R + 64; i += (NCW + NLW) * 64) *reinterpret_cast<float*>(lds + ACC0 + 4 * i) = 0.f;
    auto barrier = [&]() { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };
    if (wv >= NCW) {
        // ---------------- loader waves: slice t+1 into buffer (t+1)&1 while tile t is being consumed
        const int lw = wv - NCW;
        auto issue = [&](int t) {
            if (!XDMA) return;
            const int base = (t & 1) ? XB1 : XB0;
            for (int pc = lw; pc < NPIECE; pc += NLW) {
                int col = t * W + pc * 256 + lane * 4;
                col = col < n - 4 ? col : n - 4;                               // (the last slice: stay inside x)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x + col),
                                                 (__attribute__((address_space(3))) void*)(lds + base + pc * 1024), 16, 0, 0);
            }
        };
        issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier();
        for (int t = 0; t < ntiles; ++t) {
            if (t + 1 < ntiles) issue(t + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            barrier();
        }
    } else {
        // ---------------- compute waves
        const uint32_t* __restrict__ bi = tidx + (size_t)b * items_per_block;
        const float* __restrict__ bv = tval + (size_t)b * items_per_block;
        const uint32_t trash = (uint32_t)(R + lane + 2 * W) << 16;
        u32x4 pk[D];
        f32x4 vv[D];
        bool okf[D];
        // (tile_ptr carries D + 2 extra entries equal to the last: tiles past the end are empty, no conditionals anywhere)
        auto load = [&](int slot, int i0, int i1) {
            // chunk wv of the tile [i0, i1) (a tile holds at most NCW chunks in this prototype); lanes past its end are masked at use
            const int i = i0 + wv * CHUNK + lane * 4;
            const bool ok = i < i1;
            const int ic = ok ? i : i0;
            okf[slot] = ok;
            if (STREAM) {
                pk[slot] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(bi + ic));
                vv[slot] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(bv + ic));
            } else {
                const uint32_t z = (uint32_t)(((lane * 4 + wv * 256 + i0 * 13) * 9) % R + 2 * W) << 16 | (uint32_t)((lane * 613 + i0 * 7) % W) * 4u;
                pk[slot] = u32x4{z, z + (1u << 16), z + (2u << 16), z + (3u << 16)};
                vv[slot] = f32x4{1.f, 2.f, 3.f, 4.f};
            }
        };
#pragma unroll
        for (int u = 0; u < D; ++u) { load(u, tile_ptr[u], tile_ptr[u + 1]); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
        int q0 = tile_ptr[D], q1 = tile_ptr[D + 1];
        barrier();
        for (int t0 = 0; t0 < ntiles; t0 += D) {          // (ntiles is padded to a multiple of D by the host: empty tiles)
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int xb = (u & 1) ? XB1 : XB0;       // D is even: the parity of t0 + u is that of u
                const u32x4 k = okf[u] ? pk[u] : u32x4{trash, trash, trash, trash};
                const f32x4 v = okf[u] ? vv[u] : f32x4{0.f, 0.f, 0.f, 0.f};
                float p[4];
                uint32_t ra[4];
                const uint32_t kk[4] = {k.x, k.y, k.z, k.w};
                const float vf[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float xg = 1.0f;
                    if (GATHER) xg = *reinterpret_cast<const float*>(lds + xb + (kk[j] & 0xffffu));
                    p[j] = vf[j] * xg;
                    ra[j] = (kk[j] >> 14) & 0x3fffcu;
                }
                load(u, q0, q1);                          // the register slot is free: prefetch D tiles ahead
                q0 = q1;
                q1 = tile_ptr[t0 + u + D + 2];
                if (DEDUP) {
                    const bool e1 = ra[1] == ra[0], e2 = ra[2] == ra[1], e3 = ra[3] == ra[2];
                    float t_0 = p[0], t_1 = e1 ? t_0 + p[1] : p[1], t_2 = e2 ? t_1 + p[2] : p[2], t_3 = e3 ? t_2 + p[3] : p[3];
                    const uint32_t pra3 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)ra[3], 0x138, 0xf, 0xf, false);     // wave_shr:1
                    const float pt3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t_3), 0x138, 0xf, 0xf, false));
                    const uint32_t nra0 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)ra[0], 0x130, 0xf, 0xf, false);     // wave_shl:1
                    const float c = pra3 == ra[0] ? pt3 : 0.f;
                    t_0 += c;
                    t_1 += e1 ? c : 0.f;
                    t_2 += (e1 && e2) ? c : 0.f;
                    const uint32_t tr = (uint32_t)(ACC0 + 4 * (R + lane));
                    p[0] = t_0; p[1] = t_1; p[2] = t_2; p[3] = t_3;
                    ra[0] = e1 ? tr : ra[0];
                    ra[1] = e2 ? tr : ra[1];
                    ra[2] = e3 ? tr : ra[2];
                    ra[3] = nra0 == ra[3] ? tr : ra[3];
                }
                if (ATOM) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __hip_atomic_fetch_add(reinterpret_cast<float*>(__builtin_assume_aligned(lds + ra[j], 4)), p[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (RMW) {
                    float a[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const float*>(lds + ra[j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<float*>(lds + ra[j]) = a[j] + p[j];
                } else {
                    // keep the products alive
                    if (p[0] + p[1] + p[2] + p[3] == 1.2345f) *reinterpret_cast<float*>(lds + ra[0]) = 1.f;
                }
                barrier();
            }
        }
    }
    // epilogue stand-in: the row sums go out coalesced
    for (int i = tid; i < R; i += (NCW + NLW) * 64) y[(size_t)b * R + i] = *reinterpret_cast<const float*>(lds + ACC0 + 4 * i);
}

struct Stream {
    std::vector<uint32_t> idx;
    std::vector<float> val;
    std::vector<int32_t> tptr;
    int ntiles;
};

// one block's stream: per slice ~density*R*W items, sorted by row, runs <= 4, no row across a 256-chunk boundary
static Stream make_stream(int n, double density, unsigned seed)
{
    Stream s;
    std::mt19937 rng(seed);
    const int ntl = (n + W - 1) / W;
    s.ntiles = (ntl + D - 1) / D * D;
    s.tptr.push_back(0);
    const double mean = density * R * W;
    std::poisson_distribution<int> pois(mean);
    for (int t = 0; t < s.ntiles; ++t) {
        int cnt = t < ntl ? pois(rng) : 0;
        const int wcols = t < ntl ? std::min(W, n - t * W) : 0;
        if (wcols < W) cnt = (int)(cnt * (double)wcols / W);
        cnt = std::min(cnt, NCW * CHUNK);
        std::vector<int> rows(cnt);
        for (auto& r : rows) r = rng() % R;
        std::sort(rows.begin(), rows.end());
        // fix-ups: runs <= 4, no run across a chunk boundary
        for (int j = 1; j < cnt; ++j) {
            if (j % CHUNK == 0 && rows[j] == rows[j - 1]) rows[j] = std::min(R - 1, rows[j] + 1);
            if (j >= 4 && rows[j] == rows[j - 4]) rows[j] = std::min(R - 1, rows[j] + 1);
            if (rows[j] < rows[j - 1]) rows[j] = rows[j - 1];
        }
        for (int j = 0; j < cnt; ++j) {
            const uint32_t col = rng() % wcols;
            s.idx.push_back(((uint32_t)(rows[j] + 2 * W) << 16) | (col * 4u));
            s.val.push_back((float)((int)(rng() % 2001) - 1000) / 1000.f);
        }
        while (s.idx.size() % 4) { s.idx.push_back((uint32_t)(R + 2 * W) << 16); s.val.push_back(0.f); }
        s.tptr.push_back((int32_t)s.idx.size());
    }
    for (int e = 0; e < D + 2; ++e) s.tptr.push_back((int32_t)s.idx.size());
    return s;
}

template <int NLW, int FLAGS>
static double run(const char* name, const uint32_t* di, const float* dv, const int32_t* dt, int64_t ipb, int ntiles, const float* dx, int n,
                  float* dy, int nblk, double items)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds<NLW, FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    for (int w = 0; w < 2; ++w) k_lds<NLW, FLAGS><<<nblk, (NCW + NLW) * 64, LDS_BYTES>>>(di, dv, dt, ipb, ntiles, dx, n, dy);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) k_lds<NLW, FLAGS><<<nblk, (NCW + NLW) * 64, LDS_BYTES>>>(di, dv, dt, ipb, ntiles, dx, n, dy);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-58s loaders %d: %.3f ms  %.1f G items/s\n", name, NLW, ms, items / ms / 1e6);
    fflush(stdout);
    return ms;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10000000;
    const int nblk = argc > 2 ? atoi(argv[2]) : 501;
    const double density = 1e-5;
    Stream s = make_stream(n, density, 1234);
    const int64_t ipb = (int64_t)s.idx.size();
    printf("n %d  blocks %d  tiles/block %d  items/block %lld (%.1f per tile)  LDS %d B\n", n, nblk, s.ntiles, (long long)ipb,
           (double)ipb / s.ntiles, LDS_BYTES);
    uint32_t* di;
    float *dv, *dx, *dy;
    int32_t* dt;
    CHECK(hipMalloc(&di, ipb * 4 * nblk + 4096));
    CHECK(hipMalloc(&dv, ipb * 4 * nblk + 4096));
    CHECK(hipMalloc(&dt, s.tptr.size() * 4));
    CHECK(hipMalloc(&dx, (size_t)n * 4));
    CHECK(hipMalloc(&dy, (size_t)nblk * R * 4));
    CHECK(hipMemcpy(di, s.idx.data(), ipb * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dv, s.val.data(), ipb * 4, hipMemcpyHostToDevice));
    for (int b = 1; b < nblk; ++b) {
        CHECK(hipMemcpyAsync(di + (size_t)b * ipb, di, ipb * 4, hipMemcpyDeviceToDevice));
        CHECK(hipMemcpyAsync(dv + (size_t)b * ipb, dv, ipb * 4, hipMemcpyDeviceToDevice));
    }
    CHECK(hipMemcpy(dt, s.tptr.data(), s.tptr.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hx(n);
    std::mt19937 rng(99);
    for (auto& v : hx) v = (float)((int)(rng() % 2001) - 1000) / 500.f;
    CHECK(hipMemcpy(dx, hx.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CHECK(hipDeviceSynchronize());
    const double items = (double)ipb * nblk;
    // reference for the block
    std::vector<double> ref(R, 0.0);
    for (int t = 0; t < s.ntiles; ++t)
        for (int i = s.tptr[t]; i < s.tptr[t + 1]; ++i) {
            const int row = (int)(s.idx[i] >> 16) - 2 * W;
            if (row >= R) continue;
            ref[row] += (double)s.val[i] * hx[(size_t)t * W + (s.idx[i] & 0xffffu) / 4];
        }
    auto check = [&](const char* what) {
        std::vector<float> hy((size_t)nblk * R);
        CHECK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int b : {0, nblk / 2, nblk - 1})
            for (int r = 0; r < R; ++r) worst = std::max(worst, std::fabs(hy[(size_t)b * R + r] - ref[r]) / (1.0 + std::fabs(ref[r])));
        printf("   check %s: worst relative error %.3g %s\n", what, worst, worst < 1e-4 ? "OK" : "MISMATCH");
    };
    // full kernel
    run<2, 1 | 2 | 4 | 8 | 32>("full: dma + stream + gather + rmw + dedup", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    check("rmw+dedup");
    run<4, 1 | 2 | 4 | 8 | 32>("full: dma + stream + gather + rmw + dedup", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<1, 1 | 2 | 4 | 8 | 32>("full: dma + stream + gather + rmw + dedup", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<2, 1 | 2 | 4 | 16>("full with ds_add_f32 (no dedup)", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    check("atomic");
    // ablations (timing only)
    run<2, 2 | 4 | 8 | 32>("no x dma", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<2, 1 | 4 | 8 | 32>("no stream", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<2, 1 | 2 | 8 | 32>("no gather", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<2, 1 | 2 | 4>("no rmw", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<2, 1 | 2 | 4 | 8>("rmw without dedup (wrong sums)", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<2, 1>("x dma only", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<4, 1>("x dma only", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<2, 2>("stream only", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<2, 1 | 2>("dma + stream", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<2, 4 | 8 | 32>("LDS work only (gather + rmw + dedup, synthetic items)", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    run<2, 0>("barriers only", di, dv, dt, ipb, s.ntiles, dx, n, dy, nblk, items);
    return 0;
}
