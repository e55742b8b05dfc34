// lds_proto5.hip -- prototype of the barrier-free "LDS-resident" product kernel (round 4).
// One workgroup of NCW compute + NLW loader waves per CU.  LDS: the row block's accumulators (R rows) and a ring of NB slices
// (WS columns each) of the gathered vector, filled by LDS-DMA by the loader waves.  Every compute wave OWNS a row range of the
// block and walks its own item stream (sorted by slice, chunks of 64*IPL items re-sorted by row), so no two waves ever touch the
// same accumulator and the main loop has no workgroup barrier: the only synchronisation is the ring (FULL / DONE words in LDS).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/src/lds_proto5.hip -o tools/_bin/lds_proto5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#include <cmath>

#ifndef P_NCW
#define P_NCW 12
#endif
#ifndef P_NLW
#define P_NLW 4
#endif
#ifndef P_WS
#define P_WS 2048
#endif
#ifndef P_NB
#define P_NB 10
#endif
#ifndef P_D
#define P_D 8
#endif
#ifndef P_LDEPTH
#define P_LDEPTH 2      /* slices a loader wave keeps in flight */
#endif
constexpr int NCW = P_NCW, NLW = P_NLW, WS = P_WS, NB = P_NB, D = P_D, LDEPTH = P_LDEPTH;
constexpr int IPL = 2, CHUNK = 64 * IPL;
constexpr int R = 19968, RW = R / NCW;              // rows per block / per wave
constexpr int NPIECE = WS / 256;
constexpr int ACC0 = NB * WS * 4;                   // x ring first, accumulators behind it
constexpr int FLG0 = ACC0 + 4 * (R + 64);           // FULL[NB], DONE[NB]
constexpr int LDS_BYTES = FLG0 + 8 * NB + 16;
static_assert(R % NCW == 0, "rows per wave");
static_assert(LDS_BYTES <= 163840, "LDS");
static_assert((ACC0 / 4 + R + 64) < 65536, "row code");

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ unsigned long long g_stamps[64];

// meta per chunk (two words): w0 = slot_hi | gen_hi << 8 (wait until FULL[slot_hi] >= gen_hi), w1 = first_done_slot | ndone << 8
// flags: bit0 x DMA (+ ring protocol), bit1 stream loads, bit2 gathers, bit3 accumulator update
template <int FLAGS>
__global__ __launch_bounds__((NCW + NLW) * 64, 1) void k_ring(const uint32_t* __restrict__ tidx, const float* __restrict__ tval,
                                                             const uint32_t* __restrict__ meta, const int32_t* __restrict__ wptr,
                                                             const int32_t* __restrict__ cptr, const int32_t* __restrict__ nchp, int64_t items_per_block, int nslices,
                                                             const float* __restrict__ x, float* __restrict__ y)
{
    constexpr bool XDMA = FLAGS & 1, STREAM = FLAGS & 2, GATHER = FLAGS & 4, RMW = FLAGS & 8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    for (int i = tid; i < R + 64 + 2 * NB + 4; i += (NCW + NLW) * 64) *reinterpret_cast<uint32_t*>(lds + ACC0 + 4 * i) = 0u;
    __syncthreads();
    volatile uint32_t* full = reinterpret_cast<volatile uint32_t*>(lds + FLG0);
    volatile uint32_t* done = reinterpret_cast<volatile uint32_t*>(lds + FLG0 + 4 * NB);
    if (wv >= NCW) {
        // ---------------- loader waves: wave j brings slices j, j + NLW, ...; LDEPTH of them in flight
        if (XDMA) {
            const int lw = wv - NCW;
            __builtin_amdgcn_s_setprio(3);
            const char* gsrc = reinterpret_cast<const char*>(x) + (size_t)lane * 16;
            unsigned long long c_poll = 0, c_issue = 0, c_wait = 0, c0 = __builtin_readcyclecounter(), c1;
            auto publish = [&](int s) {                  // slice s has landed (the caller waited for it)
                if (lane == 0) full[s % NB] = (uint32_t)(s / NB + 1);
            };
            int inflight = 0;
            for (int s = lw; s < nslices; s += NLW) {
                const int slot = s % NB, gen = s / NB;
                if (gen > 0 && done[slot] < (uint32_t)(NCW * gen)) {
                    // never sit on landed-but-unpublished slices while waiting for the consumers
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    for (int k = inflight; k >= 1; --k) publish(s - k * NLW);
                    inflight = 0;
                    while (done[slot] < (uint32_t)(NCW * gen)) __builtin_amdgcn_s_sleep(2);
                }
                c1 = __builtin_readcyclecounter(); c_poll += c1 - c0; c0 = c1;
                const char* g = gsrc + (size_t)s * (WS * 4);
#pragma unroll
                for (int pc = 0; pc < NPIECE; ++pc)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + pc * 1024),
                                                     (__attribute__((address_space(3))) void*)(lds + slot * (WS * 4) + pc * 1024), 16, 0, 0);
                c1 = __builtin_readcyclecounter(); c_issue += c1 - c0; c0 = c1;
                if (++inflight == LDEPTH) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LDEPTH - 1) * NPIECE) : "memory");
                    publish(s - (LDEPTH - 1) * NLW);
                    --inflight;
                }
                c1 = __builtin_readcyclecounter(); c_wait += c1 - c0; c0 = c1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            {   // the slices still in flight
                int last = lw + ((nslices - 1 - lw) / NLW) * NLW;
                for (int k = inflight - 1; k >= 0; --k) publish(last - k * NLW);
            }
            if (b == 100 && lw == 0 && lane == 0) { g_stamps[8] = c_poll; g_stamps[9] = c_issue; g_stamps[10] = c_wait; }
        }
    } else {
        // ---------------- compute waves: phase c = x gathers of chunk c + accumulator update of chunk c-1
        const int64_t w0 = (int64_t)b * items_per_block + wptr[wv];
        const int nchunk = nchp[wv];
        const uint32_t* __restrict__ bi = tidx + w0;
        const float* __restrict__ bv = tval + w0;
        const uint32_t* __restrict__ bm = meta + 2 * (size_t)cptr[wv];
        const uint32_t tr = (uint32_t)(ACC0 + 4 * (R + lane));
        uint32_t pk[D][IPL];
        float vv[D][IPL];
        uint32_t m0[D], m1[D];
        auto load = [&](int slot, int c) {               // (the stream and the meta array carry D chunks of padding)
            const int i = c * CHUNK + lane * IPL;
            if (STREAM) {
                const u32x2 a = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(bi + i));
                const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(bv + i));
                pk[slot][0] = a.x; pk[slot][1] = a.y;
                vv[slot][0] = v.x; vv[slot][1] = v.y;
            } else {
                const uint32_t z = (uint32_t)(wv * RW + ((lane * 2 + c * 13) * 9) % RW + ACC0 / 4) << 16 | (uint32_t)((lane * 613 + c * 7) % (NB * WS));
                pk[slot][0] = z; pk[slot][1] = z + (1u << 16);
                vv[slot][0] = 1.f; vv[slot][1] = 2.f;
            }
            m0[slot] = bm[2 * c];
            m1[slot] = bm[2 * c + 1];
        };
#pragma unroll
        for (int u = 0; u < D; ++u) { load(u, u); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
        uint32_t rp[IPL];
        float pp[IPL];
#pragma unroll
        for (int j = 0; j < IPL; ++j) { rp[j] = tr; pp[j] = 0.f; }
        unsigned long long c_poll = 0, c_work = 0, c0 = __builtin_readcyclecounter(), c1;
        for (int cb = 0; cb < nchunk; cb += D) {          // (nchunk is a multiple of D: the host pads with empty chunks)
#pragma unroll
            for (int u = 0; u < D; ++u) {
                if (XDMA) {
                    // the slices this chunk is the first to touch must have landed (newest first; usually 0-2 of them)
                    uint32_t slot = m0[u] & 0xffu, gen = (m0[u] >> 8) & 0xffffu;
                    const uint32_t nnew = m0[u] >> 24;
                    for (uint32_t k = 0; k < nnew; ++k) {
                        while (full[slot] < gen) __builtin_amdgcn_s_sleep(1);
                        if (slot == 0) { slot = NB - 1; --gen; } else --slot;
                    }
                }
                c1 = __builtin_readcyclecounter(); c_poll += c1 - c0; c0 = c1;
                uint32_t kk[IPL];
                float vf[IPL], xg[IPL], a[IPL];
#pragma unroll
                for (int j = 0; j < IPL; ++j) { kk[j] = pk[u][j]; vf[j] = vv[u][j]; }
                const uint32_t md = m1[u];
#pragma unroll
                for (int j = 0; j < IPL; ++j) xg[j] = GATHER ? *reinterpret_cast<const float*>(lds + ((kk[j] & 0xffffu) << 2)) : 1.0f;
                if (RMW) {
#pragma unroll
                    for (int j = 0; j < IPL; ++j) a[j] = *reinterpret_cast<const float*>(lds + rp[j]);
                }
                load(u, cb + u + D);
                float p[IPL];
                uint32_t ra[IPL];
#pragma unroll
                for (int j = 0; j < IPL; ++j) {
                    p[j] = vf[j] * xg[j];
                    ra[j] = (kk[j] >> 14) & 0x3fffcu;
                }
                {   // equal rows are neighbours (runs <= IPL): the run's last item carries the sum, the others go to the trash slot
                    const bool e1 = ra[1] == ra[0];
                    p[1] = e1 ? p[0] + p[1] : p[1];
                    const uint32_t pra = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)ra[1], 0x138, 0xf, 0xf, false);     // wave_shr:1
                    const float ptl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p[1]), 0x138, 0xf, 0xf, false));
                    const uint32_t nra0 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)ra[0], 0x130, 0xf, 0xf, false);   // wave_shl:1
                    const float c = pra == ra[0] ? ptl : 0.f;
                    p[0] += c;
                    p[1] += e1 ? c : 0.f;
                    ra[0] = e1 ? tr : ra[0];
                    ra[1] = nra0 == ra[1] ? tr : ra[1];
                }
                if (RMW) {
#pragma unroll
                    for (int j = 0; j < IPL; ++j) *reinterpret_cast<float*>(lds + rp[j]) = a[j] + pp[j];
                } else if (p[0] + p[1] == 1.2345f) *reinterpret_cast<float*>(lds + ra[0]) = 1.f;
#pragma unroll
                for (int j = 0; j < IPL; ++j) { rp[j] = ra[j]; pp[j] = p[j]; }
                if (XDMA) {
                    // the slices this wave has left behind (their last gather was issued above; LDS executes a wave's operations in order)
                    uint32_t slot = md & 0xffu;
                    const uint32_t nd = md >> 8;
                    for (uint32_t k = 0; k < nd; ++k) {
                        if (lane == 0) __hip_atomic_fetch_add(const_cast<uint32_t*>(done + slot), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        slot = slot + 1 == NB ? 0 : slot + 1;
                    }
                }
                c1 = __builtin_readcyclecounter(); c_work += c1 - c0; c0 = c1;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (RMW) {
#pragma unroll
            for (int j = 0; j < IPL; ++j) *reinterpret_cast<float*>(lds + rp[j]) += pp[j];
        }
        if (b == 100 && (wv == 0 || wv == NCW - 1) && lane == 0) { g_stamps[wv == 0 ? 0 : 2] = c_poll; g_stamps[wv == 0 ? 1 : 3] = c_work; g_stamps[wv == 0 ? 4 : 5] = nchunk; }
    }
    __syncthreads();
    for (int i = tid; i < R; i += (NCW + NLW) * 64) y[(size_t)b * R + i] = *reinterpret_cast<const float*>(lds + ACC0 + 4 * i);
}

struct Stream {
    std::vector<uint32_t> idx, meta;
    std::vector<float> val;
    std::vector<int32_t> wptr, cptr, nch;
    int nslices;
};

static Stream make_stream(int n, double density, unsigned seed, std::vector<double>& ref, const std::vector<float>& hx)
{
    Stream s;
    std::mt19937 rng(seed);
    s.nslices = (n + WS - 1) / WS;
    std::poisson_distribution<int> pois(density * RW * WS);
    ref.assign(R, 0.0);
    s.wptr.push_back(0);
    s.cptr.push_back(0);
    struct It { int slice, row, col; float v; };
    for (int w = 0; w < NCW; ++w) {
        std::vector<It> items;
        for (int sl = 0; sl < s.nslices; ++sl) {
            const int wcols = std::min(WS, n - sl * WS);
            int cnt = (int)(pois(rng) * (double)wcols / WS);
            std::vector<It> t(cnt);
            for (auto& it : t) { it.slice = sl; it.row = w * RW + rng() % RW; it.col = rng() % wcols; it.v = (float)((int)(rng() % 2001) - 1000) / 1000.f; }
            std::sort(t.begin(), t.end(), [](const It& a, const It& b) { return a.row < b.row; });
            items.insert(items.end(), t.begin(), t.end());
        }
        for (auto& it : items) ref[it.row] += (double)it.v * hx[(size_t)it.slice * WS + it.col];
        int nch = (int)((items.size() + CHUNK - 1) / CHUNK);
        nch = (nch + D - 1) / D * D;
        std::vector<int> lo(nch + 1, s.nslices), hi(nch, 0);
        for (int c = 0; c < nch; ++c) {
            const size_t a = (size_t)c * CHUNK, e = std::min(items.size(), a + CHUNK);
            if (a < items.size()) {
                std::sort(items.begin() + a, items.begin() + e, [](const It& x, const It& y) { return x.row != y.row ? x.row < y.row : x.slice < y.slice; });
                // runs <= IPL: bump later duplicates to a free neighbouring row of the same wave (prototype only: changes the matrix, so fix ref)
                for (size_t j = a + IPL; j < e; ++j)
                    if (items[j].row == items[j - IPL].row) {
                        ref[items[j].row] -= (double)items[j].v * hx[(size_t)items[j].slice * WS + items[j].col];
                        int nr = items[j].row + 1;
                        if (nr >= (w + 1) * RW) nr = items[j].row;      // (would need a real fix in the builder; harmless here if it happens at the end)
                        items[j].row = nr;
                        ref[nr] += (double)items[j].v * hx[(size_t)items[j].slice * WS + items[j].col];
                    }
                std::stable_sort(items.begin() + a, items.begin() + e, [](const It& x, const It& y) { return x.row < y.row; });
                int l = s.nslices, h = 0;
                for (size_t j = a; j < e; ++j) { l = std::min(l, items[j].slice); h = std::max(h, items[j].slice); }
                lo[c] = l; hi[c] = h;
            } else { lo[c] = s.nslices; hi[c] = s.nslices - 1; }
        }
        // lo must be non-decreasing towards the end for the DONE accounting: slices below min(lo[c..]) are finished after chunk c-1
        for (int c = nch - 1; c >= 0; --c) lo[c] = std::min(lo[c], lo[c + 1]);
        lo[0] = 0;
        int prev_hi = -1;
        for (int c = 0; c < nch; ++c) {
            hi[c] = std::max(hi[c], prev_hi);
            const int nnew = hi[c] - prev_hi;
            prev_hi = hi[c];
            if (nnew > 255) { printf("nnew overflow\n"); exit(1); }
            if (hi[c] - lo[c] + 1 > NB - 1) { printf("chunk span %d too wide\n", hi[c] - lo[c] + 1); exit(1); }
            const int ndone = lo[c + 1] - lo[c];                 // slices finished once chunk c is through
            if (ndone > 255) { printf("ndone overflow\n"); exit(1); }
            s.meta.push_back((uint32_t)(hi[c] % NB) | (uint32_t)(hi[c] / NB + 1) << 8 | (uint32_t)nnew << 24);
            s.meta.push_back((uint32_t)(lo[c] % NB) | (uint32_t)ndone << 8);
        }
        for (int c = 0; c < D; ++c) { s.meta.push_back(0); s.meta.push_back(0); }       // (prefetch padding: gen 0 = no wait)
        for (size_t j = 0; j < (size_t)nch * CHUNK + (size_t)D * CHUNK; ++j) {
            if (j < items.size()) {
                const It& it = items[j];
                const int slot = it.slice % NB;
                s.idx.push_back(((uint32_t)(it.row + ACC0 / 4) << 16) | (uint32_t)(slot * WS + it.col));
                s.val.push_back(it.v);
            } else {
                s.idx.push_back((uint32_t)(R + (j & 63) + ACC0 / 4) << 16);
                s.val.push_back(0.f);
            }
        }
        s.wptr.push_back((int32_t)s.idx.size());
        s.cptr.push_back((int32_t)(s.meta.size() / 2));
        s.nch.push_back(nch);
    }
    // cptr counts include the D padding chunks per wave: the kernel's nchunk must exclude them
    return s;
}

template <int FLAGS>
static double run(const char* name, const uint32_t* di, const float* dv, const uint32_t* dm, const int32_t* dw, const int32_t* dc, const int32_t* dn, int64_t ipb,
                  int nslices, const float* dx, float* dy, int nblk, double items)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    for (int w = 0; w < 2; ++w) k_ring<FLAGS><<<nblk, (NCW + NLW) * 64, LDS_BYTES>>>(di, dv, dm, dw, dc, dn, ipb, nslices, dx, dy);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) k_ring<FLAGS><<<nblk, (NCW + NLW) * 64, LDS_BYTES>>>(di, dv, dm, dw, dc, dn, ipb, nslices, dx, dy);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-40s: %.3f ms  %.1f G items/s\n", name, ms, items / ms / 1e6);
    unsigned long long st[64];
    CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
    printf("      block 100, clocks per chunk: wave 0 poll %.0f work %.0f | last wave poll %.0f work %.0f | loader 0 per slice: poll %.0f issue %.0f wait %.0f\n",
           (double)st[0] / st[4], (double)st[1] / st[4], (double)st[2] / st[5], (double)st[3] / st[5], (double)st[8] / (nslices / NLW), (double)st[9] / (nslices / NLW),
           (double)st[10] / (nslices / NLW));
    fflush(stdout);
    return ms;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10000000;
    const int nblk = argc > 2 ? atoi(argv[2]) : 501;
    std::vector<float> hx((size_t)n + 2 * WS);
    std::mt19937 rng(99);
    for (auto& v : hx) v = (float)((int)(rng() % 2001) - 1000) / 500.f;
    std::vector<double> ref;
    Stream s = make_stream(n, 1e-5, 1234, ref, hx);
    const int64_t ipb = (int64_t)s.idx.size();
    // the kernel's chunk counts exclude the D padding chunks
    std::vector<int32_t> cp(s.cptr);
    printf("n %d  blocks %d  slices %d x %d cols  ring %d  waves %d+%d  items/block %lld  chunks/wave ~%d  LDS %d B\n", n, nblk, s.nslices, WS, NB, NCW, NLW, (long long)ipb,
           s.nch[0], LDS_BYTES);
    uint32_t *di, *dm;
    float *dv, *dx, *dy;
    int32_t *dw, *dc, *dn;
    CHECK(hipMalloc(&di, ipb * 4 * nblk + 4096));
    CHECK(hipMalloc(&dv, ipb * 4 * nblk + 4096));
    CHECK(hipMalloc(&dm, s.meta.size() * 4 + 4096));
    CHECK(hipMalloc(&dw, s.wptr.size() * 4));
    CHECK(hipMalloc(&dc, 2 * (NCW + 1) * 4));
    CHECK(hipMalloc(&dx, hx.size() * 4));
    CHECK(hipMalloc(&dy, (size_t)nblk * R * 4));
    CHECK(hipMemcpy(di, s.idx.data(), ipb * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dv, s.val.data(), ipb * 4, hipMemcpyHostToDevice));
    for (int b = 1; b < nblk; ++b) {
        CHECK(hipMemcpyAsync(di + (size_t)b * ipb, di, ipb * 4, hipMemcpyDeviceToDevice));
        CHECK(hipMemcpyAsync(dv + (size_t)b * ipb, dv, ipb * 4, hipMemcpyDeviceToDevice));
    }
    CHECK(hipMemcpy(dm, s.meta.data(), s.meta.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dw, s.wptr.data(), s.wptr.size() * 4, hipMemcpyHostToDevice));
    // the kernel reads nchunk = cptr[w+1] - cptr[w] and the meta base cptr[w]: give it a second array with the padding removed from the counts
    {
        // layout trick: kernel uses bm = meta + 2*cptr[w] and nchunk = cptr[w+1]-cptr[w]; to exclude the padding pass adjusted ends via a
        // per-wave pair array is more code than it is worth here: the padding chunks are EMPTY (trash items, gen 0, ndone 0), so running
        // them is harmless; but then the prefetch of the last D real chunks would read past the padding.  Give every wave 2*D of padding.
    }
    CHECK(hipMemcpy(dc, cp.data(), cp.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dn, NCW * 4));
    CHECK(hipMemcpy(dn, s.nch.data(), NCW * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipDeviceSynchronize());
    const double items = (double)ipb * nblk;
    auto check = [&](const char* what) {
        std::vector<float> hy((size_t)nblk * R);
        CHECK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int b : {0, nblk / 2, nblk - 1})
            for (int r = 0; r < R; ++r) worst = std::max(worst, std::fabs(hy[(size_t)b * R + r] - ref[r]) / (1.0 + std::fabs(ref[r])));
        printf("   check %s: worst relative error %.3g %s\n", what, worst, worst < 1e-4 ? "OK" : "MISMATCH");
    };
    run<1 | 2 | 4 | 8>("full", di, dv, dm, dw, dc, dn, ipb, s.nslices, dx, dy, nblk, items);
    check("full");
    run<2 | 4 | 8>("no x dma / ring", di, dv, dm, dw, dc, dn, ipb, s.nslices, dx, dy, nblk, items);
    run<1 | 4 | 8>("no stream", di, dv, dm, dw, dc, dn, ipb, s.nslices, dx, dy, nblk, items);
    run<4 | 8>("LDS work only", di, dv, dm, dw, dc, dn, ipb, s.nslices, dx, dy, nblk, items);
    run<2>("stream only", di, dv, dm, dw, dc, dn, ipb, s.nslices, dx, dy, nblk, items);
    run<1 | 2>("dma + ring + stream", di, dv, dm, dw, dc, dn, ipb, s.nslices, dx, dy, nblk, items);
    return 0;
}
