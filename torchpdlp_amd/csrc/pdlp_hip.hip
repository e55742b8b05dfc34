// pdlp_hip.hip -- restarted PDHG for LP on AMD Instinct MI355X (gfx950, wave64), C ABI in include/pdlp_hip.h.
//
// Design (DESIGN.md has the long form):
//   * K is stored twice (CSR of K, CSR of K'); both products are row-parallel SpMVs, no atomics.
//   * One kernel per half-iteration: a CSR "stream" SpMV whose epilogue does the projection, the
//     extrapolation / dual ascent, the eta-weighted running sums and the step-size-rule partial sums.
//     Row blocks (<= 256 rows, <= 2048 non-zeros, built once on the host) are processed by 256-thread
//     workgroups: the block's non-zeros are read fully coalesced, multiplied with the gathered vector
//     entries and staged in LDS, then 1..64 lanes per row reduce their segment (wave shuffles), and the
//     lane holding the row sum runs the epilogue.  Rows longer than 2048 get a whole workgroup.
//   * The launch is a fixed grid of <= 2048 workgroups striding over the row blocks, so the norm
//     partial sums are 2048 x 4 doubles and their final reduction is deterministic.
//   * All step-size state (eta, omega, tau, sigma, pending average weight, iteration count) lives in
//     device memory, so a whole restart period runs without a host synchronisation.
// This bandwidth-bound path uses no MFMA.  Written for gfx950 only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is resolved with dlopen when a communicator is asked for
#include <dlfcn.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "pdlp_hip.h"

namespace {

constexpr int BLOCK = 256;       // threads per workgroup (4 wave64)
constexpr int NNZ_CAP = 2048;    // non-zeros staged in LDS per row block
constexpr int ROWS_CAP = 256;    // rows per row block (one epilogue lane each at most)
constexpr int MAX_GRID = 2048;   // 256 CUs x 8 workgroups
constexpr int NACC = 4;          // partial sums a kernel may produce
constexpr int MAX_CHUNKS = 4;    // chunks of the exchange of a gathered vector (sharded problems)
constexpr int MAX_PHASE = MAX_CHUNKS + 2;
constexpr int MAX_PEER = 7;      // direct exchange (pdlp_peer_*): the other ranks of one node
constexpr int PEER_WAIT_BLOCKS = 8; // one waiting wave per XCD (k_peer_wait)
// a rank's mailbox for the direct exchange (fine-grained device memory, opened by every peer): the sequence number rank q last
// signalled, one per 64-byte line, then rank q's three sums of the step-size rule
constexpr int BOX_BYTES = 4096, BOX_FLAG_STRIDE = 16 /* uint32 */, BOX_SUMS_AT = 1024 /* bytes */, BOX_SUMS_STRIDE = 4 /* doubles */;

// indices into the device scalar block (double[PDLP_NSCAL])
enum { S_ETA = 0, S_OMEGA, S_THETA, S_TAU, S_SIGMA, S_WPEND, S_ETASUM, S_K, S_INV1PT, S_ACCEPT, S_ETABAR, S_DEN, S_ETASUM_PREV };

#define HIP_TRY(expr)                                                   \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) return PDLP_ERR_HIP_BASE - (int)e_;       \
    } while (0)

template <typename T> __device__ __forceinline__ T shfl_xor_t(T v, int m) { return __shfl_xor(v, m, 64); }

// sum of `v` over the 256-thread workgroup, valid in thread 0.  `buf` holds >= 4 entries.
// inclusive scan over the 64 lanes of a wave with DPP adds only (no LDS crossbar): Hillis-Steele inside each row of
// 16 lanes, then row_bcast:15 / row_bcast:31 carry the row totals.  All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return x;
}

// two independent inclusive scans at once, written out: the compiler fuses the DPP move into the add for one scan of such
// a pair but not reliably for the other (18 instead of 6 VALU instructions).  A DPP read of a VGPR needs 2 wait states after
// the VALU write of it: the other scan's instruction and one s_nop provide them.
__device__ __forceinline__ void wave_incl_scan2_u32(uint32_t& a, uint32_t& b)
{
#define PDLP_SCAN_STEP(ctl) "v_add_u32_dpp %0, %0, %0 " ctl "\n\tv_add_u32_dpp %1, %1, %1 " ctl "\n\ts_nop 0\n\t"
    asm("s_nop 1\n\t"
        PDLP_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        PDLP_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        PDLP_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        PDLP_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        PDLP_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        PDLP_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(a), "+v"(b));
#undef PDLP_SCAN_STEP
}

template <typename T> __device__ __forceinline__ T block_sum(T v, T* buf)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += shfl_xor_t(v, off);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) buf[w] = v;
    __syncthreads();
    return buf[0] + buf[1] + buf[2] + buf[3];
}

#include "pdlp_epilogues.inc"
#include "pdlp_kernel_csr.inc"
#include "pdlp_kernel_tiled.inc"
#include "pdlp_kernels_small.inc"
#include "pdlp_kernel_mv.inc"

// ------------------------------------------------------------------------------------------------
// RCCL, resolved at run time (single-GPU use never touches it)
// ------------------------------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;            // optional (chunked exchange): grouped in-place broadcasts
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    int last_error = 0;
};
Rccl g_rccl;

int rccl_load(const char* path)
{
    if (g_rccl.lib) return PDLP_OK;
    const char* names[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return PDLP_ERR_COMM;
    Rccl r;
    r.lib = lib;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(lib, "ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))dlsym(lib, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
    r.Broadcast = (decltype(r.Broadcast))dlsym(lib, "ncclBroadcast");
    r.GroupStart = (decltype(r.GroupStart))dlsym(lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(lib, "ncclGroupEnd");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce) return PDLP_ERR_COMM;
    g_rccl = r;
    return PDLP_OK;
}

#define RCCL_TRY(expr)                                       \
    do {                                                     \
        ncclResult_t r_ = (expr);                            \
        if (r_ != ncclSuccess) { g_rccl.last_error = (int)r_; return PDLP_ERR_COMM; } \
    } while (0)

// ------------------------------------------------------------------------------------------------
// roctx ranges (rocprofv3 --marker-trace): the counterpart of the reference's Timer (PDLP/util.py:6-27, wall-clock sections printed at
// the end of a run).  Resolved with dlopen on first use after pdlp_trace_enable -- no link-time dependency, nothing happens when
// tracing is off.  Level 2 also synchronises the given stream at both ends of a range, so that the range's wall time IS the
// GPU time of what was enqueued inside it (the per-phase table of profiles/README.md); level 1 marks the host side only.
// ------------------------------------------------------------------------------------------------
struct Roctx {
    int level = 0;
    bool tried = false;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
Roctx g_roctx;

void roctx_load()
{
    if (g_roctx.tried) return;
    g_roctx.tried = true;
    const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                           "libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"};
    for (const char* nm : names) {
        void* lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) continue;
        g_roctx.push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
        g_roctx.pop = (int (*)())dlsym(lib, "roctxRangePop");
        if (g_roctx.push && g_roctx.pop) return;
        g_roctx.push = nullptr; g_roctx.pop = nullptr;
    }
}

struct Range {         // scoped range on the handle's stream
    hipStream_t s;
    bool on;
    Range(const char* name, hipStream_t stream) : s(stream), on(g_roctx.level > 0 && g_roctx.push)
    {
        if (!on) return;
        if (g_roctx.level > 1) (void)hipStreamSynchronize(s);
        (void)g_roctx.push(name);
    }
    ~Range()
    {
        if (!on) return;
        if (g_roctx.level > 1) (void)hipStreamSynchronize(s);
        (void)g_roctx.pop();
    }
};

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct Schedule {
    int64_t* blk = nullptr;   // device, nblk+1 pairs (first row, first non-zero)
    uint32_t* rplo = nullptr; // device, rows+1: the LOW 32 bits of the row pointers.  Inside a row block the kernel only needs offsets relative
                              // to the block's first non-zero (< 2^32 apart), so it reads 4 bytes per row instead of 8: (uint32)(rp[r]) - (uint32)a
    int nblk = 0;
    int grid = 0;
    // rows longer than NNZ_CAP, cut into chunks of NNZ_CAP non-zeros
    int64_t* lch = nullptr;   // device, [2*nchunks] (first, end) non-zero of every chunk
    int32_t* lrow = nullptr;  // device, [nlong] the rows
    int32_t* lptr = nullptr;  // device, [nlong+1] their chunk ranges
    void* longpart = nullptr; // device, [nchunks] chunk sums
    int nchunks = 0, nlong = 0, lgrid = 0;
    // column-sorted row blocks (optional, attached by the caller): the CSR kernel reads each block's items sorted by column
    const uint32_t* sidx = nullptr;
    const void* sval = nullptr;
    const int32_t* cbase = nullptr;
    // panel-tiled copy (optional, attached by the caller): used instead of the CSR arrays when set
    bool tiled = false;
    pdlp_tiles t{};
    // sharded problems: the panels lying wholly inside the locally owned block of the gathered vector, [loc_pa, loc_pb),
    // can be multiplied before the all-gather of that vector has finished (pdlp_*_half_begin)
    int loc_pa = 0, loc_pb = 0;
    int slotsA = 0, slotsB = 0;   // panel groups (= partial row sum slots) of the local and of all the other panels
    bool pending = false;         // the local panels of the next product are already in rowsum[0 .. slotsA)
    bool pending_inline = false;  // ... and were launched on the handle's own stream (PDLP_OPT_BEGIN_INLINE): nothing to join
    // the exchange of the gathered vector in `nphase - 1` chunks (pdlp_set_exchange_chunks): chunk c moves elements
    // [sb[c], sb[c+1]) of EVERY rank's block; a panel belongs to the phase with which its last foreign entry arrives
    // (phase 0: the panels of the own block, phase 1 + c: chunk c).  ptab holds the panels phase by phase.
    int32_t* ptab = nullptr;      // device, room for every panel of the matrix
    int64_t ptab_cap = 0;
    int nphase = 0;               // 0: product not split
    int ph_off[MAX_PHASE] = {0}, ph_cnt[MAX_PHASE] = {0}, ph_slots[MAX_PHASE] = {0}, ph_slot0[MAX_PHASE] = {0};
    int64_t sb[MAX_PHASE] = {0};
    int chunks_done = 0;          // chunk phases of the pending product already launched (pdlp_half_chunk)
    // the RESULT of the product travels in `nrange` pieces (the plan of the exchange that follows): piece r = the rows of the row
    // blocks [rb_lo[r], rb_lo[r+1]); the last phase and the epilogue of a split product then run piece by piece (launch_mat)
    int nrange = 0;
    int rb_lo[MAX_PHASE] = {0};
};

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
inline int grid_for(int64_t n) { int64_t g = (n + BLOCK - 1) / BLOCK; return (int)(g < 1 ? 1 : (g > MAX_GRID ? MAX_GRID : g)); }

// upper bound on the number of row blocks: two consecutive blocks together exceed a cap
inline int64_t max_blocks(int64_t rows, int64_t nnz) { return 2 * (rows / ROWS_CAP + nnz / NNZ_CAP) + 4; }

// upper bounds on the chunks / long rows of a matrix with nnz non-zeros
inline int64_t max_chunks(int64_t nnz) { return 2 * (nnz / NNZ_CAP) + 2; }
inline int64_t max_long(int64_t nnz) { return nnz / NNZ_CAP + 1; }
constexpr int LONG_GRID = 64;

void build_long_rows_host(const std::vector<int64_t>& rp, int64_t rows, std::vector<int64_t>& lch, std::vector<int32_t>& lrow,
                          std::vector<int32_t>& lptr)
{
    lch.clear(); lrow.clear(); lptr.clear();
    lptr.push_back(0);
    for (int64_t r = 0; r < rows; ++r) {
        const int64_t a = rp[r], e = rp[r + 1];
        if (e - a <= NNZ_CAP) continue;
        for (int64_t c = a; c < e; c += NNZ_CAP) {
            lch.push_back(c);
            lch.push_back(c + NNZ_CAP < e ? c + NNZ_CAP : e);
        }
        lrow.push_back((int32_t)r);
        lptr.push_back((int32_t)(lch.size() / 2));
    }
}

// out = (first row, first non-zero) of every block, then (rows, nnz) as the end marker: 2 * (blocks + 1) entries
void build_schedule_host(const std::vector<int64_t>& rp, int64_t rows, std::vector<int64_t>& out)
{
    out.clear();
    int64_t r = 0;
    out.push_back(0);
    out.push_back(0);
    while (r < rows) {
        int64_t e = r;
        int64_t nnz = 0;
        while (e < rows && e - r < ROWS_CAP) {
            const int64_t len = (int64_t)rp[e + 1] - rp[e];
            if (nnz + len > NNZ_CAP) break;
            nnz += len;
            ++e;
        }
        if (e == r) e = r + 1;   // a single row longer than NNZ_CAP: its own (skipped) block, done in chunks
        out.push_back(e);
        out.push_back(rp[e]);
        r = e;
    }
}

}  // namespace

struct pdlp_solver {
    pdlp_problem p;
    hipStream_t stream;
    size_t es;                    // element size of the vectors
    bool mixed;                   // PDLP_MIXED: float32 matrix values under float64 vectors
    // delta mode (mixed precision only): every product of the iteration runs on the float32 kernels over a float32 DIFFERENCE
    // vector and is added to a float64 "anchor" product that is carried along: kxb[0] = K x_cur, ktyr = K'y (of y_cur once
    // dy_folded, else of the previous y with gdy = y_cur - y_prev still to be folded in by the next product with K')
    bool delta, anchors_valid, dy_folded;
    ncclComm_t comm;              // RCCL communicator of a sharded problem (pdlp_comm_init), or null: the caller does the exchange
    int comm_rank, comm_size;
    int xchunks;                  // chunks of the exchange of a gathered vector (pdlp_set_exchange_chunks); 1: one all-gather
    hipStream_t cstream;          // the chunks travel on this stream while the handle's stream multiplies what has arrived
    hipEvent_t ev_vec, ev_chunk[MAX_CHUNKS];
    hipEvent_t ev_ar;               // library driver: the step-size rule's all-reduce on the communication stream has finished
    hipEvent_t ev_row[MAX_CHUNKS];  // library driver: piece r of the vector a half-step is producing is final on the handle's stream
    int range_sel, range_cnt;       // >= 0: the half-step being issued covers only output piece `range_sel` of `range_cnt` (pdlp_*_half_piece)
    char* ktyr;                   // [nl] float64 running K'y
    float *gdx, *gdy;             // full-length float32 difference vectors the float32 kernels gather from
    int64_t nl, ml;               // local variable / constraint counts
    int ineq_end;                 // local rows below this index are inequalities
    Schedule sK, sKT;
    char* xb[3];                  // full-length primal buffers; roles via ix_*
    char* yb[3];
    int ix_cur, ix_prev, ix_avg;  // (x and y rotate together)
    char* xbar;
    char *x_sum, *y_sum, *x_last, *y_last;
    char* kxb[3];                 // K x caches: [0] running, [1] from KKT(cur), [2] from KKT(avg)
    char *dxf, *dyf;              // infeasibility detection: full-length x - x_prev, y - y_prev (gathered by the caller when sharded)
    char *lam_prev, *ktdy;        //   this rank's block of the previous lambda and of K'dy
    bool kx_valid, cand_valid[2];
    char* ktyb[2];                // K'y of the candidates, kept by their KKT passes: [0] current, [1] averaged iterate
    int kty_cur;                  // which of the two belongs to the CURRENT iterate after a restart (-1: see cand_valid[0])
    bool no_kty_reuse;            // PDLP_OPT_KTY_REUSE = 0: timing experiments
    int split_local, split_other; // PDLP_OPT_SPLIT_SLOTS: panel groups of a split product chosen by the caller (0: the library's rule)
    bool side_ok;                 // the library's own streams and events exist (graph replay, split products)
    bool begin_inline;            // PDLP_OPT_BEGIN_INLINE: pdlp_*_half_begin launch on the handle's stream (the caller's exchange is asynchronous)
    bool producer_pieces;         // PDLP_OPT_PRODUCER_PIECES (default on): results of split products leave piece by piece (Schedule::nrange)
    char* ws;                     // the caller's workspace (pdlp_peer_export hands it to the other ranks)
    int64_t ws_bytes;
    // direct exchange (pdlp_peer_*): the other ranks' workspaces and mailboxes, opened over HIP IPC
    struct Peer {
        bool on = false;          // connected
        bool enabled = true;      // PDLP_OPT_PEER_EXCHANGE: pdlp_iterate uses it
        bool active = false;      // inside iterate_peer: the half-steps' epilogues store into the peers
        bool loopback = false;    // timing stand-in: the "peers" are scratch buffers of this process
        bool local_first = false; // PDLP_OPT_PEER_LOCAL_FIRST: the own block's panels are multiplied between signal and wait
        bool push = false;        // PDLP_OPT_PEER_PUSH: the block leaves by a copy kernel on the side stream, beside those panels
        int rank = 0, world = 1, n = 0;      // n = world - 1 peers
        void* opened[2 * MAX_PEER] = {};     // what hipIpcCloseMemHandle wants back
        int nopened = 0;
        char* out[6][MAX_PEER] = {};         // peer i's xbar, y buffers 0 / 1 / 2, gdx, gdy -- at THIS rank's block
        uint32_t* flag[MAX_PEER] = {};       // this rank's slot in peer i's mailbox
        double* sums[MAX_PEER] = {};
        char* box = nullptr;                 // the own mailbox (fine-grained device memory)
        char* scratch = nullptr;             // loopback: the stand-in destinations
        char* scratch_host = nullptr;        // PDLP_PEER_LOOPBACK_HOST: one of them in pinned host memory (a slow link's stand-in)
        hipStream_t pstream = nullptr;       // push form: a HIGH-priority stream -- the copy kernel must get its few waves onto the
        hipEvent_t ev_push = nullptr;        //   chip before the own-block panels' launch fills every CU's registers
        int* err = nullptr;                  // host memory the wait kernel reports a timeout through
        int* err_dev = nullptr;
        uint32_t seq = 0;
        long long limit_ticks = 1000000000LL;   // 10 s of the 100 MHz clock
    } peer;
    // running products: K x (kxb[0]) is carried along by every dual half-step and both products are summed with the
    // average's weights (kx_sum, kty_sum), so a restart check evaluates K x_cur, K x_avg and K'y_avg WITHOUT products:
    // one product (K'y_cur, kept for the next primal half-step) instead of four per check
    char *kx_sum, *kty_sum;
    int64_t since_reset;          // iterations since the sums were last zeroed (set_iterate / restart)
    bool kty_tail_done;           // kty_sum already holds the term of the current y (added by the flush at a restart check)
    bool sums_broken;             // a term was lost (flush before the K'y of the current iterate existed): no running average
    bool avg_products;            // kxb[2] / ktyb[1] hold K x_avg / K'y_avg computed from the sums
    bool cur_kx_cached;           // the KKT pass of the current iterate took K x from the cache (nothing to swap on restart)
    bool no_running;              // PDLP_OPT_RUNNING_KKT = 0: every KKT pass multiplies (round-1 behaviour)
    double *partA, *partB, *red, *sc;
    void* rowsum;                 // row sums of the tiled kernel on their way to the epilogue: [groups][rs_stride]
    int64_t rs_stride;            // rows + one row block
    int rs_groups;                // panel groups the scratch has room for
    int64_t part_blocks;          // workgroups partA / partB have room for
    int last_gridA, last_gridB;   // grids of the last primal / dual launch (adaptive reduce)
    bool use_split;               // set by the half-step that may consume a pending local-panel product
    // optional (PDLP_GRAPH=1): pdlp_iterate replays two captured iterations (the buffer roles return after two) as one
    // hipGraph launch.  Captured on and replayed from the library's own stream (capture is not allowed on the
    // legacy null stream), ordered against the caller's stream with events.  One graph per (roles, mode).
    hipStream_t gstream;
    hipEvent_t ev_in, ev_out;
    bool graph_ok;
    struct IterGraph { bool valid; int ix_cur, ix_prev, adaptive; hipGraphExec_t exec; } graphs[12];
};

namespace {

void drop_graphs(pdlp_handle h)
{
    for (auto& g : h->graphs) {
        if (g.valid) (void)hipGraphExecDestroy(g.exec);
        g.valid = false;
    }
}

// the direct exchange's mappings and allocations (the peers' memory is only unmapped here, never freed)
void peer_release(pdlp_handle h)
{
    pdlp_solver::Peer& P = h->peer;
    for (int i = 0; i < P.nopened; ++i) if (P.opened[i]) (void)hipIpcCloseMemHandle(P.opened[i]);
    if (P.box) (void)hipFree(P.box);
    if (P.scratch) (void)hipFree(P.scratch);
    if (P.scratch_host) (void)hipHostFree(P.scratch_host);
    if (P.pstream) { (void)hipStreamSynchronize(P.pstream); (void)hipStreamDestroy(P.pstream); }
    if (P.ev_push) (void)hipEventDestroy(P.ev_push);
    if (P.err) (void)hipHostFree(P.err);
    (void)hipGetLastError();
    P = pdlp_solver::Peer();
}

void free_handle(pdlp_handle h)
{
    drop_graphs(h);
    if (h->comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
    if (h->gstream) { (void)hipStreamSynchronize(h->gstream); (void)hipStreamDestroy(h->gstream); }
    if (h->cstream) { (void)hipStreamSynchronize(h->cstream); (void)hipStreamDestroy(h->cstream); }
    if (h->ev_vec) (void)hipEventDestroy(h->ev_vec);
    for (auto& e : h->ev_chunk) if (e) (void)hipEventDestroy(e);
    for (auto& e : h->ev_row) if (e) (void)hipEventDestroy(e);
    if (h->ev_ar) (void)hipEventDestroy(h->ev_ar);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->ev_out) (void)hipEventDestroy(h->ev_out);
    peer_release(h);
    delete h;
}

// partial row sums of `vgroups` panel groups over `vtotal` panels in virtual order (see k_tiled_fused) into
// rowsum[slot0 .. slot0 + vgroups); the epilogue functor is not used by these launches
template <typename T, typename TV>
void launch_tiled_groups(pdlp_handle h, const Schedule& s, int rows, const void* vin, hipStream_t stream, int vgroups, int vtotal,
                         const int32_t* ptab, int slot0, int b0 = 0, int nbl = -1)
{
    if (nbl < 0) nbl = s.t.nblk - b0;                     // (default: every row block)
    if (vtotal <= 0 || vgroups <= 0 || nbl <= 0) return;
    const int groups = vgroups < vtotal ? vgroups : vtotal;       // (the kernel splits the panels evenly: no group without panels)
    StoreEpi<T> none{nullptr};
    hipLaunchKernelGGL((k_tiled_fused<T, TV, StoreEpi<T>, false>), dim3(nbl * groups), dim3(TNT), 0, stream, s.t.idx,
                       (const TV*)s.t.val, s.t.tile_ptr, s.t.blk_base, s.t.cnt, s.t.npanel, s.t.lw, s.t.rpt, rows, s.t.nblk, vtotal, ptab,
                       slot0, b0, nbl, (const T*)vin, (T*)h->rowsum, h->rs_stride, (const T*)nullptr, none, (double*)nullptr);
}

// one phase of a split product (0: the own block's panels, 1 + c: the panels completed by chunk c of the exchange), over all row
// blocks or over the row blocks [b0, b0 + nbl) of one output piece (the last phase of a product whose result travels in pieces)
template <typename T, typename TV>
void launch_phase(pdlp_handle h, const Schedule& s, int rows, const void* vin, hipStream_t stream, int phase, int b0 = 0, int nbl = -1)
{
    launch_tiled_groups<T, TV>(h, s, rows, vin, stream, s.ph_slots[phase], s.ph_cnt[phase], s.ptab + s.ph_off[phase], s.ph_slot0[phase], b0, nbl);
}

// Output pieces of a split product (Schedule::nrange > 1): rows of piece r = row blocks [rb_lo[r], rb_lo[r+1]).  The k_rowsum_epilogue
// launches of the pieces write their partial sums one after the other: piece r's first slot, and the total.
inline int range_rows_lo(const Schedule& s, int r) { return s.rb_lo[r] * TNT * s.t.rpt; }
inline int range_rows_hi(const Schedule& s, int r, int rows) { const int64_t e = (int64_t)s.rb_lo[r + 1] * TNT * s.t.rpt; return (int)(e < rows ? e : rows); }
inline int range_epi_grid(const Schedule& s, int r, int rows)
{
    const int n = range_rows_hi(s, r, rows) - range_rows_lo(s, r);
    return n > 0 ? grid_for(n) : 0;
}
inline int split_epi_grid(const Schedule& s, int rows)
{
    if (s.nrange <= 1) return grid_for(rows);
    int g = 0;
    for (int r = 0; r < s.nrange; ++r) g += range_epi_grid(s, r, rows);
    return g;
}

// one product with K (or K') over the vector vin with the epilogue fused: T = type of vin, of the row sums and of what the
// epilogue receives, TV = type of the stored matrix values
template <typename T, typename TV, class Epi>
int launch_mat(pdlp_handle h, bool transpose, const void* vin, Epi epi, double* partials)
{
    if (!h->use_split && (h->sK.pending || h->sKT.pending)) {
        // an early local-panel product that nobody is going to consume (the caller changed course): let it finish
        // before the row-sum scratch is reused
        if (!(h->sK.pending ? h->sK.pending_inline : h->sKT.pending_inline)) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_out, 0));
        h->sK.pending = h->sKT.pending = false;
        h->sK.chunks_done = h->sKT.chunks_done = 0;
    }
    const Schedule& s = transpose ? h->sKT : h->sK;
    if (s.nblk == 0) return PDLP_OK;
    if (s.tiled) {
        const int rows = (int)(transpose ? h->nl : h->ml);
        // the remainder (items the tile format could not hold) first: its row sums go to a dense vector the epilogue adds
        const T* extra = nullptr;
        if (s.t.rem_rows_n > 0 && h->range_sel > 0) {
            extra = (const T*)(sizeof(T) == 4 && h->es == 8 ? s.t.rem_extra_f32 : s.t.rem_extra);     // (computed with piece 0 of this half-step)
        } else if (s.t.rem_rows_n > 0) {
            T* ex = (T*)(sizeof(T) == 4 && h->es == 8 ? s.t.rem_extra_f32 : s.t.rem_extra);
            hipLaunchKernelGGL((k_rem_segments<T, TV>), dim3(grid_for((int64_t)s.t.rem_segs_n * 8)), dim3(BLOCK), 0, h->stream, s.t.rem_segs_n,
                               s.t.rem_sptr, s.t.rem_col, (const TV*)s.t.rem_val, (const T*)vin, (T*)s.t.rem_work);
            hipLaunchKernelGGL((k_rem_rows<T>), dim3(grid_for((int64_t)s.t.rem_rows_n * 8)), dim3(BLOCK), 0, h->stream, s.t.rem_rows_n, s.t.rem_rows,
                               s.t.rem_rptr, (const T*)s.t.rem_work, ex);
            extra = ex;
        }
        if (s.t.groups == 1 && !(s.pending && h->use_split)) {
            hipLaunchKernelGGL((k_tiled_fused<T, TV, Epi, true>), dim3(s.t.nblk), dim3(TNT), 0, h->stream, s.t.idx, (const TV*)s.t.val,
                               s.t.tile_ptr, s.t.blk_base, s.t.cnt, s.t.npanel, s.t.lw, s.t.rpt, rows, s.t.nblk, s.t.npanel,
                               (const int32_t*)nullptr, 0, 0, s.t.nblk, (const T*)vin, (T*)h->rowsum, h->rs_stride, extra, epi, partials);
        } else if (s.pending && h->use_split) {
            // the local panels were multiplied by pdlp_*_half_begin on the side stream (and the first chunks' panels by
            // pdlp_half_chunk as they arrived); now the remaining chunks' panels, then the sum over all slots in fixed order.
            // If the RESULT travels in pieces (nrange > 1: the next exchange is chunked), the last phase and the epilogue run piece
            // by piece -- the row blocks of piece 0, its epilogue, an event; then piece 1 ... -- so that a piece's collective can
            // start while the rows of the later pieces are still being multiplied.  h->range_sel >= 0: only that piece (the caller
            // issues the piece's collective after every call), else all of them.
            Schedule& sm = transpose ? h->sKT : h->sK;
            const int R = s.nrange > 1 ? s.nrange : 1, last = s.nphase - 1;
            const int r_from = h->range_sel < 0 ? 0 : h->range_sel, r_to = h->range_sel < 0 ? R : h->range_sel + 1;
            if (r_from == 0) {
                for (int ph = 1 + sm.chunks_done; ph < (R > 1 ? last : s.nphase); ++ph) launch_phase<T, TV>(h, s, rows, vin, h->stream, ph);
                sm.chunks_done = 0;
            }
            int pofs = 0;
            for (int r = 0; r < r_from && R > 1; ++r) pofs += range_epi_grid(s, r, rows);
            for (int r = r_from; r < r_to && r < R; ++r) {
                if (R > 1) launch_phase<T, TV>(h, s, rows, vin, h->stream, last, s.rb_lo[r], s.rb_lo[r + 1] - s.rb_lo[r]);
                if (r == 0 && !s.pending_inline) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_out, 0));
                const int lo = R > 1 ? range_rows_lo(s, r) : 0, hi = R > 1 ? range_rows_hi(s, r, rows) : rows;
                if (hi > lo)
                    hipLaunchKernelGGL((k_rowsum_epilogue<T, Epi>), dim3(grid_for(hi - lo)), dim3(BLOCK), 0, h->stream, (const T*)h->rowsum,
                                       s.slotsA + s.slotsB, h->rs_stride, hi, extra, epi, partials + (size_t)pofs * NACC, lo);
                pofs += R > 1 ? range_epi_grid(s, r, rows) : 0;
            }
        } else {
            launch_tiled_groups<T, TV>(h, s, rows, vin, h->stream, s.t.groups, s.t.npanel, (const int32_t*)nullptr, 0);
            hipLaunchKernelGGL((k_rowsum_epilogue<T, Epi>), dim3(grid_for(rows)), dim3(BLOCK), 0, h->stream, (const T*)h->rowsum,
                               s.t.groups, h->rs_stride, rows, extra, epi, partials);
        }
        HIP_TRY(hipGetLastError());
        return PDLP_OK;
    }
    const uint32_t* rp = s.rplo;          // (low words of the row pointers: csr_pass needs block-relative offsets only)
    const int32_t* ci = transpose ? h->p.KT_colidx : h->p.K_colidx;
    const TV* va = (const TV*)(transpose ? h->p.KT_val : h->p.K_val);
    if (s.sidx)
        hipLaunchKernelGGL((k_csr_fused<T, TV, Epi, true>), dim3(s.grid), dim3(BLOCK), 0, h->stream, s.blk, s.nblk, s.lch, s.nchunks,
                           (T*)s.longpart, rp, ci, va, s.sidx, (const TV*)s.sval, s.cbase, (const T*)vin, epi, partials);
    else
        hipLaunchKernelGGL((k_csr_fused<T, TV, Epi, false>), dim3(s.grid), dim3(BLOCK), 0, h->stream, s.blk, s.nblk, s.lch, s.nchunks,
                           (T*)s.longpart, rp, ci, va, (const uint32_t*)nullptr, (const TV*)nullptr, (const int32_t*)nullptr,
                           (const T*)vin, epi, partials);
    if (s.nlong > 0)
        hipLaunchKernelGGL((k_long_rows<T, Epi>), dim3(s.lgrid), dim3(BLOCK), 0, h->stream, s.lrow, s.lptr, s.nlong,
                           (const T*)s.longpart, epi, partials + (size_t)s.grid * NACC);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

// the product in the handle's working precision T (the matrix is float32 under float64 vectors in mixed precision)
template <typename T, class Epi>
int launch_csr(pdlp_handle h, bool transpose, const void* vin, Epi epi, double* partials)
{
    if constexpr (std::is_same<T, double>::value) {
        if (h->mixed) return launch_mat<double, float, Epi>(h, transpose, vin, epi, partials);
    }
    return launch_mat<T, T, Epi>(h, transpose, vin, epi, partials);
}

inline int grid_of(const Schedule& s, int64_t rows)
{
    if (s.nblk == 0) return 0;
    if (!s.tiled) return s.grid + (s.nlong > 0 ? s.lgrid : 0);
    if (s.t.groups == 1 && !s.pending) return s.t.nblk;
    return s.pending ? split_epi_grid(s, (int)rows) : grid_for(rows);      // split tiles: the partial sums come from k_rowsum_epilogue
}

// A half-step issued piece by piece (pdlp_*_half_piece: h->range_sel = the piece, h->range_cnt = their number).  Only a split product
// whose result travels in pieces really runs piece by piece (launch_mat); every other form of the half-step does all its work with
// piece 0 and nothing afterwards.  The state changes that end a half-step (buffer roles, counters) wait for the last piece.
struct PieceCtl { bool skip, finish; };
inline PieceCtl piece_ctl(pdlp_handle h, const Schedule& s, bool product_is_launched = true)
{
    const bool piece_mode = h->range_sel >= 0;
    const bool capable = product_is_launched && s.tiled && s.pending && s.nrange > 1;
    return PieceCtl{piece_mode && !capable && h->range_sel > 0, !piece_mode || h->range_sel >= h->range_cnt - 1};
}

// direct exchange: where the other ranks keep vector `v` (0 xbar, 1 / 2 / 3 the y buffers, 4 gdx, 5 gdy), at this rank's block
// (the half-steps pick the epilogue instantiation WITH the table only while h->peer.active: iterate_peer)
template <typename T> void peer_targets(pdlp_handle h, PeerOut<T, true>& po, int v)
{
    for (int i = 0; i < h->peer.n; ++i) po.p[i] = (T*)h->peer.out[v][i];
    po.n = h->peer.n;
}
template <typename T> void peer_targets(pdlp_handle, PeerOut<T, false>&, int) {}

template <typename T> T* xloc(pdlp_handle h, int ix) { return (T*)h->xb[ix] + h->p.col0; }
template <typename T> T* yloc(pdlp_handle h, int ix) { return (T*)h->yb[ix] + h->p.row0; }

// the primal update from a K'y that a KKT pass at this very iterate left behind: no product, one vector kernel
template <typename T, class Epi> int primal_from_kty(pdlp_handle h, int src, Epi e)
{
    if (h->nl == 0) return PDLP_OK;
    hipLaunchKernelGGL((k_rowsum_epilogue<T, Epi>), dim3(grid_for(h->nl)), dim3(BLOCK), 0, h->stream, (const T*)h->ktyb[src], 1,
                       (int64_t)0, (int)h->nl, (const T*)nullptr, e, h->partA);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

template <typename T, bool ADAPT, bool PEER> int primal_half_e(pdlp_handle h, int src, T* ksum, const PieceCtl& pc)
{
    PrimalEpi<T, ADAPT, PEER> e{xloc<T>(h, h->ix_cur), xloc<T>(h, h->ix_prev), (T*)h->xbar + h->p.col0, (const T*)h->p.c,
                                (const T*)h->p.l, (const T*)h->p.u, (T*)h->x_sum, h->sc, ksum};
    peer_targets(h, e.peer, 0);
    if (src >= 0) {
        if (ADAPT) h->last_gridA = h->nl > 0 ? grid_for(h->nl) : 0;
        return primal_from_kty<T>(h, src, e);
    }
    if (ADAPT) h->last_gridA = grid_of(h->sKT, h->nl);
    h->use_split = true;
    const int rc = launch_csr<T>(h, true, h->yb[h->ix_cur], e, h->partA);
    h->use_split = false;
    if (pc.finish) h->sKT.pending = false;
    return rc;
}

template <typename T> int primal_half_t(pdlp_handle h, int adaptive)
{
    // K'y of the current iterate may still be there from the restart check (of the current iterate if nothing moved
    // since, or of the candidate the restart adopted)
    const int src = (h->no_kty_reuse || h->graph_ok || h->sKT.pending) ? -1 : (h->kty_cur >= 0 ? h->kty_cur : (h->cand_valid[0] ? 0 : -1));
    // K'y of the previous iterate's y joins the running sum unless this is the first half-step after a reset (that y is the
    // restart point) or the restart check's flush has already added it
    // (not under graph replay: a captured launch would freeze this decision)
    T* ksum = (h->since_reset > 0 && !h->kty_tail_done && !h->sums_broken && !h->no_running && !h->graph_ok) ? (T*)h->kty_sum : nullptr;
    const PieceCtl pc = piece_ctl(h, h->sKT, src < 0);
    if (pc.skip) return PDLP_OK;                             // (all of this half-step went out with piece 0)
    // (inside the direct exchange the epilogue also stores xbar into the peers: its own instantiations)
    if (h->peer.active) return adaptive ? primal_half_e<T, true, true>(h, src, ksum, pc) : primal_half_e<T, false, true>(h, src, ksum, pc);
    return adaptive ? primal_half_e<T, true, false>(h, src, ksum, pc) : primal_half_e<T, false, false>(h, src, ksum, pc);
}

template <typename T> int refresh_kx_t(pdlp_handle h)
{
    StoreEpi<T> e{(T*)h->kxb[0]};
    int rc = launch_csr<T>(h, false, h->xb[h->ix_cur], e, h->partB);
    if (rc == PDLP_OK) h->kx_valid = true;
    return rc;
}

template <typename T, bool ADAPT, bool PEER> int dual_half_e(pdlp_handle h, T* ksum)
{
    DualEpi<T, ADAPT, PEER> e{yloc<T>(h, h->ix_cur), yloc<T>(h, h->ix_prev), (const T*)h->p.q, (T*)h->y_sum, (T*)h->kxb[0],
                              h->sc, h->ineq_end, ksum};
    peer_targets(h, e.peer, 1 + h->ix_prev);
    if (ADAPT) h->last_gridB = grid_of(h->sK, h->ml);
    h->use_split = true;
    const int rc = launch_csr<T>(h, false, h->xbar, e, h->partB);
    h->use_split = false;
    return rc;
}

template <typename T> int dual_half_t(pdlp_handle h, int adaptive)
{
    int rc;
    if (!h->kx_valid && (rc = refresh_kx_t<T>(h)) != PDLP_OK) return rc;      // K x of the current x: carried along from here on
    T* ksum = (h->sums_broken || h->no_running || h->graph_ok) ? nullptr : (T*)h->kx_sum;
    const PieceCtl pc = piece_ctl(h, h->sK);
    rc = PDLP_OK;
    if (pc.skip) {
        // (all of this half-step went out with piece 0)
    } else if (h->peer.active) {
        rc = adaptive ? dual_half_e<T, true, true>(h, ksum) : dual_half_e<T, false, true>(h, ksum);
    } else {
        rc = adaptive ? dual_half_e<T, true, false>(h, ksum) : dual_half_e<T, false, false>(h, ksum);
    }
    if (rc != PDLP_OK) { h->sK.pending = false; return rc; }
    if (!pc.finish) return PDLP_OK;                          // (more pieces of this half-step to come)
    h->sK.pending = false;
    ++h->since_reset;
    h->kty_tail_done = false; h->avg_products = false;
    const int t = h->ix_cur;   // the freshly written buffers become current, the old ones previous
    h->ix_cur = h->ix_prev;
    h->ix_prev = t;
    h->cand_valid[0] = h->cand_valid[1] = false;
    h->kty_cur = -1;
    return PDLP_OK;
}

// a fused epilogue over a vector of finished products (no matrix pass): KKT sums from running products, K'y kept by a check
template <typename T, class Epi> int vector_pass(pdlp_handle h, int64_t rows, const void* products, Epi e, double* partials)
{
    if (rows == 0) return PDLP_OK;
    hipLaunchKernelGGL((k_rowsum_epilogue<T, Epi>), dim3(grid_for(rows)), dim3(BLOCK), 0, h->stream, (const T*)products, 1, (int64_t)0,
                       (int)rows, (const T*)nullptr, e, partials);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

template <typename T, bool UNSCALE> int kkt_local_u(pdlp_handle h, int which)
{
    const int ix = which == PDLP_CUR ? h->ix_cur : (which == PDLP_AVG ? h->ix_avg : h->ix_prev);
    const T* dcol = UNSCALE ? (const T*)h->p.d_col : nullptr;
    const T* drow = UNSCALE ? (const T*)h->p.d_row : nullptr;
    int rc, gridA = grid_of(h->sKT, h->nl), gridB = grid_of(h->sK, h->ml);
    if (which == PDLP_AVG && h->avg_products) {
        // K'y_avg (ktyb[1]) and K x_avg (kxb[2]) were formed from the running sums by pdlp_compute_average: two vector passes
        KktDualEpi<T, UNSCALE> ed{xloc<T>(h, ix), (const T*)h->p.c, (const T*)h->p.l, (const T*)h->p.u, dcol, nullptr};
        if ((rc = vector_pass<T>(h, h->nl, h->ktyb[1], ed, h->partA)) != PDLP_OK) return rc;
        KktPrimalEpi<T, UNSCALE> ep{yloc<T>(h, ix), (const T*)h->p.q, drow, nullptr, h->ineq_end};
        if ((rc = vector_pass<T>(h, h->ml, h->kxb[2], ep, h->partB)) != PDLP_OK) return rc;
        gridA = h->nl > 0 ? grid_for(h->nl) : 0;
        gridB = h->ml > 0 ? grid_for(h->ml) : 0;
    } else {
        T* kx_out = which == PDLP_CUR ? (T*)h->kxb[1] : (which == PDLP_AVG ? (T*)h->kxb[2] : nullptr);
        // K'y of a candidate is the product the first primal half-step after the check needs again (same kernel, same
        // sums): keep it.  (A pass at the current iterate after a restart to the average supersedes that restart's copy.)
        T* kty_out = which == PDLP_CUR ? (T*)h->ktyb[0] : (which == PDLP_AVG ? (T*)h->ktyb[1] : nullptr);
        if (which == PDLP_CUR || (which == PDLP_AVG && h->kty_cur == 1)) h->kty_cur = -1;     // (the copy about to be overwritten)
        KktDualEpi<T, UNSCALE> ed{xloc<T>(h, ix), (const T*)h->p.c, (const T*)h->p.l, (const T*)h->p.u, dcol, kty_out};
        if ((rc = launch_csr<T>(h, true, h->yb[ix], ed, h->partA)) != PDLP_OK) return rc;
        if (which == PDLP_CUR && h->kx_valid && !h->no_running) {
            // K x of the current iterate is carried along by the dual half-steps (kxb[0]): no product
            KktPrimalEpi<T, UNSCALE> ep{yloc<T>(h, ix), (const T*)h->p.q, drow, nullptr, h->ineq_end};
            if ((rc = vector_pass<T>(h, h->ml, h->kxb[0], ep, h->partB)) != PDLP_OK) return rc;
            gridB = h->ml > 0 ? grid_for(h->ml) : 0;
            h->cur_kx_cached = true;
        } else {
            KktPrimalEpi<T, UNSCALE> ep{yloc<T>(h, ix), (const T*)h->p.q, drow, kx_out, h->ineq_end};
            if ((rc = launch_csr<T>(h, false, h->xb[ix], ep, h->partB)) != PDLP_OK) return rc;
            if (which == PDLP_CUR) h->cur_kx_cached = false;
        }
    }
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, gridA, 4, h->red, 0);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partB, gridB, 2, h->red, 4);
    HIP_TRY(hipGetLastError());
    if (which != PDLP_PREV) h->cand_valid[which == PDLP_CUR ? 0 : 1] = true;
    return PDLP_OK;
}

template <typename T> int kkt_local_t(pdlp_handle h, int which, int unscaled)
{
    return unscaled ? kkt_local_u<T, true>(h, which) : kkt_local_u<T, false>(h, which);
}

template <typename T> void kkt_finish_t(const double* r, double omega_d, double* out)
{
    // helpers.py:84-94,102-106 in the working precision
    const T p = (T)r[3], d = (T)r[5], lp = (T)r[1], un = (T)r[2];
    const T adj = d + lp + un;
    const T gap = adj - p;
    const T pr = (T)std::sqrt(r[4]), dr = (T)std::sqrt(r[0]);
    const T w = (T)omega_d, w2 = w * w;
    const T kkt = (T)std::sqrt((double)(w2 * (pr * pr) + (dr * dr) / w2 + gap * gap));
    out[0] = pr; out[1] = dr; out[2] = gap; out[3] = p; out[4] = adj; out[5] = kkt;
}

// The pieces of a chunked exchange: piece c moves elements [sb[c], sb[c+1]) of every rank's block of `B` elements (multiples of
// 64 elements: 256-byte pieces).  A function of B and the requested count alone -- every rank computes the same plan, whether
// or not its own product is split (the collectives must match on all ranks; what a rank multiplies early is its own business).
int plan_bounds(int64_t B, int xchunks, int64_t* sb /*[MAX_PHASE]*/)
{
    int C = xchunks < 1 ? 1 : (xchunks > MAX_CHUNKS ? MAX_CHUNKS : xchunks);
    if (B < (int64_t)64 * C) C = 1;
    for (int c = 0; c <= C; ++c) sb[c] = c == C ? B : (c * B / C) / 64 * 64;
    for (int c = C + 1; c < MAX_PHASE; ++c) sb[c] = B;
    return C;
}

// panel groups of the split product of one matrix (see Schedule): which panels belong to which phase of the exchange, and how
// many workgroup groups (= partial row sum slots) every phase gets
int configure_split(pdlp_handle h, bool transpose)
{
    Schedule& s = transpose ? h->sKT : h->sK;
    s.loc_pa = s.loc_pb = s.slotsA = s.slotsB = 0;
    s.pending = false;
    s.nphase = 0; s.chunks_done = 0;
    s.nrange = 0;
    if (!s.tiled) return PDLP_OK;
    const int64_t lo = transpose ? h->p.row0 : h->p.col0, hi = transpose ? h->p.row1 : h->p.col1;
    const int64_t total = transpose ? h->p.m : h->p.n;
    if (lo == 0 && hi == total) return PDLP_OK;                       // not sharded: nothing to wait for
    const int64_t W = (int64_t)1 << s.t.lw, B = hi - lo;
    const int npanel = s.t.npanel;
    const int pa = (int)((lo + W - 1) / W), pb = hi == total ? npanel : (int)(hi / W);
    const int nloc = pb - pa, nrem = npanel - nloc;
    if (nloc <= 0 || nrem <= 0 || h->rs_groups < 2 || !s.ptab || npanel > s.ptab_cap || B <= 0 || lo % B != 0) return PDLP_OK;
    const int C = plan_bounds(B, h->xchunks, s.sb);
    // The RESULT of this product (this rank's block of y for K, of xbar for K') is the input of the other product and travels in
    // the pieces of THAT exchange: elements [so[r], so[r+1]) of the block = piece r.  With more than one piece the last phase and
    // the epilogue run piece by piece (launch_mat): the row blocks that hold piece r's rows, then piece r + 1's.
    const int64_t rows_out = transpose ? h->nl : h->ml;
    const int64_t rbk = (int64_t)TNT * s.t.rpt;
    int64_t so[MAX_PHASE];
    const int R = h->producer_pieces ? plan_bounds(rows_out, h->xchunks, so) : 1;
    int nb_max = s.t.nblk;
    if (R > 1) {
        nb_max = 0;
        for (int r = 0; r <= R; ++r) {
            const int64_t b = r == R ? s.t.nblk : (so[r] + rbk - 1) / rbk;
            s.rb_lo[r] = (int)(b < s.t.nblk ? b : s.t.nblk);
        }
        for (int r = 0; r < R; ++r) nb_max = (s.rb_lo[r + 1] - s.rb_lo[r]) > nb_max ? (s.rb_lo[r + 1] - s.rb_lo[r]) : nb_max;
    }
    // a panel is complete once the last of its foreign entries has arrived
    std::vector<int> phase((size_t)npanel);
    int cnt[MAX_PHASE] = {0};
    for (int p = 0; p < npanel; ++p) {
        int ph = 0;
        if (p < pa || p >= pb) {
            const int64_t c0 = (int64_t)p * W, c1 = (c0 + W < total) ? c0 + W : total;
            ph = 1;
            for (int64_t q = c0 / B; q <= (c1 - 1) / B; ++q) {
                if (q * B == lo) continue;                               // the own block is there already
                const int64_t off_hi = ((c1 < (q + 1) * B) ? c1 : (q + 1) * B) - 1 - q * B;
                int c = 0;
                while (c + 1 < C && s.sb[c + 1] <= off_hi) ++c;
                if (1 + c > ph) ph = 1 + c;
            }
        }
        phase[(size_t)p] = ph;
        ++cnt[ph];
    }
    // Slots.  Measured on shard-shaped matrices with a spin kernel standing in for the gather (tools/split_timing.py):
    // each launch must fit ONE round of workgroups (2 per CU) or its tail costs more than the overlap gains; the
    // other panels take as many groups as fit; the local panels enough groups that a workgroup walks <= ~13 panels
    // and is done by the time the gather is.  10M x 10M: 8 ranks (2 + 8 groups) 0.402 -> 0.380 ms per half-step,
    // 4 ranks (3 + 4) 0.677 -> 0.573 ms, 2 ranks (2 + 2) 1.27 -> 1.01 ms.
    const int round_slots = 2 * 256;
    int fit = round_slots / (s.t.nblk > 0 ? s.t.nblk : 1);
    fit = fit < 1 ? 1 : fit;
    // Group counts are powers of two: the group is the fast index of blockIdx and workgroups are dealt round-robin over the 8 XCDs,
    // so with 8 (16) groups each XCD's L2 holds the panels of one (two) groups only, with 2 or 4 groups of two or four -- any other
    // count spreads every group over all XCDs and each of them pulls the whole gathered vector (measured: k_tiled_fused).
    auto pow2 = [](int g) { int p = 1; while (2 * p <= g) p *= 2; return p; };
    auto norm = [&](int g, int n) { if (n <= 0) return 0; g = g < 1 ? 1 : (g > n ? n : g); return pow2(g); };
    int a = (nloc + 12) / 13;
    a = a > fit ? fit : a;
    a = a > nloc ? nloc : a;
    int g[MAX_PHASE] = {0};
    if (C == 1) {
        int b = fit < nrem ? fit : nrem;
        if (a + b > h->rs_groups) b = h->rs_groups - a;
        if (a < 1 || b < 1) return PDLP_OK;
        int S = a + b;
        if (h->split_local >= 1 && h->split_other >= 1 && h->split_local + h->split_other <= h->rs_groups) {   // PDLP_OPT_SPLIT_SLOTS (tools)
            a = h->split_local;
            S = h->split_local + h->split_other;
        }
        g[0] = norm(a, nloc);
        g[1] = norm(S - a, nrem);
    } else {
        // every chunk's launch fills the chip by itself where it can; fewer groups per chunk when the scratch runs out
        int left = h->rs_groups - a;
        if (a < 1 || left < C) return PDLP_OK;
        g[0] = norm(a, nloc);
        int want[MAX_PHASE] = {0}, sum = 0;
        // (the last phase of a product whose result travels in pieces is launched piece by piece: each launch covers only nb_max row
        //  blocks and needs proportionally more groups to fill the chip)
        int fit_last = R > 1 && nb_max > 0 ? round_slots / nb_max : fit;     // (the phase's own group count left the chip half empty: 0.92 against 0.81 ms per iteration at 8 ranks)
        fit_last = fit_last < 1 ? 1 : fit_last;
        for (int c = 0; c < C; ++c) {
            const int f = c == C - 1 ? fit_last : fit;
            want[1 + c] = cnt[1 + c] > 0 ? (f < cnt[1 + c] ? f : cnt[1 + c]) : 0;
            sum += want[1 + c];
        }
        for (int c = 0; c < C; ++c) {
            int w = want[1 + c];
            if (sum > left && w > 0) { w = (int)((int64_t)w * left / sum); w = w < 1 ? 1 : w; }
            g[1 + c] = norm(w, cnt[1 + c]);
        }
    }
    // the table: panels phase by phase, ascending inside a phase
    std::vector<int32_t> tab((size_t)npanel);
    int off = 0, slot = 0;
    s.nphase = 1 + C;
    for (int ph = 0; ph < s.nphase; ++ph) {
        s.ph_off[ph] = off; s.ph_cnt[ph] = cnt[ph]; s.ph_slots[ph] = g[ph]; s.ph_slot0[ph] = slot;
        for (int p = 0; p < npanel; ++p)
            if (phase[(size_t)p] == ph) tab[(size_t)off++] = p;
        slot += g[ph];
    }
    if (slot > h->rs_groups) { s.nphase = 0; return PDLP_OK; }
    HIP_TRY(hipMemcpyAsync(s.ptab, tab.data(), (size_t)npanel * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));                          // (the host vector goes away)
    s.loc_pa = pa; s.loc_pb = pb;
    s.slotsA = g[0];
    s.slotsB = slot - g[0];
    s.nrange = (R > 1 && C > 1) ? R : 0;          // (one exchange piece = one all-gather: nothing to send early)
    return PDLP_OK;
}

template <typename T> int half_begin_t(pdlp_handle h, bool transpose, const void* vin)
{
    Schedule& s = transpose ? h->sKT : h->sK;
    if (!s.tiled || s.slotsA == 0 || (!h->gstream && !h->begin_inline) || s.pending) return PDLP_OK;
    const int rows = (int)(transpose ? h->nl : h->ml);
    // begin_inline (PDLP_OPT_BEGIN_INLINE): the caller has ALREADY issued the exchange asynchronously on a stream of its own, so the
    // local panels simply go onto the handle's stream and run beside it -- no side stream, no fork / join events (each cross-stream
    // dependency costs about a kernel launch on this stack); the half-step that follows then has nothing to wait for
    hipStream_t st = h->begin_inline ? h->stream : h->gstream;
    if (!h->begin_inline) {
        HIP_TRY(hipEventRecord(h->ev_in, h->stream));
        HIP_TRY(hipStreamWaitEvent(h->gstream, h->ev_in, 0));
    }
    if (h->delta) launch_phase<float, float>(h, s, rows, transpose ? (const void*)h->gdy : (const void*)h->gdx, st, 0);
    else if (std::is_same<T, double>::value && h->mixed) launch_phase<double, float>(h, s, rows, vin, st, 0);
    else launch_phase<T, T>(h, s, rows, vin, st, 0);
    if (!h->begin_inline) HIP_TRY(hipEventRecord(h->ev_out, h->gstream));
    HIP_TRY(hipGetLastError());
    s.pending = true;
    s.pending_inline = h->begin_inline;
    s.chunks_done = 0;
    return PDLP_OK;
}

// the panels that chunk `chunk` of the exchange completes, on the handle's stream (the caller has made that stream wait for the
// chunk); the last chunk's panels are launched by the half-step itself, together with the sum and the epilogue
template <typename T> int half_chunk_t(pdlp_handle h, bool transpose, const void* vin, int chunk)
{
    Schedule& s = transpose ? h->sKT : h->sK;
    if (!s.pending || s.nphase == 0) return PDLP_OK;           // the product is not split this time: the half-step does it all
    if (chunk != s.chunks_done || chunk + 2 >= s.nphase + 0) return chunk + 2 == s.nphase ? PDLP_OK : PDLP_ERR_STATE;
    const int rows = (int)(transpose ? h->nl : h->ml);
    if (h->delta) launch_phase<float, float>(h, s, rows, transpose ? (const void*)h->gdy : (const void*)h->gdx, h->stream, 1 + chunk);
    else if (std::is_same<T, double>::value && h->mixed) launch_phase<double, float>(h, s, rows, vin, h->stream, 1 + chunk);
    else launch_phase<T, T>(h, s, rows, vin, h->stream, 1 + chunk);
    HIP_TRY(hipGetLastError());
    ++s.chunks_done;
    return PDLP_OK;
}

// ---- delta mode (mixed precision) ------------------------------------------------------------------
// exact anchors: KX = K x_cur and KTY = K'y_cur by the mixed-precision kernels (float64 gathers, products and sums)
int delta_refresh(pdlp_handle h)
{
    int rc;
    StoreEpi<double> ex{(double*)h->kxb[0]};
    if ((rc = launch_mat<double, float, StoreEpi<double>>(h, false, h->xb[h->ix_cur], ex, h->partB)) != PDLP_OK) return rc;
    StoreEpi<double> ey{(double*)h->ktyr};
    if ((rc = launch_mat<double, float, StoreEpi<double>>(h, true, h->yb[h->ix_cur], ey, h->partA)) != PDLP_OK) return rc;
    h->anchors_valid = true;
    h->dy_folded = true;
    h->kx_valid = true;
    return PDLP_OK;
}

// a float64 epilogue over a float64 vector of finished products (no matrix pass)
template <class Epi> int delta_vector_pass(pdlp_handle h, int64_t rows, const double* products, Epi e, double* partials)
{
    if (rows == 0) return PDLP_OK;
    hipLaunchKernelGGL((k_rowsum_epilogue<double, Epi>), dim3(grid_for(rows)), dim3(BLOCK), 0, h->stream, products, 1, (int64_t)0,
                       (int)rows, (const double*)nullptr, e, partials);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

template <bool ADAPT, bool PEER> int delta_primal_half_a(pdlp_handle h)
{
    int rc;
    if (!h->anchors_valid && (rc = delta_refresh(h)) != PDLP_OK) return rc;
    DeltaPrimalEpi<ADAPT, PEER> e{(const double*)xloc<double>(h, h->ix_cur), xloc<double>(h, h->ix_prev), h->gdx + h->p.col0, (const double*)h->p.c,
                                  (const double*)h->p.l, (const double*)h->p.u, (double*)h->x_sum, (double*)h->ktyr, h->sc};
    peer_targets(h, e.peer, 4);
    if (h->dy_folded && !h->sKT.pending) {
        // K'y of the current y is already in the anchor (a restart check folded dy in, or the anchors are fresh): vector pass
        if (h->range_sel > 0) return PDLP_OK;                // (issued piece by piece: all of it went out with piece 0)
        h->last_gridA = h->nl > 0 ? grid_for(h->nl) : 0;
        if (h->nl == 0) return PDLP_OK;
        hipLaunchKernelGGL((k_rowsum_epilogue<float, DeltaPrimalEpi<ADAPT, PEER>>), dim3(grid_for(h->nl)), dim3(BLOCK), 0, h->stream,
                           (const float*)nullptr, 0, (int64_t)0, (int)h->nl, (const float*)nullptr, e, h->partA);
        HIP_TRY(hipGetLastError());
        return PDLP_OK;
    }
    const PieceCtl pc = piece_ctl(h, h->sKT);
    if (pc.skip) return PDLP_OK;
    h->last_gridA = grid_of(h->sKT, h->nl);
    h->use_split = true;
    rc = launch_mat<float, float, DeltaPrimalEpi<ADAPT, PEER>>(h, true, h->gdy, e, h->partA);
    h->use_split = false;
    if (pc.finish || rc != PDLP_OK) {
        h->sKT.pending = false;
        h->dy_folded = true;   // (the anchor now belongs to the current y)
    }
    return rc;
}

template <bool ADAPT, bool PEER> int delta_dual_half_a(pdlp_handle h)
{
    DeltaDualEpi<ADAPT, PEER> e{(const double*)yloc<double>(h, h->ix_cur), yloc<double>(h, h->ix_prev), h->gdy + h->p.row0, (const double*)h->p.q,
                                (double*)h->y_sum, (double*)h->kxb[0], h->sc, h->ineq_end};
    peer_targets(h, e.peer, 5);
    const PieceCtl pc = piece_ctl(h, h->sK);
    int rc = PDLP_OK;
    if (!pc.skip) {
        h->last_gridB = grid_of(h->sK, h->ml);
        h->use_split = true;
        rc = launch_mat<float, float, DeltaDualEpi<ADAPT, PEER>>(h, false, h->gdx, e, h->partB);
        h->use_split = false;
    }
    if (rc != PDLP_OK) { h->sK.pending = false; return rc; }
    if (!pc.finish) return PDLP_OK;
    h->sK.pending = false;
    const int t = h->ix_cur;
    h->ix_cur = h->ix_prev;
    h->ix_prev = t;
    h->cand_valid[0] = h->cand_valid[1] = false;
    h->dy_folded = false;      // gdy = y_cur - y_prev waits for the next product with K'
    return PDLP_OK;
}

int delta_primal_half(pdlp_handle h, int adaptive)
{
    if (h->peer.active) return adaptive ? delta_primal_half_a<true, true>(h) : delta_primal_half_a<false, true>(h);
    return adaptive ? delta_primal_half_a<true, false>(h) : delta_primal_half_a<false, false>(h);
}
int delta_dual_half(pdlp_handle h, int adaptive)
{
    if (h->peer.active) return adaptive ? delta_dual_half_a<true, true>(h) : delta_dual_half_a<false, true>(h);
    return adaptive ? delta_dual_half_a<true, false>(h) : delta_dual_half_a<false, false>(h);
}

// KKT sums of a candidate from the anchors: the current iterate needs at most the pending K'dy; the averaged / previous
// iterate two float32 products over float32(candidate - current) added to the anchors
// the current iterate's KKT sums from the anchors; UNSCALE: of the un-preconditioned problem (pdhg.py:157-161)
template <bool UNSCALE> int delta_kkt_cur(pdlp_handle h)
{
    int rc;
    typedef KktDualEpi<double, UNSCALE> KD;
    typedef KktPrimalEpi<double, UNSCALE> KP;
    KD ed{xloc<double>(h, h->ix_cur), (const double*)h->p.c, (const double*)h->p.l, (const double*)h->p.u,
          UNSCALE ? (const double*)h->p.d_col : nullptr, nullptr};
    if (!h->dy_folded) {
        AnchorEpi<KD, true> e{ed, (double*)h->ktyr};
        if ((rc = launch_mat<float, float, AnchorEpi<KD, true>>(h, true, h->gdy, e, h->partA)) != PDLP_OK) return rc;
        h->dy_folded = true;
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, grid_of(h->sKT, h->nl), 4, h->red, 0);
    } else {
        if ((rc = delta_vector_pass(h, h->nl, (const double*)h->ktyr, ed, h->partA)) != PDLP_OK) return rc;
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, h->nl > 0 ? grid_for(h->nl) : 0, 4, h->red, 0);
    }
    KP ep{yloc<double>(h, h->ix_cur), (const double*)h->p.q, UNSCALE ? (const double*)h->p.d_row : nullptr, nullptr, h->ineq_end};
    if ((rc = delta_vector_pass(h, h->ml, (const double*)h->kxb[0], ep, h->partB)) != PDLP_OK) return rc;
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partB, h->ml > 0 ? grid_for(h->ml) : 0, 2, h->red, 4);
    HIP_TRY(hipGetLastError());
    h->cand_valid[0] = true;
    return PDLP_OK;
}

int delta_kkt_local(pdlp_handle h, int which, int unscaled)
{
    int rc;
    if (!h->anchors_valid && (rc = delta_refresh(h)) != PDLP_OK) return rc;
    typedef KktDualEpi<double, false> KD;
    typedef KktPrimalEpi<double, false> KP;
    if (which == PDLP_CUR) return unscaled ? delta_kkt_cur<true>(h) : delta_kkt_cur<false>(h);
    if (unscaled) return PDLP_ERR_STATE;                  // (the driver evaluates the un-scaled problem at the current iterate only)
    if (!h->dy_folded) {
        FoldEpi f{(double*)h->ktyr};
        if ((rc = launch_mat<float, float, FoldEpi>(h, true, h->gdy, f, h->partA)) != PDLP_OK) return rc;
        h->dy_folded = true;
    }
    const int ix = which == PDLP_AVG ? h->ix_avg : h->ix_prev;
    // the full-length differences (every rank holds the complete candidate and the complete current iterate)
    hipLaunchKernelGGL(k_diff_f32, dim3(grid_for(h->p.n)), dim3(BLOCK), 0, h->stream, h->p.n, h->gdx, (const double*)h->xb[ix],
                       (const double*)h->xb[h->ix_cur]);
    hipLaunchKernelGGL(k_diff_f32, dim3(grid_for(h->p.m)), dim3(BLOCK), 0, h->stream, h->p.m, h->gdy, (const double*)h->yb[ix],
                       (const double*)h->yb[h->ix_cur]);
    // K'y and K x of the averaged iterate are kept: a restart to it adopts them as the new anchors
    KD ed{xloc<double>(h, ix), (const double*)h->p.c, (const double*)h->p.l, (const double*)h->p.u, nullptr,
          which == PDLP_AVG ? (double*)h->ktyb[1] : nullptr};
    AnchorEpi<KD, false> ea{ed, (double*)h->ktyr};
    if ((rc = launch_mat<float, float, AnchorEpi<KD, false>>(h, true, h->gdy, ea, h->partA)) != PDLP_OK) return rc;
    KP ep{yloc<double>(h, ix), (const double*)h->p.q, nullptr, which == PDLP_AVG ? (double*)h->kxb[2] : nullptr, h->ineq_end};
    AnchorEpi<KP, false> eb{ep, (double*)h->kxb[0]};
    if ((rc = launch_mat<float, float, AnchorEpi<KP, false>>(h, false, h->gdx, eb, h->partB)) != PDLP_OK) return rc;
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, grid_of(h->sKT, h->nl), 4, h->red, 0);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partB, grid_of(h->sK, h->ml), 2, h->red, 4);
    HIP_TRY(hipGetLastError());
    if (which == PDLP_AVG) h->cand_valid[1] = true;
    return PDLP_OK;
}

#define DISPATCH(h, fn, ...) ((h)->p.dtype == PDLP_F32 ? fn<float>(__VA_ARGS__) : fn<double>(__VA_ARGS__))

template <typename T> int flush_t(pdlp_handle h, int adaptive)
{
    const bool running = !h->delta && !h->sums_broken && !h->no_running && !h->graph_ok && h->since_reset > 0;
    if (adaptive) {
        // the weight of the current iterate became known only after its step-size rule: add it now
        hipLaunchKernelGGL(k_flush<T>, dim3(grid_for(h->nl)), dim3(BLOCK), 0, h->stream, h->nl, (T*)h->x_sum,
                           (const T*)xloc<T>(h, h->ix_cur), h->sc, (int)S_WPEND);
        hipLaunchKernelGGL(k_flush<T>, dim3(grid_for(h->ml)), dim3(BLOCK), 0, h->stream, h->ml, (T*)h->y_sum,
                           (const T*)yloc<T>(h, h->ix_cur), h->sc, (int)S_WPEND);
        if (running && h->kx_valid)
            hipLaunchKernelGGL(k_flush<T>, dim3(grid_for(h->ml)), dim3(BLOCK), 0, h->stream, h->ml, (T*)h->kx_sum, (const T*)h->kxb[0],
                               h->sc, (int)S_WPEND);
    }
    // K'y of the current y exists only if the KKT pass of the current iterate ran before this call (it keeps it in ktyb[0])
    if (running && !h->kty_tail_done) {
        if (h->cand_valid[0] && h->kty_cur < 0) {
            hipLaunchKernelGGL(k_flush<T>, dim3(grid_for(h->nl)), dim3(BLOCK), 0, h->stream, h->nl, (T*)h->kty_sum, (const T*)h->ktyb[0],
                               h->sc, (int)(adaptive ? S_WPEND : S_ETA));
            h->kty_tail_done = true;
        } else if (adaptive) {
            h->sums_broken = true;       // the pending weight is cleared below: that term of the sum is lost until the next restart
        }
    }
    if (adaptive) hipLaunchKernelGGL(k_clear_pending, dim3(1), dim3(1), 0, h->stream, h->sc);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

template <typename T> int average_t(pdlp_handle h)
{
    hipLaunchKernelGGL(k_average<T>, dim3(grid_for(h->nl)), dim3(BLOCK), 0, h->stream, h->nl, xloc<T>(h, h->ix_avg),
                       (const T*)h->x_sum, h->sc);
    hipLaunchKernelGGL(k_average<T>, dim3(grid_for(h->ml)), dim3(BLOCK), 0, h->stream, h->ml, yloc<T>(h, h->ix_avg),
                       (const T*)h->y_sum, h->sc);
    // the products of the average from the running sums (K is linear): K x_avg = sum w_k K x_k / sum w_k, the same for K'y
    h->avg_products = false;
    if (!h->delta && !h->sums_broken && !h->no_running && !h->graph_ok && h->since_reset > 0 && h->kty_tail_done && h->kx_valid) {
        hipLaunchKernelGGL(k_average<T>, dim3(grid_for(h->ml)), dim3(BLOCK), 0, h->stream, h->ml, (T*)h->kxb[2], (const T*)h->kx_sum, h->sc);
        hipLaunchKernelGGL(k_average<T>, dim3(grid_for(h->nl)), dim3(BLOCK), 0, h->stream, h->nl, (T*)h->ktyb[1], (const T*)h->kty_sum, h->sc);
        if (h->kty_cur == 1) h->kty_cur = -1;
        h->avg_products = true;
    }
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

template <typename T> int distance_t(pdlp_handle h)
{
    const int ga = grid_for(h->nl), gb = grid_for(h->ml);
    hipLaunchKernelGGL(k_sqdiff<T>, dim3(ga), dim3(BLOCK), 0, h->stream, h->nl, (const T*)h->x_last,
                       (const T*)xloc<T>(h, h->ix_cur), h->partA);
    hipLaunchKernelGGL(k_sqdiff<T>, dim3(gb), dim3(BLOCK), 0, h->stream, h->ml, (const T*)h->y_last,
                       (const T*)yloc<T>(h, h->ix_cur), h->partB);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, ga, 1, h->red, 0);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partB, gb, 1, h->red, 1);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

template <typename T> int spmv_t(pdlp_handle h, int transpose, const void* in_full, void* out_local)
{
    StoreEpi<T> e{(T*)out_local};
    return launch_csr<T>(h, transpose != 0, in_full, e, h->partA);
}

// dx, dy of the step just taken (cur vs prev) into this rank's blocks of the full-length buffers
template <typename T> int infeas_begin_t(pdlp_handle h)
{
    if (h->nl > 0)
        hipLaunchKernelGGL(k_sub<T>, dim3(grid_for(h->nl)), dim3(BLOCK), 0, h->stream, h->nl, (T*)h->dxf + h->p.col0,
                           (const T*)xloc<T>(h, h->ix_cur), (const T*)xloc<T>(h, h->ix_prev));
    if (h->ml > 0)
        hipLaunchKernelGGL(k_sub<T>, dim3(grid_for(h->ml)), dim3(BLOCK), 0, h->stream, h->ml, (T*)h->dyf + h->p.row0,
                           (const T*)yloc<T>(h, h->ix_cur), (const T*)yloc<T>(h, h->ix_prev));
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

// three products (K'dy, K'y with the variable-side tests fused, K dx with the constraint-side tests fused) and the
// eight partial sums of detect_infeasibility into red[0..7]
template <typename T> int infeas_local_t(pdlp_handle h, double tol)
{
    int rc;
    if ((rc = spmv_t<T>(h, 1, h->dyf, h->ktdy)) != PDLP_OK) return rc;
    InfeasDualEpi<T> ed{xloc<T>(h, h->ix_cur), xloc<T>(h, h->ix_prev), (const T*)h->p.c, (const T*)h->p.l, (const T*)h->p.u,
                        (const T*)h->ktdy, (T*)h->lam_prev, (T)tol};
    if ((rc = launch_csr<T>(h, true, h->yb[h->ix_cur], ed, h->partA)) != PDLP_OK) return rc;
    InfeasPrimalEpi<T> ep{yloc<T>(h, h->ix_cur), yloc<T>(h, h->ix_prev), (const T*)h->p.q, (T)tol, h->ineq_end};
    if ((rc = launch_csr<T>(h, false, h->dxf, ep, h->partB)) != PDLP_OK) return rc;
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, grid_of(h->sKT, h->nl), 4, h->red, 0);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partB, grid_of(h->sK, h->ml), 4, h->red, 4);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

// the decisions of enhancements.py:118-142 and :148-159 from the eight sums, in the working precision
template <typename T> int infeas_decide_t(const double* r, double tol, double* diag)
{
    const T t = (T)tol;
    const T dres = (T)std::sqrt(r[0]), lu = (T)r[1], cdx = (T)r[2], eqn = (T)std::sqrt(r[4]), qdy = (T)r[7];
    diag[0] = eqn; diag[1] = r[5]; diag[2] = cdx; diag[3] = r[3]; diag[4] = dres; diag[5] = r[6]; diag[6] = qdy; diag[7] = lu;
    if (eqn < t && r[5] == 0.0 && cdx < t && r[3] == 0.0) return 1;                      // "DUAL_INFEASIBLE"
    if (dres < t && r[6] == 0.0 && (double)qdy - (double)lu > -tol) return 2;            // "PRIMAL_INFEASIBLE"
    return 0;
}

template <typename T> int power_iteration_t(pdlp_handle h, const void* b0, int iters, void* work_n, void* work_m, double* sigma)
{
    T* b = (T*)work_n;
    T* t = (T*)work_m;
    int rc;
    HIP_TRY(hipMemcpyAsync(b, b0, (size_t)h->p.n * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    const int g = grid_for(h->p.n);
    for (int it = 0; it < iters; ++it) {                                   // helpers.py:48-50
        if ((rc = spmv_t<T>(h, 0, b, t)) != PDLP_OK) return rc;
        if ((rc = spmv_t<T>(h, 1, t, b)) != PDLP_OK) return rc;
        hipLaunchKernelGGL(k_sqdiff<T>, dim3(g), dim3(BLOCK), 0, h->stream, h->p.n, (const T*)b, (const T*)nullptr, h->partA);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, g, 1, h->red, 0);
        hipLaunchKernelGGL(k_div_by_norm<T>, dim3(g), dim3(BLOCK), 0, h->stream, h->p.n, b, h->red, 0);
    }
    if ((rc = spmv_t<T>(h, 0, b, t)) != PDLP_OK) return rc;               // helpers.py:51
    const int gm = grid_for(h->p.m);
    hipLaunchKernelGGL(k_sqdiff<T>, dim3(gm), dim3(BLOCK), 0, h->stream, h->p.m, (const T*)t, (const T*)nullptr, h->partA);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, gm, 1, h->red, 0);
    HIP_TRY(hipGetLastError());
    double r = 0.0;
    HIP_TRY(hipMemcpyAsync(&r, h->red, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *sigma = (double)(T)std::sqrt(r);
    return PDLP_OK;
}

// ---- population kernels (fishnet) ---------------------------------------------------------------
constexpr int MV_GAP_GRID = 256;
inline int mv_grid(int64_t rows, int nvp)
{
    const int64_t per_block = (int64_t)(64 / nvp) * (BLOCK / 64);
    const int64_t g = (rows + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

template <typename T, int NVP> int mv_steps_n(pdlp_handle h, int steps, double eta, double omega, double theta, T* X, T* Y, T* work)
{
    const int64_t n = h->p.n, m = h->p.m;
    const T e = (T)eta, w = (T)omega;
    const T tau = (T)(e / w), sigma = (T)(e * w);                 // (rounded like k_set_step)
    T *Xa = X, *Xb = work, *Xbar = work + n * NVP, *Ya = Y, *Yb = work + 2 * n * NVP;
    for (int s = 0; s < steps; ++s) {
        PrimalMV<T> ep{Xa, Xb, Xbar, (const T*)h->p.c, (const T*)h->p.l, (const T*)h->p.u, tau, (T)theta};
        hipLaunchKernelGGL((k_csr_mv<T, NVP, PrimalMV<T>>), dim3(mv_grid(n, NVP)), dim3(BLOCK), 0, h->stream, (int)n, h->p.KT_rowptr,
                           h->p.KT_colidx, (const T*)h->p.KT_val, (const T*)Ya, ep, (double*)nullptr);
        DualMV<T> ed{Ya, Yb, (const T*)h->p.q, sigma, h->ineq_end};
        hipLaunchKernelGGL((k_csr_mv<T, NVP, DualMV<T>>), dim3(mv_grid(m, NVP)), dim3(BLOCK), 0, h->stream, (int)m, h->p.K_rowptr,
                           h->p.K_colidx, (const T*)h->p.K_val, (const T*)Xbar, ed, (double*)nullptr);
        T* t = Xa; Xa = Xb; Xb = t;
        t = Ya; Ya = Yb; Yb = t;
    }
    HIP_TRY(hipGetLastError());
    if (steps & 1) {
        HIP_TRY(hipMemcpyAsync(X, Xa, (size_t)n * NVP * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(Y, Ya, (size_t)m * NVP * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    }
    return PDLP_OK;
}

template <typename T, int NVP> int mv_gap_n(pdlp_handle h, const T* X, const T* Y, double* work, double* gaps)
{
    const int64_t n = h->p.n, m = h->p.m;
    const int ga = mv_grid(n, NVP) < MV_GAP_GRID ? mv_grid(n, NVP) : MV_GAP_GRID, gb = mv_grid(m, NVP) < MV_GAP_GRID ? mv_grid(m, NVP) : MV_GAP_GRID;
    double* pa = work;                                   // [ga][NVP][3]
    double* pb = work + (size_t)MV_GAP_GRID * NVP * 3;   // [gb][NVP][1]
    double* out = pb + (size_t)MV_GAP_GRID * NVP;        // [NVP][3] then [NVP]
    GapMV<T> eg{X, (const T*)h->p.c, (const T*)h->p.l, (const T*)h->p.u};
    hipLaunchKernelGGL((k_csr_mv<T, NVP, GapMV<T>>), dim3(ga), dim3(BLOCK), 0, h->stream, (int)n, h->p.KT_rowptr, h->p.KT_colidx,
                       (const T*)h->p.KT_val, Y, eg, pa);
    hipLaunchKernelGGL((k_mv_dot<T, NVP>), dim3(gb), dim3(BLOCK), 0, h->stream, (int)m, (const T*)h->p.q, Y, pb);
    hipLaunchKernelGGL(k_mv_finalize, dim3(1), dim3(BLOCK), 0, h->stream, pa, ga, NVP * 3, out);
    hipLaunchKernelGGL(k_mv_finalize, dim3(1), dim3(BLOCK), 0, h->stream, pb, gb, NVP, out + NVP * 3);
    HIP_TRY(hipGetLastError());
    double r[32 * 4];
    HIP_TRY(hipMemcpyAsync(r, out, (size_t)NVP * 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int v = 0; v < NVP; ++v) {                      // get_best_pts :231-233 in the working precision
        const T p = (T)r[3 * v], lp = (T)r[3 * v + 1], un = (T)r[3 * v + 2], d = (T)r[NVP * 3 + v];
        const T adj = d + lp + un;
        gaps[v] = (double)(T)(adj - p);
    }
    return PDLP_OK;
}

template <typename T, int NVP> int mv_product_n(pdlp_handle h, const T* X, T* Y)
{
    const int64_t m = h->p.m;
    StoreMV<T> st{Y};
    hipLaunchKernelGGL((k_csr_mv<T, NVP, StoreMV<T>>), dim3(mv_grid(m, NVP)), dim3(BLOCK), 0, h->stream, (int)m, h->p.K_rowptr,
                       h->p.K_colidx, (const T*)h->p.K_val, X, st, (double*)nullptr);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

template <typename T> int mv_product_t(pdlp_handle h, int nvp, const void* X, void* Y)
{
    if (nvp == 8) return mv_product_n<T, 8>(h, (const T*)X, (T*)Y);
    if (nvp == 16) return mv_product_n<T, 16>(h, (const T*)X, (T*)Y);
    return mv_product_n<T, 32>(h, (const T*)X, (T*)Y);
}

template <typename T> int mv_steps_t(pdlp_handle h, int nvp, int steps, double eta, double omega, double theta, void* X, void* Y, void* work)
{
    if (nvp == 8) return mv_steps_n<T, 8>(h, steps, eta, omega, theta, (T*)X, (T*)Y, (T*)work);
    if (nvp == 16) return mv_steps_n<T, 16>(h, steps, eta, omega, theta, (T*)X, (T*)Y, (T*)work);
    return mv_steps_n<T, 32>(h, steps, eta, omega, theta, (T*)X, (T*)Y, (T*)work);
}

template <typename T> int mv_gap_t(pdlp_handle h, int nvp, const void* X, const void* Y, void* work, double* gaps)
{
    if (nvp == 8) return mv_gap_n<T, 8>(h, (const T*)X, (const T*)Y, (double*)work, gaps);
    if (nvp == 16) return mv_gap_n<T, 16>(h, (const T*)X, (const T*)Y, (double*)work, gaps);
    return mv_gap_n<T, 32>(h, (const T*)X, (const T*)Y, (double*)work, gaps);
}

int check_problem(const pdlp_problem* p)
{
    if (!p) return PDLP_ERR_INVALID;
    if (p->dtype != PDLP_F32 && p->dtype != PDLP_F64 && p->dtype != PDLP_MIXED) return PDLP_ERR_INVALID;
    if (p->m < 0 || p->n < 0 || p->m_ineq < 0 || p->m_ineq > p->m) return PDLP_ERR_INVALID;
    if (p->row0 < 0 || p->row1 < p->row0 || p->row1 > p->m) return PDLP_ERR_INVALID;
    if (p->col0 < 0 || p->col1 < p->col0 || p->col1 > p->n) return PDLP_ERR_INVALID;
    if (p->m >= INT32_MAX || p->n >= INT32_MAX) return PDLP_ERR_INVALID;
    return PDLP_OK;
}

// panel groups the row-sum scratch is sized for: splitting only pays when one workgroup per (largest) row block
// cannot fill 2 x 256 CUs, i.e. below about 10.5M rows
inline int64_t rowsum_groups(int64_t rows)
{
    if (rows <= (int64_t)512 * 40 * 128) return 32;               // (small shards: room for the local panels and several chunks' groups)
    return rows <= (int64_t)512 * 40 * 512 ? 8 : 1;
}

struct Carve {
    int64_t off = 0;
    int64_t take(int64_t bytes) { const int64_t o = off; off = align_up(off + bytes, 256); return o; }
};

// one layout function used by both the size query and pdlp_create
int64_t layout(const pdlp_problem* p, int64_t nnzK, int64_t nnzKT, int64_t* offs /*[48]*/)
{
    const int64_t es = p->dtype == PDLP_F32 ? 4 : 8;
    const int64_t nl = p->col1 - p->col0, ml = p->row1 - p->row0;
    Carve c;
    int k = 0;
    for (int i = 0; i < 3; ++i) offs[k++] = c.take(p->n * es);   // 0..2  xb
    for (int i = 0; i < 3; ++i) offs[k++] = c.take(p->m * es);   // 3..5  yb
    offs[k++] = c.take(p->n * es);                                // 6     xbar
    offs[k++] = c.take(nl * es);                                  // 7     x_sum
    offs[k++] = c.take(ml * es);                                  // 8     y_sum
    offs[k++] = c.take(nl * es);                                  // 9     x_last
    offs[k++] = c.take(ml * es);                                  // 10    y_last
    for (int i = 0; i < 3; ++i) offs[k++] = c.take(ml * es);     // 11..13 kx caches
    const int64_t pgrid = (int64_t)MAX_GRID * MAX_CHUNKS + LONG_GRID + (nl > ml ? nl : ml) / TNT + 2;  // CSR grid (+ long rows), one workgroup per >= 512 rows (tiled, rpt >= 1), or the epilogue launches of up to MAX_CHUNKS output pieces
    offs[k++] = c.take(pgrid * NACC * 8);                         // 14    partA
    offs[k++] = c.take(pgrid * NACC * 8);                         // 15    partB
    offs[k++] = c.take(PDLP_NRED * 8);                            // 16    red
    offs[k++] = c.take(PDLP_NSCAL * 8);                           // 17    sc
    offs[k++] = c.take((max_blocks(ml, nnzK) + 1) * 16);          // 18    schedule K  (pairs of 64-bit words)
    offs[k++] = c.take((max_blocks(nl, nnzKT) + 1) * 16);         // 19    schedule K'
    offs[k++] = c.take(rowsum_groups(nl > ml ? nl : ml) * ((nl > ml ? nl : ml) + (int64_t)TNT * TRPT_MAX_ANY) * es);   // 20  rowsum scratch
    for (int t = 0; t < 2; ++t) {                                 // 21..28 long rows of K, then of K'
        const int64_t nnz = t == 0 ? nnzK : nnzKT;
        offs[k++] = c.take(max_chunks(nnz) * 2 * 8);              //   chunk (first, end), 64-bit
        offs[k++] = c.take(max_long(nnz) * 4);                    //   rows
        offs[k++] = c.take((max_long(nnz) + 1) * 4);              //   chunk ranges
        offs[k++] = c.take(max_chunks(nnz) * es);                 //   chunk sums
    }
    offs[k++] = c.take(p->n * es);                                // 29    dx (infeasibility detection)
    offs[k++] = c.take(p->m * es);                                // 30    dy
    offs[k++] = c.take(nl * es);                                  // 31    lam_prev
    offs[k++] = c.take(nl * es);                                  // 32    K'dy
    offs[k++] = c.take(nl * es);                                  // 33    K'y from KKT(current)
    offs[k++] = c.take(nl * es);                                  // 34    K'y from KKT(average)
    const bool mixed = p->dtype == PDLP_MIXED;                    // delta mode: running K'y, float32 difference vectors
    offs[k++] = c.take(mixed ? nl * es : 0);                      // 35    ktyr
    offs[k++] = c.take(mixed ? p->n * 4 : 0);                     // 36    gdx
    offs[k++] = c.take(mixed ? p->m * 4 : 0);                     // 37    gdy
    offs[k++] = c.take(ml * es);                                  // 38    kx_sum  (running sum of w_k K x_k)
    offs[k++] = c.take(nl * es);                                  // 39    kty_sum (running sum of w_k K'y_k)
    const bool sharded = nl != p->n || ml != p->m;                // panel tables of the split products (sharded problems only)
    offs[k++] = c.take(sharded ? ((p->n >> 4) + 8) * 4 : 0);      // 40    panels of K by phase  (panel width >= 16 columns)
    offs[k++] = c.take(sharded ? ((p->m >> 4) + 8) * 4 : 0);      // 41    panels of K'
    offs[k++] = c.take((ml + 1) * 4);                             // 42    low words of K's row pointers (the CSR kernel's 4-byte reads)
    offs[k++] = c.take((nl + 1) * 4);                             // 43    ... of K''s
    return c.off;
}

int read_last_rowptr(const int64_t* rp, int64_t rows, int64_t* nnz, hipStream_t stream)
{
    // the arrays may just have been produced by kernels on the caller's stream (a non-blocking stream is not ordered
    // against the null stream's copy): read on that stream and wait
    int64_t v = 0;
    if (rows > 0) {
        HIP_TRY(hipMemcpyAsync(&v, rp + rows, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    *nnz = v;
    return PDLP_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int pdlp_abi_version(void) { return 17; }  // 17: pdlp_peer_* (direct exchange over HIP IPC), PDLP_OPT_BEGIN_INLINE; 16: pdlp_set_option (the library reads no environment variables), pdlp_mv_product, pdlp_mv_combine, pdlp_vec_sqdist, pdlp_probe_gather, pdlp_tile_limits reports the threads per workgroup, pdlp_primal_half_piece / pdlp_dual_half_piece (results of split products leave piece by piece), pdlp_trace_enable / pdlp_range_push / pdlp_range_pop (roctx), pdlp_adaptive_retry; 15: 64-bit row pointers, row-block bases of the tiles and schedule offsets (more than 2^31 non-zeros per handle); 14: pdlp_probe_stream_read; 13: chunked exchange (pdlp_set_exchange_chunks, pdlp_exchange_plan, pdlp_half_chunk); 12: count words of a tile laid out for coalesced loads; 11: pdlp_comm_load; 10: pdlp_set_anchors; 9: pdlp_attach_sorted; 8: running products, pdlp_flush_average(h, adaptive); 7: pdlp_comm_*; 6: remainder of a tiled matrix; 5: PDLP_MIXED, delta mode; 4: pdlp_tile_limits, pdlp_csr_div_cols takes nnz

const char* pdlp_strerror(int code)
{
    switch (code) {
        case PDLP_OK: return "ok";
        case PDLP_ERR_INVALID: return "invalid argument";
        case PDLP_ERR_WORKSPACE: return "workspace too small or not 256-byte aligned";
        case PDLP_ERR_STATE: return "call sequence violated";
        case PDLP_ERR_COMM:
            return g_rccl.GetErrorString && g_rccl.last_error ? g_rccl.GetErrorString((ncclResult_t)g_rccl.last_error)
                                                              : "RCCL could not be loaded or a collective failed";
        default: break;
    }
    if (code <= PDLP_ERR_HIP_BASE) return hipGetErrorString((hipError_t)(PDLP_ERR_HIP_BASE - code));
    return "unknown error";
}

int pdlp_workspace_bytes(const pdlp_problem* p, int64_t* bytes)
{
    int rc = check_problem(p);
    if (rc != PDLP_OK || !bytes) return PDLP_ERR_INVALID;
    HIP_TRY(hipSetDevice(p->device));
    int64_t nnzK = 0, nnzKT = 0, offs[48];
    if ((rc = read_last_rowptr(p->K_rowptr, p->row1 - p->row0, &nnzK, (hipStream_t)p->stream)) != PDLP_OK) return rc;
    if ((rc = read_last_rowptr(p->KT_rowptr, p->col1 - p->col0, &nnzKT, (hipStream_t)p->stream)) != PDLP_OK) return rc;
    *bytes = layout(p, nnzK, nnzKT, offs);
    return PDLP_OK;
}

int pdlp_create(pdlp_handle* out, const pdlp_problem* p, void* workspace, int64_t workspace_bytes)
{
    int rc = check_problem(p);
    if (rc != PDLP_OK || !out) return PDLP_ERR_INVALID;
    HIP_TRY(hipSetDevice(p->device));
    const int64_t nl = p->col1 - p->col0, ml = p->row1 - p->row0;
    std::vector<int64_t> rpK((size_t)ml + 1, 0), rpKT((size_t)nl + 1, 0);
    hipStream_t pstream = (hipStream_t)p->stream;
    if (ml > 0) HIP_TRY(hipMemcpyAsync(rpK.data(), p->K_rowptr, (size_t)(ml + 1) * 8, hipMemcpyDeviceToHost, pstream));
    if (nl > 0) HIP_TRY(hipMemcpyAsync(rpKT.data(), p->KT_rowptr, (size_t)(nl + 1) * 8, hipMemcpyDeviceToHost, pstream));
    HIP_TRY(hipStreamSynchronize(pstream));     // (ordered behind whatever produced the arrays on that stream)
    if (rpK[0] != 0 || rpKT[0] != 0) return PDLP_ERR_INVALID;
    int64_t offs[48];
    const int64_t need = layout(p, rpK[ml], rpKT[nl], offs);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255u)) return PDLP_ERR_WORKSPACE;

    pdlp_solver* h = new (std::nothrow) pdlp_solver();
    if (!h) return PDLP_ERR_INVALID;
    h->p = *p;
    h->stream = (hipStream_t)p->stream;
    h->es = p->dtype == PDLP_F32 ? 4 : 8;
    h->mixed = p->dtype == PDLP_MIXED;
    h->comm = nullptr; h->comm_rank = 0; h->comm_size = 1;
    h->xchunks = 1; h->cstream = nullptr; h->ev_vec = nullptr;
    for (auto& e : h->ev_chunk) e = nullptr;
    for (auto& e : h->ev_row) e = nullptr;
    h->ev_ar = nullptr;
    h->delta = false; h->anchors_valid = false; h->dy_folded = false;
    h->nl = nl;
    h->ml = ml;
    int64_t ie = p->m_ineq - p->row0;
    h->ineq_end = (int)(ie < 0 ? 0 : (ie > ml ? ml : ie));
    char* w = (char*)workspace;
    h->ws = w; h->ws_bytes = need;
    for (int i = 0; i < 3; ++i) h->xb[i] = w + offs[i];
    for (int i = 0; i < 3; ++i) h->yb[i] = w + offs[3 + i];
    h->ix_cur = 0; h->ix_prev = 1; h->ix_avg = 2;
    h->xbar = w + offs[6];
    h->x_sum = w + offs[7]; h->y_sum = w + offs[8]; h->x_last = w + offs[9]; h->y_last = w + offs[10];
    for (int i = 0; i < 3; ++i) h->kxb[i] = w + offs[11 + i];
    h->partA = (double*)(w + offs[14]); h->partB = (double*)(w + offs[15]);
    h->red = (double*)(w + offs[16]); h->sc = (double*)(w + offs[17]);
    h->sK.blk = (int64_t*)(w + offs[18]); h->sKT.blk = (int64_t*)(w + offs[19]);
    h->rowsum = (void*)(w + offs[20]);
    h->dxf = w + offs[29]; h->dyf = w + offs[30]; h->lam_prev = w + offs[31]; h->ktdy = w + offs[32];
    h->ktyb[0] = w + offs[33]; h->ktyb[1] = w + offs[34];
    h->ktyr = w + offs[35]; h->gdx = (float*)(w + offs[36]); h->gdy = (float*)(w + offs[37]);
    h->kx_sum = w + offs[38]; h->kty_sum = w + offs[39];
    if (nl != p->n || ml != p->m) {
        h->sK.ptab = (int32_t*)(w + offs[40]); h->sK.ptab_cap = (p->n >> 4) + 8;
        h->sKT.ptab = (int32_t*)(w + offs[41]); h->sKT.ptab_cap = (p->m >> 4) + 8;
    }
    h->since_reset = 0; h->kty_tail_done = false; h->avg_products = false; h->sums_broken = false; h->cur_kx_cached = false;
    h->no_running = false;
    h->kty_cur = -1;
    h->no_kty_reuse = false;
    h->split_local = h->split_other = 0;
    h->producer_pieces = true; h->begin_inline = false;
    h->range_sel = -1; h->range_cnt = 1;
    h->rs_stride = (nl > ml ? nl : ml) + (int64_t)TNT * TRPT_MAX_ANY;
    h->rs_groups = (int)rowsum_groups(nl > ml ? nl : ml);
    h->part_blocks = (int64_t)MAX_GRID * MAX_CHUNKS + LONG_GRID + (nl > ml ? nl : ml) / TNT + 2;
    h->kx_valid = false; h->cand_valid[0] = h->cand_valid[1] = false;
    h->last_gridA = h->last_gridB = 0;
    for (auto& g : h->graphs) g.valid = false;
    h->gstream = nullptr; h->ev_in = h->ev_out = nullptr;
    // opt-in (PDLP_GRAPH=1): on ROCm 7.2 / MI355X the replay measured 6-12 % SLOWER than direct launches on the small
    // LPs it was meant for (neos3-shaped: 17.8k vs 20.2k it/s; 1M x 1M, 5 nnz/row: 10.25k vs 10.86k it/s) -- the loop is
    // bound by dependent-kernel latency on the device, not by host launch cost -- and makes no difference on large ones
    const bool side = hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking) == hipSuccess &&
                      hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) == hipSuccess;
    if (!side) {                  // no side stream: no graph replay and no early local-panel products
        if (h->gstream) (void)hipStreamDestroy(h->gstream);
        h->gstream = nullptr;
        (void)hipGetLastError();
    }
    h->graph_ok = false;              // pdlp_set_option(PDLP_OPT_GRAPH) turns the replay on
    h->side_ok = side;
    h->use_split = false;

    h->sK.rplo = (uint32_t*)(w + offs[42]); h->sKT.rplo = (uint32_t*)(w + offs[43]);
    {
        std::vector<uint32_t> lo((size_t)(ml > nl ? ml : nl) + 1);
        for (int64_t i = 0; i <= ml; ++i) lo[(size_t)i] = (uint32_t)rpK[(size_t)i];
        if (hipMemcpyAsync(h->sK.rplo, lo.data(), (size_t)(ml + 1) * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { free_handle(h); return PDLP_ERR_HIP_BASE - 1; }
        for (int64_t i = 0; i <= nl; ++i) lo[(size_t)i] = (uint32_t)rpKT[(size_t)i];
        if (hipMemcpyAsync(h->sKT.rplo, lo.data(), (size_t)(nl + 1) * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { free_handle(h); return PDLP_ERR_HIP_BASE - 1; }
    }
    std::vector<int64_t> sched;
    build_schedule_host(rpK, ml, sched);
    h->sK.nblk = ml > 0 ? (int)sched.size() / 2 - 1 : 0;
    h->sK.grid = h->sK.nblk < MAX_GRID ? h->sK.nblk : MAX_GRID;
    auto upload = [&](void* dst, const void* src, size_t bytes) {      // ordered with later work on the caller's stream
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess;
    };
    rc = upload(h->sK.blk, sched.data(), sched.size() * 8) ? PDLP_OK : PDLP_ERR_HIP_BASE - 1;
    build_schedule_host(rpKT, nl, sched);
    h->sKT.nblk = nl > 0 ? (int)sched.size() / 2 - 1 : 0;
    h->sKT.grid = h->sKT.nblk < MAX_GRID ? h->sKT.nblk : MAX_GRID;
    if (rc == PDLP_OK && !upload(h->sKT.blk, sched.data(), sched.size() * 8)) rc = PDLP_ERR_HIP_BASE - 1;
    // rows longer than NNZ_CAP
    for (int t = 0; t < 2 && rc == PDLP_OK; ++t) {
        Schedule& sc = t == 0 ? h->sK : h->sKT;
        std::vector<int64_t> lch;
        std::vector<int32_t> lrow, lptr;
        build_long_rows_host(t == 0 ? rpK : rpKT, t == 0 ? ml : nl, lch, lrow, lptr);
        sc.lch = (int64_t*)(w + offs[21 + 4 * t]); sc.lrow = (int32_t*)(w + offs[22 + 4 * t]);
        sc.lptr = (int32_t*)(w + offs[23 + 4 * t]); sc.longpart = (void*)(w + offs[24 + 4 * t]);
        sc.nchunks = (int)(lch.size() / 2); sc.nlong = (int)lrow.size();
        sc.lgrid = sc.nlong > 0 ? (int)((sc.nlong + BLOCK - 1) / BLOCK < LONG_GRID ? (sc.nlong + BLOCK - 1) / BLOCK : LONG_GRID) : 0;
        const int64_t work = (int64_t)sc.nblk + sc.nchunks;
        sc.grid = (int)(work < MAX_GRID ? work : MAX_GRID);
        if (sc.nlong > 0) {
            if (!upload(sc.lch, lch.data(), lch.size() * 8) || !upload(sc.lrow, lrow.data(), lrow.size() * 4) ||
                !upload(sc.lptr, lptr.data(), lptr.size() * 4))
                rc = PDLP_ERR_HIP_BASE - 1;
        }
    }
    // zero every state vector, scratch and scalar (everything in front of the schedules)
    if (rc == PDLP_OK && hipMemsetAsync(w, 0, (size_t)offs[18], h->stream) != hipSuccess) rc = PDLP_ERR_HIP_BASE - 1;
    if (rc == PDLP_OK && hipMemsetAsync(w + offs[29], 0, (size_t)(offs[42] - offs[29]), h->stream) != hipSuccess) rc = PDLP_ERR_HIP_BASE - 1;   // (42, 43: the row pointers' low words, uploaded above)
    if (rc != PDLP_OK) { free_handle(h); return rc; }
    if (p->dtype == PDLP_F32) hipLaunchKernelGGL(k_set_step<float>, dim3(1), dim3(1), 0, h->stream, h->sc, 0.0, 1.0, 1.0, 0.0);
    else hipLaunchKernelGGL(k_set_step<double>, dim3(1), dim3(1), 0, h->stream, h->sc, 0.0, 1.0, 1.0, 0.0);
    *out = h;
    return PDLP_OK;
}

void pdlp_destroy(pdlp_handle h)
{
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    free_handle(h);
}

int pdlp_buffer_ptr(pdlp_handle h, int which, void** ptr)
{
    if (!h || !ptr) return PDLP_ERR_INVALID;
    switch (which) {
        case PDLP_BUF_X_CUR: *ptr = h->xb[h->ix_cur]; break;
        case PDLP_BUF_X_PREV: *ptr = h->xb[h->ix_prev]; break;
        case PDLP_BUF_XBAR: *ptr = h->xbar; break;
        case PDLP_BUF_X_AVG: *ptr = h->xb[h->ix_avg]; break;
        case PDLP_BUF_Y_CUR: *ptr = h->yb[h->ix_cur]; break;
        case PDLP_BUF_Y_PREV: *ptr = h->yb[h->ix_prev]; break;
        case PDLP_BUF_Y_AVG: *ptr = h->yb[h->ix_avg]; break;
        case PDLP_BUF_RED: *ptr = h->red; break;
        case PDLP_BUF_X_SUM: *ptr = h->x_sum; break;
        case PDLP_BUF_Y_SUM: *ptr = h->y_sum; break;
        case PDLP_BUF_SCALARS: *ptr = h->sc; break;
        case PDLP_BUF_DX: *ptr = h->dxf; break;
        case PDLP_BUF_DY: *ptr = h->dyf; break;
        case PDLP_BUF_LAM_PREV: *ptr = h->lam_prev; break;
        case PDLP_BUF_GDX: if (!h->mixed) return PDLP_ERR_STATE; *ptr = h->gdx; break;
        case PDLP_BUF_GDY: if (!h->mixed) return PDLP_ERR_STATE; *ptr = h->gdy; break;
        default: return PDLP_ERR_INVALID;
    }
    return PDLP_OK;
}

int pdlp_attach_tiles(pdlp_handle h, int transpose, const pdlp_tiles* t)
{
    if (!h) return PDLP_ERR_INVALID;
    Schedule& s = transpose ? h->sKT : h->sK;
    drop_graphs(h);               // captured launches name the old kernel and arrays
    if (!t) { s.tiled = false; return configure_split(h, transpose != 0); }
    const int64_t rows = transpose ? h->nl : h->ml;
    const int rpt_max = h->p.dtype == PDLP_F64 ? TileCfg<double, double>::RPT_MAX : TileCfg<float, float>::RPT_MAX;   // (mixed: float32 tiles)
    const int cap_max = h->p.dtype == PDLP_F64 ? TileCfg<double, double>::CAP : TileCfg<float, float>::CAP;
    if (t->rpt < 1 || t->rpt > rpt_max || t->cap > cap_max || t->lw < 4 || (((uint64_t)t->cap + 8u) << t->lw) > (1ull << 32)) return PDLP_ERR_INVALID;   // (slot, column) must pack into 32 bits
    const int64_t rb = (int64_t)TNT * t->rpt;
    if (t->nblk != (int)((rows + rb - 1) / rb) || t->npanel < 1) return PDLP_ERR_INVALID;
    if (t->groups < 1 || t->groups > h->rs_groups || t->groups > t->npanel) return PDLP_ERR_INVALID;
    if (t->nblk > h->part_blocks) return PDLP_ERR_INVALID;             // one slot of partial sums per workgroup
    // (every group owns at least one panel: the kernel splits the panels evenly, groups <= npanel was checked above)
    if (!t->idx || !t->val || !t->tile_ptr || !t->blk_base || !t->cnt) return PDLP_ERR_INVALID;
    if (((uintptr_t)t->idx & 15u) || ((uintptr_t)t->val & 15u) || ((uintptr_t)t->cnt & 15u)) return PDLP_ERR_INVALID;
    if (t->rem_rows_n < 0 || t->rem_segs_n < t->rem_rows_n) return PDLP_ERR_INVALID;
    if (t->rem_rows_n > 0 && (!t->rem_rows || !t->rem_rptr || !t->rem_sptr || !t->rem_col || !t->rem_val || !t->rem_work || !t->rem_extra ||
                              (h->mixed && !t->rem_extra_f32)))
        return PDLP_ERR_INVALID;
    s.t = *t;
    s.tiled = true;
    return configure_split(h, transpose != 0);
}

int pdlp_schedule_info(pdlp_handle h, int transpose, int32_t* nblk, const int64_t** blocks)
{
    if (!h || !nblk || !blocks) return PDLP_ERR_INVALID;
    const Schedule& s = transpose ? h->sKT : h->sK;
    *nblk = s.nblk;
    *blocks = s.blk;
    return PDLP_OK;
}

int pdlp_attach_sorted(pdlp_handle h, int transpose, const uint32_t* sidx, const void* sval, const int32_t* cbase)
{
    if (!h) return PDLP_ERR_INVALID;
    if ((sidx == nullptr) != (sval == nullptr) || (sidx == nullptr) != (cbase == nullptr)) return PDLP_ERR_INVALID;
    Schedule& s = transpose ? h->sKT : h->sK;
    drop_graphs(h);
    s.sidx = sidx; s.sval = sval; s.cbase = cbase;
    return PDLP_OK;
}

int pdlp_set_iterate(pdlp_handle h, const void* x_local, const void* y_local)
{
    if (!h || !x_local || !y_local) return PDLP_ERR_INVALID;
    HIP_TRY(hipMemcpyAsync(h->xb[h->ix_cur] + h->p.col0 * h->es, x_local, h->nl * h->es, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->yb[h->ix_cur] + h->p.row0 * h->es, y_local, h->ml * h->es, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->x_last, x_local, h->nl * h->es, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->y_last, y_local, h->ml * h->es, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(h->x_sum, 0, h->nl * h->es, h->stream));
    HIP_TRY(hipMemsetAsync(h->y_sum, 0, h->ml * h->es, h->stream));
    hipLaunchKernelGGL(k_reset_average, dim3(1), dim3(1), 0, h->stream, h->sc);
    HIP_TRY(hipGetLastError());
    h->kx_valid = false;
    h->cand_valid[0] = h->cand_valid[1] = false;
    h->kty_cur = -1;
    h->anchors_valid = false; h->dy_folded = false;
    HIP_TRY(hipMemsetAsync(h->kx_sum, 0, h->ml * h->es, h->stream));
    HIP_TRY(hipMemsetAsync(h->kty_sum, 0, h->nl * h->es, h->stream));
    h->since_reset = 0; h->kty_tail_done = false; h->avg_products = false; h->sums_broken = false; h->cur_kx_cached = false;
    return PDLP_OK;
}

int pdlp_get_iterate(pdlp_handle h, int which, void* x_local, void* y_local)
{
    if (!h) return PDLP_ERR_INVALID;
    const int ix = which == PDLP_CUR ? h->ix_cur : (which == PDLP_AVG ? h->ix_avg : (which == PDLP_PREV ? h->ix_prev : -1));
    if (ix < 0) return PDLP_ERR_INVALID;
    if (x_local) HIP_TRY(hipMemcpyAsync(x_local, h->xb[ix] + h->p.col0 * h->es, h->nl * h->es, hipMemcpyDeviceToDevice, h->stream));
    if (y_local) HIP_TRY(hipMemcpyAsync(y_local, h->yb[ix] + h->p.row0 * h->es, h->ml * h->es, hipMemcpyDeviceToDevice, h->stream));
    return PDLP_OK;
}

int pdlp_set_step(pdlp_handle h, double eta, double omega, double theta, int64_t iteration)
{
    if (!h || !(omega > 0.0)) return PDLP_ERR_INVALID;
    if (h->p.dtype == PDLP_F32)
        hipLaunchKernelGGL(k_set_step<float>, dim3(1), dim3(1), 0, h->stream, h->sc, eta, omega, theta, (double)iteration);
    else
        hipLaunchKernelGGL(k_set_step<double>, dim3(1), dim3(1), 0, h->stream, h->sc, eta, omega, theta, (double)iteration);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_set_omega(pdlp_handle h, double omega)
{
    if (!h || !(omega > 0.0)) return PDLP_ERR_INVALID;
    if (h->p.dtype == PDLP_F32) hipLaunchKernelGGL(k_set_omega<float>, dim3(1), dim3(1), 0, h->stream, h->sc, omega);
    else hipLaunchKernelGGL(k_set_omega<double>, dim3(1), dim3(1), 0, h->stream, h->sc, omega);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_get_scalars(pdlp_handle h, double out[PDLP_NSCAL])
{
    if (!h || !out) return PDLP_ERR_INVALID;
    HIP_TRY(hipMemcpyAsync(out, h->sc, PDLP_NSCAL * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PDLP_OK;
}

int pdlp_primal_half(pdlp_handle h, int adaptive)
{
    if (!h) return PDLP_ERR_INVALID;
    if (h->delta) return delta_primal_half(h, adaptive);
    return DISPATCH(h, primal_half_t, h, adaptive);
}

int pdlp_dual_half(pdlp_handle h, int adaptive)
{
    if (!h) return PDLP_ERR_INVALID;
    if (h->delta) return delta_dual_half(h, adaptive);
    return DISPATCH(h, dual_half_t, h, adaptive);
}

// One output piece of a half-step (see pdlp_hip.h).  Pieces in order, 0 .. pieces-1; the half-step is complete after the last.
static int half_piece(pdlp_handle h, bool dual, int adaptive, int piece, int pieces)
{
    if (!h || pieces < 1 || pieces > MAX_CHUNKS || piece < 0 || piece >= pieces) return PDLP_ERR_INVALID;
    h->range_sel = piece;
    h->range_cnt = pieces;
    const int rc = dual ? pdlp_dual_half(h, adaptive) : pdlp_primal_half(h, adaptive);
    h->range_sel = -1;
    h->range_cnt = 1;
    return rc;
}

int pdlp_primal_half_piece(pdlp_handle h, int adaptive, int piece, int pieces) { return half_piece(h, false, adaptive, piece, pieces); }
int pdlp_dual_half_piece(pdlp_handle h, int adaptive, int piece, int pieces) { return half_piece(h, true, adaptive, piece, pieces); }

int pdlp_primal_half_begin(pdlp_handle h)
{
    if (!h) return PDLP_ERR_INVALID;
    if (h->delta && !h->anchors_valid) return PDLP_OK;
    return DISPATCH(h, half_begin_t, h, true, h->yb[h->ix_cur]);
}

int pdlp_dual_half_begin(pdlp_handle h, int adaptive)
{
    if (!h) return PDLP_ERR_INVALID;
    if (h->delta && !h->anchors_valid) return PDLP_OK;
    if (!h->delta && !h->kx_valid) return PDLP_OK;        // the K x refresh ahead of this half-step uses the same scratch
    return DISPATCH(h, half_begin_t, h, false, h->xbar);
}

int pdlp_split_info(pdlp_handle h, int transpose, int32_t out[4])
{
    if (!h || !out) return PDLP_ERR_INVALID;
    const Schedule& s = transpose ? h->sKT : h->sK;
    out[0] = s.loc_pa; out[1] = s.loc_pb; out[2] = s.slotsA; out[3] = s.slotsB;
    return PDLP_OK;
}

int pdlp_set_option(pdlp_handle h, int option, int64_t value)
{
    if (!h) return PDLP_ERR_INVALID;
    if (h->sK.pending || h->sKT.pending) return PDLP_ERR_STATE;       // not in the middle of a split product
    switch (option) {
        case PDLP_OPT_RUNNING_KKT: h->no_running = value == 0; return PDLP_OK;
        case PDLP_OPT_KTY_REUSE: h->no_kty_reuse = value == 0; return PDLP_OK;
        case PDLP_OPT_GRAPH:
            drop_graphs(h);
            h->graph_ok = value != 0 && h->side_ok && !h->comm;
            return (value != 0 && !h->graph_ok) ? PDLP_ERR_STATE : PDLP_OK;
        case PDLP_OPT_BEGIN_INLINE: h->begin_inline = value != 0; return PDLP_OK;
        case PDLP_OPT_PEER_EXCHANGE: h->peer.enabled = value != 0; return PDLP_OK;
        case PDLP_OPT_PEER_LOCAL_FIRST: h->peer.local_first = value != 0; return PDLP_OK;
        case PDLP_OPT_PEER_PUSH: h->peer.push = value != 0; return PDLP_OK;
        case PDLP_OPT_PEER_TIMEOUT_MS:
            if (value < 1 || value > 3600000) return PDLP_ERR_INVALID;
            h->peer.limit_ticks = (long long)value * 100000;            // (the wait kernel counts a 100 MHz clock)
            return PDLP_OK;
        case PDLP_OPT_PRODUCER_PIECES: {
            h->producer_pieces = value != 0;
            drop_graphs(h);
            int rc = configure_split(h, false);
            if (rc == PDLP_OK) rc = configure_split(h, true);
            return rc;
        }
        case PDLP_OPT_SPLIT_SLOTS: {
            const int a = (int)(value & 0xffff), b = (int)((value >> 16) & 0xffff);
            if (value != 0 && (a < 1 || b < 1 || a + b > h->rs_groups)) return PDLP_ERR_INVALID;
            h->split_local = a;
            h->split_other = b;
            drop_graphs(h);
            int rc = configure_split(h, false);
            if (rc == PDLP_OK) rc = configure_split(h, true);
            return rc;
        }
        default: return PDLP_ERR_INVALID;
    }
}

int pdlp_set_exchange_chunks(pdlp_handle h, int chunks)
{
    if (!h || chunks < 1 || chunks > MAX_CHUNKS) return PDLP_ERR_INVALID;
    if (h->sK.pending || h->sKT.pending) return PDLP_ERR_STATE;       // not in the middle of a split product
    h->xchunks = chunks;
    drop_graphs(h);
    int rc = configure_split(h, false);
    if (rc == PDLP_OK) rc = configure_split(h, true);
    return rc;
}

int pdlp_exchange_plan(pdlp_handle h, int transpose, int32_t* nchunks, int64_t bounds[5])
{
    if (!h || !nchunks || !bounds) return PDLP_ERR_INVALID;
    // (a function of the block length and the requested count only: the same on every rank, whether or not this rank's own
    // product is split -- the pieces are collectives)
    int64_t sb[MAX_PHASE];
    *nchunks = plan_bounds(transpose ? h->ml : h->nl, h->xchunks, sb);
    for (int c = 0; c < 5; ++c) bounds[c] = sb[c];
    return PDLP_OK;
}

int pdlp_half_chunk(pdlp_handle h, int transpose, int chunk)
{
    if (!h || chunk < 0 || chunk >= MAX_CHUNKS) return PDLP_ERR_INVALID;
    if (transpose) return DISPATCH(h, half_chunk_t, h, true, h->yb[h->ix_cur], chunk);
    return DISPATCH(h, half_chunk_t, h, false, h->xbar, chunk);
}

int pdlp_tile_limits(pdlp_handle h, int32_t out[6])
{
    if (!h || !out) return PDLP_ERR_INVALID;
    out[4] = TNT;
    out[5] = 0;
    out[0] = h->rs_groups;
    out[1] = (int32_t)(h->part_blocks > INT32_MAX ? INT32_MAX : h->part_blocks);
    out[2] = h->p.dtype == PDLP_F64 ? TileCfg<double, double>::RPT_MAX : TileCfg<float, float>::RPT_MAX;
    out[3] = h->p.dtype == PDLP_F64 ? TileCfg<double, double>::CAP : TileCfg<float, float>::CAP;
    return PDLP_OK;
}

int pdlp_adaptive_retry(pdlp_handle h)
{
    if (!h) return PDLP_ERR_INVALID;
    if (h->delta || h->sK.pending || h->sKT.pending || h->graph_ok) return PDLP_ERR_STATE;   // (float32 / float64 handles, between iterations)
    // the scalars: the rejected trial's weight leaves eta_total again, nothing is pending (the trial's epilogues have folded the
    // previous iterate's weight into the sums: the repeated half-steps must add nothing), k goes back; eta keeps the rule's eta'
    hipLaunchKernelGGL(k_retry_scalars, dim3(1), dim3(1), 0, h->stream, h->sc);
    HIP_TRY(hipGetLastError());
    // the iterate: the trial wrote x+, y+ into the "previous" buffers and made them current; the old (x, y) is intact in what is now
    // the previous pair
    const int t = h->ix_cur;
    h->ix_cur = h->ix_prev;
    h->ix_prev = t;
    if (h->since_reset > 0) --h->since_reset;
    h->kx_valid = false;              // the carried K x now belongs to the rejected x+: recomputed by the next dual half-step
    h->sums_broken = true;            // (the running products of this period saw the rejected trial: the checks multiply instead)
    h->cand_valid[0] = h->cand_valid[1] = false;
    h->kty_cur = -1; h->kty_tail_done = false; h->avg_products = false;
    return PDLP_OK;
}

int pdlp_adaptive_reduce(pdlp_handle h)
{
    if (!h) return PDLP_ERR_INVALID;
    if (h->p.dtype == PDLP_F32)
        hipLaunchKernelGGL(k_adaptive_reduce_update<float>, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, h->last_gridA, h->partB,
                           h->last_gridB, h->red, h->sc, 0);
    else
        hipLaunchKernelGGL(k_adaptive_reduce_update<double>, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, h->last_gridA, h->partB,
                           h->last_gridB, h->red, h->sc, 0);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_adaptive_update(pdlp_handle h)
{
    if (!h) return PDLP_ERR_INVALID;
    if (h->p.dtype == PDLP_F32) hipLaunchKernelGGL(k_adaptive_update<float>, dim3(1), dim3(1), 0, h->stream, h->sc, h->red);
    else hipLaunchKernelGGL(k_adaptive_update<double>, dim3(1), dim3(1), 0, h->stream, h->sc, h->red);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

namespace {

int iterate_direct(pdlp_handle h, int iters, int adaptive)
{
    int rc;
    for (int it = 0; it < iters; ++it) {
        if ((rc = pdlp_primal_half(h, adaptive)) != PDLP_OK) return rc;
        if ((rc = pdlp_dual_half(h, adaptive)) != PDLP_OK) return rc;
        if (adaptive) {
            if (h->p.dtype == PDLP_F32)
                hipLaunchKernelGGL(k_adaptive_reduce_update<float>, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, h->last_gridA,
                                   h->partB, h->last_gridB, h->red, h->sc, 1);
            else
                hipLaunchKernelGGL(k_adaptive_reduce_update<double>, dim3(1), dim3(BLOCK), 0, h->stream, h->partA, h->last_gridA,
                                   h->partB, h->last_gridB, h->red, h->sc, 1);
        }
    }
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

// Graph replay was asked for (PDLP_GRAPH) and cannot be had: say so once per process -- the iteration falls back to plain launches,
// which is correct but slower on small LPs, and would otherwise show up only as a slower benchmark.
void graph_abandoned(const char* why)
{
    static bool said = false;
    if (!said) std::fprintf(stderr, "libpdlp_hip: PDLP_GRAPH: graph capture abandoned (%s); iterating with direct launches\n", why);
    said = true;
}

// the executable graph of two iterations from the current buffer roles (captured on first use), or nullptr
pdlp_solver::IterGraph* pair_graph(pdlp_handle h, int adaptive)
{
    pdlp_solver::IterGraph* slot = nullptr;
    for (auto& g : h->graphs) {
        if (g.valid && g.ix_cur == h->ix_cur && g.ix_prev == h->ix_prev && g.adaptive == adaptive) return &g;
        if (!g.valid && !slot) slot = &g;
    }
    if (!slot) return nullptr;
    // capture: the launch code runs unchanged against the library's stream; host-side roles are put back afterwards
    const int ix_cur = h->ix_cur, ix_prev = h->ix_prev, gA = h->last_gridA, gB = h->last_gridB;
    const bool kxv = h->kx_valid, c0 = h->cand_valid[0], c1 = h->cand_valid[1];
    const int64_t sr = h->since_reset;
    const bool ktd = h->kty_tail_done;
    hipStream_t user = h->stream;
    if (hipStreamBeginCapture(h->gstream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        graph_abandoned("hipStreamBeginCapture failed");
        h->graph_ok = false;
        h->sums_broken = true;        // (the running sums were not kept while replay was on: no running average before the next restart)
        return nullptr;
    }
    h->stream = h->gstream;
    const int rc = iterate_direct(h, 2, adaptive);
    h->stream = user;
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(h->gstream, &graph);
    h->ix_cur = ix_cur; h->ix_prev = ix_prev; h->last_gridA = gA; h->last_gridB = gB;
    h->kx_valid = kxv; h->cand_valid[0] = c0; h->cand_valid[1] = c1;
    h->since_reset = sr; h->kty_tail_done = ktd;
    if (rc != PDLP_OK || e != hipSuccess || !graph ||
        hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        graph_abandoned(rc != PDLP_OK ? "a launch failed during capture" : "hipStreamEndCapture / hipGraphInstantiate failed");
        h->graph_ok = false;
        h->sums_broken = true;        // (the running sums were not kept while replay was on: no running average before the next restart)
        return nullptr;
    }
    (void)hipGraphDestroy(graph);
    slot->valid = true; slot->ix_cur = ix_cur; slot->ix_prev = ix_prev; slot->adaptive = adaptive;
    return slot;
}

}  // namespace

namespace {

// all-gather of a full-length vector whose block of this rank is in place (equal blocks: rank r's block starts at r * count)
int comm_all_gather(pdlp_handle h, void* full, int64_t count, bool f32)
{
    char* base = (char*)full;
    const size_t esz = f32 ? 4 : 8;
    RCCL_TRY(g_rccl.AllGather(base + (size_t)h->comm_rank * count * esz, base, (size_t)count, f32 ? ncclFloat32 : ncclFloat64, h->comm,
                              h->stream));
    return PDLP_OK;
}

// the exchange of one gathered vector in the chunks of its product's plan: chunk c = elements [sb[c], sb[c+1]) of every rank's
// block, as one group of in-place broadcasts (one root per rank) on the communication stream; ev_chunk[c] marks its arrival
int comm_exchange_piece(pdlp_handle h, int c, const int64_t* sb, void* full, int64_t block, bool f32, hipEvent_t ready)
{
    const size_t esz = f32 ? 4 : 8;
    if (g_roctx.level > 0 && g_roctx.push) (void)g_roctx.push("pdlp: exchange piece (grouped broadcasts issued)");
    struct Pop { ~Pop() { if (g_roctx.level > 0 && g_roctx.pop) (void)g_roctx.pop(); } } pop_;
    HIP_TRY(hipStreamWaitEvent(h->cstream, ready, 0));          // this rank's part of the piece is final
    const int64_t lo = sb[c], cnt = sb[c + 1] - sb[c];
    if (cnt > 0) {
        RCCL_TRY(g_rccl.GroupStart());
        for (int q = 0; q < h->comm_size; ++q) {
            char* ptr = (char*)full + ((size_t)q * block + lo) * esz;
            const ncclResult_t r = g_rccl.Broadcast(ptr, ptr, (size_t)cnt, f32 ? ncclFloat32 : ncclFloat64, q, h->comm, h->cstream);
            if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); g_rccl.last_error = (int)r; return PDLP_ERR_COMM; }
        }
        RCCL_TRY(g_rccl.GroupEnd());
    }
    HIP_TRY(hipEventRecord(h->ev_chunk[c], h->cstream));
    return PDLP_OK;
}

int comm_exchange_chunked(pdlp_handle h, int C, const int64_t* sb, void* full, int64_t block, bool f32)
{
    const size_t esz = f32 ? 4 : 8;
    HIP_TRY(hipEventRecord(h->ev_vec, h->stream));               // this rank's block is final
    HIP_TRY(hipStreamWaitEvent(h->cstream, h->ev_vec, 0));
    for (int c = 0; c < C; ++c) {
        const int64_t lo = sb[c], cnt = sb[c + 1] - sb[c];
        if (cnt > 0) {
            RCCL_TRY(g_rccl.GroupStart());
            for (int q = 0; q < h->comm_size; ++q) {
                char* ptr = (char*)full + ((size_t)q * block + lo) * esz;
                const ncclResult_t r = g_rccl.Broadcast(ptr, ptr, (size_t)cnt, f32 ? ncclFloat32 : ncclFloat64, q, h->comm, h->cstream);
                if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); g_rccl.last_error = (int)r; return PDLP_ERR_COMM; }
            }
            RCCL_TRY(g_rccl.GroupEnd());
        }
        HIP_TRY(hipEventRecord(h->ev_chunk[c], h->cstream));
    }
    return PDLP_OK;
}

// one half-step of a sharded iteration with the exchange of its input in front: K xbar (transpose 0) or K'y (1, not after the
// last iteration of the call).  Chunked plans: the chunks travel on the communication stream, and the handle's stream multiplies
// the panels a chunk completes as soon as it has arrived; the last chunk's panels, the sum and the epilogue are the half-step.
int sharded_exchange_and_begin(pdlp_handle h, bool transpose, int adaptive, bool begin, bool pieces_sent = false)
{
    int rc;
    const bool vec32 = h->p.dtype == PDLP_F32;
    void* full = transpose ? (h->delta ? (void*)h->gdy : (void*)h->yb[h->ix_cur]) : (h->delta ? (void*)h->gdx : (void*)h->xbar);
    const int64_t block = transpose ? h->ml : h->nl;
    const bool f32 = h->delta || vec32;
    // the panels that meet this rank's own block are multiplied while the other blocks are on the wire: on the handle's own stream when
    // the pieces are under way on the communication stream already, else on the side stream (the all-gather below is in stream order)
    const bool saved_inline = h->begin_inline;
    h->begin_inline = pieces_sent;
    rc = begin ? (transpose ? pdlp_primal_half_begin(h) : pdlp_dual_half_begin(h, adaptive)) : PDLP_OK;
    h->begin_inline = saved_inline;
    if (rc != PDLP_OK) return rc;
    // (the shape of the exchange must not depend on anything rank local -- every rank issues the same collectives)
    int64_t sb[MAX_PHASE];
    const int C = plan_bounds(block, h->xchunks, sb);
    const bool chunked = C > 1 && h->cstream && g_rccl.Broadcast && g_rccl.GroupStart && g_rccl.GroupEnd;
    if (!chunked) return comm_all_gather(h, full, block, f32);
    // (pieces_sent: the half-step that produced the vector was issued piece by piece and every piece's broadcasts went out behind
    //  its rows -- sharded_half_in_pieces; only the consumer's side is left to do)
    if (!pieces_sent && (rc = comm_exchange_chunked(h, C, sb, full, block, f32)) != PDLP_OK) return rc;
    for (int c = 0; c + 1 < C; ++c) {
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_chunk[c], 0));
        if ((rc = pdlp_half_chunk(h, transpose ? 1 : 0, c)) != PDLP_OK) return rc;
    }
    HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_chunk[C - 1], 0));
    return PDLP_OK;
}

// A half-step whose result travels in pieces: piece r's rows (last phase of the product + epilogue), an event, and piece r's
// broadcasts on the communication stream behind that event -- they run while piece r + 1's rows are still being multiplied.
// Returns (through *sent) whether the pieces went out this way; if not, the caller exchanges the vector after the half-step.
// The shape of the collectives is the plan's alone: every rank issues the same groups whether or not its own product is split.
int sharded_half_in_pieces(pdlp_handle h, bool dual, int adaptive, bool* sent)
{
    int rc;
    *sent = false;
    const bool vec32 = h->p.dtype == PDLP_F32;
    // the vector this half-step writes and the exchange it feeds: primal -> xbar (input of K xbar), dual -> y (input of K'y)
    void* full = dual ? (h->delta ? (void*)h->gdy : (void*)h->yb[h->ix_prev]) : (h->delta ? (void*)h->gdx : (void*)h->xbar);
    const int64_t block = dual ? h->ml : h->nl;
    const bool f32 = h->delta || vec32;
    int64_t sb[MAX_PHASE];
    const int C = plan_bounds(block, h->xchunks, sb);
    const bool chunked = C > 1 && h->cstream && h->producer_pieces && g_rccl.Broadcast && g_rccl.GroupStart && g_rccl.GroupEnd;
    if (!chunked) return dual ? pdlp_dual_half(h, adaptive) : pdlp_primal_half(h, adaptive);
    for (int c = 0; c < C; ++c) {
        if ((rc = half_piece(h, dual, adaptive, c, C)) != PDLP_OK) return rc;
        HIP_TRY(hipEventRecord(h->ev_row[c], h->stream));
        if ((rc = comm_exchange_piece(h, c, sb, full, block, f32, h->ev_row[c])) != PDLP_OK) return rc;
    }
    *sent = true;
    return PDLP_OK;
}

// the iterations of a sharded problem with the exchange inside the library: the same sequence as PdlpEngine.iterate drives
// through torch.distributed (engine.py), all of it enqueued on the handle's streams -- one call per restart period, no host
// work between the kernels and the collectives
int iterate_sharded(pdlp_handle h, int iters, int adaptive)
{
    int rc;
    const bool vec32 = h->p.dtype == PDLP_F32;
    if (h->delta && iters > 0 && !h->anchors_valid) {
        if ((rc = comm_all_gather(h, h->xb[h->ix_cur], h->nl, vec32)) != PDLP_OK) return rc;
        if ((rc = comm_all_gather(h, h->yb[h->ix_cur], h->ml, vec32)) != PDLP_OK) return rc;
        if ((rc = delta_refresh(h)) != PDLP_OK) return rc;
    }
    for (int it = 0; it < iters; ++it) {
        bool sent = false;
        if ((rc = sharded_half_in_pieces(h, false, adaptive, &sent)) != PDLP_OK) return rc;
        if ((rc = sharded_exchange_and_begin(h, false, adaptive, true, sent)) != PDLP_OK) return rc;      // xbar (delta mode: x+ - x)
        if ((rc = sharded_half_in_pieces(h, true, adaptive, &sent)) != PDLP_OK) return rc;
        // the step-size rule's three sums: with the pieces on the communication stream the all-reduce queues up behind them there
        // and runs while the handle's stream multiplies the panels the pieces complete (same sums, same values: only the order in
        // which independent work is enqueued changes)
        const bool ar_early = adaptive && sent && h->ev_ar;
        // (the kernel that adds up this rank's three sums needs only the half-steps' partial sums: it runs while y is on the wire)
        if (adaptive && (rc = pdlp_adaptive_reduce(h)) != PDLP_OK) return rc;
        if (ar_early) {
            HIP_TRY(hipEventRecord(h->ev_vec, h->stream));
            HIP_TRY(hipStreamWaitEvent(h->cstream, h->ev_vec, 0));
            RCCL_TRY(g_rccl.AllReduce(h->red, h->red, 3, ncclFloat64, ncclSum, h->comm, h->cstream));
            HIP_TRY(hipEventRecord(h->ev_ar, h->cstream));
        }
        // the new y (delta mode: y+ - y) -- final: a rejected adaptive step is kept, quirk Q1; its product starts only if
        // another iteration follows in this call
        if ((rc = sharded_exchange_and_begin(h, true, adaptive, it + 1 < iters, sent)) != PDLP_OK) return rc;
        if (ar_early) {
            HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_ar, 0));
            if ((rc = pdlp_adaptive_update(h)) != PDLP_OK) return rc;
        } else if (adaptive) {
            RCCL_TRY(g_rccl.AllReduce(h->red, h->red, 3, ncclFloat64, ncclSum, h->comm, h->stream));
            if ((rc = pdlp_adaptive_update(h)) != PDLP_OK) return rc;
        }
    }
    if (!adaptive && iters > 0) return pdlp_fixed_advance(h, iters);
    return PDLP_OK;
}

// ---- direct exchange: no collective in the iteration -------------------------------------------------------------------------
// "everything this rank has stored into your vectors up to now is complete" to every peer (and, with_sums, this rank's three sums of
// the step-size rule): one tiny kernel behind the half-step that did the storing
int peer_signal(pdlp_handle h, bool with_sums, hipStream_t stream = nullptr)
{
    pdlp_solver::Peer& P = h->peer;
    if (!stream) stream = h->stream;
    PeerSignal sg{};
    for (int i = 0; i < P.n; ++i) { sg.flag[i] = P.flag[i]; sg.sums[i] = P.sums[i]; }
    sg.own_sums = (double*)(P.box + BOX_SUMS_AT) + (size_t)P.rank * BOX_SUMS_STRIDE;
    sg.own_flag = (uint32_t*)P.box + (size_t)P.rank * BOX_FLAG_STRIDE;
    sg.n = P.n;
    sg.seq = ++P.seq;
    if (!with_sums)
        hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, stream, sg);
    else if (h->p.dtype == PDLP_F32)      // (pdlp_adaptive_reduce and the signal in one launch)
        hipLaunchKernelGGL(k_adaptive_reduce_signal<float>, dim3(1), dim3(BLOCK), 0, stream, h->partA, h->last_gridA, h->partB,
                           h->last_gridB, h->red, h->sc, sg);
    else
        hipLaunchKernelGGL(k_adaptive_reduce_signal<double>, dim3(1), dim3(BLOCK), 0, stream, h->partA, h->last_gridA, h->partB,
                           h->last_gridB, h->red, h->sc, sg);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

// bytes per element of target vector v and the start of this rank's block in it
size_t peer_block_start(pdlp_handle h, int v)
{
    const size_t esz = v >= 4 ? 4 : h->es;
    return (size_t)(v == 0 || v == 4 ? h->p.col0 : h->p.row0) * esz;
}

// push form: behind everything the handle's stream has enqueued so far, on the side stream: this rank's block of vector v (0 xbar,
// 1..3 the y buffers, 4 gdx, 5 gdy) into every peer's copy, then the signal (with the step-size rule's sums for the y exchange)
int peer_push_and_signal(pdlp_handle h, int v, bool with_sums)
{
    pdlp_solver::Peer& P = h->peer;
    if (!P.pstream) {
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&P.pstream, hipStreamNonBlocking, greatest));
        HIP_TRY(hipEventCreateWithFlags(&P.ev_push, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(P.ev_push, h->stream));
    HIP_TRY(hipStreamWaitEvent(P.pstream, P.ev_push, 0));
    const bool x_side = v == 0 || v == 4;
    const int64_t len = x_side ? h->nl : h->ml;
    const char* base = v == 0 ? h->xbar : (v == 4 ? (const char*)h->gdx : (v == 5 ? (const char*)h->gdy : h->yb[v - 1]));
    const char* src = base + peer_block_start(h, v);
    // (a modest grid: the copy is link-bound.  Where the own-block launch fills the chip -- 2 ranks: 502 workgroups holding every
    //  register of every CU -- the copy only gets going when those workgroups retire, and a grid small enough for the slots they
    //  leave is too slow to feed a link: measured, profiles/r05_multi_gpu/README.md.  The form pays at 4 ranks.)
    const int grid = (int)((len + BLOCK - 1) / BLOCK < 128 ? (len + BLOCK - 1) / BLOCK : 128);
    if (len > 0) {
        if (v >= 4 || h->es == 4) {
            PeerPush<float> d{};
            for (int i = 0; i < P.n; ++i) d.dst[i] = (float*)P.out[v][i];
            d.n = P.n;
            hipLaunchKernelGGL(k_peer_push<float>, dim3(grid), dim3(BLOCK), 0, P.pstream, (const float*)src, d, len);
        } else {
            PeerPush<double> d{};
            for (int i = 0; i < P.n; ++i) d.dst[i] = (double*)P.out[v][i];
            d.n = P.n;
            hipLaunchKernelGGL(k_peer_push<double>, dim3(grid), dim3(BLOCK), 0, P.pstream, (const double*)src, d, len);
        }
    }
    return peer_signal(h, with_sums, P.pstream);
}

// the handle's stream waits until every peer has signalled the exchange just signalled by this rank (bounded: k_peer_wait);
// then_update: the step-size rule from all ranks' sums in the same launch
int peer_wait(pdlp_handle h, bool then_update = false)
{
    pdlp_solver::Peer& P = h->peer;
    if (!then_update)
        hipLaunchKernelGGL(k_peer_wait, dim3(PEER_WAIT_BLOCKS), dim3(64), 0, h->stream, (const uint32_t*)P.box, P.world, P.rank, P.seq, P.limit_ticks, P.err_dev);
    else if (h->p.dtype == PDLP_F32)
        hipLaunchKernelGGL(k_peer_wait_adaptive_update<float>, dim3(PEER_WAIT_BLOCKS), dim3(64), 0, h->stream, (const uint32_t*)P.box, P.world, P.rank, P.seq,
                           P.limit_ticks, P.err_dev, h->sc, h->red);
    else
        hipLaunchKernelGGL(k_peer_wait_adaptive_update<double>, dim3(PEER_WAIT_BLOCKS), dim3(64), 0, h->stream, (const uint32_t*)P.box, P.world, P.rank, P.seq,
                           P.limit_ticks, P.err_dev, h->sc, h->red);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

// The iterations of a sharded problem WITHOUT collectives: a half-step's epilogue stores its block of xbar / y (delta mode: of the
// float32 differences) into every peer's copy as it computes it (PeerOut), a one-wave kernel raises this rank's flag in the peers'
// mailboxes, a one-wave kernel waits for the peers' flags, and the next product follows (PDLP_OPT_PEER_LOCAL_FIRST: split, the panels
// that meet the own block between signal and wait -- cover for ranks that finish at different times, at the price of a launch that
// cannot fill the chip: the own block is 1/world of the panels).  The step-size rule's three sums travel with the flag of the y exchange and are added in rank
// order by every rank.  One stream, no events (the push form: a side stream for the copy kernel); per half-step the exchange adds two launches of a few microseconds to the critical
// path (tools/src/ipc_probe.hip: 5.8 us for the pair between two processes) where an all-gather adds its whole duration.
// Same arithmetic as iterate_sharded / PdlpEngine.iterate: identical bits in fixed-step mode; in adaptive mode up to the order in
// which the ranks' sums are added (rank order here, the collective's order there; two ranks: identical).
int iterate_peer(pdlp_handle h, int iters, int adaptive)
{
    int rc = PDLP_OK;
    pdlp_solver::Peer& P = h->peer;
    if (h->delta && iters > 0 && !h->anchors_valid) return PDLP_ERR_STATE;       // (the caller refreshes the anchors: that needs gathers)
    if (iters == 0) return PDLP_OK;
    // entry handshake: a peer's first block may only arrive once this rank's stream has reached the iterations -- whatever this rank
    // did with its vectors before (a restart check, a restart) is behind the flag
    if ((rc = peer_signal(h, false)) != PDLP_OK || (rc = peer_wait(h)) != PDLP_OK) return rc;
    const bool saved_inline = h->begin_inline;
    h->begin_inline = true;                       // (the own-block panels go onto the handle's stream)
    // push form: the epilogues store locally; a copy kernel on the side stream carries the block to the peers while the handle's
    // stream multiplies the own block's panels -- where those panels are a good part of the product (2, 4 ranks) they hide the
    // transfer, which the stores of an epilogue, issued in the last microseconds of a half-step on the same stream, cannot
    const bool push = P.push;
    const bool first = P.local_first || push;
    P.active = !push;
    for (int it = 0; it < iters && rc == PDLP_OK; ++it) {
        if ((rc = pdlp_primal_half(h, adaptive)) != PDLP_OK) break;              // stores xbar (x+ - x) into the peers
        if ((rc = push ? peer_push_and_signal(h, h->delta ? 4 : 0, false) : peer_signal(h, false)) != PDLP_OK) break;
        if (first && (rc = pdlp_dual_half_begin(h, adaptive)) != PDLP_OK) break;              // K's panels over the own block of xbar
        if ((rc = peer_wait(h)) != PDLP_OK) break;
        if ((rc = pdlp_dual_half(h, adaptive)) != PDLP_OK) break;                // stores the new y (y+ - y) into the peers
        // (adaptive: this rank's three sums travel with the flag; the new y is in the buffer that has just become current)
        if ((rc = push ? peer_push_and_signal(h, h->delta ? 5 : 1 + h->ix_cur, adaptive != 0) : peer_signal(h, adaptive != 0)) != PDLP_OK) break;
        if (first && it + 1 < iters && (rc = pdlp_primal_half_begin(h)) != PDLP_OK) break;
        if ((rc = peer_wait(h, adaptive != 0)) != PDLP_OK) break;                // (adaptive: and the rule from all ranks' sums)
    }
    P.active = false;
    h->begin_inline = saved_inline;
    if (rc != PDLP_OK) return rc;
    HIP_TRY(hipGetLastError());
    if (!adaptive) return pdlp_fixed_advance(h, iters);
    return PDLP_OK;
}

// what a rank tells the others (pdlp_peer_export): PDLP_PEER_INFO_BYTES opaque bytes
struct PeerInfo {
    uint32_t magic, version;
    hipIpcMemHandle_t ws, box;
    int64_t ws_off;                 // the workspace inside its allocation (an IPC handle opens at the allocation's base)
    int64_t vec_off[6];             // xbar, the three y buffers, gdx, gdy inside the workspace
    int64_t n, m, nl, ml;
    int32_t dtype, pid;
    int64_t alloc_bytes;            // size of the allocation the workspace lies in (what a peer maps)
};
static_assert(sizeof(PeerInfo) <= PDLP_PEER_INFO_BYTES, "PeerInfo must fit its published size");
constexpr uint32_t PEER_MAGIC = 0x50444c50u;

void peer_vec_offsets(pdlp_handle h, int64_t off[6])
{
    off[0] = h->xbar - h->ws;
    for (int i = 0; i < 3; ++i) off[1 + i] = h->yb[i] - h->ws;
    off[4] = (char*)h->gdx - h->ws;
    off[5] = (char*)h->gdy - h->ws;
}

int peer_own_resources(pdlp_handle h)
{
    pdlp_solver::Peer& P = h->peer;
    if (!P.box) {
        HIP_TRY(hipExtMallocWithFlags((void**)&P.box, BOX_BYTES, hipDeviceMallocFinegrained));
        HIP_TRY(hipMemsetAsync(P.box, 0, BOX_BYTES, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (!P.err) {
        HIP_TRY(hipHostMalloc((void**)&P.err, 64, hipHostMallocMapped));
        *P.err = 0;
        HIP_TRY(hipHostGetDevicePointer((void**)&P.err_dev, P.err, 0));
    }
    return PDLP_OK;
}

}  // namespace

int pdlp_peer_export(pdlp_handle h, void* info)
{
    if (!h || !info) return PDLP_ERR_INVALID;
    if (h->nl == h->p.n && h->ml == h->p.m) return PDLP_ERR_STATE;              // not sharded: nothing to exchange
    HIP_TRY(hipSetDevice(h->p.device));
    int rc;
    if ((rc = peer_own_resources(h)) != PDLP_OK) return rc;
    PeerInfo pi{};
    pi.magic = PEER_MAGIC; pi.version = (uint32_t)pdlp_abi_version();
    hipDeviceptr_t base = nullptr;
    size_t range = 0;
    HIP_TRY(hipMemGetAddressRange(&base, &range, (hipDeviceptr_t)h->ws));
    if ((char*)h->ws + h->ws_bytes > (char*)base + range) return PDLP_ERR_WORKSPACE;   // (a workspace spanning allocations cannot be exported)
    // ROCm 7.2: hipIpcOpenMemHandle never returns for an allocation whose size has bit 31 set (2000 / 5000 MB open at once, 3000 /
    // 3826 / 4000 / 6500 MB hang, 4096 / 8192 / 9000 MB open: tools/ipc_torch_probe.py).  Refuse here rather than let a peer hang.
    if (range & 0x80000000ull) return PDLP_ERR_WORKSPACE;
    HIP_TRY(hipIpcGetMemHandle(&pi.ws, (void*)base));
    HIP_TRY(hipIpcGetMemHandle(&pi.box, (void*)h->peer.box));
    pi.ws_off = h->ws - (char*)base;
    peer_vec_offsets(h, pi.vec_off);
    pi.n = h->p.n; pi.m = h->p.m; pi.nl = h->nl; pi.ml = h->ml;
    pi.dtype = h->p.dtype; pi.pid = (int32_t)getpid();
    pi.alloc_bytes = (int64_t)range;
    std::memset(info, 0, PDLP_PEER_INFO_BYTES);
    std::memcpy(info, &pi, sizeof pi);
    return PDLP_OK;
}

int pdlp_peer_connect(pdlp_handle h, int rank, int world, const void* infos, int flags)
{
    if (!h || world < 2 || world > MAX_PEER + 1 || rank < 0 || rank >= world) return PDLP_ERR_INVALID;
    if (h->nl * world != h->p.n || h->ml * world != h->p.m || h->p.col0 != (int64_t)rank * h->nl || h->p.row0 != (int64_t)rank * h->ml)
        return PDLP_ERR_INVALID;                   // equal blocks, this rank's at rank * block (torchpdlp_amd/distributed.py)
    const bool loopback = (flags & PDLP_PEER_LOOPBACK) != 0;
    if (!loopback && !infos) return PDLP_ERR_INVALID;
    HIP_TRY(hipSetDevice(h->p.device));
    pdlp_solver::Peer& P = h->peer;
    if (P.on) return PDLP_ERR_STATE;
    int rc;
    if ((rc = peer_own_resources(h)) != PDLP_OK) return rc;
    P.rank = rank; P.world = world; P.n = 0; P.loopback = loopback;
    if (loopback) {
        // timing stand-in (tools/shard_iter_timing.py): rank `rank` of `world` alone -- every "peer vector" is a scratch block of this
        // process, every flag lands in the own mailbox (the waits pass at once).  What it prices: the stores and the two launches.
        const size_t blk = (size_t)(h->nl > h->ml ? h->nl : h->ml) * 8;
        HIP_TRY(hipMalloc((void**)&P.scratch, blk * (size_t)(world - 1)));
        char* slow = nullptr;
        if (flags & PDLP_PEER_LOOPBACK_HOST) {
            // the first "peer" lives in pinned host memory: its block crosses PCIe (~55 GB/s: 5 MB in ~0.09 ms), about what the seven
            // blocks of an 8-rank exchange take over xGMI together -- so the loopback also shows how much of a slow drain of the
            // stores a schedule hides
            HIP_TRY(hipHostMalloc((void**)&P.scratch_host, blk, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void**)&slow, P.scratch_host, 0));
        }
        for (int q = 0, i = 0; q < world; ++q) {
            if (q == rank) continue;
            for (int v = 0; v < 6; ++v) P.out[v][i] = (i == 0 && slow) ? slow : P.scratch + blk * (size_t)i;
            P.flag[i] = (uint32_t*)P.box + (size_t)q * BOX_FLAG_STRIDE;
            P.sums[i] = (double*)(P.box + BOX_BYTES / 2) + (size_t)q * BOX_SUMS_STRIDE;     // (an unused part of the mailbox: the peers' slots stay zero)
            ++i;
            P.n = i;
        }
        P.on = true;
        return PDLP_OK;
    }
    int64_t mine[6];
    peer_vec_offsets(h, mine);
    for (int q = 0, i = 0; q < world; ++q) {
        if (q == rank) continue;
        PeerInfo pi;
        std::memcpy(&pi, (const char*)infos + (size_t)q * PDLP_PEER_INFO_BYTES, sizeof pi);
        if (pi.magic != PEER_MAGIC || pi.version != (uint32_t)pdlp_abi_version() || pi.n != h->p.n || pi.m != h->p.m || pi.nl != h->nl ||
            pi.ml != h->ml || pi.dtype != h->p.dtype) { peer_release(h); return PDLP_ERR_INVALID; }
        char *wsb = nullptr, *box = nullptr;
        if (hipIpcOpenMemHandle((void**)&wsb, pi.ws, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); peer_release(h); return PDLP_ERR_COMM; }
        P.opened[P.nopened++] = wsb;
        if (hipIpcOpenMemHandle((void**)&box, pi.box, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); peer_release(h); return PDLP_ERR_COMM; }
        P.opened[P.nopened++] = box;
        for (int v = 0; v < 6; ++v) P.out[v][i] = wsb + pi.ws_off + pi.vec_off[v] + peer_block_start(h, v);
        P.flag[i] = (uint32_t*)box + (size_t)rank * BOX_FLAG_STRIDE;
        P.sums[i] = (double*)(box + BOX_SUMS_AT) + (size_t)rank * BOX_SUMS_STRIDE;
        ++i;
        P.n = i;
    }
    P.on = true;
    return PDLP_OK;
}

int pdlp_peer_status(pdlp_handle h, int32_t out[4])
{
    if (!h || !out) return PDLP_ERR_INVALID;
    out[0] = h->peer.on; out[1] = h->peer.enabled;
    out[2] = h->peer.err ? __atomic_load_n(h->peer.err, __ATOMIC_RELAXED) : 0;
    out[3] = (int32_t)h->peer.seq;
    return PDLP_OK;
}

int pdlp_peer_close(pdlp_handle h)
{
    if (!h) return PDLP_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    peer_release(h);
    return PDLP_OK;
}

int pdlp_comm_load(const char* rccl_path) { return rccl_load(rccl_path); }

int pdlp_comm_unique_id(const char* rccl_path, void* id128)
{
    if (!id128) return PDLP_ERR_INVALID;
    const int rc = rccl_load(rccl_path);
    if (rc != PDLP_OK) return rc;
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return PDLP_OK;
}

int pdlp_comm_init(pdlp_handle h, const char* rccl_path, const void* id128, int rank, int nranks)
{
    if (!h || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return PDLP_ERR_INVALID;
    // equal blocks, this rank's at rank * block (the padded layout of torchpdlp_amd/distributed.py)
    if (h->nl * nranks != h->p.n || h->ml * nranks != h->p.m || h->p.col0 != (int64_t)rank * h->nl || h->p.row0 != (int64_t)rank * h->ml)
        return PDLP_ERR_INVALID;
    const int rc = rccl_load(rccl_path);
    if (rc != PDLP_OK) return rc;
    if (h->comm) { (void)g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
    HIP_TRY(hipSetDevice(h->p.device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    RCCL_TRY(g_rccl.CommInitRank(&h->comm, nranks, id, rank));
    h->comm_rank = rank; h->comm_size = nranks;
    if (!h->cstream) {           // (chunked exchange: without these it stays one all-gather on the handle's stream)
        bool ok = hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&h->ev_vec, hipEventDisableTiming) == hipSuccess;
        for (auto& e : h->ev_chunk) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        for (auto& e : h->ev_row) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&h->ev_ar, hipEventDisableTiming) == hipSuccess;
        if (!ok) { if (h->cstream) (void)hipStreamDestroy(h->cstream); h->cstream = nullptr; (void)hipGetLastError(); }
    }
    return PDLP_OK;
}

int pdlp_comm_all_gather(pdlp_handle h, int which)
{
    if (!h || !h->comm) return PDLP_ERR_STATE;
    void* p = nullptr;
    const int rc = pdlp_buffer_ptr(h, which, &p);
    if (rc != PDLP_OK) return rc;
    const bool is_x = which <= PDLP_BUF_X_AVG || which == PDLP_BUF_DX || which == PDLP_BUF_GDX;
    if (which == PDLP_BUF_RED || which == PDLP_BUF_SCALARS || which == PDLP_BUF_X_SUM || which == PDLP_BUF_Y_SUM || which == PDLP_BUF_LAM_PREV)
        return PDLP_ERR_INVALID;
    const bool f32 = h->p.dtype == PDLP_F32 || which == PDLP_BUF_GDX || which == PDLP_BUF_GDY;
    return comm_all_gather(h, p, is_x ? h->nl : h->ml, f32);
}

int pdlp_comm_all_reduce_red(pdlp_handle h)
{
    if (!h || !h->comm) return PDLP_ERR_STATE;
    RCCL_TRY(g_rccl.AllReduce(h->red, h->red, PDLP_NRED, ncclFloat64, ncclSum, h->comm, h->stream));
    return PDLP_OK;
}

int pdlp_iterate(pdlp_handle h, int iters, int adaptive)
{
    if (!h || iters < 0) return PDLP_ERR_INVALID;
    adaptive = adaptive ? 1 : 0;
    char rname[64];
    if (g_roctx.level > 0) std::snprintf(rname, sizeof rname, "pdlp: %d %s iterations", iters, adaptive ? "adaptive" : "fixed-step");
    Range range(rname, h->stream);
    if (h->peer.on && h->peer.enabled) return iterate_peer(h, iters, adaptive);
    if (h->comm) return iterate_sharded(h, iters, adaptive);
    if (h->nl != h->p.n || h->ml != h->p.m) return PDLP_ERR_STATE;   // sharded without a communicator: the caller does the exchange
    int rc, left = iters;
    if (h->graph_ok && left >= 5) {
        if (!h->kx_valid) {          // the first iteration after a reset also refreshes the K x cache: never inside a captured pair
            if ((rc = iterate_direct(h, 1, adaptive)) != PDLP_OK) return rc;
            --left;
        }
        pdlp_solver::IterGraph* g = pair_graph(h, adaptive);
        if (g) {
            HIP_TRY(hipEventRecord(h->ev_in, h->stream));
            HIP_TRY(hipStreamWaitEvent(h->gstream, h->ev_in, 0));
            for (; left >= 2; left -= 2) HIP_TRY(hipGraphLaunch(g->exec, h->gstream));
            HIP_TRY(hipEventRecord(h->ev_out, h->gstream));
            HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_out, 0));
            h->cand_valid[0] = h->cand_valid[1] = false;     // (what two direct iterations leave behind)
            if (adaptive) { h->last_gridA = grid_of(h->sKT, h->nl); h->last_gridB = grid_of(h->sK, h->ml); }
        }
    }
    if ((rc = iterate_direct(h, left, adaptive)) != PDLP_OK) return rc;
    if (!adaptive && iters > 0) {
        if (h->p.dtype == PDLP_F32) hipLaunchKernelGGL(k_fixed_advance<float>, dim3(1), dim3(1), 0, h->stream, h->sc, iters);
        else hipLaunchKernelGGL(k_fixed_advance<double>, dim3(1), dim3(1), 0, h->stream, h->sc, iters);
    }
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_fixed_advance(pdlp_handle h, int iters)
{
    if (!h || iters < 0) return PDLP_ERR_INVALID;
    if (h->p.dtype == PDLP_F32) hipLaunchKernelGGL(k_fixed_advance<float>, dim3(1), dim3(1), 0, h->stream, h->sc, iters);
    else hipLaunchKernelGGL(k_fixed_advance<double>, dim3(1), dim3(1), 0, h->stream, h->sc, iters);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_flush_average(pdlp_handle h, int adaptive)
{
    if (!h) return PDLP_ERR_INVALID;
    return DISPATCH(h, flush_t, h, adaptive ? 1 : 0);
}

int pdlp_compute_average(pdlp_handle h)
{
    if (!h) return PDLP_ERR_INVALID;
    return DISPATCH(h, average_t, h);
}

int pdlp_kkt_local(pdlp_handle h, int which, int unscaled)
{
    if (!h || which < PDLP_CUR || which > PDLP_PREV) return PDLP_ERR_INVALID;
    Range range(which == PDLP_CUR ? "pdlp: KKT pass (current)" : which == PDLP_AVG ? "pdlp: KKT pass (average)" : "pdlp: KKT pass (previous)", h->stream);
    if (unscaled && (!h->p.d_col || !h->p.d_row)) return PDLP_ERR_STATE;
    if (h->delta) return delta_kkt_local(h, which, unscaled);
    return DISPATCH(h, kkt_local_t, h, which, unscaled);
}

int pdlp_read_red(pdlp_handle h, double out[PDLP_NRED])
{
    if (!h || !out) return PDLP_ERR_INVALID;
    HIP_TRY(hipMemcpyAsync(out, h->red, PDLP_NRED * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PDLP_OK;
}

int pdlp_kkt_finish(pdlp_handle h, double omega, double out[6])
{
    if (!h || !out) return PDLP_ERR_INVALID;
    double r[PDLP_NRED];
    int rc = pdlp_read_red(h, r);
    if (rc != PDLP_OK) return rc;
    if (h->p.dtype == PDLP_F32) kkt_finish_t<float>(r, omega, out);
    else kkt_finish_t<double>(r, omega, out);
    return PDLP_OK;
}

int pdlp_restart(pdlp_handle h, int which)
{
    if (!h || (which != PDLP_CUR && which != PDLP_AVG)) return PDLP_ERR_INVALID;
    const int cand = which == PDLP_CUR ? 0 : 1;
    if (which == PDLP_AVG) {       // the averaged iterate becomes current (pdhg.py:133,137,141)
        const int t = h->ix_cur;
        h->ix_cur = h->ix_avg;
        h->ix_avg = t;
    }
    if (h->delta) {                // the anchors follow the iterate: K x and K'y of the average were kept by its KKT pass
        if (which == PDLP_AVG) {
            if (h->cand_valid[1]) {
                char* t = h->kxb[0]; h->kxb[0] = h->kxb[2]; h->kxb[2] = t;
                t = h->ktyr; h->ktyr = h->ktyb[1]; h->ktyb[1] = t;
                h->dy_folded = true;
            } else {
                h->anchors_valid = false;
            }
        }
        h->cand_valid[0] = h->cand_valid[1] = false;
        HIP_TRY(hipMemsetAsync(h->x_sum, 0, h->nl * h->es, h->stream));
        HIP_TRY(hipMemsetAsync(h->y_sum, 0, h->ml * h->es, h->stream));
        hipLaunchKernelGGL(k_reset_average, dim3(1), dim3(1), 0, h->stream, h->sc);
        HIP_TRY(hipGetLastError());
        return PDLP_OK;
    }
    if (h->cand_valid[cand]) {     // K x and K'y of the chosen point were produced by its KKT pass (or carried along)
        if (!(cand == 0 && h->cur_kx_cached)) {
            char* t = h->kxb[0];
            h->kxb[0] = h->kxb[1 + cand];
            h->kxb[1 + cand] = t;
        }
        h->kx_valid = true;
        h->kty_cur = cand;
    } else {
        h->kx_valid = false;
        if (which == PDLP_AVG) h->kty_cur = -1;
    }
    h->cand_valid[0] = h->cand_valid[1] = false;
    HIP_TRY(hipMemsetAsync(h->x_sum, 0, h->nl * h->es, h->stream));          // pdhg.py:58-60
    HIP_TRY(hipMemsetAsync(h->y_sum, 0, h->ml * h->es, h->stream));
    HIP_TRY(hipMemsetAsync(h->kx_sum, 0, h->ml * h->es, h->stream));
    HIP_TRY(hipMemsetAsync(h->kty_sum, 0, h->nl * h->es, h->stream));
    h->since_reset = 0; h->kty_tail_done = false; h->avg_products = false; h->sums_broken = false; h->cur_kx_cached = false;
    hipLaunchKernelGGL(k_reset_average, dim3(1), dim3(1), 0, h->stream, h->sc);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_restart_distance_local(pdlp_handle h)
{
    if (!h) return PDLP_ERR_INVALID;
    return DISPATCH(h, distance_t, h);
}

int pdlp_mark_restart_point(pdlp_handle h)
{
    if (!h) return PDLP_ERR_INVALID;                                          // pdhg.py:63-64
    HIP_TRY(hipMemcpyAsync(h->x_last, h->xb[h->ix_cur] + h->p.col0 * h->es, h->nl * h->es, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->y_last, h->yb[h->ix_cur] + h->p.row0 * h->es, h->ml * h->es, hipMemcpyDeviceToDevice, h->stream));
    return PDLP_OK;
}

int pdlp_infeas_reset(pdlp_handle h)
{
    if (!h) return PDLP_ERR_INVALID;
    HIP_TRY(hipMemsetAsync(h->lam_prev, 0, h->nl * h->es, h->stream));          // pdhg.py:39-40
    return PDLP_OK;
}

int pdlp_infeas_begin(pdlp_handle h)
{
    if (!h) return PDLP_ERR_INVALID;
    return DISPATCH(h, infeas_begin_t, h);
}

int pdlp_infeas_local(pdlp_handle h, double tol)
{
    if (!h) return PDLP_ERR_INVALID;
    return DISPATCH(h, infeas_local_t, h, tol);
}

int pdlp_infeas_finish(pdlp_handle h, double tol, int32_t* status, double diag[8])
{
    if (!h || !status || !diag) return PDLP_ERR_INVALID;
    double r[PDLP_NRED];
    const int rc = pdlp_read_red(h, r);
    if (rc != PDLP_OK) return rc;
    *status = h->p.dtype == PDLP_F32 ? infeas_decide_t<float>(r, tol, diag) : infeas_decide_t<double>(r, tol, diag);
    return PDLP_OK;
}

int pdlp_mv_steps(pdlp_handle h, int nvp, int steps, double eta, double omega, double theta, void* X, void* Y, void* work)
{
    if (!h || !X || !Y || !work || steps < 0 || (nvp != 8 && nvp != 16 && nvp != 32)) return PDLP_ERR_INVALID;
    if (h->nl != h->p.n || h->ml != h->p.m || h->mixed) return PDLP_ERR_STATE;
    return DISPATCH(h, mv_steps_t, h, nvp, steps, eta, omega, theta, X, Y, work);
}

int pdlp_mv_gap(pdlp_handle h, int nvp, const void* X, const void* Y, void* work, double* gaps)
{
    if (!h || !X || !Y || !work || !gaps || (nvp != 8 && nvp != 16 && nvp != 32)) return PDLP_ERR_INVALID;
    if (h->nl != h->p.n || h->ml != h->p.m || h->mixed) return PDLP_ERR_STATE;
    return DISPATCH(h, mv_gap_t, h, nvp, X, Y, work, gaps);
}

int pdlp_mv_product(pdlp_handle h, int nvp, const void* X, void* Y)
{
    if (!h || !X || !Y || (nvp != 8 && nvp != 16 && nvp != 32)) return PDLP_ERR_INVALID;
    if (h->nl != h->p.n || h->ml != h->p.m || h->mixed) return PDLP_ERR_STATE;
    return DISPATCH(h, mv_product_t, h, nvp, X, Y);
}

int pdlp_mv_combine(int dtype, int64_t rows, int j, const void* V, const void* W, int nw, void* OUT, void* stream)
{
    if (rows < 0 || j < 1 || j > 32 || nw < 1 || nw > 32 || !V || !W || !OUT || (dtype != PDLP_F32 && dtype != PDLP_F64)) return PDLP_ERR_INVALID;
    if (rows == 0) return PDLP_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PDLP_F32)
        hipLaunchKernelGGL(k_mv_combine<float>, dim3(grid_for(rows * nw)), dim3(BLOCK), 0, s, rows, j, (const float*)V, (const float*)W, nw, (float*)OUT);
    else
        hipLaunchKernelGGL(k_mv_combine<double>, dim3(grid_for(rows * nw)), dim3(BLOCK), 0, s, rows, j, (const double*)V, (const double*)W, nw, (double*)OUT);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_spmv(pdlp_handle h, int transpose, const void* in_full, void* out_local)
{
    if (!h || !in_full || !out_local) return PDLP_ERR_INVALID;
    return DISPATCH(h, spmv_t, h, transpose, in_full, out_local);
}

int pdlp_power_iteration(pdlp_handle h, const void* b0, int iters, void* work_n, void* work_m, double* sigma)
{
    if (!h || !b0 || !work_n || !work_m || !sigma || iters < 0) return PDLP_ERR_INVALID;
    Range range("pdlp: power iteration", h->stream);
    if (h->nl != h->p.n || h->ml != h->p.m) return PDLP_ERR_STATE;
    return DISPATCH(h, power_iteration_t, h, b0, iters, work_n, work_m, sigma);
}

int pdlp_set_delta(pdlp_handle h, int on)
{
    if (!h) return PDLP_ERR_INVALID;
    if (on && !h->mixed) return PDLP_ERR_STATE;           // float32 matrix values under float64 vectors only
    if ((on != 0) == h->delta) return PDLP_OK;
    drop_graphs(h);
    h->graph_ok = false;
    h->delta = on != 0;
    h->anchors_valid = false; h->dy_folded = false;
    h->kx_valid = false; h->cand_valid[0] = h->cand_valid[1] = false; h->kty_cur = -1;
    return PDLP_OK;
}

int pdlp_refresh_products(pdlp_handle h)
{
    if (!h) return PDLP_ERR_INVALID;
    Range range("pdlp: exact products (anchors / K x cache)", h->stream);
    if (h->delta) return delta_refresh(h);
    if (h->p.dtype == PDLP_F32) return refresh_kx_t<float>(h);
    return refresh_kx_t<double>(h);
}

int pdlp_set_anchors(pdlp_handle h, const void* kx_local, const void* kty_local)
{
    if (!h || !kx_local || !kty_local) return PDLP_ERR_INVALID;
    if (!h->delta) return PDLP_ERR_STATE;
    HIP_TRY(hipMemcpyAsync(h->kxb[0], kx_local, (size_t)h->ml * 8, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->ktyr, kty_local, (size_t)h->nl * 8, hipMemcpyDeviceToDevice, h->stream));
    h->anchors_valid = true; h->dy_folded = true; h->kx_valid = true;
    h->cand_valid[0] = h->cand_valid[1] = false;
    return PDLP_OK;
}

int pdlp_delta_state(pdlp_handle h, int32_t out[3])
{
    if (!h || !out) return PDLP_ERR_INVALID;
    out[0] = h->delta; out[1] = h->anchors_valid; out[2] = h->dy_folded;
    return PDLP_OK;
}

int pdlp_probe_stream_read(const void* buf, int64_t bytes, int reps, void* stream, double* gb_per_s)
{
    if (!buf || bytes < ((int64_t)1 << 24) || reps < 1 || !gb_per_s || ((uintptr_t)buf & 15u)) return PDLP_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const size_t n16 = (size_t)bytes / 16;
    const int grid = 512;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    unsigned* sink = (unsigned*)const_cast<void*>(buf);           // (never written: see the kernel)
    hipLaunchKernelGGL(k_probe_read, dim3(grid), dim3(512), 0, s, (const probe_u32x4*)buf, n16, sink);
    HIP_TRY(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_probe_read, dim3(grid), dim3(512), 0, s, (const probe_u32x4*)buf, n16, sink);
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIP_TRY(hipGetLastError());
    const size_t per = n16 / grid, read16 = per / (4 * 512) * (4 * 512) * grid;      // what the kernel really loads
    *gb_per_s = (double)read16 * 16.0 * reps / ((double)ms * 1e-3) / 1e9;
    return PDLP_OK;
}

int pdlp_trace_enable(int level)
{
    if (level < 0 || level > 2) return PDLP_ERR_INVALID;
    if (level > 0) {
        roctx_load();
        if (!g_roctx.push) { g_roctx.level = 0; return PDLP_ERR_STATE; }       // no roctx library on this machine
    }
    g_roctx.level = level;
    return PDLP_OK;
}

int pdlp_range_push(const char* name, void* stream)
{
    if (!name) return PDLP_ERR_INVALID;
    if (g_roctx.level > 0 && g_roctx.push) {
        if (g_roctx.level > 1) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        (void)g_roctx.push(name);
    }
    return PDLP_OK;
}

int pdlp_range_pop(void* stream)
{
    if (g_roctx.level > 0 && g_roctx.pop) {
        if (g_roctx.level > 1) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        (void)g_roctx.pop();
    }
    return PDLP_OK;
}

int pdlp_probe_gather(void* scratch, int64_t scratch_bytes, int64_t table_entries, int reps, void* stream, double* gitems_per_s)
{
    if (!scratch || table_entries < 1 || table_entries > (int64_t)1 << 31 || reps < 1 || !gitems_per_s || ((uintptr_t)scratch & 255u))
        return PDLP_ERR_INVALID;
    const int64_t tbytes = align_up(table_entries * 4 + 256, 256);
    const int64_t items = (scratch_bytes - tbytes) / 8 / NNZ_CAP * NNZ_CAP;
    if (items < (int64_t)NNZ_CAP * 64) return PDLP_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    float* table = (float*)scratch;
    uint32_t* idx = (uint32_t*)((char*)scratch + tbytes);
    float* val = (float*)(idx + items);
    HIP_TRY(hipMemsetAsync(table, 0, (size_t)tbytes, s));
    hipLaunchKernelGGL(k_probe_fill, dim3(grid_for(items)), dim3(BLOCK), 0, s, idx, val, items, (uint32_t)table_entries);
    const int64_t nblk = items / NNZ_CAP;
    const int grid = (int)(nblk < MAX_GRID ? nblk : MAX_GRID);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    float* sink = table + table_entries;               // (inside the padding of the table; never written)
    hipLaunchKernelGGL(k_probe_gather, dim3(grid), dim3(BLOCK), 0, s, idx, val, table, items, sink);
    HIP_TRY(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_probe_gather, dim3(grid), dim3(BLOCK), 0, s, idx, val, table, items, sink);
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIP_TRY(hipGetLastError());
    *gitems_per_s = (double)items * reps / ((double)ms * 1e-3) / 1e9;
    return PDLP_OK;
}

// ---- Ruiz building blocks -------------------------------------------------------------------------
int pdlp_csr_row_scale_factors(int dtype, int64_t rows, const int64_t* rowptr, const void* val, double eps, void* norm, void* stream)
{
    if (rows < 0 || (dtype != PDLP_F32 && dtype != PDLP_F64)) return PDLP_ERR_INVALID;
    if (rows == 0) return PDLP_OK;
    const int g = grid_for(rows * 8);
    if (dtype == PDLP_F32)
        hipLaunchKernelGGL(k_row_scale_factors<float>, dim3(g), dim3(BLOCK), 0, (hipStream_t)stream, rows, rowptr, (const float*)val,
                           (float)eps, (float*)norm);
    else
        hipLaunchKernelGGL(k_row_scale_factors<double>, dim3(g), dim3(BLOCK), 0, (hipStream_t)stream, rows, rowptr, (const double*)val,
                           eps, (double*)norm);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_csr_div_rows(int dtype, int64_t rows, const int64_t* rowptr, void* val, const void* norm, void* stream)
{
    if (rows < 0 || (dtype != PDLP_F32 && dtype != PDLP_F64)) return PDLP_ERR_INVALID;
    if (rows == 0) return PDLP_OK;
    const int g = grid_for(rows * 8);
    if (dtype == PDLP_F32)
        hipLaunchKernelGGL(k_div_rows<float>, dim3(g), dim3(BLOCK), 0, (hipStream_t)stream, rows, rowptr, (float*)val, (const float*)norm);
    else
        hipLaunchKernelGGL(k_div_rows<double>, dim3(g), dim3(BLOCK), 0, (hipStream_t)stream, rows, rowptr, (double*)val, (const double*)norm);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_csr_div_cols(int dtype, int64_t nnz, const int32_t* colidx, void* val, const void* norm_full, void* stream)
{
    if (nnz < 0 || (dtype != PDLP_F32 && dtype != PDLP_F64)) return PDLP_ERR_INVALID;
    if (nnz == 0) return PDLP_OK;
    const int g = grid_for(nnz);
    if (dtype == PDLP_F32)
        hipLaunchKernelGGL(k_div_cols<float>, dim3(g), dim3(BLOCK), 0, (hipStream_t)stream, nnz, colidx, (float*)val,
                           (const float*)norm_full);
    else
        hipLaunchKernelGGL(k_div_cols<double>, dim3(g), dim3(BLOCK), 0, (hipStream_t)stream, nnz, colidx, (double*)val,
                           (const double*)norm_full);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_vec_muldiv(int dtype, int64_t len, void* a, const void* b, int op, void* stream)
{
    if (len < 0 || (dtype != PDLP_F32 && dtype != PDLP_F64) || (op != 0 && op != 1)) return PDLP_ERR_INVALID;
    if (len == 0) return PDLP_OK;
    if (dtype == PDLP_F32)
        hipLaunchKernelGGL(k_muldiv<float>, dim3(grid_for(len)), dim3(BLOCK), 0, (hipStream_t)stream, len, (float*)a, (const float*)b, op);
    else
        hipLaunchKernelGGL(k_muldiv<double>, dim3(grid_for(len)), dim3(BLOCK), 0, (hipStream_t)stream, len, (double*)a, (const double*)b, op);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_vec_project_lambda(int dtype, int64_t len, const void* g, const void* l, const void* u, void* out, void* stream)
{
    if ((dtype != PDLP_F32 && dtype != PDLP_F64) || len < 0 || (len > 0 && (!g || !l || !u || !out))) return PDLP_ERR_INVALID;
    if (len == 0) return PDLP_OK;
    if (dtype == PDLP_F32)
        hipLaunchKernelGGL(k_project_lambda<float>, dim3(grid_for(len)), dim3(BLOCK), 0, (hipStream_t)stream, len, (const float*)g,
                           (const float*)l, (const float*)u, (float*)out);
    else
        hipLaunchKernelGGL(k_project_lambda<double>, dim3(grid_for(len)), dim3(BLOCK), 0, (hipStream_t)stream, len, (const double*)g,
                           (const double*)l, (const double*)u, (double*)out);
    HIP_TRY(hipGetLastError());
    return PDLP_OK;
}

int pdlp_vec_max_dev_from_one(int dtype, int64_t len, const void* v, void* work8, double* out, void* stream)
{
    if (len < 0 || !work8 || !out || (dtype != PDLP_F32 && dtype != PDLP_F64)) return PDLP_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(work8, 0, 8, s));
    if (len > 0) {
        if (dtype == PDLP_F32)
            hipLaunchKernelGGL(k_max_dev_from_one<float>, dim3(grid_for(len)), dim3(BLOCK), 0, s, len, (const float*)v, (double*)work8);
        else
            hipLaunchKernelGGL(k_max_dev_from_one<double>, dim3(grid_for(len)), dim3(BLOCK), 0, s, len, (const double*)v, (double*)work8);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(out, work8, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return PDLP_OK;
}

int pdlp_vec_sqdist(int dtype, int64_t len, const void* a, const void* b, void* work, double* out, void* stream)
{
    if (len < 0 || !work || !out || (len > 0 && (!a || !b)) || (dtype != PDLP_F32 && dtype != PDLP_F64)) return PDLP_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    *out = 0.0;
    if (len == 0) return PDLP_OK;
    // partial sums of <= 256 workgroups at work[b * NACC], added in fixed order by one workgroup into work[256 * NACC]
    const int64_t want = (len + BLOCK - 1) / BLOCK;
    const int grid = (int)(want < 256 ? want : 256);
    double* part = (double*)work;
    if (dtype == PDLP_F32)
        hipLaunchKernelGGL(k_sqdiff<float>, dim3(grid), dim3(BLOCK), 0, s, len, (const float*)a, (const float*)b, part);
    else
        hipLaunchKernelGGL(k_sqdiff<double>, dim3(grid), dim3(BLOCK), 0, s, len, (const double*)a, (const double*)b, part);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, s, (const double*)part, grid, 1, part + 256 * NACC, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, part + 256 * NACC, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return PDLP_OK;
}

}  // extern "C"
