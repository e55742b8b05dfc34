/* pdlp_hip.h -- C ABI of libpdlp_hip.so: the restarted-PDHG hot path for AMD Instinct MI355X (gfx950).
 *
 * Drop-in boundary for SimplySnap/torchPDLP's solver hot path.  The reference has no FFI; its
 * boundary is the set of Python functions pdlp_algorithm() calls.  Each
 * entry point below names the reference function (file:line under /root/reference) whose
 * arithmetic it replaces.  All pointers in pdlp_problem and the workspace are DEVICE pointers
 * (e.g. torch.Tensor.data_ptr()); the library never allocates or frees device memory, never
 * copies the problem data, and enqueues all work on the HIP stream given at creation.
 *
 * LP form (reference PDLP/util.py:76-84):
 *     min c'x  s.t.  K[:m_ineq] x >= q[:m_ineq],  K[m_ineq:] x = q[m_ineq:],  l <= x <= u
 *
 * Layout: K is held twice, as CSR of K (rows = constraints) and CSR of K' (rows = variables),
 * int32 row pointers / column indices, values in the working precision.  One process per GPU:
 * a rank owns rows [row0,row1) of K (with y, q) and rows [col0,col1) of K' (with x, c, l, u);
 * vectors that are gathered from (x, xbar, y and their averaged copies) live in full-length
 * buffers which the caller all-gathers between the half-steps (single GPU: the block is the
 * whole range and no exchange is needed).
 *
 * Every function returns PDLP_OK (0) or a negative error code (pdlp_strerror()).
 * A handle is not thread-safe; use one handle per GPU rank.
 */
#ifndef PDLP_HIP_H
#define PDLP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDLP_OK 0
#define PDLP_ERR_INVALID (-1)      /* bad argument / inconsistent sizes            */
#define PDLP_ERR_WORKSPACE (-2)    /* workspace too small or misaligned            */
#define PDLP_ERR_STATE (-3)        /* call sequence violated                       */
#define PDLP_ERR_COMM (-4)         /* RCCL could not be loaded, or a collective failed */
#define PDLP_ERR_HIP_BASE (-1000)  /* -(1000 + hipError_t) for HIP runtime errors  */

/* working precision: PDLP_F32 / PDLP_F64 = every vector and matrix value in that type; PDLP_MIXED = float32 matrix values
 * (K_val, KT_val, tile values) under float64 vectors, products and sums -- for tolerances below float32 resolution on
 * matrices whose entries are float32 numbers (the reference itself is float32 end to end, util.py:240-246) */
enum { PDLP_F32 = 0, PDLP_F64 = 1, PDLP_MIXED = 2 };

/* which iterate an evaluation refers to (reference pdhg.py:118-125) */
enum { PDLP_CUR = 0, PDLP_AVG = 1, PDLP_PREV = 2 };

/* device buffers the caller may need to view (collectives, reading results) */
enum {
    PDLP_BUF_X_CUR = 0, PDLP_BUF_X_PREV = 1, PDLP_BUF_XBAR = 2, PDLP_BUF_X_AVG = 3,     /* full length n */
    PDLP_BUF_Y_CUR = 4, PDLP_BUF_Y_PREV = 5, PDLP_BUF_Y_AVG = 6,                       /* full length m */
    PDLP_BUF_RED = 7,        /* double[PDLP_NRED]: partial sums awaiting a cross-rank all-reduce          */
    PDLP_BUF_X_SUM = 8, PDLP_BUF_Y_SUM = 9,   /* local length: eta-weighted sums (pdhg.py:107-108)         */
    PDLP_BUF_SCALARS = 10,   /* double[PDLP_NSCAL]: eta, omega, theta, tau, sigma, w_pending, eta_sum, k  */
    PDLP_BUF_DX = 11,        /* full length n: x - x_prev of the last step (infeasibility detection)      */
    PDLP_BUF_DY = 12,        /* full length m: y - y_prev                                                 */
    PDLP_BUF_LAM_PREV = 13,  /* local length: the previous lambda of the infeasibility detector (pdhg.py:40,101) */
    PDLP_BUF_GDX = 14,       /* PDLP_MIXED: full length n, FLOAT32: x+ - x of the last primal half-step (delta mode gathers from it) */
    PDLP_BUF_GDY = 15        /* PDLP_MIXED: full length m, FLOAT32: y+ - y of the last dual half-step                              */
};
#define PDLP_NRED 8
#define PDLP_NSCAL 16

typedef struct pdlp_problem {
    int32_t dtype;              /* PDLP_F32 | PDLP_F64 | PDLP_MIXED (float32 K_val / KT_val, float64 vectors)           */
    int32_t device;             /* HIP device ordinal                                                       */
    int64_t m, n;               /* global constraint / variable counts                                      */
    int64_t m_ineq;             /* the first m_ineq constraints are ">=" rows (util.py:250-261)             */
    int64_t row0, row1;         /* this rank's rows of K   (0,m on a single GPU)                            */
    int64_t col0, col1;         /* this rank's rows of K'  (0,n on a single GPU)                            */
    const int64_t* K_rowptr;    /* [row1-row0+1], K_rowptr[0] == 0; 64-bit: a handle may hold more than 2^31 non-zeros */
    const int32_t* K_colidx;    /* [nnz_K]  global column indices in [0,n)                                  */
    const void* K_val;          /* [nnz_K]                                                                  */
    const int64_t* KT_rowptr;   /* [col1-col0+1]                                                            */
    const int32_t* KT_colidx;   /* [nnz_KT] global row indices in [0,m)                                     */
    const void* KT_val;         /* [nnz_KT]                                                                 */
    const void* c;              /* [col1-col0]                                                              */
    const void* l;              /* [col1-col0]  -inf allowed                                                */
    const void* u;              /* [col1-col0]  +inf allowed                                                */
    const void* q;              /* [row1-row0]                                                              */
    const void* d_col;          /* [col1-col0] Ruiz column scaling or NULL (pdhg.py:157-161)                */
    const void* d_row;          /* [row1-row0] Ruiz row scaling or NULL                                     */
    void* stream;               /* hipStream_t all work is enqueued on (0 = null stream)                    */
} pdlp_problem;

/* Optional panel-tiled copy of one of the two matrices (layout and builder: torchpdlp_amd/tiled.py).
 * Rows are cut into blocks of 512*rpt rows (one workgroup each; lane l of wave w owns rows w*64*rpt + i*64 + l,
 * i < rpt, of the block), columns into panels of 2^lw entries; a tile is (row block, panel).  Items of a tile are sorted
 * by column; item = value + ((slot << lw) | local column), slot = rank of the item in row order inside the
 * tile; tiles are padded to multiples of 256 items with (value 0, slot = number of real items, column 0), and
 * every group of 256 sorted items is stored interleaved (position 4*lane + j holds sorted item 64*j + lane) so
 * that one wave's j-th gathers are 64 consecutive sorted items.  cnt holds, per (tile, thread), five (float32) or
 * three (float64) 32-bit words of 4-bit item counts (<= 15) of the thread's rows (nibble i = row i of the thread);
 * any 64 consecutive rows (one i of one wave) may hold at most 255 items of a tile -- the kernel scans the lanes'
 * counts in 8-bit fields.  float32 (and PDLP_MIXED, whose tiles are float32): rpt <= 40, cap <= 16384; float64: rpt <= 24,
 * cap <= 8192.  (slot, column) must pack into 32 bits: (cap + 8) << lw <= 2^32, i.e. lw <= 17 at the full float32 capacity
 * (the builder uses 16: a 256 KB panel stays L2-resident on every XCD and keeps the gathered lines dense). */
typedef struct pdlp_tiles {
    int32_t lw, rpt, cap;       /* panel = 2^lw columns; rows per thread              ; most items per tile  */
    int32_t nblk, npanel;       /* row blocks, column panels                                               */
    int32_t groups;             /* workgroups sharing a row block, each walking ceil(npanel/groups) panels; 1 = the
                                   epilogue is fused, > 1 (<= 8, <= npanel) = partial row sums + k_rowsum_epilogue */
    const uint32_t* idx;        /* [items]                                                                 */
    const void* val;            /* [items] in the problem's precision                                      */
    const int32_t* tile_ptr;    /* [nblk][npanel + 1] item offsets of a row block's tiles RELATIVE to the block's first item,
                                   multiples of 256 (32-bit: what a thread indexes with) ...                               */
    const int64_t* blk_base;    /* [nblk] ... and the position of every row block's first item in idx / val (64-bit: one
                                   matrix copy may hold more than 2^31 items)                                              */
    const uint32_t* cnt;        /* 4-bit counts, 16-byte aligned: [tiles][512][4] words 0..3, then [tiles][cw-4][512]; cw = 5|3                                                 */
    /* the remainder: items the tiles could not hold (more than 15 of a row in one tile, more than 255 of 64 consecutive rows,
     * more than cap in a tile), as segments of <= 512 items of the rows that have any; rem_rows_n = 0: none */
    int32_t rem_rows_n, rem_segs_n;
    const int32_t* rem_rows;    /* [rem_rows_n]     rows (local) with a remainder, ascending                */
    const int32_t* rem_rptr;    /* [rem_rows_n + 1] their segment ranges                                    */
    const int32_t* rem_sptr;    /* [rem_segs_n + 1] item ranges of the segments                             */
    const int32_t* rem_col;     /* [items]          column of every remainder item                          */
    const void* rem_val;        /* [items]          value, in the precision of the matrix                   */
    void* rem_work;             /* [rem_segs_n]     scratch, 8 bytes each                                   */
    void* rem_extra;            /* [rows] in the working precision, ZERO-initialised: the remainder's row sums (only the rows in
                                   rem_rows are ever written)                                               */
    void* rem_extra_f32;        /* PDLP_MIXED only: the same in float32 for the delta-mode products          */
} pdlp_tiles;

typedef struct pdlp_solver* pdlp_handle;

const char* pdlp_strerror(int code);
/* library ABI version; bumped on any signature change */
int pdlp_abi_version(void);

/* ---- lifetime ------------------------------------------------------------------------------- */
/* bytes of device workspace pdlp_create needs for this problem (state vectors + scratch) */
int pdlp_workspace_bytes(const pdlp_problem* p, int64_t* bytes);
/* Builds the row-block schedule of both CSR copies (reads the two row-pointer arrays back to the
 * host once) and carves the caller's workspace (256-byte aligned device memory).  x = y = 0. */
int pdlp_create(pdlp_handle* out, const pdlp_problem* p, void* workspace, int64_t workspace_bytes);
void pdlp_destroy(pdlp_handle h);
/* Use the panel-tiled copy for every product with K (transpose = 0) or K' (transpose = 1) from now on;
 * NULL goes back to the CSR arrays.  The arrays must outlive the handle and be 16-byte aligned.
 * Same results up to summation order; several times faster when the gathered vector exceeds the L2. */
int pdlp_attach_tiles(pdlp_handle h, int transpose, const pdlp_tiles* t);
/* The CSR kernel's row blocks (built by pdlp_create: <= 256 rows, <= 2048 non-zeros each): *blocks = device array of
 * (first row, first non-zero) pairs, nblk + 1 of them.  For building the column-sorted copy below. */
int pdlp_schedule_info(pdlp_handle h, int transpose, int32_t* nblk, const int64_t** blocks);
/* Optional column-sorted copy of the items of K (transpose = 0) or K' (1) for the CSR kernel: inside every row block the items
 * are sorted by column; sval = the values in that order (precision of the matrix), sidx = (slot << 21) | (column - cbase[block]),
 * slot = the item's position in CSR order inside its block.  A wave's gathers then cover consecutive sorted items -- a few cache
 * lines on banded / block-structured matrices (whose tiles would be nearly empty: torchpdlp_amd/tiled.py) instead of one per
 * lane.  cbase[block] < 0: the block's columns span more than 2^21 and it is read in CSR order.  NULLs detach.  Same results up
 * to nothing: the reduction order is the CSR one. */
int pdlp_attach_sorted(pdlp_handle h, int transpose, const uint32_t* sidx, const void* sval, const int32_t* cbase);
/* device address of one of the PDLP_BUF_* buffers (inside the workspace) */
int pdlp_buffer_ptr(pdlp_handle h, int which, void** ptr);

/* ---- state ---------------------------------------------------------------------------------- */
/* Copies this rank's block of the iterate from device memory (x: col1-col0, y: row1-row0 values)
 * and zeroes the averaging sums.  Replaces the x/y initialisation of pdhg.py:31-36. */
int pdlp_set_iterate(pdlp_handle h, const void* x_local, const void* y_local);
int pdlp_get_iterate(pdlp_handle h, int which, void* x_local, void* y_local);   /* device destinations */
/* step size eta, primal weight omega, extrapolation theta (pdhg.py:22-25); rounded to the working
 * precision on the device exactly as the reference's 0-dim tensors are. iteration = global k so far. */
int pdlp_set_step(pdlp_handle h, double eta, double omega, double theta, int64_t iteration);
/* new primal weight after a restart (primal_weight_update enhancements.py:77); keeps the device eta */
int pdlp_set_omega(pdlp_handle h, double omega);
/* host copy of {eta, omega, theta, tau, sigma, w_pending, eta_sum, k} (synchronises the stream) */
int pdlp_get_scalars(pdlp_handle h, double out[PDLP_NSCAL]);

/* ---- the PDHG iteration --------------------------------------------------------------------- */
/* primal half-step: x+ = clamp(x - (eta/omega)(c - K'y), l, u); xbar = x+ + theta(x+ - x)
 * -- fixed_one_step_pdhg step.py:25-30 / adaptive_one_step_pdhg step.py:74-82 -- as ONE kernel:
 * CSR(K') SpMV with the projection, extrapolation and eta-weighted sum fused into its epilogue.
 * Needs y (PDLP_BUF_Y_CUR) complete over [0,m). Writes this rank's block of x+ and xbar. */
int pdlp_primal_half(pdlp_handle h, int adaptive);
/* dual half-step: y+ = y + eta*omega (q - K xbar); y+[:m_ineq] = max(.,0) -- step.py:33-38 / :85-90 --
 * CSR(K) SpMV with the projection and sums fused.  Needs xbar complete over [0,n).  Afterwards
 * the new iterate becomes PDLP_CUR and the old one PDLP_PREV (pdhg.py:77-78). */
int pdlp_dual_half(pdlp_handle h, int adaptive);
/* Sharded problems, optional: start the coming half-step's product on the panels that lie wholly inside THIS rank's
 * block of the gathered vector (the block the preceding half-step just wrote) on a library-owned side stream, so
 * that it overlaps the all-gather of the other ranks' blocks.  Call right after the half-step that produced the
 * block and before the all-gather; the matching pdlp_primal_half / pdlp_dual_half then multiplies the remaining
 * panels, adds the partial row sums in fixed order and runs the fused update.  No-ops (return PDLP_OK) when the
 * matrix is not tiled, the problem is not sharded, or (dual, adaptive) the K x cache has to be refreshed first.
 * The reference has no counterpart (single device); the result equals the unsplit product up to summation order. */
int pdlp_primal_half_begin(pdlp_handle h);
int pdlp_dual_half_begin(pdlp_handle h, int adaptive);
/* what the two calls above will do for K (transpose = 0) or K' (1): out = {first local panel, end of the local panels,
 * panel groups of the local panels, panel groups of the others}; all zero = the product is not split */
int pdlp_split_info(pdlp_handle h, int transpose, int32_t out[4]);
/* what pdlp_attach_tiles accepts for this handle: out = {most panel groups (the row-sum scratch has that many slots),
 * most row blocks (slots of per-workgroup partial sums), most rows per thread, most items per tile, threads per workgroup of the
 * tiled kernel (a row block is that many times rpt rows), 0 (reserved)} */
int pdlp_tile_limits(pdlp_handle h, int32_t out[6]);
/* The INTENDED loop of the adaptive step (SURVEY quirk Q1's optional flag; the live package takes ONE trial and keeps a rejected step,
 * /root/reference/PDLP/primal_dual_hybrid_gradient_step.py:71,110-115 -- the loop it was meant to be is
 * /root/reference/enhancements/test_ass.py:322-363): after an adaptive iteration whose trial was REJECTED (scalar "accepted" = 0)
 * this discards the trial -- the old (x, y) is current again, its weight leaves the average, k goes back, the step size stays the
 * shrunk eta' -- so that the iteration can be issued again.  Off the default path (pdlp_algorithm(adaptive_retry=True)); float32 and
 * float64 handles; K x of the old x is recomputed by the next half-step and the running products of the period are dropped. */
int pdlp_adaptive_retry(pdlp_handle h);
/* adaptive rule, part 1: reduce this rank's ||dx||^2, ||dy||^2, dy'K dx into PDLP_BUF_RED[0..2]
 * (all-reduce them across ranks before part 2) -- step.py:91-96 */
int pdlp_adaptive_reduce(pdlp_handle h);
/* adaptive rule, part 2: eta_bar, eta', accept/reject, eta <- eta', weight of this iterate in the
 * average, k <- k+1 -- step.py:99-115 incl. the single-trial quirk, pdhg.py:107-112 */
int pdlp_adaptive_update(pdlp_handle h);
/* `iters` whole iterations without host synchronisation: primal half, dual half and, when adaptive, the step-size rule
 * (pdhg.py:76-112).  Single-rank problems, or sharded ones after pdlp_comm_init (the exchange then happens inside). */
int pdlp_iterate(pdlp_handle h, int iters, int adaptive);
/* fixed step, multi-rank driver: eta_total += eta and k += 1, `iters` times (pdhg.py:76,109);
 * pdlp_iterate does this itself */
int pdlp_fixed_advance(pdlp_handle h, int iters);

/* ---- sharded problems: the exchange inside the library (RCCL over xGMI) ------------------------------------------------
 * One process per GPU; rank r owns block r of the constraints and of the variables (equal, padded blocks -- the layout of
 * torchpdlp_amd/distributed.py).  The reference is single-device, so these have no counterpart there; SURVEY.md 8b sketched
 * them as pdlp_create_sharded(..., ncclComm_t, rank, nranks).  RCCL is resolved with dlopen on first use (rccl_path: the
 * librccl.so the process already has loaded, e.g. PyTorch's; NULL = search the default names), so single-GPU use never needs it.
 * pdlp_comm_load: dlopen + symbol lookup only (rank local, no communication): call it on every rank and agree on the results
 *   BEFORE any rank enters pdlp_comm_init, which is collective -- a rank that cannot load RCCL would leave the others waiting.
 * pdlp_comm_unique_id: rank 0 creates the id, the caller broadcasts its 128 bytes to all ranks (any transport).
 * pdlp_comm_init: ncclCommInitRank on the handle's device.  From then on pdlp_iterate runs whole sharded iterations --
 *   half-steps, the all-gathers of xbar and y (float32 differences in delta mode) overlapped with the local panels' products,
 *   and the 3-double all-reduce of the step-size rule -- on the handle's stream with no host work in between.
 * pdlp_comm_all_gather / pdlp_comm_all_reduce_red: the same collectives for the caller-driven parts (KKT passes, set-up). */
/* Chunked exchange (round 3).  One all-gather per product exposes the whole transfer minus the 1/P of the product that needs no
 * foreign entries.  With `chunks` = C > 1 the vector a product gathers travels as C pieces -- piece c = elements
 * [bounds[c], bounds[c+1]) of EVERY rank's block (pdlp_exchange_plan) -- and the panels a piece completes are multiplied while
 * the next piece is on the wire: pdlp_*_half_begin, then for c = 0 .. C-2: [piece c has arrived on the handle's stream]
 * pdlp_half_chunk(h, transpose, c), then [piece C-1 has arrived] pdlp_dual_half / pdlp_primal_half (last piece's panels, the sum of
 * all partial row sums in fixed order, the epilogue).  transpose = 0: K xbar (pieces of PDLP_BUF_XBAR, or PDLP_BUF_GDX in delta
 * mode), 1: K'y (PDLP_BUF_Y_CUR / PDLP_BUF_GDY).  Tiled matrices of sharded handles only; otherwise the plan is one piece and
 * pdlp_half_chunk does nothing.  pdlp_iterate on a handle with a communicator does all of this itself (pieces as grouped
 * in-place broadcasts on a communication stream of its own). */
int pdlp_set_exchange_chunks(pdlp_handle h, int chunks /* 1..4 */);
/* Producer side of the chunked exchange (round 5).  With C > 1 pieces the RESULT of a split product -- this rank's block of xbar
 * (primal half-step) or of y (dual half-step) -- also LEAVES in pieces: the last phase of the product and the epilogue run over the
 * row blocks that hold piece 0's rows first, then piece 1's ..., so that piece c's collective can start while the rows of the later
 * pieces are still being multiplied (until round 4 a half-step had to finish completely before the first byte of its block left).
 * pdlp_primal_half_piece / pdlp_dual_half_piece issue ONE piece of the half-step on the handle's stream: call them for
 * piece = 0 .. pieces-1 in order (pieces = the plan of the exchange that FOLLOWS: pdlp_exchange_plan(h, 0) for the primal, (h, 1) for
 * the dual half-step) and start piece c's collective after the c-th call; the half-step is complete (buffer roles, counters) after
 * the last call.  A half-step that is not a split tiled product (CSR kernel, K'y kept by a restart check, ...) does all its work
 * with piece 0.  Same arithmetic as pdlp_primal_half / pdlp_dual_half on the same handle (which run all pieces back to back).
 * pdlp_iterate on a handle with a communicator does this itself.  PDLP_OPT_PRODUCER_PIECES = 0 switches it off (A/B timing). */
int pdlp_primal_half_piece(pdlp_handle h, int adaptive, int piece, int pieces);
int pdlp_dual_half_piece(pdlp_handle h, int adaptive, int piece, int pieces);
/* Switches of the handle that tests and tools flip (the library itself reads no environment variables; the Python host layer maps
 * its PDLP_* test knobs onto these).  None has a counterpart in the reference; the defaults are the product path.
 *   PDLP_OPT_RUNNING_KKT  1 (default): restart checks take K x_avg, K'y_avg from running sums (pdhg.py:118-125 costs one product
 *                         instead of four); 0: every KKT pass multiplies
 *   PDLP_OPT_KTY_REUSE    1 (default): the first primal half-step after a restart check reuses the check's K'y; 0: multiplies again
 *   PDLP_OPT_GRAPH        0 (default) / 1: pdlp_iterate replays captured pairs of iterations as hipGraph launches (single-rank handles;
 *                         PDLP_ERR_STATE when the handle cannot capture)
 *   PDLP_OPT_SPLIT_SLOTS  0 (default): the library's rule; local | other << 16: panel groups of a split product (tools/split_timing.py)
 *   PDLP_OPT_PRODUCER_PIECES 1 (default): with a chunked exchange the result of a split product leaves piece by piece
 *                         (pdlp_*_half_piece); 0: the half-step finishes before its block is exchanged (round-4 behaviour)
 *   PDLP_OPT_BEGIN_INLINE 0 (default): pdlp_*_half_begin multiply the local panels on a side stream of the library (forked from and
 *                         joined to the handle's stream with events), so that they run beside a BLOCKING exchange on the handle's
 *                         stream; 1: on the handle's stream itself -- for callers whose exchange is already under way on a stream of
 *                         its own (an asynchronous collective): the same kernels without the two cross-stream dependencies
 *   PDLP_OPT_PEER_EXCHANGE 1 (default): a handle connected by pdlp_peer_connect iterates with the direct exchange; 0: with whatever it
 *                         would use without (its RCCL communicator, or the caller's loop)
 *   PDLP_OPT_PEER_TIMEOUT_MS how long a wait of the direct exchange spins before it gives up (default 10 000)
 *   PDLP_OPT_PEER_LOCAL_FIRST 0 (default): direct exchange = signal, wait, the whole product; 1: the product is split and the panels that
 *                         meet the own block run between signal and wait (they hide ranks finishing at different times but cannot
 *                         fill the chip: 1/world of the panels) -- same sums up to the grouping of the partial row sums
 *   PDLP_OPT_PEER_PUSH    0 (default) / 1: the epilogues store locally and a copy kernel on the library's side stream carries the block
 *                         to the peers (then the signal) WHILE the handle's stream multiplies the own block's panels -- with 2 or 4
 *                         ranks those panels are a half / a quarter of the product and hide the transfer over the links, which
 *                         stores issued by the epilogue (the last microseconds of a half-step, same stream) cannot; same results as
 *                         PDLP_OPT_PEER_LOCAL_FIRST */
enum { PDLP_OPT_RUNNING_KKT = 0, PDLP_OPT_KTY_REUSE = 1, PDLP_OPT_GRAPH = 2, PDLP_OPT_SPLIT_SLOTS = 3, PDLP_OPT_PRODUCER_PIECES = 4,
       PDLP_OPT_BEGIN_INLINE = 5, PDLP_OPT_PEER_EXCHANGE = 6, PDLP_OPT_PEER_TIMEOUT_MS = 7, PDLP_OPT_PEER_LOCAL_FIRST = 8,
       PDLP_OPT_PEER_PUSH = 9 };
int pdlp_set_option(pdlp_handle h, int option, int64_t value);
int pdlp_exchange_plan(pdlp_handle h, int transpose, int32_t* nchunks, int64_t bounds[5]);
int pdlp_half_chunk(pdlp_handle h, int transpose, int chunk);
/* Direct exchange: the iterations of a sharded problem WITHOUT collectives (north_star: "sharded across the 8 GPUs of one node ...
 * over xGMI"; SURVEY section 8e "hand-rolled P2P copies over xGMI with IPC buffers"; the reference has one device, PDLP/main.py:45-54).
 * Every rank exports its workspace and a small mailbox over HIP IPC; once connected, the epilogue of a half-step stores its block of
 * xbar / y (delta mode: of the float32 differences) into every peer's copy of that vector AS IT COMPUTES IT -- between GPUs these are
 * xGMI stores that ride on the product, 35 MB per half-step and rank at 10M x 10M / 8 ranks against the 1 GB the product streams --, a
 * one-wave kernel raises the rank's sequence number in the peers' mailboxes, and a one-wave kernel with a BOUNDED spin waits for the
 * peers' numbers before the part of the next product that needs their blocks.  The step-size rule's three sums (step.py:85-96) travel
 * with the flag of the y exchange and are added in rank order.  pdlp_iterate then is one call per restart period with no collective,
 * no second stream and no event in it; everything outside the iterations (KKT sums, gathers for restart checks) stays with the caller.
 *   pdlp_peer_export    this rank's PDLP_PEER_INFO_BYTES of connection data (IPC handles, offsets, shape); the caller distributes them
 *                       (an all-gather of bytes over its process group).  The workspace must lie inside ONE device allocation --
 *                       and PDLP_ERR_WORKSPACE if that allocation's size has bit 31 set (2-4 GiB, 6-8 GiB, ...): on ROCm 7.2
 *                       hipIpcOpenMemHandle never returns for such an allocation (tools/ipc_torch_probe.py); give the workspace an
 *                       allocation of its own of another size (PdlpEngine does: a private pool, padded to the next 4 GiB).
 *   pdlp_peer_connect   `infos` = the world x PDLP_PEER_INFO_BYTES bytes of all ranks in rank order (equal blocks, this rank's at
 *                       rank * block; at most 8 ranks).  PDLP_ERR_INVALID: shapes disagree; PDLP_ERR_COMM: a handle would not open.
 *                       flags = PDLP_PEER_LOOPBACK (infos may be null): a timing stand-in on ONE process -- the "peers" are scratch
 *                       blocks, the flags land in the own mailbox (tools/shard_iter_timing.py); results are those of one rank alone.
 *   pdlp_peer_status    out = { connected, enabled, 0 or 1 + the rank a wait gave up on, exchanges issued }.  A wait that gave up
 *                       lets the stream run on with incomplete vectors: check after synchronising, before using any result.
 *   pdlp_peer_close     unmaps the peers' memory, frees the mailbox (pdlp_destroy does the same).
 * Identical bits to the other two drivers in fixed-step mode; in adaptive mode up to the order of the ranks' three sums (two ranks:
 * identical).  Delta mode: the anchors must be valid on entry (PDLP_ERR_STATE otherwise: pdlp_refresh_products needs gathers). */
#define PDLP_PEER_INFO_BYTES 256
#define PDLP_PEER_LOOPBACK 1
#define PDLP_PEER_LOOPBACK_HOST 2      /* with PDLP_PEER_LOOPBACK: one stand-in peer in pinned host memory (a slow link's stand-in) */
int pdlp_peer_export(pdlp_handle h, void* info);
int pdlp_peer_connect(pdlp_handle h, int rank, int world, const void* infos, int flags);
int pdlp_peer_status(pdlp_handle h, int32_t out[4]);
int pdlp_peer_close(pdlp_handle h);
int pdlp_comm_load(const char* rccl_path);
int pdlp_comm_unique_id(const char* rccl_path, void* id128);
int pdlp_comm_init(pdlp_handle h, const char* rccl_path, const void* id128, int rank, int nranks);
int pdlp_comm_all_gather(pdlp_handle h, int which /* PDLP_BUF_* of a full-length vector */);
int pdlp_comm_all_reduce_red(pdlp_handle h);

/* ---- delta mode (PDLP_MIXED handles; no counterpart in the float32 reference) --------------------------------------
 * on != 0: from now on every product of an iteration and of a KKT pass runs on the FLOAT32 kernels over a float32 difference
 * vector (PDLP_BUF_GDX = x+ - x, PDLP_BUF_GDY = y+ - y, or candidate - current in a KKT pass) and is added to a float64
 * "anchor" product the handle carries along (K x and K'y of the current iterate):
 *     K xbar = K x + (1 + theta) K dx,   K'y+ = K'y + K'dy        (step.py:25-38 / :74-96 in exact arithmetic)
 * so the rounding of a product scales with the step, eps32 ||K|| ||dx||, instead of with the iterate, and the iteration
 * streams 8 instead of 12 bytes per non-zero.  pdlp_refresh_products recomputes the anchors exactly (float64 gathers,
 * products and sums); the handle does it by itself after pdlp_set_iterate; call it after every restart to bound the drift.
 * Sharded problems: all-gather PDLP_BUF_GDX before pdlp_dual_half and PDLP_BUF_GDY before pdlp_primal_half (instead of xbar
 * and y), and x, y of the current iterate before pdlp_refresh_products. */
int pdlp_set_delta(pdlp_handle h, int on);
int pdlp_refresh_products(pdlp_handle h);
/* The anchors from outside: kx_local = K x (this rank's constraints), kty_local = K'y (this rank's variables) of the CURRENT iterate,
 * float64 device arrays.  For matrices that are not float32-valued (any float64 K, a Ruiz-scaled K): the handle then holds K
 * ROUNDED to float32 -- the products of the iterations only ever see difference vectors, where the rounding of K costs what the
 * float32 products cost anyway, 6e-8 ||K|| ||dx|| -- and the caller evaluates the anchors with the true float64 matrix (a second,
 * float64 handle) after every restart instead of calling pdlp_refresh_products. */
int pdlp_set_anchors(pdlp_handle h, const void* kx_local, const void* kty_local);
/* out = {delta mode on, anchors valid, the pending dy is folded into K'y} */
int pdlp_delta_state(pdlp_handle h, int32_t out[3]);

/* ---- restart machinery ---------------------------------------------------------------------- */
/* Closes the averaging period before a restart check.  adaptive != 0: adds the not-yet-accumulated weight of the current
 * iterate to the sums (adaptive mode defers it by one step because the weight is only known after the step-size rule).
 * Both modes: the handle also keeps running sums of w_k K x_k and w_k K'y_k (K is linear: they are K x_avg and K'y_avg up to
 * the division), so that the check needs no product for the averaged iterate; K'y of the CURRENT y exists only once
 * pdlp_kkt_local(PDLP_CUR) has run, so call that first -- otherwise this call falls back to products for the average. */
int pdlp_flush_average(pdlp_handle h, int adaptive);
/* x_avg = x_sum / eta_sum, y_avg = y_sum / eta_sum for this rank's block -- pdhg.py:118-119 */
int pdlp_compute_average(pdlp_handle h);
/* KKT pass at PDLP_CUR / PDLP_AVG / PDLP_PREV: two fused kernels -- SpMVs, or vector passes where the product is already
 * there (K x of the current iterate is carried along by the dual half-steps; K x_avg and K'y_avg come from the running sums
 * after pdlp_flush_average + pdlp_compute_average: a check then costs ONE product, K'y_cur, instead of four) -- accumulate
 * {dual_res^2, l_dual'max(lam,0), u_dual'min(lam,0), c'x, primal_res^2, q'y} of this rank's block
 * into PDLP_BUF_RED[0..5] -- compute_residuals_and_duality_gap helpers.py:53-96 with
 * project_lambda_box helpers.py:3-39.  unscaled != 0 evaluates the un-preconditioned problem
 * (needs d_col/d_row) -- pdhg.py:157-161.  The x and y of `which` must be complete (all-gathered). */
int pdlp_kkt_local(pdlp_handle h, int which, int unscaled);
/* reads PDLP_BUF_RED (after the caller's all-reduce), returns
 * {primal_residual, dual_residual, duality_gap (signed), prim_obj, adjusted_dual, KKT_error}
 * -- helpers.py:84-94, KKT_error helpers.py:98-108.  Synchronises the stream. */
int pdlp_kkt_finish(pdlp_handle h, double omega, double out[6]);
/* make PDLP_AVG (or keep PDLP_CUR) the current iterate and zero the sums -- pdhg.py:57-60,131-142 */
int pdlp_restart(pdlp_handle h, int which);
/* ||x - x_last_restart||^2, ||y - y_last_restart||^2 of this rank's block into PDLP_BUF_RED[0..1]
 * -- primal_weight_update enhancements.py:74-75; pdlp_restart keeps the previous restart point,
 * so call this after it and before pdlp_mark_restart_point */
int pdlp_restart_distance_local(pdlp_handle h);
/* x_last_restart, y_last_restart <- current iterate -- pdhg.py:63-64 */
int pdlp_mark_restart_point(pdlp_handle h);
int pdlp_read_red(pdlp_handle h, double out[PDLP_NRED]);   /* synchronises the stream */

/* ---- infeasibility detection (opt-in) ------------------------------------------------------- */
/* Replaces detect_infeasibility (enhancements.py:80-161) and the lambda bookkeeping around it
 * (primal_dual_hybrid_gradient.py:39-40,89-101) for the step just taken (PDLP_CUR against PDLP_PREV).
 * Call order per iteration: pdlp_infeas_begin -> [all-gather PDLP_BUF_DX, PDLP_BUF_DY when sharded] ->
 * pdlp_infeas_local -> [all-reduce PDLP_BUF_RED] -> pdlp_infeas_finish.
 * pdlp_infeas_reset: lam_prev = 0 (pdhg.py:39-40); a new handle starts that way. */
int pdlp_infeas_reset(pdlp_handle h);
/* this rank's blocks of dx = x - x_prev (enhancements.py:108) and dy = y - y_prev (:109) into PDLP_BUF_DX / PDLP_BUF_DY */
int pdlp_infeas_begin(pdlp_handle h);
/* K'dy, then lam = project_lambda_box(c - K'y) (pdhg.py:90) with dlam, the dual residual K'dy - dlam (:146), the
 * bound products (:150-157), c'dx (:124) and the per-variable bound test (:128-139) fused into the product's epilogue,
 * then K dx with the equality norm (:118), the inequality test (:121), dy_in >= -tol (:148) and q'dy (:149) fused.
 * Needs y, PDLP_BUF_DX and PDLP_BUF_DY complete.  Sums of this rank's block -> PDLP_BUF_RED[0..7]; lam_prev <- lam. */
int pdlp_infeas_local(pdlp_handle h, double tol);
/* reads PDLP_BUF_RED (after the caller's all-reduce) and decides as enhancements.py:118-142 / :148-159:
 * *status = 0 (None), 1 ("DUAL_INFEASIBLE"), 2 ("PRIMAL_INFEASIBLE");
 * diag = { ||K_eq dx||, #{K_in dx < -tol}, c'dx, #{bound test fails}, ||K'dy - dlam||, #{dy_in < -tol}, q'dy,
 *          l_f'dlam_minus + u_f'dlam_plus }.  Synchronises the stream. */
int pdlp_infeas_finish(pdlp_handle h, double tol, int32_t* status, double diag[8]);

/* ---- populations of points (fishnet warm start, spectral_casting.py:65-159) -------------------- */
/* nvp (8, 16 or 32) points advance together; X is [n][nvp], Y is [m][nvp], row-major device arrays in the problem's
 * precision (single GPU).  One pass over each matrix serves all nvp points: an item contributes to nvp row sums with
 * one coalesced line of the population matrix.
 * pdlp_mv_steps: `steps` fixed-step PDHG iterations on every point, in place -- PDHG_step spectral_casting.py:254-293
 *   (= fixed_one_step_pdhg per column).  work: (2 n + m) * nvp values.
 * pdlp_mv_gap: gaps[v] = adjusted dual objective - primal objective of point v (signed) -- get_best_pts
 *   spectral_casting.py:215-234.  work: 256 * nvp * 4 + nvp * 4 doubles.  gaps: host array of nvp.  Synchronises. */
int pdlp_mv_steps(pdlp_handle h, int nvp, int steps, double eta, double omega, double theta, void* X, void* Y, void* work);
int pdlp_mv_gap(pdlp_handle h, int nvp, const void* X, const void* Y, void* work, double* gaps);
/* pdlp_mv_product: Y = K X for all nvp points in one pass over K -- `pts_y = K @ pts` spectral_casting.py:100.
 * pdlp_mv_combine: OUT[r][t] = sum_p V[r][p] * W[p][t] for t < nw -- the random convex combinations of the breeding rounds,
 *   `new_x = pts @ w` spectral_casting.py:133-141 (V is [rows][j] row-major, W [j][nw], OUT [rows][nw]; j <= 32; no handle). */
int pdlp_mv_product(pdlp_handle h, int nvp, const void* X, void* Y);
int pdlp_mv_combine(int dtype, int64_t rows, int j, const void* V, const void* W, int nw, void* OUT, void* stream);

/* ---- plain products (power iteration helpers.py:41-51, tests) ------------------------------- */
/* out_local = K in_full (transpose=0, out has row1-row0 values) or K' in_full (transpose=1) */
int pdlp_spmv(pdlp_handle h, int transpose, const void* in_full, void* out_local);
/* spectral_norm_estimate_torch helpers.py:41-51 with the start vector given (single rank) */
int pdlp_power_iteration(pdlp_handle h, const void* b0, int iters, void* work_n, void* work_m, double* sigma);

/* ---- tracing: roctx ranges for rocprofv3 --marker-trace ------------------------------------------ */
/* The reference times its sections with a wall-clock Timer (PDLP/util.py:6-27).  Here the phases of a solve are roctx ranges:
 * inside the library around pdlp_iterate ("pdlp: N adaptive iterations"), every KKT pass, the exact product refresh, the power
 * iteration and the pieces of a chunked exchange; the host layer adds "restart check", "restart work", "Ruiz sweep", "tile build"
 * with pdlp_range_push / pdlp_range_pop.  level 0: off (default: no library is loaded, no call is made); 1: ranges on the host
 * timeline; 2: the stream is synchronised at both ends of a range, so a range's duration is the GPU time of its work (the
 * per-phase table of profiles/README.md).  The roctx library is resolved with dlopen (librocprofiler-sdk-roctx.so, else
 * libroctx64.so); PDLP_ERR_STATE when there is none. */
int pdlp_trace_enable(int level);
int pdlp_range_push(const char* name, void* stream);
int pdlp_range_pop(void* stream);

/* Measurement aid (no counterpart in the reference): GB/s at which this GPU streams `bytes` (>= 16 MiB, 16-byte aligned, zero
 * filled) with the tiled kernel's access pattern -- 512 workgroups, each its own contiguous slice, four 16-byte non-temporal loads
 * in flight per thread; one warm-up launch, then `reps` timed ones (HIP events on `stream`).  bench.py reports it beside the
 * nominal 8 TB/s as `roofline.measured_read_ceiling`. */
int pdlp_probe_stream_read(const void* buf, int64_t bytes, int reps, void* stream, double* gb_per_s);
/* The other roof (no counterpart in the reference either): G items/s of "8 bytes of item stream + ONE random 4-byte gather per item"
 * over a zero table of `table_entries` floats, in the CSR kernel's launch shape (256-thread workgroups, blocks of 2048 items, all
 * item loads of a block in one round, then all gathers).  A CSR product whose gathers share no cache lines (5 non-zeros per row
 * over 1M columns) cannot run faster than this, whatever HBM could stream: bench.py reports it as `roofline.gather_ceiling` when
 * the working set lives in the Infinity Cache.  scratch: 256-byte aligned, >= 4 * table_entries + 1 MiB + 8 bytes per item to be
 * streamed (the library fills it: table, then uniform random columns and unit values). */
int pdlp_probe_gather(void* scratch, int64_t scratch_bytes, int64_t table_entries, int reps, void* stream, double* gitems_per_s);

/* ---- Ruiz equilibration on CSR (ruiz_precondition enhancements.py:4-71) ---------------------- */
/* norm[i] = sqrt(max_p |val[p]|) over row i, replaced by 1 when < eps (:49-50 / :54-55) */
int pdlp_csr_row_scale_factors(int dtype, int64_t rows, const int64_t* rowptr, const void* val, double eps,
                               void* norm, void* stream);
/* val[p] /= norm[row(p)]                      (:52 / :57 on the copy whose rows are being scaled) */
int pdlp_csr_div_rows(int dtype, int64_t rows, const int64_t* rowptr, void* val, const void* norm, void* stream);
/* val[p] /= norm_full[colidx[p]]              (the same scaling applied to the transposed copy)   */
int pdlp_csr_div_cols(int dtype, int64_t nnz, const int32_t* colidx, void* val, const void* norm_full, void* stream);
/* elementwise helpers for D /= norm, c*D, l/D ... (:51,:56,:64-67); op: 0 a*=b, 1 a/=b */
int pdlp_vec_muldiv(int dtype, int64_t len, void* a, const void* b, int op, void* stream);
/* out = project_lambda_box(g) (helpers.py:3-39): 0 where l = -inf and u = +inf, min(g,0) where only l = -inf,
 * max(g,0) where only u = +inf, g otherwise.  (Inside the solver this projection is fused into the KKT and
 * infeasibility epilogues; this entry point serves the reference-named operator.) */
int pdlp_vec_project_lambda(int dtype, int64_t len, const void* g, const void* l, const void* u, void* out, void* stream);
/* max_i |1 - v[i]| (the early-exit test :60-61); host result, synchronises */
int pdlp_vec_max_dev_from_one(int dtype, int64_t len, const void* v, void* work8, double* out, void* stream);
/* sum_i (a[i] - b[i])^2, accumulated in double: the two distances of primal_weight_update (enhancements.py:74-75) for the
 * reference-named operator (inside the solver: pdlp_restart_distance_local).  work: 1040 doubles; host result, synchronises */
int pdlp_vec_sqdist(int dtype, int64_t len, const void* a, const void* b, void* work, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PDLP_HIP_H */
