// Internal declarations shared by the translation units of libsafe_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "safe_hip.h"

#define SAFE_WAVE 64
#define SAFE_PAD_U16 0xFFFFu

void safe_set_error(const char *fmt, ...);
// SAFE_HIP_TRACE=1: host-side time stamps (ms since the first call) on stderr
void safe_trace(const char *what);

// hipStreamSynchronize, or a sleeping wait on a blocking event when blocking waits are on (ctx.hip)
hipError_t safe_stream_sync(hipStream_t s);
// flags for events the host may wait on: adds hipEventBlockingSync while blocking waits are on
unsigned safe_event_flags(unsigned base);
bool safe_blocking_sync_selected();
// A diagnostic switch that makes a kernel skip work (timing experiments: SAFE_HIP_BITS_DBG, SAFE_HIP_MFMA_DBG*) is in effect: says so
// on stderr, once per switch -- results of such a run are WRONG by construction and must not be mistaken for the product's.
void safe_warn_diagnostic(const char *name);
bool safe_hidden_regs_checked();                // buildinfo.cpp: the build's disassembly check of the id-stream registers passed   // blocking (sleeping) host waits are on: do not spin

#define SAFE_HIP_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            safe_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                           __LINE__);                                                          \
            return SAFE_E_HIP;                                                                 \
        }                                                                                      \
    } while (0)

#define SAFE_REQUIRE(cond, ...)                                                                \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            safe_set_error(__VA_ARGS__);                                                       \
            return SAFE_E_INVALID;                                                             \
        }                                                                                      \
    } while (0)

#define SAFE_TRY(expr)                                                                         \
    do {                                                                                       \
        int _rc = (expr);                                                                      \
        if (_rc != SAFE_OK) return _rc;                                                        \
    } while (0)

struct safe_perms;

struct KernelStat {
    std::string name;
    double total_ms = 0.0;
    int64_t launches = 0;
    double busy_ms = 0.0;           // union of the launches' intervals (consecutive launches overlap on two streams); 0 = not measured
};

struct safe_perms;
// The context's draw thread (rng.cpp): one persistent host thread runs the sequential part of every seeded stream of the
// context (np.random.seed / np.random.permutation, safepy/safe_extras.py:46,58).  A thread created per call started on whatever
// idle core the scheduler found -- clock and caches cold -- and one step in twenty drew 1.5-2 x slower; the persistent thread
// spins briefly for its next job before it sleeps, so back-to-back calls never pay a wake-up either.
struct DrawWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;                 // job posted / worker idle again
    std::atomic<safe_perms *> job{nullptr};     // posted, not yet taken
    bool busy = false;                          // (under mu) a job is running
    bool quit = false;
};
void draw_worker_shutdown(safe_ctx *ctx);

struct safe_ctx {
    int device = 0;
    DrawWorker *draw_worker = nullptr, *draw_worker2 = nullptr;    // (the second one runs the twin chain: safe_perms::twin)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;           // permutation-table generation (overlaps the enrichment kernels)
    hipStream_t side_stream = nullptr;          // second enrichment stream: consecutive spans overlap their tails
    hipStream_t more_streams[2] = {nullptr, nullptr};   // third and fourth (bit-sliced kernel: deeper overlap of consecutive launches)
    int num_cu = 0;
    int64_t hbm_bytes = 0;
    char arch[64] = {0};
    hipEvent_t t0 = nullptr, t1 = nullptr;      // safe_timer_*
    hipEvent_t k0 = nullptr, k1 = nullptr;      // dominant-kernel timing
    KernelStat last_kernel;
    int last_slices = 0;                        // i8 slices of the last matrix-core permutation test (2 / 4 / 6)
    void *diag_prof = nullptr;                  // (diagnostic builds) per-phase cycle counters of the matrix-core kernel
    int last_core_slices = 0;                   // slices its matrix-core kernel multiplied (3 of 6 in the filtered form)
    int64_t last_undecided = 0;                 // filtered form: compares decided by k_mfma_resolve (negative: list overflow, six-slice rerun)
    // grow-only scratch buffers reused across calls (hipMalloc of >100 MB costs milliseconds)
    struct safe_perms *perm_cache = nullptr;    // buffers of the last destroyed permutation handle, reused by the next
    struct PermRing *ring = nullptr;            // node-shared permutation stream (safe_ctx_share_stream), or NULL
    // packed <= / >= counters of the last integer-counter permutation kernel (scratch slot 0):
    // u32 [packed_m][packed_n_pad] = (#less << 16 | #greater); layout 0 = SELL positions,
    // 1 = block order of the MFMA kernel, -1 = none (safe_export_packed_counts)
    const unsigned int *packed_counts = nullptr;
    int64_t packed_n_pad = 0, packed_m = 0, packed_perms = 0;
    int packed_layout = -1;
    // Exchange overlap of the sharded step (safe_set_exchange_chunks; sharding.py): the bit-sliced launches run the LAST part of
    // the permutations column chunk by column chunk, so that a chunk's counters are final -- and can travel -- while the later
    // chunks still compute.  xc_want / xc_cols: what the caller armed; xc_made / xc_bounds / xc_events: what the last call did.
    static constexpr int XC_MAX = 8;
    int xc_want = 0;
    int64_t xc_cols = 0;
    void (*xc_callback)(void *) = nullptr;      // called once every launch of the call is enqueued (before the call's own sync)
    void *xc_user = nullptr;
    int xc_made = 0;
    int64_t xc_bounds[XC_MAX + 1] = {};         // column boundaries of the chunks (this rank's columns: clipped to packed_m)
    int64_t xc_tail_perms = 0;                  // permutations the column-chunked launches covered
    hipEvent_t xc_events[XC_MAX] = {};          // chunk k's counters are final
    std::vector<double> nes_tab_host;           // safe_outputs_from_packed_slabs: the table it uploaded last (no sync per call)
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    // large device-to-host copies into pageable memory (safe_memcpy_d2h): a ring of pinned slots the DMA engine fills while host
    // threads copy finished slots into the destination (and take its first-touch page faults in parallel)
    static constexpr int D2H_SLOTS = 8;
    static constexpr size_t D2H_SLOT_BYTES = size_t(4) << 20;
    void *d2h_ring = nullptr;
    std::mutex d2h_mu;                          // the ring serves one read-back at a time
    hipEvent_t d2h_events[D2H_SLOTS] = {};
    std::vector<hipEvent_t> ev_timing, ev_plain;   // reused per-launch events (creating 20 per call costs ~0.1 ms)
    std::vector<std::pair<size_t, void *>> block_cache;   // small device blocks of destroyed handles (ctx_block_alloc)
    static constexpr int N_SCRATCH = 21;
    void *scratch[N_SCRATCH] = {};
    size_t scratch_bytes[N_SCRATCH] = {};
};

// returns a device buffer of at least `bytes` from slot `slot`, valid until the next request
// for the same slot; contents are undefined
int ctx_scratch(safe_ctx *ctx, int slot, size_t bytes, void **out);
// grow-only pinned host buffer (device-to-host copies into it go through the DMA engines: no copy kernel
// that would have to wait for a CU while a persistent kernel holds all of them)
int ctx_pinned(safe_ctx *ctx, size_t bytes, void **out);
// `count` reusable events of the context (timing-enabled, or hipEventDisableTiming); valid until the next request of the same kind
int ctx_events(safe_ctx *ctx, bool timing, size_t count, hipEvent_t **out);
void perms_cache_drop(safe_ctx *ctx);   // frees ctx->perm_cache (rng.cpp)
// small per-handle device blocks (row flags, column sums): hipMalloc + hipFree cost tens of microseconds each and a handle
// is made per compute_pvalues pass, so freed blocks wait in the context for the next handle of the same shape
int kernel_stat_from_events(safe_ctx *ctx, hipEvent_t *ev, int64_t n_launch);   // enrich.hip
int ctx_block_alloc(safe_ctx *ctx, size_t bytes, void **out);
void ctx_block_free(safe_ctx *ctx, void *p, size_t bytes);

// RAII-less device buffer helper: all frees go through the owning handle's destroy.
// calls of hipMalloc / hipHostMalloc made by the library so far (safe_alloc_count): a timed step is expected to make none
extern std::atomic<long long> g_alloc_calls;

template <typename T>
static inline int dev_alloc(T **p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T));
    if (e != hipSuccess) {
        safe_set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        return SAFE_E_NOMEM;
    }
    return SAFE_OK;
}

struct BitsQueues {
    int off[9];                                                           // tasks [off[q], off[q+1]) belong to queue q
};
// Task lists of the bit-sliced permutation kernel for one launch plan (launch_bits): a function of the membership structure and of
// `key` only, so the handle keeps the last one (building them took 90 us of every compute_pvalues pass)
struct BitsTaskPlan {
    std::vector<int64_t> key;
    std::vector<int4> tasks;                                              // the lists back to back
    std::vector<int64_t> list_span, list_first, list_count, launch_list;
    std::vector<BitsQueues> list_queues;
};

// Membership structure.  Canonical form: bit matrix; derived: CSR and SELL-64 (sliced
// ELLPACK with rows sorted by descending count so each 64-row slice is nearly uniform).
struct safe_nbr {
    safe_ctx *ctx = nullptr;
    int64_t n = 0;
    int64_t nnz = 0;
    int64_t max_count = 0;
    int64_t words = 0;              // 64-bit words per bit-matrix row
    uint64_t *bits = nullptr;       // [n][words]
    int32_t *row_ptr = nullptr;     // [n+1]
    int32_t *col = nullptr;         // [nnz]
    // SELL-64
    int64_t n_slices = 0;
    int32_t *sell_row = nullptr;    // [n_slices*64] original row id, -1 = padding lane
    int32_t *sell_pos = nullptr;    // [n] position of row i in sell_row (inverse map)
    int64_t *slice_off = nullptr;   // [n_slices+1] offset into sell_col (entries)
    int32_t *slice_width = nullptr; // [n_slices]
    int32_t *sell_col = nullptr;    // [slice_off[n_slices]] column id, n = padding (zero row)
    int64_t sell_entries = 0;
    uint16_t *sell_col2 = nullptr;  // [sell_entries + 1024] 2*column id as u16 (n < 32768 only): LDS byte offsets of a u16 table
    uint16_t *sell_col2b = nullptr; // the same list in blocked order: 8 members of a lane adjacent (one 16-byte load per lane and block)
    std::vector<int32_t> h_slice_width;
    BitsTaskPlan bits_plan;         // (launch_bits' cache)
    void *bits_plan_pinned = nullptr;              // the plan's task lists in pinned host memory (grow-only): uploaded from there every
    size_t bits_plan_pinned_bytes = 0;             // pass -- an async copy from pageable memory pins the pages first (an ioctl that
                                                   // was seen to take milliseconds once in a few hundred passes)
    std::vector<int64_t> h_slice_off;
    std::vector<int32_t> h_row_count;   // host copy of per-row counts
    double *dist = nullptr;         // optional [n][n]
    // CSR of the transpose (column k -> rows i with A[i,k] = 1), built on first use
    int32_t *at_ptr = nullptr;      // [n+1]
    int32_t *at_col = nullptr;      // [nnz], unordered inside a column
    // Block-sparse form for the MFMA permutation kernel (mfma.hip), built on first use:
    // nodes in a locality-preserving order (Hilbert curve over the layout when one is known,
    // Cuthill-McKee over the membership graph otherwise); 256-row groups x 32-column blocks,
    // only blocks holding at least one member are stored, as 256 32-bit row words each.
    std::vector<double> h_xy;       // [n][2] layout, if known (safe_nbr_euclidean / safe_nbr_set_layout)
    bool blocks_ready = false;
    int64_t bs_groups = 0;          // number of 256-row groups
    int64_t bs_blocks = 0;          // stored blocks (every group padded to a multiple of 4)
    int64_t bs_pieces = 0;          // 32-row x 32-column pieces of the stored blocks that hold at least one member (the only ones multiplied)
    int64_t bs_src = 0;             // length of a source-row map: (ceil(n/32)+1)*32, the last block is padding
    int32_t *bs_order = nullptr;    // [bs_src] node at ordered position u (n = padding -> zero attribute row)
    int32_t *bs_rowmap = nullptr;   // [bs_groups*256] node at ordered row u, -1 = padding
    int32_t *bs_rowcnt = nullptr;   // [bs_groups*256] members of that node's neighborhood (-1 = padding row)
    int32_t *bs_grpmax = nullptr;   // [bs_groups] largest neighborhood of the group's rows
    uint4 *bs_bits4 = nullptr;      // the membership bits again, [super-step of 4 blocks][256 rows][4]: one 16-byte load per row and super-step
    uint4 *bs_bits4p = nullptr;     // bs_bits4 with the bits of every word in operand order: bit 8 b + 4 h + j = member 16 h + 4 j + b of the block
                                    // (k_permtest_mfma_g: register j of lane half h is (word >> (4 h + j)) & 0x01010101)
    int64_t bs_max_group_blocks = 0;   // blocks of the longest group
    int32_t *bs_ptr = nullptr;      // [bs_groups+1] first block of a group
    int32_t *bs_kb = nullptr;       // [bs_blocks] ordered column block of a stored block
    uint32_t *bs_bits = nullptr;    // [bs_blocks][256] membership bits of the block's 256 rows
    std::vector<int32_t> h_bs_ptr;
    std::vector<int32_t> h_bs_rowmap;   // host copy of bs_rowmap
    // task list of the matrix-core COUNT kernel (one i8 plane per tile): (row group, column group) pairs in eight per-XCD queues;
    // a function of the block structure and the number of column groups only, kept for the next call (counts_setup)
    std::vector<int2> counts_tasks;
    int32_t counts_qoff[9] = {0};
    int64_t counts_tasks_grp = -1;
};

// Hypergeometric epilogue by table lookup (enrich.hip: k_hyp_table builds tab): instead of a count X
// a kernel writes p = tab[(nid[row] * n_kid + kid[col]) * xs + X], -log10 p, the binarised value and
// the per-attribute enriched counts (safe.py:596-608, 468-472).  tab == NULL: plain counts as f64
// into pvalues_pos (compute_neighborhood_score 'sum' of 0/1 data).
struct HypLookup {
    const int32_t *nid;        // [n] index of the row's neighborhood size among the distinct sizes
    const int32_t *kid;        // [mloc] index of the column's annotation count among the distinct counts
    const double2 *tab;        // [n_nid][n_kid][xs] of (p, -log10 p)
    int64_t n_kid, xs;
    double p_cut;              // binarisation as a bound on p (nes_p_cut)
    double *dummy;             // [64] write-only slots for the padding rows / columns of branch-free epilogues
    double *pvalues_pos, *nes, *nes_binary;
    unsigned int *enriched;
    unsigned int *cnt16;       // split form of the matrix-core counts: packed u16 counts [group][position][32][6], else NULL
    unsigned int *xmax;        // split form: largest count of the call (written by the count kernel, read by the emit kernel)
    int dbg = 0;               // diagnostic builds of the matrix-core permutation kernel (SAFE_HIP_MFMA_DBG; wrong results)
};

// |nes| > -log10(enrichment_threshold) (safe.py:468-470) for nes = -log10(p), p in [0, 1], restated as a
// bound on p itself: hit <=> p < p_cut with p_cut = the smallest double whose -log10 (host libm, the one
// NumPy calls) does NOT exceed the threshold.  log10 is monotone, so the decision for a given p is the
// reference's whatever the device's own log10 rounds to.
static inline double nes_p_cut(double enrichment_threshold) {
    const double thr = -std::log10(enrichment_threshold);
    auto hit = [&](double p) { return -std::log10(p) > thr; };
    double p = enrichment_threshold;
    for (int it = 0; it < 4096 && hit(p); ++it) p = std::nextafter(p, 1.0);
    for (int it = 0; it < 4096 && !hit(std::nextafter(p, 0.0)); ++it) p = std::nextafter(p, 0.0);
    return p;
}

// outputs of the permutation-test kernels (enrich.hip, mfma.hip)
struct PermOut {
    double *ns;            // [n][mloc] or NULL
    double *counts_neg;    // raw-count mode
    double *counts_pos;
    double *pvalues_neg;   // full mode
    double *pvalues_pos;
    double *nes;
    double *nes_binary;
    unsigned int *enriched;   // [mloc] u32 column counters
    const double *nes_table;  // [P+1]
    double nes_threshold;     // -log10(enrichment_threshold)
    int sign_mode;
    int mode;                 // 0 = score only, 1 = raw counts, 2 = full post-processing
    int64_t ld;               // row pitch of the output matrices in elements (0 = the column count of the call)
};

struct safe_attr {
    safe_ctx *ctx = nullptr;
    int64_t n = 0, m = 0;
    int dtype = SAFE_DTYPE_F64;
    int64_t row_stride = 0, col_stride = 0;
    const void *raw = nullptr;      // device
    bool owns_raw = false;
    uint8_t *row_flags = nullptr;   // [n] device, 1 = row has >= 1 non-NaN value
    std::vector<uint8_t> h_row_flags;   // host copy of row_flags (kept in step: safe_attr_row_flags answers without a device sync)
    bool flags_ready = false;
    bool stats_ready = false;
    int64_t n_other = 0, max_nan_col = 0, n_rows_with_value = 0, n_non_integer = 0;
    double *col_sum = nullptr;      // [m] nansum per column (device)
    double max_abs = 0.0;           // max |value| over non-NaN entries
    // binary matrices only: support lists (CSC of the ones), built on first use
    int32_t *sup_ptr = nullptr;     // [m+1]
    int32_t *sup_row = nullptr;     // [n_ones]
    int64_t n_ones = 0;
    std::vector<int32_t> h_sup_ptr; // host copy
};

struct DrawStream;   // host MT19937 draw stream (rng.cpp)

struct safe_perms {
    safe_ctx *ctx = nullptr;
    int64_t n = 0;
    int64_t count = 0;
    int64_t k = 0;                  // number of movable rows (indx_vals)
    std::vector<int32_t> h_movable;
    DrawStream *stream = nullptr;
    // pipeline positions (in permutations): drawn >= enqueued on the GPU
    int64_t generated = 0, enqueued = 0;
    std::vector<int64_t> stages;                   // pipeline stage boundaries of this handle (perms_stage_plan(count))
    std::vector<hipEvent_t> chunk_done;            // recorded on ctx->aux_stream after each enqueued chunk
    std::vector<uint32_t> h_local;                 // the draw thread's private buffer: the targets of one shuffle (L1-resident)
    // the draw thread: runs the sequential MT19937 / rejection stream chunk by chunk into the pinned staging buffers, at most
    // kStage chunks ahead of the uploads, independent of the thread that launches kernels
    std::thread drawer;                            // a thread of its own (only when the context's draw worker is busy with another handle)
    bool on_worker = false;                        // this handle's draws run (or ran) on ctx->draw_worker
    bool worker_done = false;                      // (under the worker's mutex) ... and have ended
    // The TWIN: a second draw thread runs the very same chain (same seed, same words) on another core into buffers of its own,
    // and whichever thread finishes a chunk first publishes it.  The chain is sequential and cannot be sped up, but on a shared
    // host one thread runs 1.5-2 x slower for a millisecond or two now and then (a neighbour on its sibling hardware thread, a
    // pre-emption) -- the idea was that both threads stumbling in the same chunk is rare.  Measured: no better than one thread
    // (rng.cpp, perms_create_impl), so it is opt-in (SAFE_HIP_DRAW_TWIN=1).
    bool twin = false;
    bool on_worker2 = false, worker_done2 = false;
    DrawStream *stream2 = nullptr;
    void *h_stage2[3] = {nullptr, nullptr, nullptr};
    hipEvent_t staged2[3] = {nullptr, nullptr, nullptr};
    std::vector<uint32_t> h_local2;
    std::vector<uint8_t> chunk_src;                // (under draw_mu) which thread's staging buffer holds chunk c
    int64_t twin_wins = 0;                         // chunks the twin published first (diagnostics)
    std::mutex draw_mu;
    std::condition_variable draw_cv;
    int64_t drawn_chunks = 0;                      // chunks whose targets are complete in h_stage[c % kStage]
    std::atomic<int64_t> drawn_chunks_pub{0};      // the same, readable without the mutex (the launcher spins on it)
    int64_t enqueued_chunks = 0;                   // chunks whose upload has been queued (staged[c % kStage] recorded)
    bool draw_stop = false;
    bool draw_failed = false;        // the draw thread gave up (under draw_mu): waiters return an error instead of waiting for ever
    static constexpr int kStage = 3;
    void *h_stage[kStage] = {nullptr, nullptr, nullptr};   // pinned: accepted swap targets of a chunk, [chunk][target_width] u16 (k <= 65535) or u32
    hipEvent_t staged[kStage] = {nullptr, nullptr, nullptr};
    size_t stage_bytes = 0;                        // capacity of each staging buffer (sized for k = n)
    int target_bytes = 2;                          // bytes per swap target on the wire (2 when k <= 65535)
    int64_t target_width = 0;                      // targets per permutation row: k - 1 rounded up to 8
    void *d_targets = nullptr;                     // device copy of the chunk being replayed (even chunks)
    void *d_targets_odd = nullptr;                 // ... odd chunks: a chunk's upload and replay run beside its predecessor's scan
    int32_t *d_maps_odd[2] = {nullptr, nullptr};   // row maps / scan ping-pong of the odd chunks
    hipEvent_t replayed[2] = {nullptr, nullptr};   // by chunk parity: the replay has written its row maps (recorded on the replay stream)
    hipEvent_t movpos_ready = nullptr;             // d_movpos uploaded (aux stream) -- the replay stream waits for it once per handle
    int32_t *d_big = nullptr;                      // k > 65535 only: [chunk][k] position arrays of the global-memory replay
    int32_t *h_movpos = nullptr;                   // pinned [2n]: staging of d_movpos
    int32_t *d_maps[2] = {nullptr, nullptr};       // [chunk][n+1] row maps / scan ping-pong
    int32_t *d_cur = nullptr;       // [n+1] running composition (last emitted row)
    int32_t *table = nullptr;       // [count][n+1] device; entry n is the padding row (== n)
    // 16-bit copy of the table, rows padded to a multiple of 8 entries (n < 65535 only)
    uint16_t *table16 = nullptr;
    int64_t stride16 = 0;
    // transposed inverse permutations, built on first use (n < 65535 only):
    // inverse_t[r * inv_stride + p] = position k with table[p][k] == r; inv_stride = padded count
    uint16_t *inverse_t = nullptr;
    int64_t inv_stride = 0;
    bool from_table = false;        // rows supplied by the caller (safe_perms_create_from_table): no stream, complete from the start
    // node-shared stream (ring.h; safe_perms_create_shared): local rank 0 of a node publishes every chunk's row maps,
    // the other ranks fetch them instead of drawing (no draw thread, no swap workers on those ranks)
    bool device_gen = false;        // tables generated on the device (safe_perms_create_device): no host stream at all
    int32_t *d_movpos = nullptr;    // [n] movable rows (first k used) | [n] position of a row in that list (-1: fixed)
    struct PermRing *ring = nullptr;               // the context's ring while this handle takes part in a shared call, else NULL
    bool ring_consumer = false;
    // host-side timing of the stream (safe_perms_timing), ms since the handle was created
    double t_created_s = 0.0, draw_busy_ms = 0.0, drawn_all_ms = 0.0, enqueued_all_ms = 0.0, ring_wait_ms = 0.0;
};

// launch-geometry helpers
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

int safe_attr_prepare(safe_attr *attr);   // row flags + stats (attr.hip)
int nbr_finalize_from_bits(safe_nbr *nbr);   // bits -> CSR + SELL (nbr.hip)
int nbr_build_transpose(safe_nbr *nbr);      // at_ptr / at_col (nbr.hip)
int attr_build_support(safe_attr *attr);     // sup_ptr / sup_row of a binary matrix (attr.hip)
int perms_build_inverse(safe_perms *perms);  // inverse tables (rng.cpp)
int perms_generate_until(safe_perms *perms, int64_t upto);   // enqueue table rows [generated, upto) on aux_stream (rng.cpp)
// host/GPU pipeline stages of the permutation stream for `count` permutations: boundaries b[0] = 0 < b[1] < ... < b.back() = count
// (a short first stage, 128 each, short last stages -- rng.cpp)
std::vector<int64_t> perms_stage_plan(int64_t count);
// launch boundaries of the permutation kernels: starts[c] .. starts[c+1]; the default follows the
// stream's pipeline stages (so the first launch can start after 32 permutations have been drawn),
// SAFE_HIP_BITS_SPAN=<n> forces uniform spans.  *span = the longest launch.
// merge > 1: after the three start-up stages (32, 96, 128 permutations) a launch covers `merge` stages -- fewer,
// longer launches once the stream is ahead of the kernels.
static inline std::vector<int64_t> perm_launch_starts(const safe_perms *perms, int64_t *span, int merge = 1) {
    const int64_t P = perms->count;
    std::vector<int64_t> starts;
    int64_t uniform = 0;
    if (const char *e = getenv("SAFE_HIP_BITS_SPAN")) uniform = std::max<int64_t>(16, atoll(e));
    if (uniform > 0) {
        for (int64_t p = 0; p < P; p += uniform) starts.push_back(p);
    } else {
        const std::vector<int64_t> &plan = perms->stages;          // the handle's own stages (host pipeline, or even spans for device tables)
        const int64_t nc = static_cast<int64_t>(plan.size()) - 1;
        for (int64_t c = 0; c < nc; ++c)
            if (c < 3 || merge <= 1 || (c - 3) % merge == 0) starts.push_back(plan[c]);
    }
    if (starts.empty()) starts.push_back(0);
    starts.push_back(std::max<int64_t>(P, 0));
    *span = 1;
    for (size_t c = 0; c + 1 < starts.size(); ++c) *span = std::max<int64_t>(*span, starts[c + 1] - starts[c]);
    return starts;
}
int perms_wait(safe_perms *perms, int64_t upto, hipStream_t s);   // make stream s wait until rows [0, upto) exist (rng.cpp)
// counters [column][position] (#less << 16 | #greater) -> outputs; rowmap[position] = row or -1 (enrich.hip)
int enrich_finalize_counts(safe_ctx *ctx, const unsigned int *counts, int64_t n_pad, const int32_t *rowmap, int64_t mloc,
                           int64_t n_perm, const PermOut &out, const double *ns_direct, hipStream_t on = nullptr, bool pk20 = false);
// MFMA (i8, exact fixed point) form of the permutation test for quantitative attributes (mfma.hip)
bool mfma_applicable(const safe_ctx *ctx, const safe_nbr *nbr, const safe_attr *attr, const safe_perms *perms, bool z);
int launch_mfma(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, safe_perms *perms, int64_t col0, int64_t col1, bool z,
                const PermOut &out, bool *declined);
// X = A . B0 for 0/1 attributes on the matrix cores (block-sparse, one i8 plane per 32-column tile),
// written through `hl` (mfma.hip); sets ctx->last_kernel and the k0/k1 timing events
int launch_mfma_counts(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, int64_t col0, int64_t col1, const HypLookup &hl);
struct MfmaCountsSplit;
bool mfma_counts_split_applicable(const safe_nbr *nbr);
int mfma_counts_split_begin(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, int64_t col0, int64_t col1, MfmaCountsSplit **out);
int mfma_counts_split_rows(safe_ctx *ctx, safe_nbr *nbr, MfmaCountsSplit *st, const int32_t *h_nid, hipStream_t hs, void *pinned_stage);
int mfma_counts_split_emit(safe_ctx *ctx, safe_nbr *nbr, MfmaCountsSplit *st, const HypLookup &hl);
void mfma_counts_split_free(MfmaCountsSplit *st);
const unsigned int *mfma_counts_split_xmax(const MfmaCountsSplit *st);   // device: largest count of the call
void nbr_free_blocks(safe_nbr *nbr);
