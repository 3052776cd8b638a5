// Row-permutation stream of run_permutations (safepy/safe_extras.py:46-58): legacy NumPy
// RandomState = MT19937 seeded by init_genrand; np.random.permutation = Fisher-Yates from the
// top with masked-rejection bounded integers; permutations applied
// cumulatively to the rows that hold at least one value.
//
// Split of the work.  The only inherently sequential part is the draw stream: how many
// 32-bit outputs a shuffle consumes depends on its rejections, so permutation q+1 cannot
// start before q has finished drawing, and the state a draw leaves behind depends on every
// word before it (two runs of the rule started a few words apart never meet again: measured,
// tools/ubench/chain_merge.c).  One host thread therefore produces just the accepted swap
// targets j[q][i] (draws.cpp: bulk MT19937 generation + batch-resolved rejection, ~0.3 ns per
// draw) straight into pinned staging memory, 2 bytes per target.  Everything else runs on the
// device, chunk by chunk, overlapping the draw thread's next chunk and the enrichment kernels
// of the previous one:
//   k_replay_targets   one wave per permutation: the Fisher-Yates swaps replayed on an LDS copy
//                      of the positions, 64 steps at a time (steps that touch a location a lower
//                      lane of the batch also touches are settled in lane order afterwards), then
//                      the row map M_q
//   k_scan_round x log2(chunk), k_emit_rows
//                      cur_q = cur_{q-1} o M_q is a prefix product under composition: a
//                      log-depth parallel scan, emitting the composed table rows
#include <algorithm>
#include <random>
#include <condition_variable>
#include <mutex>
#include <thread>

#include <pthread.h>

#include <chrono>

#include "common.h"
#include "draws.h"
#include "ring.h"

static inline void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}
static double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static uint32_t entropy_seed() {
    std::random_device rd;
    return rd();
}

// --------------------------------------------------------------------------------------
// device side
// --------------------------------------------------------------------------------------
// Composition of the chunk's row maps.  The host hands over, for every permutation q of the
// chunk, the full-length map M_q (M_q[i] = i for rows that do not move, M_q[indx_vals[t]] =
// shuffled[t]); the composed table is cur_q = cur_{q-1} o M_q, i.e. a prefix product under
// composition.  Composition is associative, so the prefix is a log-depth scan
// (Hillis-Steele): X_q <- X_{q-d} o X_q for d = 1, 2, 4, ...; every round is a fully parallel
// gather, then cur_q = cur_base o X_q.
__global__ __launch_bounds__(256) void k_scan_round(const int32_t *__restrict__ xin, int32_t *__restrict__ xout,
                                                    int64_t cnt, int64_t stride, int64_t d) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t q = blockIdx.y;
    if (i >= stride) return;
    const int32_t v = xin[q * stride + i];
    xout[q * stride + i] = q >= d ? xin[(q - d) * stride + v] : v;
}

__global__ __launch_bounds__(256) void k_emit_rows(const int32_t *__restrict__ x, const int32_t *__restrict__ cur_base,
                                                   int64_t cnt, int64_t n, int32_t *__restrict__ table,
                                                   uint16_t *__restrict__ table16, int64_t stride16) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t q = blockIdx.y, stride = n + 1;
    if (i < stride) {
        const int32_t v = cur_base[x[q * stride + i]];
        table[q * stride + i] = v;
        if (table16) table16[q * stride16 + i] = static_cast<uint16_t>(v);
    } else if (table16 && i < stride16) {
        table16[q * stride16 + i] = static_cast<uint16_t>(n);
    }
}

// the last scan round and the emit in one launch (a chunk's chain is launch-latency bound: every launch less is ~6 us
// sooner for the tables of the pipeline's first stages)
__global__ __launch_bounds__(256) void k_scan_emit(const int32_t *__restrict__ xin, int64_t d, const int32_t *__restrict__ cur_base,
                                                   int64_t cnt, int64_t n, int32_t *__restrict__ table,
                                                   uint16_t *__restrict__ table16, int64_t stride16) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t q = blockIdx.y, stride = n + 1;
    if (i < stride) {
        int32_t v = xin[q * stride + i];
        if (q >= d) v = xin[(q - d) * stride + v];
        v = cur_base[v];
        table[q * stride + i] = v;
        if (table16) table16[q * stride16 + i] = static_cast<uint16_t>(v);
    } else if (table16 && i < stride16) {
        table16[q * stride16 + i] = static_cast<uint16_t>(n);
    }
}

// 16-bit copy of a caller-supplied table, rows padded with the padding row's id
__global__ __launch_bounds__(256) void k_table16(const int32_t *__restrict__ table, int64_t stride, uint16_t *__restrict__ table16,
                                                  int64_t stride16, int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t q = blockIdx.y;
    if (i < stride16) table16[q * stride16 + i] = static_cast<uint16_t>(i < stride ? table[q * stride + i] : n);
}

__global__ void k_iota(int32_t *p, int64_t count) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < count) p[i] = static_cast<int32_t>(i);
}

__global__ void k_invert_perms(const int32_t *__restrict__ table, int64_t stride, int64_t total, int64_t inv_stride,
                               uint16_t *__restrict__ inverse_t) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int64_t p = idx / stride, k = idx % stride;
    inverse_t[static_cast<int64_t>(table[idx]) * inv_stride + p] = static_cast<uint16_t>(k);
}

// --------------------------------------------------------------------------------------
// Device-side stream for UNSEEDED runs (random_seed=None, the reference's default: safe.py:88, np.random.seed(None) at
// safe_extras.py:46 seeds from OS entropy, so there is no stream to reproduce).  The reference shuffles the rows in place once
// per iteration (safe_extras.py:58): cur_q = cur_{q-1} o M_q with independent uniform M_q, and a uniform permutation composed
// with anything independent of it is uniform and independent of the past -- the tables cur_1 .. cur_P are i.i.d. uniform
// permutations of the movable rows.  So every table row is generated directly and independently, one wave per permutation
// (k_perms_device below: a scatter shuffle, 64 buckets + Fisher-Yates inside each), all random bits from Philox4x32-10 keyed
// by the call's 64-bit key with counters that name (permutation, element or bucket / step) -- every draw is independent of
// every other -- and bounded draws without bias (Lemire's multiply-shift with rejection).  Every rank of a sharded run
// generates the SAME tables from the agreed key: no host thread, no exchange, nothing sequential across permutations.
// The algorithm is restated by the tests' CPU checker (its device_stream_tables) and compared bit for bit.
// --------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011): four 32-bit words from a 128-bit counter and a 64-bit key
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&w)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    w[0] = c0, w[1] = c1, w[2] = c2, w[3] = c3;
}

// bucket of element e of permutation q: six bits of a byte of the Philox words of counter (q, e >> 4, 0xB0C7, 0)
__device__ __forceinline__ uint32_t device_bucket_word(uint32_t q, uint32_t e, uint32_t key0, uint32_t key1, uint32_t &cached_block,
                                                       uint32_t (&w)[4]) {
    if ((e >> 4) != cached_block) {
        cached_block = e >> 4;
        philox4x32_10(q, cached_block, 0xB0C7u, 0u, key0, key1, w);
    }
    const uint32_t word = (e >> 2) & 3u;
    const uint32_t v = word == 0 ? w[0] : word == 1 ? w[1] : word == 2 ? w[2] : w[3];
    return (v >> (8u * (e & 3u))) & 63u;
}

// the draw of step i of bucket b of permutation q: uniform on [0, i] by Lemire's multiply-shift; the word is word (i & 3) of
// counter (q, b << 16 | i >> 2, 0x5AFE, 0); the few low products that would bias the draw (probability (i + 1) / 2^32) are
// rejected and the draw falls back to the words of counters (q, b << 16 | i, 0xFA11, r), r = 0, 1, ..., in order
__device__ __forceinline__ uint32_t device_draw(uint32_t q, uint32_t b, uint32_t i, uint32_t key0, uint32_t key1) {
    const uint32_t range = i + 1u, thresh = (0u - range) % range;
    uint32_t w[4];
    philox4x32_10(q, (b << 16) | (i >> 2), 0x5AFEu, 0u, key0, key1, w);
    const uint32_t first = (i & 3u) == 0 ? w[0] : (i & 3u) == 1 ? w[1] : (i & 3u) == 2 ? w[2] : w[3];
    uint64_t m = static_cast<uint64_t>(first) * range;
    if (static_cast<uint32_t>(m) >= thresh) return static_cast<uint32_t>(m >> 32);
    for (uint32_t r = 0;; ++r) {
        philox4x32_10(q, (b << 16) | i, 0xFA11u, r, key0, key1, w);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            m = static_cast<uint64_t>(w[t]) * range;
            if (static_cast<uint32_t>(m) >= thresh) return static_cast<uint32_t>(m >> 32);
        }
    }
}

// One wave = one permutation, shuffled in three parallel steps (Rao-Sandelius scatter shuffle): every element draws one of
// 64 buckets; a stable counting sort groups the elements by bucket (lane l owns a contiguous range of elements, so "stable" =
// ascending element index: deterministic); lane b then Fisher-Yates-shuffles bucket b in place -- a chain of ~k / 64 dependent
// LDS steps instead of k.  Buckets laid end to end are a uniform permutation: for a target order, the elements' buckets are
// determined by the bucket sizes (probability 64^-k) and every bucket must come out in the target's order (1 / size! each),
// whatever the target.
__global__ __launch_bounds__(64) void k_perms_device(int64_t n, int64_t k, int64_t count, uint32_t key0, uint32_t key1,
                                                     const int32_t *__restrict__ mov, const int32_t *__restrict__ pos_of,
                                                     int32_t *__restrict__ table, uint16_t *__restrict__ table16, int64_t stride16) {
    extern __shared__ uint32_t lds32[];                               // hist [64 lanes][64 buckets] u32 | x [kpad] u16
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    uint32_t *hist = lds32;
    uint16_t *x = reinterpret_cast<uint16_t *>(lds32 + 64 * 64);
    const uint32_t ku = static_cast<uint32_t>(k), ce = (ku + 63u) / 64u;
    const uint32_t e0 = min(ku, lane * ce), e1 = min(ku, e0 + ce);
    uint32_t *my_hist = hist + lane * 64;
    for (int b = 0; b < 64; ++b) my_hist[b] = 0;
    uint32_t cached = 0xFFFFFFFFu, w[4] = {0u, 0u, 0u, 0u};
    for (uint32_t e = e0; e < e1; ++e) my_hist[device_bucket_word(q, e, key0, key1, cached, w)] += 1u;
    __syncthreads();
    // lane b: size of bucket b, its start (exclusive scan over the buckets), and every lane's first slot inside it
    uint32_t total = 0;
    for (int l = 0; l < 64; ++l) total += hist[l * 64 + lane];
    uint32_t start = total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(start, d);
        if (lane >= d) start += up;
    }
    start -= total;
    uint32_t run = start;
    for (int l = 0; l < 64; ++l) {
        const uint32_t c = hist[l * 64 + lane];
        hist[l * 64 + lane] = run;
        run += c;
    }
    __syncthreads();
    cached = 0xFFFFFFFFu;
    for (uint32_t e = e0; e < e1; ++e) {
        const uint32_t b = device_bucket_word(q, e, key0, key1, cached, w);
        const uint32_t at = my_hist[b];
        my_hist[b] = at + 1u;
        x[at] = static_cast<uint16_t>(e);
    }
    __syncthreads();
    if (total >= 2u) {                                                // Fisher-Yates from the top inside bucket `lane`
        uint16_t *xb = x + start;
        for (uint32_t i = total - 1u; i >= 1u; --i) {
            const uint32_t j = device_draw(q, static_cast<uint32_t>(lane), i, key0, key1);
            const uint16_t xi = xb[i], xj = xb[j];
            xb[i] = xj;
            xb[j] = xi;
        }
    }
    __syncthreads();
    const int64_t stride = n + 1;
    int32_t *row = table + static_cast<int64_t>(q) * stride;
    uint16_t *row16 = table16 ? table16 + static_cast<int64_t>(q) * stride16 : nullptr;
    for (int64_t r = lane; r < (row16 ? stride16 : stride); r += 64) {
        int32_t v = static_cast<int32_t>(n);                           // entry n (and the 16-bit row's padding) = the padding row
        if (r < n) {
            const int32_t t = pos_of[r];
            v = t < 0 ? static_cast<int32_t>(r) : mov[x[t]];          // safe_extras.py:58: the row at indx_vals[t] is now old row perm[t]
        }
        if (r < stride) row[r] = v;
        if (row16) row16[r] = static_cast<uint16_t>(v);
    }
    (void)count;
}

// --------------------------------------------------------------------------------------
// chunked generation
// --------------------------------------------------------------------------------------
// permutations per host/GPU pipeline stage (every per-stage buffer holds this many)
static constexpr int64_t kChunk = 128;
// Stage boundaries of the host / GPU pipeline for `count` permutations: a ramp [0, 16), [16, 64), [64, 192), then 128-permutation
// stages, and a SHORT last stage (the final stages are cut to <= 96 and 32 permutations: what the last stage holds runs after the
// draws have ended).  The ramp: the first kernel launch waits for the first stage's draws, replay and scan, so it is short; the
// launch that covers a short stage costs more per permutation, so the next ones grow quickly.  Seeded step at configs[1], medians
// on one box (tools/exp_ab.sh): 64 | 128 ... 3.28-3.32 ms; 32 | 128 ... 3.22-3.24; 16 | 64 | 192 ... 3.18-3.21; 16 | 48 | 128 3.31-3.36;
// 8 | 32 | 96 | 224 3.38.
std::vector<int64_t> perms_stage_plan(int64_t count) {
    std::vector<int64_t> b;
    b.push_back(0);
    if (count <= 0) return b;
    // SAFE_HIP_STAGES="32,96,224": the first boundaries by hand (A/B of the pipeline's fill), 128-permutation stages after them
    static const std::vector<int64_t> head = [] {
        std::vector<int64_t> h;
        if (const char *e = getenv("SAFE_HIP_STAGES"))
            for (const char *c = e; *c;) {
                char *end = nullptr;
                const long long v = strtoll(c, &end, 10);
                if (end == c) break;
                if (v > (h.empty() ? 0 : h.back())) h.push_back(v);
                c = *end ? end + 1 : end;
            }
        if (h.empty()) h = {16, 64, 192};
        return h;
    }();
    for (int64_t q : head)
        if (q < count) b.push_back(q);
    for (int64_t q = head.back() + kChunk; q < count; q += kChunk) b.push_back(q);
    const int64_t last = b.back();
    // the tail: ... | <= 96 | 32.  (Round 6, on a box whose chain was slowed to 5 us per permutation so that the stages behind the last
    // draw are what is measured: ... | <= 112 | 32 | 16 made the step LONGER, 5.73-5.86 -> 5.91-5.94 ms -- a launch costs ~150 us
    // plus 2 us per permutation whatever its size (its heaviest slice group's tasks are its critical path), so more and
    // shorter stages behind the last draw add more than they take away; a short tail joined to its predecessor was no better.
    // With the chain paced to 2.55 us per permutation -- the driver's box of round 5: step 3.55 ms = chain + 1.0, kernels busy 2.93 --
    // tails of 64 | 64 | 40, 64 | 48 | 32 | 24 and 64-permutation stages over the last 300 permutations measured 3.60 / 3.76 / 3.67 ms
    // (fast chain: 3.12 -> 3.19 / 3.40 / 3.32), and finer tasks (8 or 4 permutations at least) 3.57 / 3.76: at that chain rate the
    // kernels' 2.5 ms of work and the chain's 2.55 ms are both critical, and only less kernel work would shorten the step.)
    if (count - last > 48 && count > kChunk) b.push_back(count - 32);
    b.push_back(count);
    // every per-stage buffer (pinned staging, ring slot, row maps, targets) holds kChunk permutations: whatever SAFE_HIP_STAGES
    // and the tail rule asked for, no stage may be longer
    std::vector<int64_t> cut;
    cut.push_back(0);
    for (size_t i = 1; i < b.size(); ++i) {
        while (b[i] - cut.back() > kChunk) cut.push_back(cut.back() + kChunk);
        cut.push_back(b[i]);
    }
    return cut;
}
static int64_t stage_begin(const safe_perms *p, int64_t ci) { return p->stages[std::min<size_t>(ci, p->stages.size() - 1)]; }
static int64_t stage_count(const safe_perms *p) { return static_cast<int64_t>(p->stages.size()) - 1; }
static int64_t chunk_of(const safe_perms *p, int64_t perm) {
    return static_cast<int64_t>(std::upper_bound(p->stages.begin(), p->stages.end(), perm) - p->stages.begin()) - 1;
}
static int64_t chunk_end(const safe_perms *p, int64_t ci) { return stage_begin(p, ci + 1); }

// --------------------------------------------------------------------------------------
// Replay of the accepted swap targets (np.random.permutation = legacy shuffle: for i = k-1 .. 1: swap(a[i], a[j_i]);
// safepy/safe_extras.py:58) on the device.  One wave per permutation; the positions 0..k-1 live in LDS as 16-bit values.
// A shuffle is one dependent chain of k - 1 swaps, but 64 consecutive steps rarely touch a common location (their own
// positions i are distinct, only the targets j can coincide: ~64^2 / k pairs per batch), so a batch runs in two phases:
//   every lane stamps its two locations in a hashed tag table with an atomic min of (batch, lane) -- the LOWEST lane that
//   touches a location owns it; a lane that owns both of its locations shares neither with a lower lane, so its swap commutes
//   with every lower lane's and runs at once; the other lanes (they share a location, or just a tag slot, with a lower lane)
//   run afterwards one by one in lane order.
// Order is kept exactly where it matters: between two steps that share a location the lower one is either unflagged (first
// phase) or earlier in the second phase.  Stamps decrease from batch to batch, so the table is never cleared.
// --------------------------------------------------------------------------------------
static const int kReplayBlock = 1024;      // targets staged in LDS at a time (16 batches)
// ROWIDS: the LDS array holds the ROW that sits at each position (n <= 65535: a row id fits 16 bits) -- the row map then needs
// no gather through `mov` at the end; otherwise it holds the original position and the rows are looked up when the map is written.
// `mov` and `pos_of` are padded to a multiple of 4 entries with -1 (16-byte loads).
template <bool ROWIDS>
__global__ __launch_bounds__(64) void k_replay_targets(const uint16_t *__restrict__ targets, int64_t width, int64_t n, int64_t k,
                                                       const int32_t *__restrict__ mov, const int32_t *__restrict__ pos_of,
                                                       int32_t *__restrict__ maps, uint32_t hash_mask) {
    extern __shared__ uint32_t lds32[];                               // tag [hash_mask + 1] u32 | tb [1024] u16 | a [kpad] u16
    uint32_t *tag = lds32;
    uint16_t *tb = reinterpret_cast<uint16_t *>(lds32 + hash_mask + 1);
    uint16_t *a = tb + kReplayBlock;
    const uint32_t lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    for (uint32_t t = lane; t <= hash_mask; t += 64) tag[t] = 0xFFFFFFFFu;
    if (ROWIDS) {
#pragma unroll 4
        for (uint32_t t = 4 * lane; t < static_cast<uint32_t>(k); t += 256) {      // (a[] is padded to a multiple of 4)
            const int4 m = *reinterpret_cast<const int4 *>(mov + t);
            a[t] = static_cast<uint16_t>(m.x), a[t + 1] = static_cast<uint16_t>(m.y);
            a[t + 2] = static_cast<uint16_t>(m.z), a[t + 3] = static_cast<uint16_t>(m.w);
        }
    } else {
        for (uint32_t t = lane; t < static_cast<uint32_t>(k); t += 64) a[t] = static_cast<uint16_t>(t);
    }
    // One wave: its LDS instructions execute in issue order, so a lane sees what another lane wrote in an earlier statement
    // without a barrier (the compiler keeps may-alias LDS accesses in program order; wave_barrier() only pins the schedule).
    // No __syncthreads() in this kernel: its fence would also wait for the target loads that are kept in flight below.
    __builtin_amdgcn_wave_barrier();
    const uint32_t steps = k > 0 ? static_cast<uint32_t>(k - 1) : 0u;
    // The targets of 1024 steps are staged in LDS; the next block's 16 targets per lane are loaded into registers before the
    // block's 16 batches run and stored behind them, so a global-memory round trip is paid once per row, not once per batch.
    // (Rows are 16-byte aligned and the buffer has a block of slack behind the last row: the last block may read past its row.)
    const uint4 *src = reinterpret_cast<const uint4 *>(targets + q * width) + 2 * lane;
    uint4 nx0 = src[0], nx1 = src[1];
    for (uint32_t b0 = 0; b0 < steps; b0 += kReplayBlock) {
        reinterpret_cast<uint4 *>(tb)[2 * lane] = nx0;
        reinterpret_cast<uint4 *>(tb)[2 * lane + 1] = nx1;
        __builtin_amdgcn_wave_barrier();
        if (b0 + kReplayBlock < steps) {
            src += kReplayBlock / 8;
            nx0 = src[0];
            nx1 = src[1];
        }
        const uint32_t b_end = min(steps, b0 + kReplayBlock);
        for (uint32_t s0 = b0; s0 < b_end; s0 += 64) {
            const uint32_t st = s0 + lane, batch = s0 >> 6;
            const bool valid = st < steps;
            const uint32_t i = valid ? static_cast<uint32_t>(k - 1) - st : 0u;
            const uint32_t j = valid ? static_cast<uint32_t>(tb[st - b0]) : 0u;
            const uint32_t stamp = ((0x03FFFFFFu - batch) << 6) | lane;
            if (valid) {
                atomicMin(&tag[i & hash_mask], stamp);
                atomicMin(&tag[j & hash_mask], stamp);
            }
            __builtin_amdgcn_wave_barrier();
            const bool mine = valid && tag[i & hash_mask] == stamp && tag[j & hash_mask] == stamp;
            uint16_t x = 0, y = 0;
            if (mine) {
                x = a[i];
                y = a[j];
            }
            __builtin_amdgcn_wave_barrier();
            if (mine) {
                a[i] = y;
                a[j] = x;
            }
            __builtin_amdgcn_wave_barrier();
            // The other lanes, in lane order, without an LDS round trip per step: each reads its two values once (current with
            // respect to the first phase), then the steps are replayed in REGISTERS -- step t broadcasts (i, j, x, y); a later lane
            // whose position or target is step t's target now holds what t moved there; an earlier lane whose target is written
            // again by t (as its target or as its own position) gives up that write.  One write of the survivors at the end.
            const bool late = valid && !mine;
            uint64_t later = __ballot(late);
            if (later) {
                uint32_t xr = 0, yr = 0;
                if (late) {
                    xr = a[i];
                    yr = a[j];
                }
                bool write_j = late;
                while (later) {                                        // (uniform)
                    const int t = __builtin_ctzll(later);
                    later &= later - 1ull;
                    const uint32_t it = __builtin_amdgcn_readlane(i, t), jt = __builtin_amdgcn_readlane(j, t);
                    const uint32_t xt = __builtin_amdgcn_readlane(xr, t);
                    if (late && lane > static_cast<uint32_t>(t)) {   // (a later step's position or target can only meet jt: its i and j are < it)
                        if (i == jt) xr = xt;
                        if (j == jt) yr = xt;
                    }
                    if (lane < static_cast<uint32_t>(t) && (j == jt || j == it)) write_j = false;
                }
                __builtin_amdgcn_wave_barrier();
                if (late) {
                    a[i] = static_cast<uint16_t>(yr);
                    if (write_j) a[j] = static_cast<uint16_t>(xr);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // the row map: rows that hold no value stay (entry n: the padding row); safe_extras.py:58: the row at indx_vals[t] is now
    // old row indx_vals[position that ended at t]
    int32_t *row = maps + q * (n + 1);
#pragma unroll 4
    for (int64_t r4 = 4 * lane; r4 < n; r4 += 256) {
        const int4 p4 = *reinterpret_cast<const int4 *>(pos_of + r4);
        const int32_t ps[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (r4 + e < n) {
                int32_t v = static_cast<int32_t>(r4 + e);
                if (ps[e] >= 0) v = ROWIDS ? static_cast<int32_t>(a[ps[e]]) : mov[a[ps[e]]];
                row[r4 + e] = v;
            }
    }
    if (lane == 0) row[n] = static_cast<int32_t>(n);
}

// k > 65535 movable rows (16-bit positions and LDS no longer hold a shuffle): the same replay, one lane per permutation on a
// global-memory array.  Slow (a dependent global load per step) and never on a measured path; kept so that the library
// has no size at which the stream leaves the device.
__global__ __launch_bounds__(64) void k_replay_targets_big(const uint32_t *__restrict__ targets, int64_t width, int64_t n, int64_t k,
                                                           const int32_t *__restrict__ mov, const int32_t *__restrict__ pos_of,
                                                           int32_t *__restrict__ maps, int32_t *__restrict__ scratch) {
    const int64_t q = blockIdx.x;
    int32_t *a = scratch + q * k;
    for (int64_t t = threadIdx.x; t < k; t += 64) a[t] = static_cast<int32_t>(t);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t *tg = targets + q * width;
        for (int64_t i = k - 1, st = 0; i > 0; --i, ++st) {
            const int64_t j = tg[st];
            const int32_t x = a[i], y = a[j];
            a[i] = y;
            a[j] = x;
        }
    }
    __syncthreads();
    int32_t *row = maps + q * (n + 1);
    for (int64_t r = threadIdx.x; r <= n; r += 64) {
        int32_t v = static_cast<int32_t>(r);
        if (r < n) {
            const int32_t t = pos_of[r];
            if (t >= 0) v = mov[a[t]];
        }
        row[r] = v;
    }
}

static size_t chunk_target_bytes(const safe_perms *p, int64_t cnt) { return static_cast<size_t>(cnt) * p->target_width * p->target_bytes; }

// enqueue the GPU part of a chunk whose swap targets are ready in h_stage[ci % kStage] (or come from the node's ring)
static int enqueue_chunk(safe_perms *p, int64_t ci) {
    safe_ctx *ctx = p->ctx;
    hipStream_t gs = ctx->aux_stream;
    const int64_t n = p->n, k = p->k, stride = n + 1;
    const int64_t q0 = stage_begin(p, ci), q1 = chunk_end(p, ci), cnt = q1 - q0;
    const int b = static_cast<int>(ci % safe_perms::kStage);
    const size_t bytes = chunk_target_bytes(p, cnt);
    SAFE_REQUIRE(cnt <= kChunk, "pipeline stage of %lld permutations exceeds the stage buffers (%lld)", (long long)cnt, (long long)kChunk);
    if (p->ring_consumer) {
        // the node's producer drew this chunk: block until it is published, copy it into this rank's pinned staging buffer
        // (free once the upload of kStage chunks ago has completed)
        if (ci >= safe_perms::kStage) SAFE_HIP_CHECK(hipEventSynchronize(p->staged[b]));
        SAFE_TRY(ring_fetch(p->ring, ci, p->h_stage[b], bytes, &p->ring_wait_ms));
        safe_trace("  gen: chunk fetched from the node's stream");
    } else if (p->ring) {
        SAFE_TRY(ring_publish(p->ring, ci, p->h_stage[b], bytes));
    }
    // Two streams: upload + replay of chunk c on the replay stream, scan + emit on the table stream.  The replays are independent
    // of each other (only the scan composes with the previous chunk's last row), so chunk c's upload and replay run beside chunk
    // c - 1's scan: in the ramp of the pipeline (stages of 16 / 48 / 128 permutations, each waiting for its draws) the tables of the
    // second and third stage are ready 65-85 us earlier.  Buffers alternate by the chunk's parity.
    constexpr bool one_stream = false;                 // (replay on the table stream: measured 3.18-3.20 against 3.13-3.14 ms, round 5)
    const int par = one_stream ? 0 : static_cast<int>(ci & 1);
    hipStream_t rs = one_stream ? gs : ctx->more_streams[0];
    void *d_tg = par ? p->d_targets_odd : p->d_targets;
    if (!one_stream) {
        if (ci == 0) SAFE_HIP_CHECK(hipStreamWaitEvent(rs, p->movpos_ready, 0));
        if (ci >= 2) SAFE_HIP_CHECK(hipStreamWaitEvent(rs, p->chunk_done[ci - 2], 0));       // this parity's row maps are free again
    }
    const void *h_src = p->h_stage[b];
    if (p->twin) {
        std::lock_guard<std::mutex> lk(p->draw_mu);
        if (p->chunk_src[ci]) h_src = p->h_stage2[b];                // the twin finished this chunk first
    }
    SAFE_HIP_CHECK(hipMemcpyAsync(d_tg, h_src, bytes, hipMemcpyHostToDevice, rs));
    SAFE_HIP_CHECK(hipEventRecord(p->staged[b], rs));
    if (p->twin) SAFE_HIP_CHECK(hipEventRecord(p->staged2[b], rs));    // (both threads' buffers of this slot are free again after this upload)
    {
        std::lock_guard<std::mutex> lk(p->draw_mu);                  // the draw thread may fill this buffer again once the upload is done
        p->enqueued_chunks = std::max<int64_t>(p->enqueued_chunks, ci + 1);
    }
    p->draw_cv.notify_all();
    int32_t *xa = par ? p->d_maps_odd[0] : p->d_maps[0], *xb = par ? p->d_maps_odd[1] : p->d_maps[1];
    const int32_t *d_mov = p->d_movpos, *d_pos = p->d_movpos + ((n + 3) & ~int64_t(3));
    if (k <= 65535) {
        uint32_t hash_mask = 63u;                              // tag slots: the next power of two >= k (no aliasing), at most 8192 (4096 when the positions need most of the LDS)
        while (hash_mask + 1 < static_cast<uint32_t>(k) && hash_mask < (k <= 32768 ? 8191u : 4095u)) hash_mask = 2 * hash_mask + 1;
        const int64_t kpad = (std::max<int64_t>(k, 4) + 3) & ~int64_t(3);
        const size_t lds = (static_cast<size_t>(hash_mask) + 1) * sizeof(uint32_t) + static_cast<size_t>(kReplayBlock + kpad) * sizeof(uint16_t);
        auto kernel = n <= 65535 ? k_replay_targets<true> : k_replay_targets<false>;
        static std::atomic<size_t> lds_set[2][16] = {};              // by kernel form and device (the attribute call costs ~3 us of the launcher's time per chunk)
        std::atomic<size_t> &set = lds_set[n <= 65535 ? 1 : 0][ctx->device & 15];
        if (set.load(std::memory_order_relaxed) < lds) {
            SAFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            set.store(lds, std::memory_order_relaxed);
        }
        hipLaunchKernelGGL(kernel, dim3(cnt), dim3(64), lds, rs, static_cast<const uint16_t *>(d_tg), p->target_width, n, k,
                           d_mov, d_pos, xa, hash_mask);
    } else {
        if (!p->d_big) SAFE_TRY(dev_alloc(&p->d_big, static_cast<size_t>(kChunk) * p->n));
        hipLaunchKernelGGL(k_replay_targets_big, dim3(cnt), dim3(64), 0, rs, static_cast<const uint32_t *>(d_tg),
                           p->target_width, n, k, d_mov, d_pos, xa, p->d_big);
    }
    if (!one_stream) {
        SAFE_HIP_CHECK(hipEventRecord(p->replayed[par], rs));
        SAFE_HIP_CHECK(hipStreamWaitEvent(gs, p->replayed[par], 0));
    }
    const dim3 grid(ceil_div(std::max<int64_t>(stride, p->stride16), 256), cnt), block(256);
    // the base of this chunk's compositions = the last row of the previous chunk, read where it lies in the table (the identity
    // for the first chunk: d_cur)
    const int32_t *cur_base = q0 > 0 ? p->table + (q0 - 1) * stride : p->d_cur;
    int32_t *t_rows = p->table + q0 * stride;
    uint16_t *t16_rows = p->table16 ? p->table16 + q0 * p->stride16 : nullptr;
    if (cnt == 1) {
        hipLaunchKernelGGL(k_emit_rows, grid, block, 0, gs, xa, cur_base, cnt, n, t_rows, t16_rows, p->stride16);
    } else {
        for (int64_t d = 1; d < cnt; d <<= 1) {
            if (2 * d >= cnt) {                                          // the last round carries the emit
                hipLaunchKernelGGL(k_scan_emit, grid, block, 0, gs, xa, d, cur_base, cnt, n, t_rows, t16_rows, p->stride16);
            } else {
                hipLaunchKernelGGL(k_scan_round, grid, block, 0, gs, xa, xb, cnt, stride, d);
                std::swap(xa, xb);
            }
        }
    }
    SAFE_HIP_CHECK(hipGetLastError());
    if (static_cast<int64_t>(p->chunk_done.size()) <= ci) p->chunk_done.resize(ci + 1, nullptr);
    if (!p->chunk_done[ci]) SAFE_HIP_CHECK(hipEventCreateWithFlags(&p->chunk_done[ci], safe_event_flags(hipEventDisableTiming)));
    SAFE_HIP_CHECK(hipEventRecord(p->chunk_done[ci], gs));
    p->enqueued = q1;
    if (q1 >= p->count) p->enqueued_all_ms = 1e3 * (wall_s() - p->t_created_s);
    safe_trace("  gen: chunk enqueued");
    return SAFE_OK;
}

static void drawer_main(safe_perms *p, int w = 0) {
    pthread_setname_np(pthread_self(), w ? "safe-draw2" : "safe-draw");
    (void)hipSetDevice(p->ctx->device);
    const int64_t k = p->k;
    const int64_t n_chunks = stage_count(p);
    const int64_t steps = std::max<int64_t>(k - 1, 0);
    // w = 1: the twin (safe_perms::twin) -- its own generator state (same seed), private buffer, staging buffers and events
    uint32_t *h = w ? p->h_local2.data() : p->h_local.data();
    DrawStream *gen = w ? p->stream2 : p->stream;
    void *const *stage = w ? p->h_stage2 : p->h_stage;
    hipEvent_t *staged = w ? p->staged2 : p->staged;
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int b = static_cast<int>(c % safe_perms::kStage);
        {
            std::unique_lock<std::mutex> lk(p->draw_mu);
            p->draw_cv.wait(lk, [&] { return p->draw_stop || c < p->enqueued_chunks + safe_perms::kStage; });
            if (p->draw_stop) return;
        }
        // the staging buffer's previous chunk (c - kStage) must have left for the device (its upload was queued: see the wait above)
        const int64_t q0 = stage_begin(p, c), cnt = chunk_end(p, c) - q0;
        if ((c >= safe_perms::kStage && hipEventSynchronize(staged[b]) != hipSuccess) || cnt > kChunk) {
            // the launcher waits on draw_cv for this chunk: tell it, or it waits for ever
            {
                std::lock_guard<std::mutex> lk(p->draw_mu);
                p->draw_failed = true;
            }
            p->draw_cv.notify_all();
            return;
        }
        safe_trace("    drawer: buffer free, drawing");
        const double t_draw = wall_s();
        char *dst = static_cast<char *>(stage[b]);
        for (int64_t q = 0; q < cnt; ++q) {
            draw_stream_targets(gen, k, h);                         // into a buffer that stays in L1, then packed onto the wire
            if (p->target_bytes == 2) draws_pack_u16(reinterpret_cast<uint16_t *>(dst) + q * p->target_width, h, static_cast<size_t>(steps));
            else memcpy(reinterpret_cast<uint32_t *>(dst) + q * p->target_width, h, static_cast<size_t>(steps) * sizeof(uint32_t));
            if ((q & 15) == 15) {
                std::lock_guard<std::mutex> lk(p->draw_mu);
                if (p->draw_stop) return;
            }
        }
        safe_trace("    drawer: chunk drawn");
        {
            std::lock_guard<std::mutex> lk(p->draw_mu);
            if (p->drawn_chunks <= c) {                               // first to finish this chunk (always, without a twin)
                if (p->twin) p->chunk_src[c] = static_cast<uint8_t>(w);
                p->twin_wins += w;
                p->drawn_chunks = c + 1;
                p->drawn_chunks_pub.store(c + 1, std::memory_order_release);
                p->draw_busy_ms += 1e3 * (wall_s() - t_draw);
                if (c + 1 == n_chunks) p->drawn_all_ms = 1e3 * (wall_s() - p->t_created_s);
            }
        }
        p->draw_cv.notify_all();
    }
}

// CPUs the draw threads of this process may run on (safe_set_draw_cpus; empty = wherever the process may)
static std::mutex g_draw_cpu_mu;
static std::vector<int> g_draw_cpus;

int safe_set_draw_cpus(const int *cpus, int count) {
    SAFE_REQUIRE(count == 0 || cpus, "safe_set_draw_cpus: NULL argument");
    std::lock_guard<std::mutex> lk(g_draw_cpu_mu);
    g_draw_cpus.assign(cpus, cpus + std::max(count, 0));
    return SAFE_OK;
}

// which = 0: the chain thread, 1: its twin.  With a twin the list is split in two halves (the launcher lists whole physical
// cores one after the other: safe_set_draw_cpus): the two chains must not share a core -- on sibling hardware threads BOTH run a
// third slower and the twin is worse than no twin.
static void draw_thread_apply_cpus(int which = 0, bool split = false) {
    std::lock_guard<std::mutex> lk(g_draw_cpu_mu);
    if (g_draw_cpus.empty()) return;
    size_t lo = 0, hi = g_draw_cpus.size();
    if (split && hi >= 2) {
        const size_t mid = hi / 2;
        if (which == 0) hi = mid;
        else lo = mid;
    }
    cpu_set_t set;
    CPU_ZERO(&set);
    for (size_t i = lo; i < hi; ++i)
        if (g_draw_cpus[i] >= 0 && g_draw_cpus[i] < CPU_SETSIZE) CPU_SET(g_draw_cpus[i], &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);      // (best effort: a refused mask leaves the thread where it was)
}

static void draw_worker_main(safe_ctx *ctx, int which) {
    DrawWorker *w = which ? ctx->draw_worker2 : ctx->draw_worker;
    pthread_setname_np(pthread_self(), which ? "safe-draw2" : "safe-draw");
    static const bool twin_on = getenv("SAFE_HIP_DRAW_TWIN") && !strcmp(getenv("SAFE_HIP_DRAW_TWIN"), "1");
    draw_thread_apply_cpus(which, twin_on && !safe_blocking_sync_selected());
    (void)hipSetDevice(ctx->device);
    constexpr double spin_s = 1.5e-3;                                      // how long the idle worker polls before it sleeps (20 ms measured the same)
    for (;;) {
        safe_perms *p = nullptr;
        if (!safe_blocking_sync_selected()) {
            const double t0 = wall_s();
            while (!(p = w->job.exchange(nullptr, std::memory_order_acq_rel)) && wall_s() - t0 < spin_s) cpu_relax();
        }
        if (!p) {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->quit || (p = w->job.exchange(nullptr, std::memory_order_acq_rel)) != nullptr; });
            if (!p) return;                                                // quit
        }
        drawer_main(p, which);
        {
            std::lock_guard<std::mutex> lk(w->mu);
            (which ? p->worker_done2 : p->worker_done) = true;
            w->busy = false;
        }
        w->cv.notify_all();
    }
}

void draw_worker_shutdown(safe_ctx *ctx) {
    for (DrawWorker **slot : {&ctx->draw_worker, &ctx->draw_worker2}) {
        DrawWorker *w = *slot;
        if (!w) continue;
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->quit = true;
        }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        delete w;
        *slot = nullptr;
    }
}

static bool worker_post(safe_ctx *ctx, DrawWorker **slot, int which, safe_perms *p) {
    if (!*slot) {
        *slot = new DrawWorker;
        (*slot)->th = std::thread(draw_worker_main, ctx, which);
    }
    DrawWorker *w = *slot;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        if (w->busy) return false;                                    // (one stream at a time per worker)
        w->busy = true;
        // Published under the mutex: the worker evaluates its wait predicate under the same mutex, so a store + notify can
        // never fall between its last look and its block (a lost wake-up would leave the seeded call waiting forever).
        w->job.store(p, std::memory_order_release);
    }
    w->cv.notify_all();                                               // (a polling worker sees the store; a sleeping one this)
    return true;
}

static void drawer_start(safe_perms *p) {
    p->drawn_chunks = p->enqueued_chunks = 0;
    p->drawn_chunks_pub.store(0, std::memory_order_release);
    p->draw_stop = false;
    p->draw_failed = false;
    p->on_worker = p->on_worker2 = false;
    p->worker_done = p->worker_done2 = false;
    p->twin_wins = 0;
    if (p->count <= 0 || p->ring_consumer) return;
    safe_ctx *ctx = p->ctx;
    // the context's persistent worker; a thread of the handle's own only when that worker is busy with another live handle
    // (a thread per call measured 2.5 % slower and with 5-10 ms outliers, round 5)
    if (worker_post(ctx, &ctx->draw_worker, 0, p)) {
        p->on_worker = true;
        if (p->twin) {
            if (worker_post(ctx, &ctx->draw_worker2, 1, p)) p->on_worker2 = true;
            // (a busy second worker -- another live handle -- simply leaves this call without a twin: chunk_src stays 0)
        }
        return;
    }
    p->drawer = std::thread([p] {
        draw_thread_apply_cpus();
        drawer_main(p, 0);
    });
}

static void drawer_stop(safe_perms *p) {
    if (p->on_worker) {
        {
            std::lock_guard<std::mutex> lk(p->draw_mu);
            p->draw_stop = true;
        }
        p->draw_cv.notify_all();
        {
            DrawWorker *w = p->ctx->draw_worker;
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return p->worker_done; });
        }
        if (p->on_worker2) {
            DrawWorker *w = p->ctx->draw_worker2;
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return p->worker_done2; });
        }
        p->on_worker = p->on_worker2 = false;
        return;
    }
    if (!p->drawer.joinable()) return;
    {
        std::lock_guard<std::mutex> lk(p->draw_mu);
        p->draw_stop = true;
    }
    p->draw_cv.notify_all();
    p->drawer.join();
}

// Pipeline per chunk c:  draw thread: targets(c+1 ..)  ||  GPU: upload + replay + scan(c)  ||  enrichment kernels(c-1).
// On return every row < upto has been enqueued on ctx->aux_stream.  The calling thread only moves chunks along
// (drawn -> GPU); the draws themselves run on p->drawer.
int perms_generate_until(safe_perms *p, int64_t upto) {
    upto = std::min<int64_t>(upto, p->count);
    while (p->enqueued < upto) {
        const int64_t ci = chunk_of(p, p->enqueued);
        if (!p->ring_consumer) {                             // (a consumer's chunks come from the node's producer: no draws here)
            // The draw thread needs ~0.25 ms per chunk and this thread has nothing else to do meanwhile.  Sleeping on the
            // condition variable costs a futex wake-up per chunk, and on a shared host that wake-up was seen to take 4-6 ms
            // once in a few hundred steps (tools/probe/trace_outlier.py) -- so spin on the counter first (bounded; with
            // blocking waits selected -- several ranks on few cores -- sleep at once).
            if (!safe_blocking_sync_selected()) {
                const double t_spin = wall_s();
                while (p->drawn_chunks_pub.load(std::memory_order_acquire) <= ci && wall_s() - t_spin < 2e-3) cpu_relax();
            }
            std::unique_lock<std::mutex> lk(p->draw_mu);
            p->draw_cv.wait(lk, [&] { return p->drawn_chunks > ci || p->draw_failed; });
            if (p->drawn_chunks <= ci) {
                lk.unlock();
                safe_set_error("the draw thread of the seeded stream stopped (staging buffer wait failed)");
                return SAFE_E_HIP;
            }
            p->generated = chunk_end(p, p->drawn_chunks - 1);
            lk.unlock();
            safe_trace("  gen: chunk drawn");
        }
        SAFE_TRY(enqueue_chunk(p, ci));
    }
    return SAFE_OK;
}

// stream s will not run past this point before table rows [0, upto) are complete
int perms_wait(safe_perms *p, int64_t upto, hipStream_t s) {
    SAFE_TRY(perms_generate_until(p, upto));
    upto = std::min<int64_t>(upto, p->count);
    if (upto <= 0) return SAFE_OK;
    const int64_t ci = chunk_of(p, upto - 1);
    SAFE_HIP_CHECK(hipStreamWaitEvent(s, p->chunk_done[ci], 0));
    return SAFE_OK;
}

int perms_build_inverse(safe_perms *perms) {
    if (perms->inverse_t) return SAFE_OK;
    SAFE_REQUIRE(perms->n < 65535, "perms_build_inverse: n too large for 16-bit positions");
    safe_ctx *ctx = perms->ctx;
    SAFE_TRY(perms_wait(perms, perms->count, ctx->stream));
    const int64_t stride = perms->n + 1, total = perms->count * stride;
    perms->inv_stride = ((perms->count + 15) / 16) * 16 + 32;      // chunked prefetch may read two chunks ahead
    const size_t elems = static_cast<size_t>(stride) * perms->inv_stride;
    SAFE_TRY(dev_alloc(&perms->inverse_t, elems));
    SAFE_HIP_CHECK(hipMemsetAsync(perms->inverse_t, 0, elems * sizeof(uint16_t), ctx->stream));
    if (total)
        hipLaunchKernelGGL(k_invert_perms, dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, perms->table, stride,
                           total, perms->inv_stride, perms->inverse_t);
    SAFE_HIP_CHECK(hipGetLastError());
    return SAFE_OK;
}

static void perms_free(safe_perms *p) {
    if (!p) return;
    drawer_stop(p);
    if (p->ring) ring_end_call(p->ring);
    p->ring = nullptr;
    for (int b = 0; b < safe_perms::kStage; ++b) {
        if (p->h_stage[b]) (void)hipHostFree(p->h_stage[b]);
        if (p->staged[b]) (void)hipEventDestroy(p->staged[b]);
        if (p->h_stage2[b]) (void)hipHostFree(p->h_stage2[b]);
        if (p->staged2[b]) (void)hipEventDestroy(p->staged2[b]);
    }
    for (int b = 0; b < 2; ++b) {
        (void)hipFree(p->d_maps[b]);
        (void)hipFree(p->d_maps_odd[b]);
        if (p->replayed[b]) (void)hipEventDestroy(p->replayed[b]);
    }
    if (p->movpos_ready) (void)hipEventDestroy(p->movpos_ready);
    (void)hipFree(p->d_targets_odd);
    if (p->h_movpos) (void)hipHostFree(p->h_movpos);
    (void)hipFree(p->d_targets);
    (void)hipFree(p->d_big);
    for (hipEvent_t e : p->chunk_done)
        if (e) (void)hipEventDestroy(e);
    (void)hipFree(p->d_cur);
    (void)hipFree(p->d_movpos);
    (void)hipFree(p->table);
    (void)hipFree(p->table16);
    (void)hipFree(p->inverse_t);
    if (p->stream) draw_stream_free(p->stream);
    if (p->stream2) draw_stream_free(p->stream2);
    delete p;
}

void perms_cache_drop(safe_ctx *ctx) {
    if (ctx->perm_cache) {
        perms_free(ctx->perm_cache);
        ctx->perm_cache = nullptr;
    }
}

extern "C" {

int safe_rng_permutations_host(uint32_t seed, const int64_t *values, int64_t n_items, int64_t count, int64_t *out) {
    SAFE_REQUIRE(n_items >= 0 && count >= 0, "safe_rng_permutations_host: negative size");
    SAFE_REQUIRE(n_items == 0 || count == 0 || (values && out), "safe_rng_permutations_host: NULL argument");
    SAFE_REQUIRE(n_items < (1ll << 32), "safe_rng_permutations_host: n_items too large");
    DrawStream *ds = draw_stream_new(seed);
    std::vector<uint32_t> j(std::max<int64_t>(n_items, 1));
    for (int64_t c = 0; c < count; ++c) {
        int64_t *dst = out + c * n_items;
        memcpy(dst, values, n_items * sizeof(int64_t));
        draw_stream_targets(ds, n_items, j.data());
        for (int64_t i = n_items - 1, st = 0; i > 0; --i, ++st) std::swap(dst[i], dst[j[st]]);
    }
    draw_stream_free(ds);
    return SAFE_OK;
}

// the movable rows and their positions in that list, on the device (both the replay of a seeded stream and the device stream
// write the row maps through them)
static int upload_movpos(safe_perms *p) {
    const int64_t n = p->n, k = p->k, n_pad = (n + 3) & ~int64_t(3);      // two halves of n_pad entries, padded with -1
    int32_t *h = p->h_movpos;                                  // (the previous handle's upload has completed: destroy synchronises)
    for (int64_t i = 0; i < 2 * n_pad; ++i) h[i] = -1;
    for (int64_t t = 0; t < k; ++t) {
        h[t] = p->h_movable[t];
        h[n_pad + p->h_movable[t]] = static_cast<int32_t>(t);
    }
    SAFE_HIP_CHECK(hipMemcpyAsync(p->d_movpos, h, static_cast<size_t>(2 * n_pad) * sizeof(int32_t), hipMemcpyHostToDevice, p->ctx->aux_stream));
    if (p->movpos_ready) SAFE_HIP_CHECK(hipEventRecord(p->movpos_ready, p->ctx->aux_stream));
    return SAFE_OK;
}

// whole table on the device (k_perms_device); every pipeline stage is complete once the kernel has run
static int perms_generate_on_device(safe_perms *p, uint64_t key) {
    safe_ctx *ctx = p->ctx;
    const int64_t n = p->n, k = p->k, count = p->count;
    SAFE_REQUIRE(k <= 65535, "device permutation stream: %lld movable rows (16-bit positions hold 65535)", (long long)k);
    int32_t *d_mov = p->d_movpos, *d_pos = p->d_movpos + ((n + 3) & ~int64_t(3));
    hipStream_t gs = ctx->aux_stream;
    const int64_t kpad = (k + 1) & ~int64_t(1);
    const size_t lds = 64 * 64 * sizeof(uint32_t) + static_cast<size_t>(std::max<int64_t>(kpad, 2)) * sizeof(uint16_t);
    int lds_limit = 0;
    SAFE_HIP_CHECK(hipDeviceGetAttribute(&lds_limit, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device));
    SAFE_REQUIRE(lds <= static_cast<size_t>(lds_limit), "device permutation stream: %lld movable rows need %zu bytes of LDS, a workgroup of this "
                 "device gets %d (gfx950: 160 KiB); use the NumPy-compatible stream (SAFE_HIP_DEVICE_STREAM=0)", (long long)k, lds, lds_limit);
    SAFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_perms_device), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
    hipLaunchKernelGGL(k_perms_device, dim3(count), dim3(64), lds, gs, n, k, count, static_cast<uint32_t>(key),
                       static_cast<uint32_t>(key >> 32), d_mov, d_pos, p->table, p->table16, p->stride16);
    SAFE_HIP_CHECK(hipGetLastError());
    const int64_t n_chunks = stage_count(p);
    if (static_cast<int64_t>(p->chunk_done.size()) < n_chunks) p->chunk_done.resize(n_chunks, nullptr);
    for (int64_t c = 0; c < n_chunks; ++c) {
        if (!p->chunk_done[c]) SAFE_HIP_CHECK(hipEventCreateWithFlags(&p->chunk_done[c], safe_event_flags(hipEventDisableTiming)));
        SAFE_HIP_CHECK(hipEventRecord(p->chunk_done[c], gs));
    }
    SAFE_HIP_CHECK(hipMemcpyAsync(p->d_cur, p->table + (count - 1) * (n + 1), (n + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, gs));
    p->generated = p->enqueued = count;
    p->enqueued_all_ms = 1e3 * (wall_s() - p->t_created_s);
    return SAFE_OK;
}

// FNV-1a over the movable flags: consumers of a shared stream check that they mark the same rows as the producer
static uint64_t movable_fingerprint(const uint8_t *movable_host, int64_t n) {
    uint64_t h = 1469598103934665603ull;
    for (int64_t i = 0; i < n; ++i) h = (h ^ (movable_host[i] ? 1u : 0u)) * 1099511628211ull;
    return h;
}

static int perms_create_impl(safe_ctx *ctx, int64_t n, const uint8_t *movable_host, int64_t num_permutations, int has_seed,
                             uint32_t seed, bool shared, const char *who, safe_perms **out, bool device_gen = false,
                             uint64_t device_key = 0) {
    SAFE_REQUIRE(ctx && movable_host && out, "%s: NULL argument", who);
    SAFE_REQUIRE(n >= 1 && n < (1ll << 31) - 1, "%s: n out of range", who);
    SAFE_REQUIRE(num_permutations >= 0, "%s: negative permutation count", who);
    *out = nullptr;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    safe_trace("perms_create: enter");
    // a shared call goes through the node's ring when a chunk fits it at least twice; every rank of the node decides this
    // from (n, ring capacity) alone, hence identically.  Otherwise (and for empty streams) this rank draws for itself.
    // (a slot holds a chunk's swap targets; sized for k = n so that the decision does not depend on the rows' values)
    const int64_t slot_bytes = kChunk * ((n + 7) & ~int64_t(7)) * (n <= 65535 ? 2 : 4);
    PermRing *ring = !device_gen && shared && ctx->ring && num_permutations > 0 && ring_slots_for(ctx->ring, slot_bytes) >= 2 ? ctx->ring : nullptr;
    safe_perms *p = nullptr;
    bool reused = false;
    if (ctx->perm_cache && ctx->perm_cache->n == n && ctx->perm_cache->count == num_permutations) {
        p = ctx->perm_cache;               // same shape as the last destroyed handle: keep its buffers
        ctx->perm_cache = nullptr;
        reused = true;
        p->h_movable.clear();
    } else {
        perms_cache_drop(ctx);
        p = new safe_perms();
    }
    p->ctx = ctx;
    p->n = n;
    p->count = num_permutations;
    for (int64_t i = 0; i < n; ++i)
        if (movable_host[i]) p->h_movable.push_back(static_cast<int32_t>(i));
    p->k = static_cast<int64_t>(p->h_movable.size());
    p->ring = ring;
    p->ring_consumer = ring != nullptr && !ring_is_producer(ring);
    p->device_gen = device_gen;
    const uint32_t stream_seed = has_seed ? seed : entropy_seed();
    p->stream = (p->ring_consumer || device_gen) ? nullptr : draw_stream_new(stream_seed);
    // the twin chain (see safe_perms::twin), OPT-IN (SAFE_HIP_DRAW_TWIN=1; single-process seeded calls with polling waits).  Measured on
    // the bench host, eight driver-style runs each (5 + 20 steps, tools/probe/jitter_ab.sh): without twin mean 3.09-3.22 ms (3.157 on
    // average), worst step 3.19-4.07; with the twin on a core of its own 3.12-3.39 (3.184), worst 3.23-4.76 -- the slow steps are
    // not one thread stumbling (both chains slow down together: something host-wide), and two threads on sibling hardware
    // threads of one core are much worse (median 3.66).  Not the default.
    const char *twin_env = getenv("SAFE_HIP_DRAW_TWIN");                  // (read per handle: tests switch it on and off)
    p->twin = p->stream != nullptr && ring == nullptr && twin_env && !strcmp(twin_env, "1") && !safe_blocking_sync_selected();
    if (p->stream2) draw_stream_free(p->stream2);
    p->stream2 = p->twin ? draw_stream_new(stream_seed) : nullptr;
    p->generated = p->enqueued = 0;
    p->stages = perms_stage_plan(num_permutations);
    if (device_gen) {
        // no host pipeline to follow: the whole table is there before the first launch, so the launches are even spans of
        // 200 permutations -- a task of the bit-sliced kernel counts up to 255 permutations before it must flush its counters,
        // and fewer, longer tasks flush less (tools/probe/span_sweep.py: 10 000 permutations 24.8 ms at 128 per launch, 24.1 at
        // 200, 24.3 at 250, 25.3 at 334 where a launch's tasks are split again)
        const int64_t even = 200;
        p->stages.clear();
        for (int64_t q = 0; q < num_permutations; q += even) p->stages.push_back(q);
        p->stages.push_back(num_permutations);
        if (p->stages.size() >= 3 && num_permutations - p->stages[p->stages.size() - 2] < even / 4)
            p->stages.erase(p->stages.end() - 2);                        // a short tail joins its predecessor
    }
    p->t_created_s = wall_s();
    p->draw_busy_ms = p->drawn_all_ms = p->enqueued_all_ms = p->ring_wait_ms = 0.0;
    const int64_t k = p->k, stride = n + 1, rows = std::max<int64_t>(num_permutations, 1);
    int rc = SAFE_OK;
    do {
        if (!reused) {
            if ((rc = dev_alloc(&p->table, static_cast<size_t>(rows) * stride)) != SAFE_OK) break;
            if (n < 65535) {
                p->stride16 = (stride + 7) / 8 * 8;
                if ((rc = dev_alloc(&p->table16, static_cast<size_t>(rows) * p->stride16)) != SAFE_OK) break;
            }
            if ((rc = dev_alloc(&p->d_cur, stride)) != SAFE_OK) break;
            if ((rc = dev_alloc(&p->d_maps[0], kChunk * stride)) != SAFE_OK) break;
            if ((rc = dev_alloc(&p->d_maps[1], kChunk * stride)) != SAFE_OK) break;
            if ((rc = dev_alloc(&p->d_movpos, static_cast<size_t>(2 * n + 8))) != SAFE_OK) break;
            // staging of a chunk's swap targets, sized for k = n (a reused handle may see another set of movable rows)
            p->stage_bytes = static_cast<size_t>(slot_bytes);
            if ((rc = dev_alloc(reinterpret_cast<char **>(&p->d_targets), p->stage_bytes + 2 * kReplayBlock * sizeof(uint16_t))) != SAFE_OK) break;   // (+ a block of slack: see k_replay_targets)
        }
        if (!device_gen) {                                   // the odd chunks' buffers (a handle first used with the device stream has none yet)
            if (!p->d_maps_odd[0] && (rc = dev_alloc(&p->d_maps_odd[0], kChunk * stride)) != SAFE_OK) break;
            if (!p->d_maps_odd[1] && (rc = dev_alloc(&p->d_maps_odd[1], kChunk * stride)) != SAFE_OK) break;
            if (!p->d_targets_odd &&
                (rc = dev_alloc(reinterpret_cast<char **>(&p->d_targets_odd), p->stage_bytes + 2 * kReplayBlock * sizeof(uint16_t))) != SAFE_OK)
                break;
        }
        p->target_bytes = k <= 65535 ? 2 : 4;
        p->target_width = (std::max<int64_t>(k - 1, 1) + 7) & ~int64_t(7);
        hipError_t e = hipSuccess;
        if (!p->h_movpos) g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
        if (!p->h_movpos) e = hipHostMalloc(reinterpret_cast<void **>(&p->h_movpos), static_cast<size_t>(2 * n + 8) * sizeof(int32_t), hipHostMallocDefault);
        for (int b = 0; b < 2 && e == hipSuccess; ++b)
            if (!p->replayed[b]) e = hipEventCreateWithFlags(&p->replayed[b], safe_event_flags(hipEventDisableTiming));
        if (e == hipSuccess && !p->movpos_ready) e = hipEventCreateWithFlags(&p->movpos_ready, safe_event_flags(hipEventDisableTiming));
        for (int b = 0; b < safe_perms::kStage && e == hipSuccess; ++b) {
            if (!p->h_stage[b] && !device_gen) g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
            if (!p->h_stage[b] && !device_gen) e = hipHostMalloc(&p->h_stage[b], p->stage_bytes, hipHostMallocDefault);
            if (e == hipSuccess && !p->staged[b]) e = hipEventCreateWithFlags(&p->staged[b], safe_event_flags(hipEventDisableTiming));
        }
        if (!p->ring_consumer && !device_gen) p->h_local.resize(std::max<int64_t>(k, 1) + 64);
        if (p->twin) {
            p->h_local2.resize(std::max<int64_t>(k, 1) + 64);
            p->chunk_src.assign(p->stages.size() + 1, 0);
            for (int b = 0; b < safe_perms::kStage && e == hipSuccess; ++b) {
                if (!p->h_stage2[b]) {
                    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
                    e = hipHostMalloc(&p->h_stage2[b], p->stage_bytes, hipHostMallocDefault);
                }
                if (e == hipSuccess && !p->staged2[b]) e = hipEventCreateWithFlags(&p->staged2[b], safe_event_flags(hipEventDisableTiming));
            }
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_iota, dim3(ceil_div(stride, 256)), dim3(256), 0, ctx->aux_stream, p->d_cur, stride);
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            safe_set_error("%s: %s", who, hipGetErrorString(e));
            rc = SAFE_E_HIP;
            break;
        }
        if (ring) {
            RingCall call;
            call.n = n;
            call.k = k;
            call.count = num_permutations;
            call.chunk_rows = kChunk;
            call.movable_hash = movable_fingerprint(movable_host, n);
            rc = p->ring_consumer ? ring_join_call(ring, call, slot_bytes) : ring_begin_call(ring, call, slot_bytes);
            if (rc != SAFE_OK) p->ring = nullptr;            // (no call is open on the ring: nothing to end)
        }
    } while (0);
    if (rc == SAFE_OK) rc = upload_movpos(p);
    if (rc == SAFE_OK && device_gen && num_permutations > 0) rc = perms_generate_on_device(p, device_key);
    if (rc != SAFE_OK) {
        perms_free(p);
        return rc;
    }
    if (!device_gen) drawer_start(p);
    safe_trace(reused ? "perms_create: done (buffers reused)" : "perms_create: done");
    *out = p;
    return SAFE_OK;
}

int safe_perms_create(safe_ctx *ctx, int64_t n, const uint8_t *movable_host, int64_t num_permutations, int has_seed,
                      uint32_t seed, safe_perms **out) {
    return perms_create_impl(ctx, n, movable_host, num_permutations, has_seed, seed, false, "safe_perms_create", out);
}

int safe_perms_create_shared(safe_ctx *ctx, int64_t n, const uint8_t *movable_host, int64_t num_permutations, int has_seed,
                             uint32_t seed, safe_perms **out) {
    return perms_create_impl(ctx, n, movable_host, num_permutations, has_seed, seed, true, "safe_perms_create_shared", out);
}

int safe_perms_create_device(safe_ctx *ctx, int64_t n, const uint8_t *movable_host, int64_t num_permutations, uint64_t key,
                             safe_perms **out) {
    return perms_create_impl(ctx, n, movable_host, num_permutations, 0, 0u, false, "safe_perms_create_device", out, true, key);
}

int safe_ctx_share_stream(safe_ctx *ctx, const char *name, int local_rank, int local_world, int64_t capacity_bytes) {
    SAFE_REQUIRE(ctx && name, "safe_ctx_share_stream: NULL argument");
    SAFE_REQUIRE(ctx->ring == nullptr, "safe_ctx_share_stream: this context already shares a stream (safe_ctx_unshare_stream first)");
    if (local_world <= 1) return SAFE_OK;                      // a node with one rank has nobody to share with
    perms_cache_drop(ctx);
    return ring_open(name, local_rank, local_world, capacity_bytes, 20.0, &ctx->ring);     // (the ranks call this together: 20 s covers their skew)
}

int safe_ctx_unshare_stream(safe_ctx *ctx) {
    SAFE_REQUIRE(ctx, "safe_ctx_unshare_stream: NULL argument");
    if (ctx->ring) ring_close(ctx->ring);
    ctx->ring = nullptr;
    return SAFE_OK;
}

int safe_perms_timing(safe_perms *perms, double *out5) {
    SAFE_REQUIRE(perms && out5, "safe_perms_timing: NULL argument");
    {
        std::lock_guard<std::mutex> lk(perms->draw_mu);
        out5[0] = perms->draw_busy_ms;
        out5[1] = perms->drawn_all_ms;
    }
    out5[2] = perms->enqueued_all_ms;
    out5[3] = perms->ring_wait_ms;
    out5[4] = perms->device_gen ? 3.0 : perms->ring ? (perms->ring_consumer ? 2.0 : 1.0) : 0.0;
    return SAFE_OK;
}

int safe_perms_twin_stats(safe_perms *perms, int *twin_active, int64_t *chunks, int64_t *chunks_won_by_twin) {
    SAFE_REQUIRE(perms && twin_active && chunks && chunks_won_by_twin, "safe_perms_twin_stats: NULL argument");
    std::lock_guard<std::mutex> lk(perms->draw_mu);
    *twin_active = perms->twin ? 1 : 0;
    *chunks = perms->drawn_chunks;
    *chunks_won_by_twin = perms->twin_wins;
    return SAFE_OK;
}

// A handle over finished tables: `count` composed rows of n + 1 entries at `src` (host memory, or device memory that is
// complete on the aux stream when this is called).  No draw stream, no pipeline: every stage is done at once.
static int perms_table_handle(safe_ctx *ctx, int64_t n, int64_t count, const int32_t *src, hipMemcpyKind kind, const char *who,
                              safe_perms **out) {
    const int64_t stride = n + 1, rows = std::max<int64_t>(count, 1);
    safe_perms *p = new safe_perms();
    p->ctx = ctx;
    p->n = n;
    p->count = count;
    p->from_table = true;
    p->k = n;
    p->generated = p->enqueued = count;
    int rc = SAFE_OK;
    do {
        if ((rc = dev_alloc(&p->table, static_cast<size_t>(rows) * stride)) != SAFE_OK) break;
        hipError_t e = count ? hipMemcpyAsync(p->table, src, static_cast<size_t>(count) * stride * sizeof(int32_t), kind, ctx->aux_stream)
                             : hipMemsetAsync(p->table, 0, static_cast<size_t>(rows) * stride * sizeof(int32_t), ctx->aux_stream);
        if (e == hipSuccess && n < 65535) {
            p->stride16 = (stride + 7) / 8 * 8;
            if ((rc = dev_alloc(&p->table16, static_cast<size_t>(rows) * p->stride16)) != SAFE_OK) break;
            if ((rc = dev_alloc(&p->d_cur, stride)) != SAFE_OK) break;
            hipLaunchKernelGGL(k_iota, dim3(ceil_div(stride, 256)), dim3(256), 0, ctx->aux_stream, p->d_cur, stride);
            // grid.y is limited to 65535 blocks: row blocks of at most that many (the reference places no cap on num_permutations)
            for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
                const int64_t rb = std::min<int64_t>(65535, rows - r0);
                const dim3 grid(ceil_div(std::max<int64_t>(stride, p->stride16), 256), rb), block(256);
                hipLaunchKernelGGL(k_table16, grid, block, 0, ctx->aux_stream, p->table + r0 * stride, stride,
                                   p->table16 + r0 * p->stride16, p->stride16, n);
            }
            e = hipGetLastError();
        }
        p->stages = perms_stage_plan(count);
        const int64_t n_chunks = stage_count(p);
        p->chunk_done.assign(std::max<int64_t>(n_chunks, 1), nullptr);
        for (size_t c = 0; c < p->chunk_done.size() && e == hipSuccess; ++c) {
            e = hipEventCreateWithFlags(&p->chunk_done[c], safe_event_flags(hipEventDisableTiming));
            if (e == hipSuccess) e = hipEventRecord(p->chunk_done[c], ctx->aux_stream);
        }
        if (e == hipSuccess) e = safe_stream_sync(ctx->aux_stream);    // `src` may be host memory of the caller's call
        if (e != hipSuccess) {
            safe_set_error("%s: %s", who, hipGetErrorString(e));
            rc = SAFE_E_HIP;
        }
    } while (0);
    if (rc != SAFE_OK) {
        perms_free(p);
        return rc;
    }
    *out = p;
    return SAFE_OK;
}

int safe_perms_create_from_table(safe_ctx *ctx, int64_t n, int64_t num_permutations, const int32_t *perm_idx_host,
                                 safe_perms **out) {
    SAFE_REQUIRE(ctx && out && (perm_idx_host || num_permutations == 0), "safe_perms_create_from_table: NULL argument");
    SAFE_REQUIRE(n >= 1 && n < (1ll << 31) - 1, "safe_perms_create_from_table: n out of range");
    SAFE_REQUIRE(num_permutations >= 0, "safe_perms_create_from_table: negative permutation count");
    *out = nullptr;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    // every row must be a permutation of 0..n-1 (safe_extras.py:58 moves rows, it never duplicates one)
    const int64_t stride = n + 1, rows = std::max<int64_t>(num_permutations, 1);
    std::vector<int32_t> staged(static_cast<size_t>(rows) * stride, 0);
    {
        std::vector<uint8_t> seen(n);
        for (int64_t p = 0; p < num_permutations; ++p) {
            std::fill(seen.begin(), seen.end(), 0);
            const int32_t *row = perm_idx_host + p * n;
            for (int64_t i = 0; i < n; ++i) {
                const int32_t v = row[i];
                if (v < 0 || v >= n || seen[v]) {
                    safe_set_error("safe_perms_create_from_table: row %lld is not a permutation of 0..%lld (entry %lld = %d)",
                                   (long long)p, (long long)(n - 1), (long long)i, v);
                    return SAFE_E_VALUE;
                }
                seen[v] = 1;
            }
            memcpy(staged.data() + p * stride, row, n * sizeof(int32_t));
            staged[p * stride + n] = static_cast<int32_t>(n);           // the padding row maps to itself
        }
    }
    return perms_table_handle(ctx, n, num_permutations, staged.data(), hipMemcpyHostToDevice, "safe_perms_create_from_table", out);
}

// Permutations [p0, p1) of an existing handle as a handle of their own (device-to-device copy of the finished rows): the
// permutation-axis split -- every rank draws the ONE stream of the call (safe_extras.py:46, 58: the permutations are
// cumulative, so each rank needs all of its predecessors' draws anyway) and tests its own range of it.
int safe_perms_slice(safe_perms *perms, int64_t p0, int64_t p1, safe_perms **out) {
    SAFE_REQUIRE(perms && out, "safe_perms_slice: NULL argument");
    SAFE_REQUIRE(0 <= p0 && p0 <= p1 && p1 <= perms->count, "safe_perms_slice: range [%lld,%lld) outside [0,%lld)", (long long)p0,
                 (long long)p1, (long long)perms->count);
    *out = nullptr;
    safe_ctx *ctx = perms->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    // The node's producer publishes chunks as it enqueues them.  A producer that only needs [p0, p1) for itself would leave the
    // consumers (whose ranges lie further down the stream) waiting until its handle is destroyed, i.e. until its own kernels are
    // done: the split would run one rank after the other.  So it moves the WHOLE stream along before it takes its slice.
    if (perms->ring && !perms->ring_consumer) SAFE_TRY(perms_generate_until(perms, perms->count));
    SAFE_TRY(perms_wait(perms, p1, ctx->aux_stream));
    return perms_table_handle(ctx, perms->n, p1 - p0, perms->table + p0 * (perms->n + 1), hipMemcpyDeviceToDevice, "safe_perms_slice", out);
}

int safe_perms_destroy(safe_perms *perms) {
    if (!perms) return SAFE_OK;
    (void)hipSetDevice(perms->ctx->device);
    safe_ctx *ctx = perms->ctx;
    if (perms->from_table) {                              // no stream, no staging buffers: nothing worth caching
        (void)safe_stream_sync(ctx->aux_stream);
        (void)safe_stream_sync(ctx->side_stream);
        (void)safe_stream_sync(ctx->stream);
        perms_free(perms);
        return SAFE_OK;
    }
    int rc = SAFE_OK;
    if (perms->ring && !perms->ring_consumer)             // the node's producer publishes the WHOLE stream, whatever it used itself
        rc = perms_generate_until(perms, perms->count);
    drawer_stop(perms);
    if (perms->ring) ring_end_call(perms->ring);
    perms->ring = nullptr;
    perms->ring_consumer = false;
    (void)safe_stream_sync(ctx->aux_stream);
    (void)safe_stream_sync(ctx->side_stream);
    (void)safe_stream_sync(ctx->stream);
    // keep the allocations for the next handle of the same shape (hipMalloc / hipHostMalloc / hipFree of
    // ~30 MB per call cost more than a millisecond)
    perms_cache_drop(ctx);
    if (perms->stream) draw_stream_free(perms->stream);
    perms->stream = nullptr;
    if (perms->stream2) draw_stream_free(perms->stream2);
    perms->stream2 = nullptr;
    (void)hipFree(perms->inverse_t);
    perms->inverse_t = nullptr;
    ctx->perm_cache = perms;
    return rc;
}

int safe_perms_read(safe_perms *perms, int64_t p0, int64_t p1, int32_t *out_host) {
    SAFE_REQUIRE(perms && out_host, "safe_perms_read: NULL argument");
    SAFE_REQUIRE(0 <= p0 && p0 <= p1 && p1 <= perms->count, "safe_perms_read: range [%lld,%lld) outside [0,%lld)",
                 (long long)p0, (long long)p1, (long long)perms->count);
    safe_ctx *ctx = perms->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    SAFE_TRY(perms_wait(perms, p1, ctx->stream));
    const int64_t stride = perms->n + 1;
    if (p1 > p0)
        SAFE_HIP_CHECK(hipMemcpy2DAsync(out_host, perms->n * sizeof(int32_t), perms->table + p0 * stride,
                                        stride * sizeof(int32_t), perms->n * sizeof(int32_t), p1 - p0,
                                        hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    return SAFE_OK;
}

}  // extern "C"
