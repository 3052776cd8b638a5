// Enrichment on gfx950: neighborhood score, permutation test and hypergeometric test.
//
// Replaces compute_neighborhood_score / run_permutations (safepy/safe_extras.py:6-70) and
// compute_pvalues_by_randomization / compute_pvalues_by_hypergeom + the binarisation of
// compute_pvalues (safepy/safe.py:468-608, FDR branch excluded).
//
// Formulation.  The reference evaluates every permutation as a dense product
// A[N,N] . B[perm][N,M] (np.dot -> dgemm, 2*N^2*M flop) although A is ~1-3 % dense.  Here
// the membership is held as SELL-64 (64-row slices, column-major inside a slice, rows
// sorted by count) and a permuted score is a gather-sum
//     S_p[i, j] = sum_{k in nbr(i)} B0[cur_p[k], j]
// over a column tile of B0 that stays resident on chip for all P permutations; the
// <= / >= comparisons against the observed score and the counters stay in registers, and
// p-values / NES / binarisation are produced in the kernel epilogue -- S_p never exists in
// HBM.  All arithmetic is f64 like the reference's dgemm (the comparison counts are only
// reproducible at f64); compiled with -ffp-contract=off so the z-score's EXX - M*M is not
// contracted.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "common.h"

// --------------------------------------------------------------------------------------
// Column tiles of the attribute matrix: Bt[tile][row 0..n][plane][BN] f64, NaN -> 0; row n
// is the all-zero padding row the SELL padding entries point at.  Planes: 0 = B0;
// z-score adds 1 = B0^2 (squared in B's own dtype like np.power(B, 2),
// safe_extras.py:24) and 2 = not-NaN indicator (safe_extras.py:13).
// --------------------------------------------------------------------------------------
template <typename T, int BN, int PLANES>
__global__ __launch_bounds__(256) void k_tile_prep(const void *__restrict__ raw, int64_t n, int64_t rs, int64_t cs,
                                                   int64_t col0, int64_t mloc, int64_t n_tiles,
                                                   double *__restrict__ bt) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t total = n_tiles * (n + 1) * BN;
    if (idx >= total) return;
    const int c = static_cast<int>(idx % BN);
    const int64_t row = (idx / BN) % (n + 1);
    const int64_t tile = idx / (static_cast<int64_t>(BN) * (n + 1));
    const int64_t j = tile * BN + c;
    double v = 0.0, v2 = 0.0, nn = 0.0;
    if (row < n && j < mloc) {
        const T x = reinterpret_cast<const T *>(raw)[row * rs + (col0 + j) * cs];
        if (x == x) {
            v = static_cast<double>(x);
            v2 = static_cast<double>(static_cast<T>(x * x));
            nn = 1.0;
        }
    }
    double *dst = bt + ((tile * (n + 1) + row) * PLANES) * BN + c;
    dst[0] = v;
    if (PLANES == 3) {
        dst[BN] = v2;
        dst[2 * BN] = nn;
    }
}

// --------------------------------------------------------------------------------------
// score from the accumulated planes (safe_extras.py:15-31)
// --------------------------------------------------------------------------------------
template <int BN, bool Z>
__device__ __forceinline__ void finish_score(const double (&acc)[Z ? 3 : 1][BN], double (&score)[BN]) {
#pragma unroll
    for (int c = 0; c < BN; ++c) {
        if constexpr (!Z) {
            score[c] = acc[0][c];
        } else {
            const double cnt = acc[2][c];
            const double mean = acc[0][c] / cnt;
            const double exx = acc[1][c] / cnt;
            const double sd = sqrt(exx - mean * mean);
            double s = mean / sd;
            if (sd == 0.0) s = __longlong_as_double(0x7FF8000000000000ll);
            if (cnt < 3.0) s = __longlong_as_double(0x7FF8000000000000ll);
            score[c] = s;
        }
    }
}


// --------------------------------------------------------------------------------------
// K5 (general f64 form): one workgroup = (column tile, slice group); its 4 waves walk
// 64-row slices; inside a slice every lane owns one neighborhood (row) and BN columns.
// Workgroups of one tile are placed on one XCD (blockIdx % 8) so the tile is fetched into
// a single L2.
// --------------------------------------------------------------------------------------
template <int BN, bool Z>
__global__ __launch_bounds__(256) void k_permtest_gather(
    const int32_t *__restrict__ sell_row, const int64_t *__restrict__ slice_off,
    const int32_t *__restrict__ slice_width, const int32_t *__restrict__ sell_col, int64_t n_slices, int64_t n,
    const double *__restrict__ bt, int64_t n_tiles, int n_groups, const int32_t *__restrict__ table, int64_t n_perm,
    int64_t mloc, PermOut out) {
    constexpr int PLANES = Z ? 3 : 1;
    constexpr int ROWLEN = PLANES * BN;
    const int64_t b = blockIdx.x;
    const int64_t xcd = b & 7, q = b >> 3;
    const int64_t tile = (q / n_groups) * 8 + xcd;
    const int group = static_cast<int>(q % n_groups);
    if (tile >= n_tiles) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double *tb = bt + tile * (n + 1) * ROWLEN;
    const int64_t stride = n + 1;
    const int64_t jbase = tile * BN;

    for (int64_t s = group + static_cast<int64_t>(n_groups) * wave; s < n_slices; s += static_cast<int64_t>(n_groups) * 4) {
        const int32_t row = sell_row[s * 64 + lane];
        const int32_t *cols = sell_col + slice_off[s] + lane;
        const int wdt = slice_width[s];

        double acc[PLANES][BN];
        double obs[BN];
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int c = 0; c < BN; ++c) acc[pl][c] = 0.0;
        for (int t = 0; t < wdt; ++t) {
            const double *r = tb + static_cast<int64_t>(cols[t * 64]) * ROWLEN;
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                for (int c = 0; c < BN; ++c) acc[pl][c] += r[pl * BN + c];
        }
        finish_score<BN, Z>(acc, obs);

        unsigned int cneg[BN], cpos[BN];
#pragma unroll
        for (int c = 0; c < BN; ++c) cneg[c] = cpos[c] = 0;

        for (int64_t p = 0; p < n_perm; ++p) {
            const int32_t *cur = table + p * stride;
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                for (int c = 0; c < BN; ++c) acc[pl][c] = 0.0;
#pragma unroll 2
            for (int t = 0; t < wdt; ++t) {
                const double *r = tb + static_cast<int64_t>(cur[cols[t * 64]]) * ROWLEN;
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                    for (int c = 0; c < BN; ++c) acc[pl][c] += r[pl * BN + c];
            }
            double sc[BN];
            finish_score<BN, Z>(acc, sc);
#pragma unroll
            for (int c = 0; c < BN; ++c) {
                cneg[c] += sc[c] <= obs[c];      // safe_extras.py:65
                cpos[c] += sc[c] >= obs[c];      // safe_extras.py:66
            }
        }

        // ---- epilogue: everything compute_pvalues derives from the counters -------------
        const bool live = row >= 0;
        const int64_t o = static_cast<int64_t>(live ? row : 0) * mloc + jbase;
#pragma unroll
        for (int c = 0; c < BN; ++c) {
            const bool ok = live && (jbase + c < mloc);
            const bool obs_nan = obs[c] != obs[c];
            if (ok && out.ns) out.ns[o + c] = obs[c];
            if (out.mode == 1) {
                // run_permutations returns plain counts (no NaN masking, safe_extras.py:70)
                if (ok) {
                    out.counts_neg[o + c] = static_cast<double>(cneg[c]);
                    out.counts_pos[o + c] = static_cast<double>(cpos[c]);
                }
            } else if (out.mode == 2) {
                const double qnan = __longlong_as_double(0x7FF8000000000000ll);
                // safe.py:528-533: counts[isnan(ns)] = nan; p = counts / P
                const double pn = obs_nan ? qnan : static_cast<double>(cneg[c]) / static_cast<double>(n_perm);
                const double pp = obs_nan ? qnan : static_cast<double>(cpos[c]) / static_cast<double>(n_perm);
                // safe.py:546-554 via the caller's -log10 table (k/P has P+1 possible values)
                const double en = obs_nan ? qnan : out.nes_table[cneg[c]];
                const double ep = obs_nan ? qnan : out.nes_table[cpos[c]];
                double nes = ep - en;
                if (out.sign_mode == SAFE_SIGN_HIGHEST) nes = ep;
                if (out.sign_mode == SAFE_SIGN_LOWEST) nes = en;
                // safe.py:468-470
                const bool hit = (nes == nes) && (fabs(nes) > out.nes_threshold);
                if (ok) {
                    out.pvalues_neg[o + c] = pn;
                    out.pvalues_pos[o + c] = pp;
                    out.nes[o + c] = nes;
                    out.nes_binary[o + c] = hit ? 1.0 : 0.0;
                }
                const unsigned long long bal = __ballot(ok && hit);
                if (lane == 0 && bal) atomicAdd(&out.enriched[jbase + c], static_cast<unsigned int>(__popcll(bal)));
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// K5 (binary attributes, sparse form).  For 0/1 data the permuted neighborhood sum is a
// set intersection,   S_p[i,j] = | nbr(i)  n  { inv_p[r] : B[r,j] = 1 } |,
// so instead of gathering nbr(i) for every (i,j,p) -- nnz(A)*M adds per permutation, almost
// all of them adding 0 at GO-like densities (~1 %) -- the kernel SCATTERS: for every 1 of
// column j it adds 1 to the neighborhoods that contain its permuted position
// (nnz(A)*nnz(B)/N increments per permutation) and only ever looks at touched
// neighborhoods.  Untouched entries have S_p = 0, whose contribution is known
// (S_p <= S_obs always; S_p >= S_obs iff S_obs == 0), and because a count only grows
// inside one permutation each threshold is crossed at most once, at a known increment:
//     counts_neg = P - #{p : S_p > S_obs}  = P - #{p : some increment made S_p == S_obs + 1}
//     counts_pos = S_obs == 0 ? P : #{p : some increment made S_p == S_obs}.
// Integer arithmetic: bit-exact against the reference's f64 sums of 0/1 values.
//
// One workgroup owns one attribute at a time (dynamic queue, largest attribute first); its
// per-node state lives in LDS: EP[i] (epoch-tagged running count), SO[i] = S_obs (u16),
// CT[i] = #greater<<16 | #reached.  The running count is a DOWN-counter updated with one
// LDS atomicDec(addr, D_p): "old > D_p ? D_p : old - 1" -- D_p = D_0 - 4096*p, so a value
// left over from an earlier permutation (> D_p) is lazily reset by the very increment that
// first touches it.  One walk per permutation, no reset pass; no S_p, no counter ever
// touches HBM.
// --------------------------------------------------------------------------------------
#define SC_BATCH 16
#define SC_D0 0xFFFF0000u
#define SC_EPOCH 4096u
#define SC_CHUNK 16          // permutations per prefetched chunk of the transposed inverse table

// Enumerates up to SC_BATCH 64-wide segments of the transposed-membership rows held one per
// lane (beg/end), level by level (segment 0 of every row, then segment 1 of the long rows,
// ...), and issues the column loads.  All control is wave-uniform (scalar): no scan, no LDS.
struct SegCursor {
    unsigned long long mask;
    int level;
    bool done;
};

__device__ __forceinline__ SegCursor seg_begin(int nseg) {
    SegCursor c;
    c.level = 0;
    c.mask = __ballot(nseg > 0);
    c.done = c.mask == 0;
    return c;
}

__device__ __forceinline__ void seg_issue(SegCursor &c, int32_t beg, int32_t end, int nseg, int lane,
                                          const int32_t *__restrict__ at_col, int32_t (&node)[SC_BATCH]) {
#pragma unroll
    for (int u = 0; u < SC_BATCH; ++u) {
        node[u] = -1;
        if (!c.done && c.mask == 0) {
            ++c.level;
            c.mask = __ballot(nseg > c.level);
            c.done = c.mask == 0;
        }
        if (!c.done) {
            const int row = __ffsll(static_cast<unsigned long long>(c.mask)) - 1;
            c.mask &= c.mask - 1;
            const int32_t b = __builtin_amdgcn_readlane(beg, row);
            const int32_t e = __builtin_amdgcn_readlane(end, row);
            const int32_t t = b + (c.level << 6) + lane;
            if (t < e) node[u] = at_col[t];
        }
    }
    if (!c.done && c.mask == 0) {            // so that `done` is exact after a full batch
        ++c.level;
        c.mask = __ballot(nseg > c.level);
        c.done = c.mask == 0;
    }
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_permtest_scatter(
    int64_t n, int64_t n_perm, const uint16_t *__restrict__ inv_t, int64_t inv_stride,
    const int32_t *__restrict__ at_ptr, const int32_t *__restrict__ at_col, const int32_t *__restrict__ sup_ptr,
    const int32_t *__restrict__ sup_row, int64_t col0, const int32_t *__restrict__ order, int64_t mloc,
    unsigned int *__restrict__ queue, PermOut out) {
    extern __shared__ unsigned int lds[];
    unsigned int *EP = lds;                                            // [n]
    unsigned int *CT = lds + n;                                        // [n]
    unsigned int *slot_box = lds + 2 * n;                              // [4]
    unsigned short *SO = reinterpret_cast<unsigned short *>(lds + 2 * n + 4);   // [n]
    constexpr int NT = 64 * WAVES;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

    // LDS stage of one batch: SC_BATCH returning atomics in flight together
    auto count_batch = [&](const int32_t (&node)[SC_BATCH], unsigned int dp) {
        unsigned int so[SC_BATCH], old[SC_BATCH];
#pragma unroll
        for (int u = 0; u < SC_BATCH; ++u) so[u] = node[u] >= 0 ? SO[node[u]] : 0u;
#pragma unroll
        for (int u = 0; u < SC_BATCH; ++u) old[u] = node[u] >= 0 ? atomicDec(&EP[node[u]], dp) : 0u;
#pragma unroll
        for (int u = 0; u < SC_BATCH; ++u) {
            const unsigned int v = old[u] > dp ? 1u : dp - old[u] + 2u;     // running count after this increment
            const unsigned int inc = (static_cast<unsigned int>(v == so[u] + 1u) << 16) | static_cast<unsigned int>(v == so[u]);
            if (node[u] >= 0 && inc) atomicAdd(&CT[node[u]], inc);
        }
    };

    for (;;) {
        if (threadIdx.x == 0) *slot_box = atomicAdd(queue, 1u);
        __syncthreads();
        const int64_t slot = *slot_box;
        __syncthreads();
        if (slot >= mloc) break;
        const int j = order[slot];                                     // local column index
        const int32_t sbeg = sup_ptr[col0 + j];
        const int nrows = sup_ptr[col0 + j + 1] - sbeg;

        for (int64_t i = threadIdx.x; i < n; i += NT) {
            EP[i] = 0xFFFFFFFFu;
            CT[i] = 0;
        }
        // this wave's share of the support rows: row q = round*NT + wave + WAVES*lane
        const int q0 = wave + WAVES * lane;
        const bool has0 = q0 < nrows;
        const int32_t r0 = has0 ? sup_row[sbeg + q0] : 0;
        __syncthreads();

        // ---- observed counts: identity permutation (safe.py:496-499) ----------------------
        for (int base = 0; base < nrows; base += NT) {
            const int q = base + q0;
            int32_t beg = 0, end = 0;
            if (q < nrows) {
                const int32_t r = sup_row[sbeg + q];
                beg = at_ptr[r];
                end = at_ptr[r + 1];
            }
            const int nseg = (end - beg + 63) >> 6;
            SegCursor c = seg_begin(nseg);
            while (!c.done) {
                int32_t node[SC_BATCH];
                seg_issue(c, beg, end, nseg, lane, at_col, node);
#pragma unroll
                for (int u = 0; u < SC_BATCH; ++u)
                    if (node[u] >= 0) atomicAdd(&CT[node[u]], 1u);
            }
        }
        __syncthreads();
        for (int64_t i = threadIdx.x; i < n; i += NT) {
            SO[i] = static_cast<unsigned short>(CT[i]);
            CT[i] = 0;
        }
        __syncthreads();

        // ---- permutations, software pipelined for the round-0 rows ------------------------
        // kq: 32-entry queue of this lane's upcoming positions k = inv_p[r0] (16 bits each);
        // refilled one chunk ahead from the transposed inverse table.
        const uint16_t *my_inv = inv_t + static_cast<int64_t>(r0) * inv_stride;
        unsigned int kq[16], kn[8];
        {
            const uint4 a0 = *reinterpret_cast<const uint4 *>(my_inv);
            const uint4 a1 = *reinterpret_cast<const uint4 *>(my_inv + 8);
            const uint4 a2 = *reinterpret_cast<const uint4 *>(my_inv + 16);
            const uint4 a3 = *reinterpret_cast<const uint4 *>(my_inv + 24);
            kq[0] = a0.x; kq[1] = a0.y; kq[2] = a0.z; kq[3] = a0.w; kq[4] = a1.x; kq[5] = a1.y; kq[6] = a1.z; kq[7] = a1.w;
            kq[8] = a2.x; kq[9] = a2.y; kq[10] = a2.z; kq[11] = a2.w; kq[12] = a3.x; kq[13] = a3.y; kq[14] = a3.z; kq[15] = a3.w;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) kn[i] = 0;

        int32_t beg = 0, end = 0, begn = 0, endn = 0;
        int32_t node[SC_BATCH];
        SegCursor cur;
        {   // prologue: p = 0 rows and first batch of column loads; p = 1 row pointers
            if (has0) {
                const int32_t k = kq[0] & 0xFFFFu;
                beg = at_ptr[k];
                end = at_ptr[k + 1];
                const int32_t k1 = kq[0] >> 16;
                begn = at_ptr[k1];
                endn = at_ptr[k1 + 1];
            }
            cur = seg_begin((end - beg + 63) >> 6);
            seg_issue(cur, beg, end, (end - beg + 63) >> 6, lane, at_col, node);
        }

        for (int64_t p = 0; p < n_perm; ++p) {
            const unsigned int dp = SC_D0 - SC_EPOCH * static_cast<unsigned int>(p);
            const int nseg = (end - beg + 63) >> 6;

            // (a) row pointers for p+2 are not needed yet; issue those for... p+1 were issued one
            //     iteration ago (begn/endn).  Advance the queue: element 0 becomes k_{p+1}.
#pragma unroll
            for (int i = 0; i < 15; ++i) kq[i] = __builtin_amdgcn_alignbit(kq[i + 1], kq[i], 16);
            kq[15] >>= 16;
            if (((p + 1) & (SC_CHUNK - 1)) == 0) {
                // the upper half is now empty: move the prefetched chunk in, prefetch the next one
#pragma unroll
                for (int i = 0; i < 8; ++i) kq[8 + i] = kn[i];
            }
            if ((p & (SC_CHUNK - 1)) == 0) {
                const int64_t c2 = (p / SC_CHUNK + 2) * SC_CHUNK;        // chunk that becomes the upper half next time
                const uint4 a0 = *reinterpret_cast<const uint4 *>(my_inv + c2);
                const uint4 a1 = *reinterpret_cast<const uint4 *>(my_inv + c2 + 8);
                kn[0] = a0.x; kn[1] = a0.y; kn[2] = a0.z; kn[3] = a0.w; kn[4] = a1.x; kn[5] = a1.y; kn[6] = a1.z; kn[7] = a1.w;
            }
            // row pointers for p+2 (k_{p+2} is queue element 1 after the shift)
            int32_t beg2 = 0, end2 = 0;
            if (has0) {
                const int32_t k2 = kq[0] >> 16;
                beg2 = at_ptr[k2];
                end2 = at_ptr[k2 + 1];
            }

            // (b) LDS stage of p: first batch was loaded during the previous iteration
            count_batch(node, dp);
            while (!cur.done) {                                          // long tails: not pipelined
                seg_issue(cur, beg, end, nseg, lane, at_col, node);
                count_batch(node, dp);
            }
            // rows beyond the first NT of very large attributes: plain path
            for (int base = NT; base < nrows; base += NT) {
                const int q = base + q0;
                int32_t b2 = 0, e2 = 0;
                if (q < nrows) {
                    const int32_t r = sup_row[sbeg + q];
                    const int32_t k = inv_t[static_cast<int64_t>(r) * inv_stride + p];
                    b2 = at_ptr[k];
                    e2 = at_ptr[k + 1];
                }
                const int ns2 = (e2 - b2 + 63) >> 6;
                SegCursor c2 = seg_begin(ns2);
                while (!c2.done) {
                    seg_issue(c2, b2, e2, ns2, lane, at_col, node);
                    count_batch(node, dp);
                }
            }

            // (c) column loads of p+1's first batch (its row pointers arrived during (b))
            beg = begn;
            end = endn;
            begn = beg2;
            endn = end2;
            if (p + 1 < n_perm) {
                const int nsegn = (end - beg + 63) >> 6;
                cur = seg_begin(nsegn);
                seg_issue(cur, beg, end, nsegn, lane, at_col, node);
            }
            if (WAVES > 1) __syncthreads();     // all waves of a workgroup share the epoch
        }
        __syncthreads();

        // ---- epilogue (same outputs as the gather kernel) ---------------------------------
        unsigned int hits = 0;
        for (int64_t i = threadIdx.x; i < n; i += NT) {
            const unsigned int so = SO[i];
            const unsigned int ct = CT[i];
            const unsigned int cneg = static_cast<unsigned int>(n_perm) - (ct >> 16);
            const unsigned int cpos = so == 0 ? static_cast<unsigned int>(n_perm) : (ct & 0xFFFFu);
            const int64_t o = i * mloc + j;
            if (out.ns) out.ns[o] = static_cast<double>(so);
            if (out.mode == 1) {
                out.counts_neg[o] = static_cast<double>(cneg);
                out.counts_pos[o] = static_cast<double>(cpos);
            } else if (out.mode == 2) {
                const double en = out.nes_table[cneg], ep = out.nes_table[cpos];
                double nes = ep - en;
                if (out.sign_mode == SAFE_SIGN_HIGHEST) nes = ep;
                if (out.sign_mode == SAFE_SIGN_LOWEST) nes = en;
                const bool hit = (nes == nes) && (fabs(nes) > out.nes_threshold);
                out.pvalues_neg[o] = static_cast<double>(cneg) / static_cast<double>(n_perm);
                out.pvalues_pos[o] = static_cast<double>(cpos) / static_cast<double>(n_perm);
                out.nes[o] = nes;
                out.nes_binary[o] = hit ? 1.0 : 0.0;
                hits += hit;
            }
        }
        if (out.mode == 2) {
            for (int off = 32; off > 0; off >>= 1) hits += __shfl_down(hits, off);
            if (lane == 0 && hits) atomicAdd(&out.enriched[j], hits);
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------
// K5 (binary attributes, bit-sliced form).  64 attributes ride in one 64-bit word per node:
// T[r] = bits of B[r, 64*wg .. 64*wg+63] (NaN -> 0).  A lane owns one neighborhood (SELL-64
// slice row) and adds the words of its members with carry-save adders into VERTICAL
// counters s[level] (bit b of s[l] = bit l of the sum for attribute b): ~4.4 bitwise ops per
// member per 32 attributes instead of 32 adds.  The comparison with the observed sum and
// the two permutation counters are bit-sliced too, so every instruction works on 32
// attributes x 64 neighborhoods.  Exact integer arithmetic; no atomics; cost independent of
// the attribute density.  A workgroup = 4 adjacent slices x one 64-attribute word group;
// it keeps the word column T (8 B per node) and the current permutation (2 B per node,
// double buffered) in LDS; one barrier per permutation.
// --------------------------------------------------------------------------------------
#define BT_LV 10               // levels of a neighborhood sum: max row count < 1024

// 16-byte vector view of the u16 permutation rows: must be may_alias, the rows are read back
// as unsigned short (without it TBAA lets the compiler move those reads across the refills)
typedef uint4 __attribute__((may_alias)) uint4_alias;

// gfx950 has a three-input bitwise instruction (v_bitop3_b32, 8-bit truth table on
// a = 0xF0, b = 0xCC, c = 0xAA): a full adder is two instructions.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

// adds eight one-bit-per-attribute words into the low three levels of the vertical counter
// and returns the carry into level 3 (the "eights")
template <int LV>
__device__ __forceinline__ uint32_t vadd8(uint32_t (&s)[LV], const uint32_t (&x)[8]) {
    uint32_t t2a = maj3(s[0], x[0], x[1]);
    s[0] = xor3(s[0], x[0], x[1]);
    uint32_t t2b = maj3(s[0], x[2], x[3]);
    s[0] = xor3(s[0], x[2], x[3]);
    const uint32_t t4a = maj3(s[1], t2a, t2b);
    s[1] = xor3(s[1], t2a, t2b);
    t2a = maj3(s[0], x[4], x[5]);
    s[0] = xor3(s[0], x[4], x[5]);
    t2b = maj3(s[0], x[6], x[7]);
    s[0] = xor3(s[0], x[6], x[7]);
    const uint32_t t4b = maj3(s[1], t2a, t2b);
    s[1] = xor3(s[1], t2a, t2b);
    const uint32_t t8 = maj3(s[2], t4a, t4b);
    s[2] = xor3(s[2], t4a, t4b);
    return t8;
}

template <int LV>
__device__ __forceinline__ void vripple(uint32_t (&s)[LV], uint32_t t8) {
#pragma unroll
    for (int l = 3; l < LV; ++l) {
        const uint32_t c = s[l] & t8;
        s[l] ^= t8;
        t8 = c;
    }
}

// One pass over a lane's neighborhood: s = sum over members of T[cur[member]] (vertical).
// cols2 holds 2*member id (the LDS byte offset into the u16 permutation row); with SCALED the
// row itself holds 8*row id (the LDS byte offset into T, which starts at LDS address 0), so a
// member costs two LDS reads and one address add.  `first` = the first block's ids (the same
// for every permutation, loaded once per task); the ids of block b+1 are fetched as soon as
// block b's look-ups have been issued.  The carry into the eights is rippled only when some
// lane of the wave has one (sums rarely reach 8 on sparse annotations).
typedef const __attribute__((address_space(3))) unsigned short *lds_u16_ptr;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x2 *lds_u2_ptr;

template <bool IDENT, bool SCALED>
__device__ __forceinline__ void bits_accumulate(const uint16_t *__restrict__ cols2, int wdt, const uint32_t (&first)[8],
                                                uint32_t cur_addr, uint32_t t_addr, uint32_t (&s0)[BT_LV],
                                                uint32_t (&s1)[BT_LV]) {
    // cur_addr / t_addr: absolute LDS byte addresses of the permutation row and of T; with
    // SCALED the row entries already are absolute addresses of T rows
#pragma unroll
    for (int l = 0; l < BT_LV; ++l) s0[l] = s1[l] = 0;
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = first[u];
    const uint16_t *pc = cols2 + 8 * 64;                                              // next block's ids
    for (int t0 = 0; t0 < wdt; t0 += 8, pc += 8 * 64) {
        uint32_t r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (IDENT) r[u] = (c[u] << 2) + t_addr;                                   // 2*id -> address of T[id]
            else {
                const uint32_t v = *(lds_u16_ptr)(uintptr_t)(cur_addr + c[u]);
                r[u] = SCALED ? v : (v << 3) + t_addr;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = pc[u * 64];                                // (the array has a tail)
        uint32_t x0[8], x1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32x2 w = *(lds_u2_ptr)(uintptr_t)(r[u]);
            x0[u] = w.x;
            x1[u] = w.y;
        }
        const uint32_t e0 = vadd8(s0, x0);
        const uint32_t e1 = vadd8(s1, x1);
        if (__builtin_amdgcn_ballot_w64((e0 | e1) != 0)) {
            vripple(s0, e0);
            vripple(s1, e1);
        }
    }
}

// bit-sliced counter c[0..CL) += mask, with the carries out of the low three levels parked
// in `pend` (a position wraps at most once per 8 increments) and rippled every 8th call
// (a level's two operations are pinned in place -- "+v" -- like the sums' carry ripple: left to itself the compiler copies all
// CL levels of all four counters to a second register set around the three levels it updates and back again on the loop path,
// 32 64-bit moves per permutation, 40 % of the per-permutation instructions)
template <int CL>
__device__ __forceinline__ void vcount(uint32_t (&c)[CL], uint32_t &pend, uint32_t m) {
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        uint32_t k;
        asm("v_and_b32 %0, %1, %2\n\tv_xor_b32 %1, %1, %2" : "=&v"(k), "+v"(c[l]) : "v"(m));
        m = k;
    }
    pend |= m;
}

template <int CL>
__device__ __forceinline__ void vflush(uint32_t (&c)[CL], uint32_t &pend) {
    uint32_t m = pend;
#pragma unroll
    for (int l = 3; l < CL; ++l) {
        uint32_t k;
        asm("v_and_b32 %0, %1, %2\n\tv_xor_b32 %1, %1, %2" : "=&v"(k), "+v"(c[l]) : "v"(m));
        m = k;
    }
    pend = 0;
}

// A workgroup's copy of one word column into LDS: T[r] = src[r], r = 0 .. n.  Sixteen loads per thread are in flight before the
// first LDS store (the plain loop compiled to load - wait - store per 256 rows: 16 L2 latencies in a row at N = 3971, 12-16 us
// of every task, all four waves waiting at the barrier behind it).
template <int NT = 256>
__device__ __forceinline__ void load_word_column(uint2 *__restrict__ T, const uint2 *__restrict__ src, int64_t n) {
    constexpr int UNL = NT > 256 ? 4 : 16;
    for (int64_t r0 = threadIdx.x; r0 <= n; r0 += NT * UNL) {
        uint2 v[UNL];
#pragma unroll
        for (int u = 0; u < UNL; ++u) {
            const int64_t r = r0 + u * NT;
            v[u] = src[r <= n ? r : n];
        }
#pragma unroll
        for (int u = 0; u < UNL; ++u) {
            const int64_t r = r0 + u * NT;
            if (r <= n) T[r] = v[u];
        }
    }
}

template <int LEVELS>
__device__ __forceinline__ unsigned int vextract(const uint32_t (&c)[LEVELS], int bit) {
    unsigned int v = 0;
#pragma unroll
    for (int l = 0; l < LEVELS; ++l) v |= ((c[l] >> bit) & 1u) << l;
    return v;
}

// In-register transpose of a 32 x 32 bit matrix (rows = words): afterwards bit l of a[b] is
// what bit b of a[l] was.  Used to turn vertical counters (one word per level) into one
// value per attribute: 80 masked swaps instead of 32 x levels bit extractions.
__device__ __forceinline__ void transpose32(uint32_t (&a)[32]) {
#pragma unroll
    for (int j = 16; j != 0; j >>= 1) {
        const uint32_t m = j == 16 ? 0x0000FFFFu : j == 8 ? 0x00FF00FFu : j == 4 ? 0x0F0F0F0Fu : j == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if ((k & j) == 0) {
                const uint32_t t = ((a[k] >> j) ^ a[k + j]) & m;
                a[k] ^= t << j;
                a[k + j] ^= t;
            }
        }
    }
}

// WV = waves (= adjacent slices of a task) per workgroup.  4 is the form of the small networks; 16 (round 6) is for networks whose
// word column + permutation rows leave room for ONE workgroup per CU (N > 8190: 12 bytes per node): the sixteen waves share that
// one copy and the SIMDs keep four waves each, where four-wave workgroups ran one wave per SIMD.
template <int CL, bool SCALED, int WV = 4>
__global__ __launch_bounds__(64 * WV) void k_permtest_bits(
    int64_t n, int64_t n_perm, const uint16_t *__restrict__ cur16, int64_t stride16,
    const int32_t *__restrict__ sell_row, const int64_t *__restrict__ slice_off,
    const int32_t *__restrict__ slice_width, const uint16_t *__restrict__ sell_col2, int64_t n_slices,
    const uint2 *__restrict__ bbits, int64_t n_tasks, const int4 *__restrict__ tasks, int64_t p_base, int64_t p_limit,
    unsigned int *__restrict__ queue, int64_t mloc, unsigned int *__restrict__ gl_counts, int64_t n_pad,
    double *__restrict__ ns_out) {
    extern __shared__ unsigned int lds[];
    const int64_t t_words = 2 * ((n + 2) & ~int64_t(1));               // T: (n+1) uint2 at LDS address 0, 16-B padded
    uint2 *T = reinterpret_cast<uint2 *>(lds);
    unsigned short *CUR = reinterpret_cast<unsigned short *>(lds + t_words);     // [2][stride16]
    unsigned int *slot_box = lds + t_words + stride16;                  // after the two u16 buffers
    // absolute LDS byte addresses (the dynamic segment need not start at 0)
    const uint32_t t_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned int *)lds);
    const uint32_t cur_bytes0 = t_addr + static_cast<uint32_t>(t_words * 4);
    const uint32_t cur_bytes1 = cur_bytes0 + static_cast<uint32_t>(stride16 * 2);
    const uint32_t t_addr2 = t_addr | (t_addr << 16);                    // both u16 halves
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int vec_per_row = static_cast<int>(stride16 / 8);              // uint4 (8 x u16) per table row
    constexpr int NT = 64 * WV;

    // a table row (8 x u16 per vector) scaled to T byte offsets
    auto scale_row = [=](uint4 v) {
        if (SCALED) {                         // 8 * row + address of T, per u16 half (no carries: < 65536)
            v.x = ((v.x << 3) & 0xFFF8FFF8u) + t_addr2;
            v.y = ((v.y << 3) & 0xFFF8FFF8u) + t_addr2;
            v.z = ((v.z << 3) & 0xFFF8FFF8u) + t_addr2;
            v.w = ((v.w << 3) & 0xFFF8FFF8u) + t_addr2;
        }
        return v;
    };

    for (;;) {
        if (threadIdx.x == 0) *slot_box = atomicAdd(queue, 1u);
        __syncthreads();
        const int64_t slot = *slot_box;
        __syncthreads();
        if (slot >= n_tasks) break;
        const int4 task = tasks[slot];
        const int wg = task.x, sg = task.y;
        // this task's permutations (task ranges are relative to the launch's chunk)
        const int64_t p_begin = p_base + task.z;
        const int64_t p_end = p_base + task.w < p_limit ? p_base + task.w : p_limit;
        if (p_end <= p_begin) continue;                                   // a launch shorter than the task grid's span

        load_word_column<NT>(T, bbits + static_cast<int64_t>(wg) * (n + 1), n);
        if (p_end > p_begin)
            for (int v = threadIdx.x; v < vec_per_row; v += NT)
                reinterpret_cast<uint4_alias *>(CUR)[v] =
                    scale_row(reinterpret_cast<const uint4_alias *>(cur16 + p_begin * stride16)[v]);

        const int64_t s = static_cast<int64_t>(sg) * WV + wave;
        const bool active = s < n_slices;
        const int32_t row = active ? sell_row[s * 64 + lane] : -1;
        const uint16_t *cols2 = sell_col2 + (active ? slice_off[s] : 0) + lane;
        const int wdt = active ? slice_width[s] : 0;
        uint32_t first[8];                                               // 2 * member id of the first block
#pragma unroll
        for (int u = 0; u < 8; ++u) first[u] = cols2[u * 64];
        __syncthreads();

        uint32_t o0[BT_LV], o1[BT_LV];                                   // observed sums (safe.py:496-499)
        bits_accumulate<true, SCALED>(cols2, wdt, first, 0u, t_addr, o0, o1);

        uint32_t g0[CL], g1[CL], l0[CL], l1[CL];                          // #(S_p > S_obs), #(S_p < S_obs)
        uint32_t gp0 = 0, gp1 = 0, lp0 = 0, lp1 = 0;
#pragma unroll
        for (int l = 0; l < CL; ++l) g0[l] = g1[l] = l0[l] = l1[l] = 0;

        for (int64_t p = p_begin; p < p_end; ++p) {
            const int64_t rel = p - p_begin;
            const uint32_t cur_base = (rel & 1) ? cur_bytes1 : cur_bytes0;
            // next permutation's row: global -> registers now, registers -> LDS after the compute
            uint4 nxt = make_uint4(0, 0, 0, 0);
            const bool fetch = (p + 1 < p_end) && (static_cast<int>(threadIdx.x) < vec_per_row);
            if (fetch) nxt = reinterpret_cast<const uint4_alias *>(cur16 + (p + 1) * stride16)[threadIdx.x];

            uint32_t s0[BT_LV], s1[BT_LV];
            bits_accumulate<false, SCALED>(cols2, wdt, first, cur_base, t_addr, s0, s1);

            // bit-sliced compare as two borrow chains, least significant level first:
            // lt = borrow out of (S - O), gt = borrow out of (O - S); per level
            // borrow' = (s != o) ? subtrahend bit : borrow   -- one v_bitop3_b32 each
            uint32_t gt0 = 0, gt1 = 0, lt0 = 0, lt1 = 0;
#pragma unroll
            for (int l = 0; l < BT_LV; ++l) {
                // f(s, o, b) = (s != o) ? o : b  -> 0x8E ;  (s != o) ? s : b -> 0xB2
                lt0 = __builtin_amdgcn_bitop3_b32(s0[l], o0[l], lt0, 0x8E);
                gt0 = __builtin_amdgcn_bitop3_b32(s0[l], o0[l], gt0, 0xB2);
                lt1 = __builtin_amdgcn_bitop3_b32(s1[l], o1[l], lt1, 0x8E);
                gt1 = __builtin_amdgcn_bitop3_b32(s1[l], o1[l], gt1, 0xB2);
            }
            vcount<CL>(g0, gp0, gt0);
            vcount<CL>(g1, gp1, gt1);
            vcount<CL>(l0, lp0, lt0);
            vcount<CL>(l1, lp1, lt1);
            if ((rel & 7) == 7) {
                vflush<CL>(g0, gp0);
                vflush<CL>(g1, gp1);
                vflush<CL>(l0, lp0);
                vflush<CL>(l1, lp1);
            }

            if (vec_per_row > NT) {                                      // rows longer than one vector per thread: strided copy
                for (int v = threadIdx.x + NT; v < vec_per_row; v += NT)
                    if (p + 1 < p_end)
                        reinterpret_cast<uint4_alias *>(CUR + ((rel + 1) & 1) * stride16)[v] =
                            scale_row(reinterpret_cast<const uint4_alias *>(cur16 + (p + 1) * stride16)[v]);
            }
            if (fetch) reinterpret_cast<uint4_alias *>(CUR + ((rel + 1) & 1) * stride16)[threadIdx.x] = scale_row(nxt);
            __syncthreads();
        }
        vflush<CL>(g0, gp0);
        vflush<CL>(g1, gp1);
        vflush<CL>(l0, lp0);
        vflush<CL>(l1, lp1);

        // ---- epilogue: un-slice the counters of this permutation range (bit-matrix transpose:
        //      rows 0..15 = #greater levels, rows 16..31 = #less levels -> word b = less<<16 | greater
        //      of attribute b) and add them to the totals, laid out [attribute][SELL position] so
        //      that the 64 lanes of a wave update one contiguous 256-byte run
        const bool live = row >= 0;
        const int64_t spos = s * 64 + lane;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t m[32];
#pragma unroll
            for (int l = 0; l < 16; ++l) {
                m[l] = l < CL ? (half ? g1[l < CL ? l : 0] : g0[l < CL ? l : 0]) : 0u;
                m[16 + l] = l < CL ? (half ? l1[l < CL ? l : 0] : l0[l < CL ? l : 0]) : 0u;
            }
            transpose32(m);
#pragma unroll
            for (int bit = 0; bit < 32; ++bit) {
                const int64_t jc = static_cast<int64_t>(wg) * 64 + half * 32 + bit;
                if (jc < mloc && active && m[bit]) atomicAdd(&gl_counts[jc * n_pad + spos], m[bit]);
            }
        }
        if (ns_out && p_begin == 0 && live) {
            const int64_t obase = static_cast<int64_t>(row) * mloc;
#pragma unroll
            for (int half = 0; half < 2; ++half)
                for (int bit = 0; bit < 32; ++bit) {
                    const int64_t jc = static_cast<int64_t>(wg) * 64 + half * 32 + bit;
                    if (jc >= mloc) break;
                    ns_out[obase + jc] = static_cast<double>(half ? vextract<BT_LV>(o1, bit) : vextract<BT_LV>(o0, bit));
                }
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------
// K5 bit-sliced form with PRE-PERMUTED member lists (the default when 8*(N+1) < 65536).
// For every permutation p of a span a small kernel writes ids_p[e] = 8 * cur_p[member e]
// for all SELL entries (u16: the LDS byte offset of the member's word pair).  The main
// kernel then streams those lists from L2/MALL -- coalesced 128-byte loads, one block ahead
// -- and needs a single LDS gather per member; the permutation row never enters LDS, so there
// is no shared per-permutation state, no barrier inside a task, and half the LDS traffic.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_permute_cols(const uint16_t *__restrict__ cur16, int64_t stride16,
                                                      const uint16_t *__restrict__ sell_col2, int64_t entries,
                                                      int64_t entries_pad, int64_t p0, int64_t count, uint32_t pad_off,
                                                      uint16_t *__restrict__ out, int diag_banks = 0, int sh = 3) {
    // sh: ids are written as id << sh -- 3: the LDS byte offset of the member's word pair (8 (N + 1) < 65536); 1: 2 * id for the
    // larger networks of k_permtest_bits_pre<.., 16, 2>, whose byte offsets do not fit 16 bits (the kernel shifts once more)
    // one permutation row (<= 16 KB) staged in LDS per block, 4096 member entries per block: the
    // random 2-byte reads hit LDS instead of L2 sectors
    extern __shared__ uint16_t row[];
    const int64_t q = blockIdx.y;
    if (q >= count) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(cur16 + (p0 + q) * stride16);
    for (int64_t v = threadIdx.x; v < stride16 / 8; v += 256) reinterpret_cast<uint4 *>(row)[v] = src[v];
    __syncthreads();
    // eight entries per thread and trip: one 16-byte load of ids, eight LDS look-ups, one 16-byte store (2-byte global
    // accesses made this kernel as slow as the main one once that got faster: it runs on the few CUs left to it)
    const int64_t e0 = static_cast<int64_t>(blockIdx.x) * 4096;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t e = e0 + (j * 256 + threadIdx.x) * 8;
        if (e >= entries_pad) break;
        uint4 o;
        if (e + 8 <= entries) {
            const uint4 c = *reinterpret_cast<const uint4 *>(sell_col2 + e);
            uint32_t a0 = row[(c.x & 0xFFFFu) >> 1], a1 = row[c.x >> 17], a2 = row[(c.y & 0xFFFFu) >> 1], a3 = row[c.y >> 17];
            uint32_t a4 = row[(c.z & 0xFFFFu) >> 1], a5 = row[c.z >> 17], a6 = row[(c.w & 0xFFFFu) >> 1], a7 = row[c.w >> 17];
            if (diag_banks) {
                // diagnostic (WRONG results): the low five bits of every member's row become the lane's -- in each gather instruction
                // the 32 lanes of a half-wave then hit 32 different bank pairs: what would conflict-free gathers be worth?
                const uint32_t lb = static_cast<uint32_t>((e >> 3) & 31);
                auto fix = [&](uint32_t a) { return (a & ~31u) | lb; };
                a0 = fix(a0), a1 = fix(a1), a2 = fix(a2), a3 = fix(a3);
                a4 = fix(a4), a5 = fix(a5), a6 = fix(a6), a7 = fix(a7);
            }
            o.x = (a0 << sh) | (a1 << (16 + sh));
            o.y = (a2 << sh) | (a3 << (16 + sh));
            o.z = (a4 << sh) | (a5 << (16 + sh));
            o.w = (a6 << sh) | (a7 << (16 + sh));
        } else {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = e + u < entries ? static_cast<uint32_t>(row[sell_col2[e + u] >> 1]) << sh : pad_off;
            o.x = (v[0] & 0xFFFFu) | (v[1] << 16);
            o.y = (v[2] & 0xFFFFu) | (v[3] << 16);
            o.z = (v[4] & 0xFFFFu) | (v[5] << 16);
            o.w = (v[6] & 0xFFFFu) | (v[7] << 16);
        }
        *reinterpret_cast<uint4 *>(out + q * entries_pad + e) = o;
    }
}

// ids: u16 LDS byte offsets (relative to T) of the members, SELL layout; SHIFT = 2 turns the
// resident 2*id list into 8*id (observed pass), 0 takes pre-permuted offsets as they are
template <int SHIFT, int LV = BT_LV>
__device__ __forceinline__ void bits_accumulate_ids(const uint16_t *__restrict__ ids, int wdt, uint32_t t_addr,
                                                    uint32_t (&s0)[LV], uint32_t (&s1)[LV]) {
#pragma unroll
    for (int l = 0; l < LV; ++l) s0[l] = s1[l] = 0;
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = ids[u * 64];
    const uint16_t *pc = ids + 8 * 64;
    for (int t0 = 0; t0 < wdt; t0 += 8, pc += 8 * 64) {
        uint32_t r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = (c[u] << SHIFT) + t_addr;
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = pc[u * 64];                                // next block (lists have a tail)
        uint32_t x0[8], x1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32x2 w = *(lds_u2_ptr)(uintptr_t)(r[u]);
            x0[u] = w.x;
            x1[u] = w.y;
        }
        const uint32_t e0 = vadd8(s0, x0);
        const uint32_t e1 = vadd8(s1, x1);
        if (__builtin_amdgcn_ballot_w64((e0 | e1) != 0)) {
            vripple(s0, e0);
            vripple(s1, e1);
        }
    }
}

// WV = 16, PSHIFT = 2 (round 6): networks of 8191 .. 20 470 nodes -- T alone (8 bytes per node) fills most of a CU's LDS, so ONE
// workgroup of sixteen waves shares it (four waves per SIMD as in the small form), and the permuted lists hold 2 * id (the byte
// offset 8 * id no longer fits 16 bits; the shift that is left costs one operation per member).
// LVS = levels of the vertical sums: BT_LV (neighborhoods below 1024 members); the sixteen-wave form also exists with eleven (below
// 2048: a few hub neighborhoods no longer send a whole call to the scatter or matrix-core kernels).
template <int CL, int WV = 4, int PSHIFT = 0, int LVS = BT_LV>
__global__ __launch_bounds__(64 * WV, WV == 4 ? (CL <= 8 ? 4 : 3) : 1) void k_permtest_bits_pre(
    int64_t n, const uint16_t *__restrict__ ids_p, int64_t entries_pad, const int32_t *__restrict__ sell_row,
    const int64_t *__restrict__ slice_off, const int32_t *__restrict__ slice_width,
    const uint16_t *__restrict__ sell_col2, int64_t n_slices, const uint2 *__restrict__ bbits, int64_t n_tasks,
    const int4 *__restrict__ tasks, int64_t p_base, int64_t p_limit, unsigned int *__restrict__ queue, int64_t mloc,
    unsigned int *__restrict__ gl_counts, int64_t n_pad, double *__restrict__ ns_out) {
    extern __shared__ unsigned int lds[];
    const int64_t t_words = 2 * ((n + 2) & ~int64_t(1));               // T: (n+1) uint2, 16-B padded
    uint2 *T = reinterpret_cast<uint2 *>(lds);
    unsigned int *slot_box = lds + t_words;
    const uint32_t t_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned int *)lds);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

    for (;;) {
        if (threadIdx.x == 0) *slot_box = atomicAdd(queue, 1u);
        __syncthreads();
        const int64_t slot = *slot_box;
        __syncthreads();
        if (slot >= n_tasks) break;
        const int4 task = tasks[slot];
        const int wg = task.x, sg = task.y;
        const int64_t p_begin = p_base + task.z;
        const int64_t p_end = p_base + task.w < p_limit ? p_base + task.w : p_limit;
        if (p_end <= p_begin) continue;                                   // a launch shorter than the task grid's span

        load_word_column<64 * WV>(T, bbits + static_cast<int64_t>(wg) * (n + 1), n);
        const int64_t s = static_cast<int64_t>(sg) * WV + wave;
        const bool active = s < n_slices;
        const int32_t row = active ? sell_row[s * 64 + lane] : -1;
        const int64_t my_off = (active ? slice_off[s] : 0) + lane;
        const int wdt = active ? slice_width[s] : 0;
        __syncthreads();                                                  // T is complete; waves are independent from here

        uint32_t o0[LVS], o1[LVS];                                   // observed sums (safe.py:496-499)
        bits_accumulate_ids<2, LVS>(sell_col2 + my_off, wdt, t_addr, o0, o1);

        uint32_t g0[CL], g1[CL], l0[CL], l1[CL];                          // #(S_p > S_obs), #(S_p < S_obs)
        uint32_t gp0 = 0, gp1 = 0, lp0 = 0, lp1 = 0;
#pragma unroll
        for (int l = 0; l < CL; ++l) g0[l] = g1[l] = l0[l] = l1[l] = 0;

        for (int64_t p = p_begin; p < p_end; ++p) {
            uint32_t s0[LVS], s1[LVS];
            bits_accumulate_ids<PSHIFT, LVS>(ids_p + (p - p_base) * entries_pad + my_off, wdt, t_addr, s0, s1);
            uint32_t gt0 = 0, gt1 = 0, lt0 = 0, lt1 = 0;
#pragma unroll
            for (int l = 0; l < LVS; ++l) {
                // f(s, o, b) = (s != o) ? o : b  -> 0x8E ;  (s != o) ? s : b -> 0xB2
                lt0 = __builtin_amdgcn_bitop3_b32(s0[l], o0[l], lt0, 0x8E);
                gt0 = __builtin_amdgcn_bitop3_b32(s0[l], o0[l], gt0, 0xB2);
                lt1 = __builtin_amdgcn_bitop3_b32(s1[l], o1[l], lt1, 0x8E);
                gt1 = __builtin_amdgcn_bitop3_b32(s1[l], o1[l], gt1, 0xB2);
            }
            vcount<CL>(g0, gp0, gt0);
            vcount<CL>(g1, gp1, gt1);
            vcount<CL>(l0, lp0, lt0);
            vcount<CL>(l1, lp1, lt1);
            if (((p - p_begin) & 7) == 7) {
                vflush<CL>(g0, gp0);
                vflush<CL>(g1, gp1);
                vflush<CL>(l0, lp0);
                vflush<CL>(l1, lp1);
            }
        }
        vflush<CL>(g0, gp0);
        vflush<CL>(g1, gp1);
        vflush<CL>(l0, lp0);
        vflush<CL>(l1, lp1);

        const bool live = row >= 0;
        const int64_t spos = s * 64 + lane;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t m[32];
#pragma unroll
            for (int l = 0; l < 16; ++l) {
                m[l] = l < CL ? (half ? g1[l < CL ? l : 0] : g0[l < CL ? l : 0]) : 0u;
                m[16 + l] = l < CL ? (half ? l1[l < CL ? l : 0] : l0[l < CL ? l : 0]) : 0u;
            }
            transpose32(m);
#pragma unroll
            for (int bit = 0; bit < 32; ++bit) {
                const int64_t jc = static_cast<int64_t>(wg) * 64 + half * 32 + bit;
                if (jc < mloc && active && m[bit]) atomicAdd(&gl_counts[jc * n_pad + spos], m[bit]);
            }
        }
        if (ns_out && p_begin == 0 && live) {
            const int64_t obase = static_cast<int64_t>(row) * mloc;
#pragma unroll
            for (int half = 0; half < 2; ++half)
                for (int bit = 0; bit < 32; ++bit) {
                    const int64_t jc = static_cast<int64_t>(wg) * 64 + half * 32 + bit;
                    if (jc >= mloc) break;
                    ns_out[obase + jc] = static_cast<double>(half ? vextract<LVS>(o1, bit) : vextract<LVS>(o0, bit));
                }
        }
        __syncthreads();                                                  // before T is overwritten by the next task
    }
}

// --------------------------------------------------------------------------------------
// The pre-permuted form for networks of 20 478 .. 32 767 nodes (round 6): the 8-byte word column no longer fits a CU's LDS, a
// column of 32-attribute HALF words (4 bytes per node) does.  A task is (word group, half, sixteen adjacent slices, permutation
// range): task.x = 2 * word group + half; T holds .x or .y of the word pairs; the permuted lists hold the ids themselves
// (k_permute_cols with shift 0: 4 * id does not fit 16 bits), the observed pass reads the resident 2 * id list.  Per 64 attributes
// the id stream and the gathers run twice, the carry-save adds once: ~1.4 x the cost of the full-word form.
// --------------------------------------------------------------------------------------
typedef const __attribute__((address_space(3))) unsigned int *lds_u32_ptr;

template <int SHIFT, int LV>
__device__ __forceinline__ void bits_accumulate_ids32(const uint16_t *__restrict__ ids, int wdt, uint32_t t_addr, uint32_t (&s0)[LV]) {
#pragma unroll
    for (int l = 0; l < LV; ++l) s0[l] = 0;
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = ids[u * 64];
    const uint16_t *pc = ids + 8 * 64;
    for (int t0 = 0; t0 < wdt; t0 += 8, pc += 8 * 64) {
        uint32_t r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = (c[u] << SHIFT) + t_addr;
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = pc[u * 64];                                // next block (lists have a tail)
        uint32_t x0[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x0[u] = *(lds_u32_ptr)(uintptr_t)(r[u]);
        const uint32_t e0 = vadd8(s0, x0);
        if (__builtin_amdgcn_ballot_w64(e0 != 0)) vripple(s0, e0);
    }
}

template <int CL, int WV, int LVS>
__global__ __launch_bounds__(64 * WV, 1) void k_permtest_bits_pre32(
    int64_t n, const uint16_t *__restrict__ ids_p, int64_t entries_pad, const int32_t *__restrict__ sell_row,
    const int64_t *__restrict__ slice_off, const int32_t *__restrict__ slice_width,
    const uint16_t *__restrict__ sell_col2, int64_t n_slices, const uint2 *__restrict__ bbits, int64_t n_tasks,
    const int4 *__restrict__ tasks, int64_t p_base, int64_t p_limit, unsigned int *__restrict__ queue, int64_t mloc,
    unsigned int *__restrict__ gl_counts, int64_t n_pad, double *__restrict__ ns_out) {
    extern __shared__ unsigned int lds[];
    const int64_t t_words = (n + 4) & ~int64_t(3);                      // T: (n+1) half words, 16-B padded
    unsigned int *T = lds;
    unsigned int *slot_box = lds + t_words;
    const uint32_t t_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned int *)lds);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    constexpr int NT = 64 * WV;

    for (;;) {
        if (threadIdx.x == 0) *slot_box = atomicAdd(queue, 1u);
        __syncthreads();
        const int64_t slot = *slot_box;
        __syncthreads();
        if (slot >= n_tasks) break;
        const int4 task = tasks[slot];
        const int wg = task.x >> 1, half = task.x & 1, sg = task.y;
        const int64_t p_begin = p_base + task.z;
        const int64_t p_end = p_base + task.w < p_limit ? p_base + task.w : p_limit;
        if (p_end <= p_begin) continue;                                   // a launch shorter than the task grid's span

        {
            const uint2 *src = bbits + static_cast<int64_t>(wg) * (n + 1);
            for (int64_t r = threadIdx.x; r <= n; r += NT) T[r] = half ? src[r].y : src[r].x;
        }
        const int64_t s = static_cast<int64_t>(sg) * WV + wave;
        const bool active = s < n_slices;
        const int32_t row = active ? sell_row[s * 64 + lane] : -1;
        const int64_t my_off = (active ? slice_off[s] : 0) + lane;
        const int wdt = active ? slice_width[s] : 0;
        __syncthreads();                                                  // T is complete; waves are independent from here

        uint32_t o0[LVS];                                                // observed sums (safe.py:496-499)
        bits_accumulate_ids32<1, LVS>(sell_col2 + my_off, wdt, t_addr, o0);           // 2 * id -> 4 * id

        uint32_t g0[CL], l0[CL];                                         // #(S_p > S_obs), #(S_p < S_obs)
        uint32_t gp0 = 0, lp0 = 0;
#pragma unroll
        for (int l = 0; l < CL; ++l) g0[l] = l0[l] = 0;

        for (int64_t p = p_begin; p < p_end; ++p) {
            uint32_t s0[LVS];
            bits_accumulate_ids32<2, LVS>(ids_p + (p - p_base) * entries_pad + my_off, wdt, t_addr, s0);   // id -> 4 * id
            uint32_t gt0 = 0, lt0 = 0;
#pragma unroll
            for (int l = 0; l < LVS; ++l) {
                lt0 = __builtin_amdgcn_bitop3_b32(s0[l], o0[l], lt0, 0x8E);
                gt0 = __builtin_amdgcn_bitop3_b32(s0[l], o0[l], gt0, 0xB2);
            }
            vcount<CL>(g0, gp0, gt0);
            vcount<CL>(l0, lp0, lt0);
            if (((p - p_begin) & 7) == 7) {
                vflush<CL>(g0, gp0);
                vflush<CL>(l0, lp0);
            }
        }
        vflush<CL>(g0, gp0);
        vflush<CL>(l0, lp0);

        const bool live = row >= 0;
        const int64_t spos = s * 64 + lane;
        {
            uint32_t m[32];
#pragma unroll
            for (int l = 0; l < 16; ++l) {
                m[l] = l < CL ? g0[l < CL ? l : 0] : 0u;
                m[16 + l] = l < CL ? l0[l < CL ? l : 0] : 0u;
            }
            transpose32(m);
#pragma unroll
            for (int bit = 0; bit < 32; ++bit) {
                const int64_t jc = static_cast<int64_t>(wg) * 64 + half * 32 + bit;
                if (jc < mloc && active && m[bit]) atomicAdd(&gl_counts[jc * n_pad + spos], m[bit]);
            }
        }
        if (ns_out && p_begin == 0 && live) {
            const int64_t obase = static_cast<int64_t>(row) * mloc;
            for (int bit = 0; bit < 32; ++bit) {
                const int64_t jc = static_cast<int64_t>(wg) * 64 + half * 32 + bit;
                if (jc >= mloc) break;
                ns_out[obase + jc] = static_cast<double>(vextract<LVS>(o0, bit));
            }
        }
        __syncthreads();                                                  // before T is overwritten by the next task
    }
}

// --------------------------------------------------------------------------------------
// K5 bit-sliced form, BLOCKED member lists (the default when 8*(N+1) < 65536).  Same arithmetic as
// k_permtest_bits_pre; what changed is how a lane gets at its members and how much it carries:
//   * ids of 8 members are adjacent (k_sell_blocked16 / k_permute_cols on the blocked list): ONE 16-byte
//     load per lane and block instead of eight 2-byte loads -- an eighth of the vector-memory instructions,
//     four id registers instead of eight, no address adds (the u16 IS the LDS address: T sits at LDS
//     address 0, checked once);
//   * the number of levels of the vertical sums follows the slice: a neighborhood of w members cannot sum
//     past w, so a wave whose slice is <= 8 / 56 / 248 members wide runs with 4 / 6 / 8 levels instead of 10
//     (rows are sorted by size: most slices are narrow) -- shorter compare chains, fewer live registers;
//   * DBG (diagnostics, tools/bits_ablate.py): bit 0 skips the LDS gathers, bit 1 the counter flush,
//     bit 2 the compare / count step, bit 6 flushes with plain stores instead of atomics -- wrong results, used to see what the
//     time goes to.  (Round 4: the no-flush build is 9 % faster, the plain-store build is not: with the flush gone the compiler also
//     drops the upper counter levels and their ripples as dead code -- the atomics themselves cost nothing measurable.)
// --------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int LV>
__device__ __forceinline__ uint32_t vadd8_lv(uint32_t (&s)[LV], const uint32_t (&x)[8]) {
    uint32_t t2a = maj3(s[0], x[0], x[1]);
    s[0] = xor3(s[0], x[0], x[1]);
    uint32_t t2b = maj3(s[0], x[2], x[3]);
    s[0] = xor3(s[0], x[2], x[3]);
    const uint32_t t4a = maj3(s[1], t2a, t2b);
    s[1] = xor3(s[1], t2a, t2b);
    t2a = maj3(s[0], x[4], x[5]);
    s[0] = xor3(s[0], x[4], x[5]);
    t2b = maj3(s[0], x[6], x[7]);
    s[0] = xor3(s[0], x[6], x[7]);
    const uint32_t t4b = maj3(s[1], t2a, t2b);
    s[1] = xor3(s[1], t2a, t2b);
    const uint32_t t8 = maj3(s[2], t4a, t4b);
    s[2] = xor3(s[2], t4a, t4b);
    return t8;
}

// (the two operations of a level are pinned in place: left to itself the compiler turns the chain into a parallel-prefix
// form on fresh registers and moves all levels back afterwards -- 16 moves around 14 operations on a path taken in about a
// third of the blocks)
template <int LV>
__device__ __forceinline__ void vripple_lv(uint32_t (&s)[LV], uint32_t t8) {
#pragma unroll
    for (int l = 3; l < LV; ++l) {
        uint32_t c;
        asm("v_and_b32 %0, %1, %2\n\tv_xor_b32 %1, %1, %2" : "=&v"(c), "+v"(s[l]) : "v"(t8));
        t8 = c;
    }
}

// The same in two stages (classes of eight and more levels): a carry into the eights almost never travels past the thirty-twos on
// sparse annotations, so levels 3 and 4 are rippled first and the rest only when some lane still carries.
template <int LV>
__device__ __forceinline__ uint32_t vripple_low(uint32_t (&s)[LV], uint32_t t8) {
#pragma unroll
    for (int l = 3; l < 5 && l < LV; ++l) {
        uint32_t c;
        asm("v_and_b32 %0, %1, %2\n\tv_xor_b32 %1, %1, %2" : "=&v"(c), "+v"(s[l]) : "v"(t8));
        t8 = c;
    }
    return t8;
}
template <int LV>
__device__ __forceinline__ void vripple_high(uint32_t (&s)[LV], uint32_t t32) {
#pragma unroll
    for (int l = 5; l < LV; ++l) {
        uint32_t c;
        asm("v_and_b32 %0, %1, %2\n\tv_xor_b32 %1, %1, %2" : "=&v"(c), "+v"(s[l]) : "v"(t32));
        t32 = c;
    }
}

// ids: this lane's blocks (uint4 = 8 x u16), 64 uint4 apart; SHIFT = 2 turns the resident 2*id list into 8*id.
// The id words are fetched TWO blocks ahead into two register quads that swap roles (no copies): a block's adds take
// 100-200 cycles, a load from L2 / MALL several hundred -- one block of look-ahead left every wave waiting at vmcnt(0)
// at the top of each block (the lists have a tail of two blocks, so the look-ahead never leaves the buffer).
// GATHER: 1 = the real thing; 0 = no LDS reads (diagnostic); 2 = LDS reads at conflict-free addresses (diagnostic: bits 3..7 of
// every address replaced by the lane's number mod 32 -- the same number of gathers, every bank pair used once per lane group)
template <int LV, int SHIFT, int GATHER, bool RIP2 = false>
__device__ __forceinline__ void blk_add8(u32x4 &c, const u32x4 *__restrict__ refill, uint32_t (&s0)[LV], uint32_t (&s1)[LV]) {
    uint32_t a[8];
    a[0] = (c.x & 0xFFFFu) << SHIFT;
    a[1] = (c.x >> 16) << SHIFT;
    a[2] = (c.y & 0xFFFFu) << SHIFT;
    a[3] = (c.y >> 16) << SHIFT;
    a[4] = (c.z & 0xFFFFu) << SHIFT;
    a[5] = (c.z >> 16) << SHIFT;
    a[6] = (c.w & 0xFFFFu) << SHIFT;
    a[7] = (c.w >> 16) << SHIFT;
    c = *refill;                                                                          // the block after next
    if (GATHER == 2) {
        const uint32_t lane_bits = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 31u) << 3;
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = (a[u] & 0xFF00u) | lane_bits;
    }
    uint32_t x0[8], x1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (GATHER) {
            const u32x2 w = *(lds_u2_ptr)(uintptr_t)(a[u]);
            x0[u] = w.x;
            x1[u] = w.y;
        } else {
            x0[u] = a[u];
            x1[u] = a[u] >> 3;
        }
    }
    const uint32_t e0 = vadd8_lv<LV>(s0, x0);
    const uint32_t e1 = vadd8_lv<LV>(s1, x1);
    if (__builtin_amdgcn_ballot_w64((e0 | e1) != 0)) {
        if constexpr (RIP2 && LV >= 8) {
            const uint32_t c0 = vripple_low<LV>(s0, e0), c1 = vripple_low<LV>(s1, e1);
            if (__builtin_amdgcn_ballot_w64((c0 | c1) != 0)) {
                vripple_high<LV>(s0, c0);
                vripple_high<LV>(s1, c1);
            }
        } else {
            vripple_lv<LV>(s0, e0);
            vripple_lv<LV>(s1, e1);
        }
    }
}

// (the clobber lists below name registers that are RESERVED in the stream kernels -- that is the point: see the next comment; the
// check that the compiler keeps out of them is tests/test_abi.py's disassembly of the shipped code object)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// The stream form of blk_add8 (k_permtest_bits_blk's permutation loop).  The two id quads live in v[112:115] (Q = 0) and
// v[116:119] (Q = 1), registers the compiler does not know about: the kernel is built with 112 registers (amdgpu_num_vgpr) and the
// clobber lists below make the allocation 120 (not 128: at 4 x 120 registers a SIMD keeps room for a wave of the table kernels of
// the next pipeline stage next to this kernel's four; with 4 x 128 they queued behind it and the seeded step got slower).  A quad is refilled IN PLACE with the block after next right after its ids were
// extracted, and the wait is ours: the quads are fetched in block order, so when block g is due exactly one younger fetch (block
// g + 1's) may still be in flight -- vmcnt(1); whatever else the compiler has in flight only makes that wait stricter (vmcnt
// retires in order).  As C++ values (`c = *refill`, blk_add8) the compiler loaded the refill into fresh registers and copied them
// over at the loop latch behind `s_waitcnt vmcnt(0)`: every second block waited for ids requested one block earlier -- an L2 round
// trip against a block's 160-600 clocks -- and a quad written by an asm statement is no better: the allocator copies it around
// BEFORE the wait.  `base` is wave-uniform (scalar registers), `lane_off` the block's byte offset plus the lane's.
template <int Q>
__device__ __forceinline__ void stream_fetch(const u32x4 *base, uint32_t lane_off) {
    if (Q == 0) asm volatile("global_load_dwordx4 v[112:115], %0, %1" : : "v"(lane_off), "s"(base) : "v112", "v113", "v114", "v115");
    else asm volatile("global_load_dwordx4 v[116:119], %0, %1" : : "v"(lane_off), "s"(base) : "v116", "v117", "v118", "v119");
}

template <int LV, int GATHER, bool RIP2, int Q>
__device__ __forceinline__ void blk_add8s(const u32x4 *base, uint32_t lane_off, uint32_t (&s0)[LV], uint32_t (&s1)[LV]) {
    uint32_t a[8];
    if (Q == 0)
        asm volatile("s_waitcnt vmcnt(1)\n\tv_and_b32 %0, 0xffff, v112\n\tv_lshrrev_b32 %1, 16, v112\n\tv_and_b32 %2, 0xffff, v113\n\t"
                     "v_lshrrev_b32 %3, 16, v113\n\tv_and_b32 %4, 0xffff, v114\n\tv_lshrrev_b32 %5, 16, v114\n\t"
                     "v_and_b32 %6, 0xffff, v115\n\tv_lshrrev_b32 %7, 16, v115"
                     : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7]));
    else
        asm volatile("s_waitcnt vmcnt(1)\n\tv_and_b32 %0, 0xffff, v116\n\tv_lshrrev_b32 %1, 16, v116\n\tv_and_b32 %2, 0xffff, v117\n\t"
                     "v_lshrrev_b32 %3, 16, v117\n\tv_and_b32 %4, 0xffff, v118\n\tv_lshrrev_b32 %5, 16, v118\n\t"
                     "v_and_b32 %6, 0xffff, v119\n\tv_lshrrev_b32 %7, 16, v119"
                     : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7]));
    stream_fetch<Q>(base, lane_off);
    if (GATHER == 2) {
        const uint32_t lane_bits = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 31u) << 3;
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = (a[u] & 0xFF00u) | lane_bits;
    }
    uint32_t x0[8], x1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (GATHER) {
            const u32x2 w = *(lds_u2_ptr)(uintptr_t)(a[u]);
            x0[u] = w.x;
            x1[u] = w.y;
        } else {
            x0[u] = a[u];
            x1[u] = a[u] >> 3;
        }
    }
    const uint32_t e0 = vadd8_lv<LV>(s0, x0);
    const uint32_t e1 = vadd8_lv<LV>(s1, x1);
    if (__builtin_amdgcn_ballot_w64((e0 | e1) != 0)) {
        if constexpr (RIP2 && LV >= 8) {
            const uint32_t c0 = vripple_low<LV>(s0, e0), c1 = vripple_low<LV>(s1, e1);
            if (__builtin_amdgcn_ballot_w64((c0 | c1) != 0)) {
                vripple_high<LV>(s0, c0);
                vripple_high<LV>(s1, c1);
            }
        } else {
            vripple_lv<LV>(s0, e0);
            vripple_lv<LV>(s1, e1);
        }
    }
}

// The stream's block in two halves, the LDS gathers ONE HALF AHEAD of the adds: when the adds of a block's first four members
// run the gathers of its last four are in flight, and the gathers of the NEXT block's first four (X0 / X1, carried into the next
// call) are issued before the last four are added -- a wave no longer sits out a whole LDS round trip per block (the diagnostic
// build without gathers ran as fast with eight more vector operations per block: that round trip was what the gathers cost).
// Q = the quad of this block; the next block's is the other one, waited for (vmcnt(1), as in blk_add8s) when its first half is due.
constexpr int HALF_MAX_LV = 8;            // (the classes of nine and ten levels have no registers for the second set of gathered words)
template <int Q>
__device__ __forceinline__ void stream_ids_lo(uint32_t (&a)[4]) {             // ids 0..3 of quad Q, after waiting for its fetch
    if (Q == 0)
        asm volatile("s_waitcnt vmcnt(1)\n\tv_and_b32 %0, 0xffff, v112\n\tv_lshrrev_b32 %1, 16, v112\n\tv_and_b32 %2, 0xffff, v113\n\t"
                     "v_lshrrev_b32 %3, 16, v113" : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]));
    else
        asm volatile("s_waitcnt vmcnt(1)\n\tv_and_b32 %0, 0xffff, v116\n\tv_lshrrev_b32 %1, 16, v116\n\tv_and_b32 %2, 0xffff, v117\n\t"
                     "v_lshrrev_b32 %3, 16, v117" : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]));
}
template <int Q>
__device__ __forceinline__ void stream_ids_hi(uint32_t (&a)[4]) {             // ids 4..7 (the quad has landed: its first half was used)
    if (Q == 0)
        asm volatile("v_and_b32 %0, 0xffff, v114\n\tv_lshrrev_b32 %1, 16, v114\n\tv_and_b32 %2, 0xffff, v115\n\t"
                     "v_lshrrev_b32 %3, 16, v115" : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]));
    else
        asm volatile("v_and_b32 %0, 0xffff, v118\n\tv_lshrrev_b32 %1, 16, v118\n\tv_and_b32 %2, 0xffff, v119\n\t"
                     "v_lshrrev_b32 %3, 16, v119" : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]));
}
template <int GATHER>
__device__ __forceinline__ void gather4(uint32_t (&a)[4], uint32_t (&x0)[4], uint32_t (&x1)[4]) {
    if (GATHER == 2) {
        const uint32_t lane_bits = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 31u) << 3;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = (a[u] & 0xFF00u) | lane_bits;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (GATHER) {
            const u32x2 w = *(lds_u2_ptr)(uintptr_t)(a[u]);
            x0[u] = w.x;
            x1[u] = w.y;
        } else {
            x0[u] = a[u];
            x1[u] = a[u] >> 3;
        }
    }
}
// four members into the sums: the carry into the fours comes back
template <int LV>
__device__ __forceinline__ uint32_t vadd4_lv(uint32_t (&s)[LV], const uint32_t (&x)[4]) {
    const uint32_t t2a = maj3(s[0], x[0], x[1]);
    s[0] = xor3(s[0], x[0], x[1]);
    const uint32_t t2b = maj3(s[0], x[2], x[3]);
    s[0] = xor3(s[0], x[2], x[3]);
    const uint32_t t4 = maj3(s[1], t2a, t2b);
    s[1] = xor3(s[1], t2a, t2b);
    return t4;
}
template <int LV, int GATHER, bool RIP2, int Q>
__device__ __forceinline__ void blk_step(const u32x4 *base, uint32_t lane_off, uint32_t (&s0)[LV], uint32_t (&s1)[LV], uint32_t (&X0)[4],
                                         uint32_t (&X1)[4]) {
    uint32_t a[4], Y0[4], Y1[4];
    stream_ids_hi<Q>(a);
    stream_fetch<Q>(base, lane_off);                                         // the block after next, into this block's quad
    gather4<GATHER>(a, Y0, Y1);
    const uint32_t f0 = vadd4_lv<LV>(s0, X0), f1 = vadd4_lv<LV>(s1, X1);
    stream_ids_lo<1 - Q>(a);
    gather4<GATHER>(a, X0, X1);                                              // the next block's first half
    const uint32_t h0 = vadd4_lv<LV>(s0, Y0), h1 = vadd4_lv<LV>(s1, Y1);
    const uint32_t e0 = maj3(s0[2], f0, h0), e1 = maj3(s1[2], f1, h1);
    s0[2] = xor3(s0[2], f0, h0);
    s1[2] = xor3(s1[2], f1, h1);
    if (__builtin_amdgcn_ballot_w64((e0 | e1) != 0)) {
        if constexpr (RIP2 && LV >= 8) {
            const uint32_t c0 = vripple_low<LV>(s0, e0), c1 = vripple_low<LV>(s1, e1);
            if (__builtin_amdgcn_ballot_w64((c0 | c1) != 0)) {
                vripple_high<LV>(s0, c0);
                vripple_high<LV>(s1, c1);
            }
        } else {
            vripple_lv<LV>(s0, e0);
            vripple_lv<LV>(s1, e1);
        }
    }
}

template <int LV, int SHIFT, int GATHER, bool RIP2 = false>
__device__ __forceinline__ void blk_sum(const u32x4 *__restrict__ ids, int lane, int nblk, uint32_t (&s0)[LV], uint32_t (&s1)[LV]) {
    // `ids` is the slice's first block, the same for the whole wave (scalar registers); the lane is the offset
#pragma unroll
    for (int l = 0; l < LV; ++l) s0[l] = s1[l] = 0;
    u32x4 ca = ids[lane], cb = ids[64 + lane];
    const u32x4 *pc = ids + 128 + lane;
    int b = 0;
    constexpr int STEP = GATHER == 3 ? 0 : 128;            // GATHER 3 (diagnostic): every id load re-reads the slice's first blocks (L1 hits)
    for (; b + 1 < nblk; b += 2, pc += STEP) {
        blk_add8<LV, SHIFT, GATHER, RIP2>(ca, pc, s0, s1);
        blk_add8<LV, SHIFT, GATHER, RIP2>(cb, pc + 64, s0, s1);
    }
    if (b < nblk) blk_add8<LV, SHIFT, GATHER, RIP2>(ca, pc, s0, s1);
}

// one wave, one task: observed sums, then every permutation of the task's range; counters come back in g / l
// (levels above CL stay zero), the observed sums in oo (levels above LV zero)
// CLT <= CL: levels the task really counts with (a task of at most 2^CLT - 1 permutations); OBSMEM: the observed sums are
// re-read from memory (L1 / L2 resident: 2 * LV coalesced 256-byte loads per permutation) in the compare step instead of
// being held in 2 * LV registers for the whole task -- the wide classes trade them for a fifth wave per SIMD.
template <int LV, int CLT, int DBG, bool OBSMEM>
__device__ __forceinline__ void blk_task_core(const uint32_t *__restrict__ obs, const u32x4 *__restrict__ perm_ids, int64_t perm_stride,
                                              int lane, int nblk, int np, uint32_t (&g0)[CLT], uint32_t (&g1)[CLT], uint32_t (&l0)[CLT],
                                              uint32_t (&l1)[CLT]) {
    constexpr int GATHER = (DBG & 1) ? 0 : (DBG & 8) ? 2 : (DBG & 16) ? 3 : 1;
    // observed sums of this (word group, slice): computed once per call by k_bits_observed, vertical like the permuted sums
    uint32_t o0[OBSMEM ? 1 : LV], o1[OBSMEM ? 1 : LV];
    if (!OBSMEM) {
#pragma unroll
        for (int l = 0; l < LV; ++l) {
            o0[OBSMEM ? 0 : l] = obs[l * 64 + lane];
            o1[OBSMEM ? 0 : l] = obs[(BT_LV + l) * 64 + lane];
        }
    }
    uint32_t gp0 = 0, gp1 = 0, lp0 = 0, lp1 = 0;
    constexpr bool STREAM = (DBG & 256) == 0;                         // (the five-waves build passes bit 8: it has no registers 112-119)
    // STREAM (the default): the task's blocks -- permutation after permutation -- are ONE stream through two id quads that are
    // refilled IN PLACE with the block after next (blk_add8s), so a permutation's first blocks were requested while the previous
    // permutation was still being added.  !STREAM (SAFE_HIP_BITS_DBG=256, the form of rounds 2-4, same results): every permutation
    // starts its own look-ahead (blk_sum) and waits for its first block's ids -- an L2 round trip per permutation and slice.
    u32x4 ca{}, cb{};
    const bool single = !STREAM && LV == 4 && nblk == 1 && GATHER != 3;
    if (single) {
        ca = perm_ids[lane];
        cb = perm_ids[(np > 1 ? perm_stride : 0) + lane];
    }
    // the stream's fetch position, two blocks ahead of the block being added: a wave-uniform byte offset from the task's first
    // block (scalar registers; the id lists of one launch are far below 4 GB), added to the lane's own offset per fetch
    const uint32_t pf_hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(reinterpret_cast<uint64_t>(perm_ids) >> 32)));
    const uint32_t pf_lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(reinterpret_cast<uint64_t>(perm_ids))));
    const u32x4 *pf_base = reinterpret_cast<const u32x4 *>((static_cast<uint64_t>(pf_hi) << 32) | static_cast<uint64_t>(pf_lo));
    uint32_t pf = 0;
    int pf_left = nblk, pf_perms = np;                                      // blocks left in the fetch position's permutation, permutations left
    const uint32_t pf_stride = static_cast<uint32_t>(perm_stride) * 16u, pf_wrap = pf_stride - static_cast<uint32_t>(nblk) * 1024u;
    const uint32_t lane_off = static_cast<uint32_t>(lane) * 16u;
    // (the wrap is a real branch -- the empty asm statement keeps the compiler from turning it into a dozen scalar selects per block)
    auto pf_next = [&]() __attribute__((always_inline)) {
        if (GATHER == 3) return;
        pf += 1024u;
        if (__builtin_expect(--pf_left == 0, 0)) {
            asm volatile("");
            pf_left = nblk;
            pf += pf_wrap;
            if (--pf_perms == 0) {                                           // past the task's end: the last permutation's blocks again (never used)
                pf_perms = 1;
                pf -= pf_stride;
            }
        }
    };
    // what follows a permutation's sums: compare with the observed sums, count
    auto settle = [&](int p, uint32_t (&s0)[LV], uint32_t (&s1)[LV]) __attribute__((always_inline)) {
        if (DBG & 4) {
            g0[0] ^= s0[0] ^ s0[LV - 1];
            g1[0] ^= s1[0] ^ s1[LV - 1];
            return;
        }
        uint32_t gt0 = 0, gt1 = 0, lt0 = 0, lt1 = 0;
#pragma unroll
        for (int l = 0; l < LV; ++l) {
            const uint32_t a0 = OBSMEM ? obs[l * 64 + lane] : o0[OBSMEM ? 0 : l];
            const uint32_t a1 = OBSMEM ? obs[(BT_LV + l) * 64 + lane] : o1[OBSMEM ? 0 : l];
            // f(s, o, b) = (s != o) ? o : b  -> 0x8E ;  (s != o) ? s : b -> 0xB2
            lt0 = __builtin_amdgcn_bitop3_b32(s0[l], a0, lt0, 0x8E);
            gt0 = __builtin_amdgcn_bitop3_b32(s0[l], a0, gt0, 0xB2);
            lt1 = __builtin_amdgcn_bitop3_b32(s1[l], a1, lt1, 0x8E);
            gt1 = __builtin_amdgcn_bitop3_b32(s1[l], a1, gt1, 0xB2);
        }
        vcount<CLT>(g0, gp0, gt0);
        vcount<CLT>(g1, gp1, gt1);
        vcount<CLT>(l0, lp0, lt0);
        vcount<CLT>(l1, lp1, lt1);
        if ((p & 7) == 7) {
            vflush<CLT>(g0, gp0);
            vflush<CLT>(g1, gp1);
            vflush<CLT>(l0, lp0);
            vflush<CLT>(l1, lp1);
        }
    };
    if (STREAM) {
        // Blocks alternate between the quads (block g of the stream sits in quad g & 1), in straight-line code: an `if (quad)` around
        // one block body made the two arms keep the sums in different registers, ten moves per block.  A slice of an even number of
        // blocks is pairs all the way; with an odd number two permutations make one period: pairs, A | B, pairs.
        constexpr bool R2 = (DBG & 32) == 0;
        constexpr bool HALF = (DBG & 512) == 0 && LV <= HALF_MAX_LV;        // (bit 9: the gathers of a block all at its start, as before)
        stream_fetch<0>(pf_base, pf + lane_off);
        pf_next();
        stream_fetch<1>(pf_base, pf + lane_off);
        pf_next();
        uint32_t X0[4] = {0, 0, 0, 0}, X1[4] = {0, 0, 0, 0};
        if (HALF) {
            uint32_t a[4];
            stream_ids_lo<0>(a);
            gather4<GATHER>(a, X0, X1);
        }
        auto blockA = [&](uint32_t (&s0)[LV], uint32_t (&s1)[LV]) __attribute__((always_inline)) {
            if (HALF) blk_step<LV, GATHER, R2, 0>(pf_base, pf + lane_off, s0, s1, X0, X1);
            else blk_add8s<LV, GATHER, R2, 0>(pf_base, pf + lane_off, s0, s1);
            pf_next();
        };
        auto blockB = [&](uint32_t (&s0)[LV], uint32_t (&s1)[LV]) __attribute__((always_inline)) {
            if (HALF) blk_step<LV, GATHER, R2, 1>(pf_base, pf + lane_off, s0, s1, X0, X1);
            else blk_add8s<LV, GATHER, R2, 1>(pf_base, pf + lane_off, s0, s1);
            pf_next();
        };
        const int pairs = nblk >> 1;
        auto add_pairs = [&](uint32_t (&s0)[LV], uint32_t (&s1)[LV]) __attribute__((always_inline)) {
            for (int i = 0; i < pairs; ++i) {
                blockA(s0, s1);
                blockB(s0, s1);
            }
        };
        if (nblk & 1) {
            int p = 0;
            for (; p + 1 < np; p += 2) {
                uint32_t s0[LV], s1[LV];
#pragma unroll
                for (int l = 0; l < LV; ++l) s0[l] = s1[l] = 0;
                add_pairs(s0, s1);
                blockA(s0, s1);
                settle(p, s0, s1);
#pragma unroll
                for (int l = 0; l < LV; ++l) s0[l] = s1[l] = 0;
                blockB(s0, s1);
                add_pairs(s0, s1);
                settle(p + 1, s0, s1);
            }
            if (p < np) {
                uint32_t s0[LV], s1[LV];
#pragma unroll
                for (int l = 0; l < LV; ++l) s0[l] = s1[l] = 0;
                add_pairs(s0, s1);
                blockA(s0, s1);
                settle(p, s0, s1);
            }
        } else {
            for (int p = 0; p < np; ++p) {
                uint32_t s0[LV], s1[LV];
#pragma unroll
                for (int l = 0; l < LV; ++l) s0[l] = s1[l] = 0;
                add_pairs(s0, s1);
                settle(p, s0, s1);
            }
        }
    } else {
        for (int p = 0; p < np; ++p, perm_ids += perm_stride) {
            uint32_t s0[LV], s1[LV];
            if (single) {
#pragma unroll
                for (int l = 0; l < LV; ++l) s0[l] = s1[l] = 0;
                const u32x4 *ahead = perm_ids + (p + 2 < np ? 2 * perm_stride : 0) + lane;
                if (p & 1) blk_add8<LV, 0, GATHER>(cb, ahead, s0, s1);
                else blk_add8<LV, 0, GATHER>(ca, ahead, s0, s1);
            } else
                blk_sum<LV, 0, GATHER, (DBG & 32) == 0>(perm_ids, lane, nblk, s0, s1);
            settle(p, s0, s1);
        }
    }
    vflush<CLT>(g0, gp0);
    vflush<CLT>(g1, gp1);
    vflush<CLT>(l0, lp0);
    vflush<CLT>(l1, lp1);
    // (the two fetches past the stream's end: landed before anything else is asked of the vector-memory counter)
    if (STREAM) asm volatile("s_waitcnt vmcnt(0)" : : : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
}
#pragma clang diagnostic pop

template <int LV, int CL, int DBG, int CLT = CL, bool OBSMEM = false>
__device__ __forceinline__ void blk_task(const uint32_t *__restrict__ obs, const u32x4 *__restrict__ perm_ids, int64_t perm_stride,
                                         int lane, int nblk, int np, uint32_t (&G0)[CL], uint32_t (&G1)[CL], uint32_t (&L0)[CL],
                                         uint32_t (&L1)[CL]) {
    if constexpr (CLT == CL) {
        blk_task_core<LV, CL, DBG, OBSMEM>(obs, perm_ids, perm_stride, lane, nblk, np, G0, G1, L0, L1);      // (zeroed by the caller)
    } else {
        uint32_t g0[CLT], g1[CLT], l0[CLT], l1[CLT];
#pragma unroll
        for (int l = 0; l < CLT; ++l) g0[l] = g1[l] = l0[l] = l1[l] = 0;
        blk_task_core<LV, CLT, DBG, OBSMEM>(obs, perm_ids, perm_stride, lane, nblk, np, g0, g1, l0, l1);
#pragma unroll
        for (int l = 0; l < CLT; ++l) {
            G0[l] = g0[l];
            G1[l] = g1[l];
            L0[l] = l0[l];
            L1[l] = l1[l];
        }
    }
}

// Observed neighborhood sums of every (word group, slice), once per call (safe.py:496-499): vertical counters
// obs[word group][slice][2 x BT_LV levels][64 lanes] for the permutation kernels' compare step, and the scores
// themselves (`ns`).  One workgroup = one word group x four adjacent slices.
template <int LV>
__device__ __forceinline__ void observed_wave(const u32x4 *__restrict__ ids, int lane, int nblk, uint32_t *__restrict__ obs,
                                              uint32_t (&oo0)[BT_LV], uint32_t (&oo1)[BT_LV]) {
    uint32_t o0[LV], o1[LV];
    blk_sum<LV, 2, 1>(ids, lane, nblk, o0, o1);
#pragma unroll
    for (int l = 0; l < BT_LV; ++l) {
        oo0[l] = l < LV ? o0[l < LV ? l : 0] : 0u;
        oo1[l] = l < LV ? o1[l < LV ? l : 0] : 0u;
        obs[l * 64 + lane] = oo0[l];
        obs[(BT_LV + l) * 64 + lane] = oo1[l];
    }
}

__global__ __launch_bounds__(256) void k_bits_observed(int64_t n, const int32_t *__restrict__ sell_row,
                                                       const int64_t *__restrict__ slice_off,
                                                       const int32_t *__restrict__ slice_width,
                                                       const uint16_t *__restrict__ sell_col2b, int64_t n_slices,
                                                       const uint2 *__restrict__ bbits, int64_t mloc,
                                                       uint32_t *__restrict__ obs, double *__restrict__ ns_out) {
    extern __shared__ unsigned int lds[];
    uint2 *T = reinterpret_cast<uint2 *>(lds);
    if ((uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned int *)lds) != 0u) __builtin_trap();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t wg = blockIdx.x, s = static_cast<int64_t>(blockIdx.y) * 4 + wave;
    load_word_column(T, bbits + wg * (n + 1), n);
    __syncthreads();
    if (s >= n_slices) return;
    const int wdt = __builtin_amdgcn_readfirstlane(slice_width[s]);
    const u32x4 *ids = reinterpret_cast<const u32x4 *>(sell_col2b) + slice_off[s] / 8;
    uint32_t *my_obs = obs + (wg * n_slices + s) * (2 * BT_LV * 64);
    uint32_t oo0[BT_LV], oo1[BT_LV];
    if (wdt <= 8) observed_wave<4>(ids, lane, wdt >> 3, my_obs, oo0, oo1);
    else if (wdt <= 56) observed_wave<6>(ids, lane, wdt >> 3, my_obs, oo0, oo1);
    else if (wdt <= 248) observed_wave<8>(ids, lane, wdt >> 3, my_obs, oo0, oo1);
    else if (wdt <= 504) observed_wave<9>(ids, lane, wdt >> 3, my_obs, oo0, oo1);
    else observed_wave<BT_LV>(ids, lane, wdt >> 3, my_obs, oo0, oo1);
    if (ns_out) {
        // ns (safe.py:514-515) is [node][attribute]: the lanes hold one row each, so the 64 x 64 values of the wave go through
        // LDS (u16, rows 33 dwords apart: no bank conflicts either way) and leave as 512 contiguous bytes of one row per store
        const int32_t row = sell_row[s * 64 + lane];
        uint16_t *tile = reinterpret_cast<uint16_t *>(lds + 2 * ((n + 2) & ~int64_t(1)) + 4) + wave * (64 * 66);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int bit = 0; bit < 32; ++bit)
                tile[lane * 66 + half * 32 + bit] = static_cast<uint16_t>(half ? vextract<BT_LV>(oo1, bit) : vextract<BT_LV>(oo0, bit));
        const int64_t jc = wg * 64 + lane;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
            const int32_t row_r = __shfl(row, r);
            if (row_r >= 0 && jc < mloc) ns_out[static_cast<int64_t>(row_r) * mloc + jc] = static_cast<double>(tile[r * 66 + lane]);
        }
    }
}

// Un-slices the counters of one (wave, work item) -- bit-matrix transpose: rows 0..15 = #greater levels, rows 16..31 =
// #less levels -> word b = less << 16 | greater of attribute b -- and adds them to the totals [attribute][SELL position]
// (the 64 lanes of a wave update one contiguous 256-byte run).  The address walks down the attributes in a vector
// register pair: 64 scalar base addresses held at once do not fit the scalar file.
template <int CL, bool PLAIN_STORE = false>
__device__ __forceinline__ void flush_counters(const uint32_t (&g0)[CL], const uint32_t (&g1)[CL], const uint32_t (&l0)[CL],
                                               const uint32_t (&l1)[CL], unsigned int *__restrict__ gl_counts, int64_t col0,
                                               int64_t mloc, int64_t n_pad, int64_t spos, bool active) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t m[32];
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            m[l] = l < CL ? (half ? g1[l < CL ? l : 0] : g0[l < CL ? l : 0]) : 0u;
            m[16 + l] = l < CL ? (half ? l1[l < CL ? l : 0] : l0[l < CL ? l : 0]) : 0u;
        }
        transpose32(m);
        const int64_t c_first = col0 + half * 32;
        const int n_valid = static_cast<int>(mloc - c_first < 32 ? (mloc - c_first > 0 ? mloc - c_first : 0) : 32);
        // (the OFFSET walks in a vector register pair, the base stays the kernel argument: laundering the pointer itself made it a
        // generic one, and the 64 updates became flat atomics -- both wait counters, the slower path)
        // (TWO positions per 8-byte atomic -- the even lane of a lane pair adding attribute `bit` of both positions, the odd lane
        // attribute `bit + 1` -- was built in round 6 because the flush is 30 kclk of a wave-task waiting for its 64 updates to be
        // accepted: global_atomic_add_x2 made it 170-280 kclk and the seeded step 4.4 ms instead of 3.2; four-byte updates stay)
        int64_t off = c_first * n_pad + spos;
#pragma unroll
        for (int bit = 0; bit < 32; ++bit) {
            asm volatile("" : "+v"(off));
            if (PLAIN_STORE) {                                    // (diagnostic, wrong results: what the atomics themselves cost)
                if (bit < n_valid && active && m[bit]) gl_counts[off] = m[bit];
            } else if (bit < n_valid && active && m[bit]) atomicAdd(gl_counts + off, m[bit]);
            off += n_pad;
        }
    }
}

// Diagnostic (SAFE_HIP_BITS_DBG bit 7, results stay correct): every wave records (task slot, wave, start, end, block-iterations) of
// each task it ran in this buffer -- s_memtime: shader clocks on gfx9 -- and launch_bits prints how long the tasks took, in clocks
// per block-iteration and by slice width (clocks of different XCDs are not synchronised: only durations are used).
constexpr int BLK_TRACE_MAX = 1 << 17;
__device__ unsigned long long g_blk_trace[BLK_TRACE_MAX * 8];
__device__ unsigned int g_blk_trace_n;

// WPS = waves per SIMD the kernel is built for: 4 (128 registers; every class keeps its observed sums in registers and counts
// with CL levels) or 5 (96 registers: the classes of more than 56 members count with five levels -- their tasks hold at most
// 31 permutations -- and re-read the observed sums; five workgroups per CU when T fits five times).
template <int CL, int DBG, int WPS>
__device__ __forceinline__ void bits_blk_body(

    int64_t n, const uint16_t *__restrict__ ids_p, int64_t entries_pad, const int32_t *__restrict__ sell_row,
    const int64_t *__restrict__ slice_off, const int32_t *__restrict__ slice_width,
    const uint32_t *__restrict__ obs, int64_t n_slices, const uint2 *__restrict__ bbits, BitsQueues qs,
    const int4 *__restrict__ tasks, int64_t p_base, int64_t p_limit, unsigned int *__restrict__ queue, int64_t mloc,
    unsigned int *__restrict__ gl_counts, int64_t n_pad) {
    extern __shared__ unsigned int lds[];
    const int64_t t_words = 2 * ((n + 2) & ~int64_t(1));               // T: (n+1) uint2, 16-B padded
    uint2 *T = reinterpret_cast<uint2 *>(lds);
    int *slot_box = reinterpret_cast<int *>(lds + t_words);
    // the member ids ARE LDS addresses of T rows: T must sit at LDS address 0 (it does: no static LDS in this kernel)
    if ((uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned int *)lds) != 0u) __builtin_trap();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // One task queue per XCD (workgroups are dealt to the 8 XCDs round-robin: blockIdx & 7).  All word groups of one
    // (slice group, permutation range) sit in ONE queue, so the member-id lists of that range -- read by every one of its
    // 69 word-group tasks -- are fetched into one XCD's L2, not into all eight (the id stream was 8 x 41 MB of fabric
    // reads per launch); a workgroup whose queue has run dry takes from the others.
    const int home = blockIdx.x & 7;
    int tried = 0;                                                        // (thread 0's copy is the one that counts)

    for (;;) {
        if (threadIdx.x == 0) {
            int slot = -1;
            while (tried < 8) {
                const int q = (home + tried) & 7;
                const unsigned int got = atomicAdd(queue + q, 1u);
                if (got < static_cast<unsigned int>(qs.off[q + 1] - qs.off[q])) {
                    slot = qs.off[q] + static_cast<int>(got);
                    break;
                }
                ++tried;
            }
            *slot_box = slot;
        }
        __syncthreads();
        const int slot = *slot_box;
        __syncthreads();
        if (slot < 0) break;
        const int4 task = tasks[slot];
        const int wg = task.x;
        // a task = four adjacent slices (one per wave) of one word group over a permutation range
        const int64_t s = static_cast<int64_t>(task.y) * 4 + wave;
        const bool active = s < n_slices;
        const int64_t p_begin = p_base + task.z;
        int64_t p_end = p_base + task.w < p_limit ? p_base + task.w : p_limit;
        if (p_end < p_begin || !active) p_end = p_begin;                  // (nothing to do for this wave; it still joins the barriers)

        unsigned long long t_begin = 0;
        if (DBG & 128) t_begin = __builtin_readcyclecounter();
        load_word_column(T, bbits + static_cast<int64_t>(wg) * (n + 1), n);
        const int64_t my_blk = (active ? slice_off[s] : 0) / 8;             // in uint4 units (slice offsets are multiples of 512); wave-uniform
        const int wdt = __builtin_amdgcn_readfirstlane(active ? slice_width[s] : 0);
        const int nblk = wdt >> 3;
        const int np = static_cast<int>(p_end - p_begin);
        const uint32_t *my_obs = obs + (static_cast<int64_t>(wg) * n_slices + (active ? s : 0)) * (2 * BT_LV * 64);
        const u32x4 *perm_ids = reinterpret_cast<const u32x4 *>(ids_p + (p_begin - p_base) * entries_pad) + my_blk;
        const int64_t perm_stride = entries_pad / 8;
        __syncthreads();                                                  // T is complete; waves are independent from here
        unsigned long long t_ready = 0, t_counted = 0;
        if (DBG & 128) t_ready = __builtin_readcyclecounter();

        uint32_t g0[CL], g1[CL], l0[CL], l1[CL];                          // #(S_p > S_obs), #(S_p < S_obs)
#pragma unroll
        for (int l = 0; l < CL; ++l) g0[l] = g1[l] = l0[l] = l1[l] = 0;
        constexpr int CLW = WPS > 4 ? 5 : CL;                             // counter levels of the wide classes
        constexpr int CLN = WPS > 4 ? 6 : CL;                             // ... and of the narrow ones (tasks of at most 63 permutations)
        constexpr bool OM = WPS > 4;                                      // the wide classes' observed sums come from memory
        // a neighborhood of wdt members cannot sum past wdt: levels by slice width (wave-uniform branch)
        if (np <= 0) {
        } else if (DBG & 1024) {       // (diagnostic, WRONG results: every class counts with four levels -- what bounding the levels by the attributes' carrier counts could gain at most)
            blk_task<4, CL, DBG, CLN>(my_obs, perm_ids, perm_stride, lane, nblk, np, g0, g1, l0, l1);
        } else if (DBG & 2048) {       // (the same with six levels)
            blk_task<6, CL, DBG, CLN>(my_obs, perm_ids, perm_stride, lane, nblk, np, g0, g1, l0, l1);
        } else if (wdt <= 8) blk_task<4, CL, DBG, CLN>(my_obs, perm_ids, perm_stride, lane, nblk, np, g0, g1, l0, l1);
        else if (wdt <= 56) blk_task<6, CL, DBG, CLN>(my_obs, perm_ids, perm_stride, lane, nblk, np, g0, g1, l0, l1);
        else if (wdt <= 248) blk_task<8, CL, DBG, CLW, OM>(my_obs, perm_ids, perm_stride, lane, nblk, np, g0, g1, l0, l1);
        else if (wdt <= 504) blk_task<9, CL, DBG, CLW, OM>(my_obs, perm_ids, perm_stride, lane, nblk, np, g0, g1, l0, l1);
        else blk_task<BT_LV, CL, DBG, CLW, OM>(my_obs, perm_ids, perm_stride, lane, nblk, np, g0, g1, l0, l1);

        if (DBG & 128) {
            asm volatile("" : "+v"(g0[0]), "+v"(l0[0]));                  // (the counters exist: the counting is over)
            t_counted = __builtin_readcyclecounter();
        }
        const int64_t spos = s * 64 + lane;
        if (!(DBG & 2)) {
            if (np > 0) flush_counters<CL, (DBG & 64) != 0>(g0, g1, l0, l1, gl_counts, static_cast<int64_t>(wg) * 64, mloc, n_pad, spos, active);
        } else if (active && (g0[0] | g1[0] | l0[0] | l1[0]) == 0xDEADBEEFu) {
            gl_counts[spos] = g0[1] ^ l0[1];                              // (keeps the counters alive in the diagnostic build)
        }
        if ((DBG & 128) && lane == 0) {
            const unsigned int at = atomicAdd(&g_blk_trace_n, 1u);
            if (at < BLK_TRACE_MAX) {
                g_blk_trace[8 * at] = (static_cast<unsigned long long>(blockIdx.x) << 32) | (static_cast<unsigned long long>(slot) << 2) | wave;
                g_blk_trace[8 * at + 1] = t_begin;
                g_blk_trace[8 * at + 2] = __builtin_readcyclecounter();
                g_blk_trace[8 * at + 3] = (static_cast<unsigned long long>(nblk) << 32) | static_cast<unsigned int>(np);
                g_blk_trace[8 * at + 4] = t_ready;
                g_blk_trace[8 * at + 5] = t_counted;
            }
        }
        __syncthreads();                                                  // before T is overwritten by the next task
    }
}

// The two builds of the body.  The default one holds the id stream's two quads in registers 112-119 behind the compiler's back
// (blk_add8s): it is compiled with 112 registers (the attribute counts in pairs on gfx90a and later: 56), and the clobber lists of the
// stream's asm statements make the allocation 120.
// The plain one (five waves per SIMD, SAFE_HIP_BITS_DBG bit 8) runs without the stream.
template <int CL, int DBG>
__global__ __launch_bounds__(256, 4) __attribute__((amdgpu_num_vgpr(56))) void k_permtest_bits_blk(

    int64_t n, const uint16_t *__restrict__ ids_p, int64_t entries_pad, const int32_t *__restrict__ sell_row,
    const int64_t *__restrict__ slice_off, const int32_t *__restrict__ slice_width,
    const uint32_t *__restrict__ obs, int64_t n_slices, const uint2 *__restrict__ bbits, BitsQueues qs,
    const int4 *__restrict__ tasks, int64_t p_base, int64_t p_limit, unsigned int *__restrict__ queue, int64_t mloc,
    unsigned int *__restrict__ gl_counts, int64_t n_pad) {
    bits_blk_body<CL, DBG & ~256, 4>(n, ids_p, entries_pad, sell_row, slice_off, slice_width, obs, n_slices, bbits, qs, tasks, p_base, p_limit, queue, mloc, gl_counts, n_pad);
}
template <int CL, int DBG, int WPS>
__global__ __launch_bounds__(256, WPS) void k_permtest_bits_blk_plain(

    int64_t n, const uint16_t *__restrict__ ids_p, int64_t entries_pad, const int32_t *__restrict__ sell_row,
    const int64_t *__restrict__ slice_off, const int32_t *__restrict__ slice_width,
    const uint32_t *__restrict__ obs, int64_t n_slices, const uint2 *__restrict__ bbits, BitsQueues qs,
    const int4 *__restrict__ tasks, int64_t p_base, int64_t p_limit, unsigned int *__restrict__ queue, int64_t mloc,
    unsigned int *__restrict__ gl_counts, int64_t n_pad) {
    bits_blk_body<CL, DBG | 256, WPS>(n, ids_p, entries_pad, sell_row, slice_off, slice_width, obs, n_slices, bbits, qs, tasks, p_base, p_limit, queue, mloc, gl_counts, n_pad);
}

// counts -> everything compute_pvalues derives from them (safe.py:528-554, 468-472).
// The counters are [column][SELL position]; the outputs are [row][column].  A block takes a
// 64 x 64 tile: coalesced reads along the SELL positions, transpose through LDS, coalesced
// 512-byte writes along the columns of each row.
// DIRECT = false: counters hold (#less << 16 | #greater)  (bit-sliced kernel)
// DIRECT = true : counters hold (#>=   << 16 | #<=) and NaN observed scores matter (f64 kernel)
// MODE = out.mode (1 raw counts, 2 everything, 3 NES only, 4 any subset), a template parameter so that the row loop has no
// branches; TAB_LDS: the NES table (P + 1 doubles) is staged in LDS.  The row loop then holds no global load at all (row ids
// and the table come from LDS), so its stores stream: the first form loaded the row id, the two table entries and (DIRECT) the
// observed score per row with `s_waitcnt vmcnt(0)` between them -- vmcnt retires in order, so every row also waited for the
// previous row's stores: 158 us for 0.63 GB.
// Tile shape: FIN_TP SELL positions x FIN_TC columns.  A 64 x 64 tile wrote 512-byte runs (3.6 TB/s); with 16 x 512 a wave
// writes 4 KiB of one output row back to back (the counter reads become 64-byte pieces, but they are an eighth of the bytes).
constexpr int FIN_TP = 16, FIN_TC = 512;
// pv_lds: behind the NES table in LDS sits a second one, k / P for k = 0..P (each entry one correctly rounded division, as the
// reference's counts / num_permutations, safe.py:532-533): two look-ups per output instead of two f64 divisions (~60 VALU
// instructions per output against four stores)
// PK20 (exchanged counters of other ranks, P <= 1023): a column is n_pad / 2 words + n_pad / 2 bytes -- two 20-bit pairs
// (#less << 10 | #greater) in 40 bits, low 32 bits in the word, high 8 in the byte (safe_export_packed_chunk_narrow)
template <bool DIRECT, int MODE, bool TAB_LDS, bool PK20 = false>
__global__ __launch_bounds__(256) void k_counts_finalize(const unsigned int *__restrict__ counts, int64_t n_pad,
                                                         const int32_t *__restrict__ sell_row,
                                                         const double *__restrict__ ns, int64_t mloc, int64_t n_perm,
                                                         PermOut out, int pv_lds) {
    __shared__ unsigned int tile[FIN_TC][FIN_TP + 1];
    __shared__ unsigned int part[4][FIN_TC];
    __shared__ int32_t rows[FIN_TP];
    extern __shared__ double tab_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t spos0 = static_cast<int64_t>(blockIdx.x) * FIN_TP, c0 = static_cast<int64_t>(blockIdx.y) * FIN_TC;
    {
        // thread t: position t % 16 of columns t / 16 + 16 k -- sixteen loads in flight, two rounds
        const int p = threadIdx.x & (FIN_TP - 1), cq = threadIdx.x / FIN_TP;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned int v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int64_t c = c0 + cq + 16 * (half * 16 + k);
                if (PK20) {
                    const unsigned int *col = counts + (c < mloc ? c : mloc - 1) * (n_pad / 8 * 5);
                    const int64_t pos = spos0 + p, i = pos >> 1;
                    const unsigned long long x = static_cast<unsigned long long>(col[i]) |
                                                 (static_cast<unsigned long long>(reinterpret_cast<const unsigned char *>(col + n_pad / 2)[i]) << 32);
                    const unsigned int pair = static_cast<unsigned int>(x >> (20 * (pos & 1))) & 0xFFFFFu;
                    v[k] = ((pair >> 10) << 16) | (pair & 0x3FFu);
                } else {
                    v[k] = counts[(c < mloc ? c : mloc - 1) * n_pad + spos0 + p];
                }
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) tile[cq + 16 * (half * 16 + k)][p] = v[k];
        }
    }
    if (threadIdx.x < FIN_TP) rows[threadIdx.x] = sell_row[spos0 + threadIdx.x];
    if (TAB_LDS && MODE != 1)
        for (int64_t i = threadIdx.x; i <= n_perm; i += 256) {
            tab_lds[i] = out.nes_table[i];
            if (pv_lds) tab_lds[n_perm + 1 + i] = static_cast<double>(i) / static_cast<double>(n_perm);
        }
    __syncthreads();
    const double *tab = TAB_LDS ? tab_lds : out.nes_table;
    const double *pv = (TAB_LDS && pv_lds) ? tab_lds + n_perm + 1 : nullptr;
    const unsigned int P = static_cast<unsigned int>(n_perm);
    const double p_f = static_cast<double>(P);
    unsigned int hits[FIN_TC / 64];
#pragma unroll
    for (int ct = 0; ct < FIN_TC / 64; ++ct) hits[ct] = 0;
    for (int ss = wave; ss < FIN_TP; ss += 4) {
        const int32_t row = rows[ss];
        if (row < 0) continue;
#pragma unroll
        for (int ct = 0; ct < FIN_TC / 64; ++ct) {
            const int64_t c = c0 + ct * 64 + lane;
            if (c >= mloc) continue;
            const unsigned int v = tile[ct * 64 + lane][ss];
            const int64_t o = static_cast<int64_t>(row) * (out.ld ? out.ld : mloc) + c;      // (ld: a column block of a wider matrix)
            unsigned int cneg, cpos;
            bool obs_nan = false;
            if (DIRECT) {
                cneg = v & 0xFFFFu;
                cpos = v >> 16;
                const double obs = ns[static_cast<int64_t>(row) * mloc + c];
                obs_nan = obs != obs;
            } else {
                cneg = P - (v & 0xFFFFu);           // #(S_p <= S_obs) = P - #greater
                cpos = P - (v >> 16);               // #(S_p >= S_obs) = P - #less
            }
            if (MODE == 1) {
                out.counts_neg[o] = static_cast<double>(cneg);
                out.counts_pos[o] = static_cast<double>(cpos);
            } else if (MODE == 3) {                                  // NES only (all-gathered counters of other ranks)
                const double en = tab[cneg], ep = tab[cpos];
                out.nes[o] = out.sign_mode == SAFE_SIGN_HIGHEST ? ep : out.sign_mode == SAFE_SIGN_LOWEST ? en : ep - en;
            } else if (MODE == 4) {                                  // any subset of the matrices from all-gathered 'sum' counters
                const double en = tab[cneg], ep = tab[cpos];
                const double nes = out.sign_mode == SAFE_SIGN_HIGHEST ? ep : out.sign_mode == SAFE_SIGN_LOWEST ? en : ep - en;
                if (out.pvalues_neg) out.pvalues_neg[o] = pv ? pv[cneg] : static_cast<double>(cneg) / p_f;
                if (out.pvalues_pos) out.pvalues_pos[o] = pv ? pv[cpos] : static_cast<double>(cpos) / p_f;
                if (out.nes) out.nes[o] = nes;
                if (out.nes_binary) out.nes_binary[o] = fabs(nes) > out.nes_threshold ? 1.0 : 0.0;
            } else if (MODE == 2) {
                const double qnan = __longlong_as_double(0x7FF8000000000000ll);
                const double en = obs_nan ? qnan : tab[cneg], ep = obs_nan ? qnan : tab[cpos];
                double nes = ep - en;
                if (out.sign_mode == SAFE_SIGN_HIGHEST) nes = ep;
                if (out.sign_mode == SAFE_SIGN_LOWEST) nes = en;
                const bool hit = (nes == nes) && (fabs(nes) > out.nes_threshold);
                out.pvalues_neg[o] = obs_nan ? qnan : pv ? pv[cneg] : static_cast<double>(cneg) / p_f;
                out.pvalues_pos[o] = obs_nan ? qnan : pv ? pv[cpos] : static_cast<double>(cpos) / p_f;
                out.nes[o] = nes;
                out.nes_binary[o] = hit ? 1.0 : 0.0;
                hits[ct] += hit;
            }
        }
    }
    if (MODE == 2) {
#pragma unroll
        for (int ct = 0; ct < FIN_TC / 64; ++ct) part[wave][ct * 64 + lane] = hits[ct];
        __syncthreads();
        for (int cc = threadIdx.x; cc < FIN_TC; cc += 256) {
            const unsigned int t = part[0][cc] + part[1][cc] + part[2][cc] + part[3][cc];
            if (t && c0 + cc < mloc) atomicAdd(&out.enriched[c0 + cc], t);
        }
    }
}

// ns_direct != NULL: the counters hold (#>= << 16 | #<=) against the observed scores in ns_direct (NaN there = no test)
int enrich_finalize_counts(safe_ctx *ctx, const unsigned int *counts, int64_t n_pad, const int32_t *rowmap, int64_t mloc,
                           int64_t n_perm, const PermOut &out, const double *ns_direct, hipStream_t on, bool pk20) {
    const hipStream_t fin_stream = on ? on : ctx->stream;
    const dim3 grid(n_pad / FIN_TP, ceil_div(mloc, FIN_TC));
    const size_t tab_bytes = static_cast<size_t>(n_perm + 1) * sizeof(double);
    const bool tab_lds = out.mode != 1 && tab_bytes <= 20 * 1024;          // (next to 43 KB of static LDS)
    const int pv_lds = tab_lds && 2 * tab_bytes <= 20 * 1024 ? 1 : 0;                                  // k / P table behind the NES table
    const size_t dyn = tab_lds ? (pv_lds ? 2 : 1) * tab_bytes : 0;
#define FIN(D, M, L) hipLaunchKernelGGL((k_counts_finalize<D, M, L>), grid, dim3(256), dyn, fin_stream, counts, n_pad, rowmap, ns_direct, mloc, n_perm, out, pv_lds)
#define FIN_MODE(D, L)                      \
    do {                                    \
        if (out.mode == 1) FIN(D, 1, false); \
        else if (out.mode == 2) FIN(D, 2, L); \
        else if (out.mode == 3) FIN(D, 3, L); \
        else if (out.mode == 4) FIN(D, 4, L); \
    } while (0)
    if (pk20) {                                        // (exchanged slabs only: mode 4, no observed scores)
        if (ns_direct || out.mode != 4 || n_perm > 1023) {
            safe_set_error("enrich_finalize_counts: 20-bit counter pairs serve the exchanged 'sum' counters of at most 1023 permutations");
            return SAFE_E_INVALID;
        }
        if (tab_lds) hipLaunchKernelGGL((k_counts_finalize<false, 4, true, true>), grid, dim3(256), dyn, fin_stream, counts, n_pad, rowmap, ns_direct, mloc, n_perm, out, pv_lds);
        else hipLaunchKernelGGL((k_counts_finalize<false, 4, false, true>), grid, dim3(256), dyn, fin_stream, counts, n_pad, rowmap, ns_direct, mloc, n_perm, out, pv_lds);
    } else if (ns_direct) {
        if (tab_lds) FIN_MODE(true, true);
        else FIN_MODE(true, false);
    } else {
        if (tab_lds) FIN_MODE(false, true);
        else FIN_MODE(false, false);
    }
#undef FIN_MODE
#undef FIN
    SAFE_HIP_CHECK(hipGetLastError());
    return SAFE_OK;
}

// bbits[wg][r] = 64 attribute bits of row r for word group wg (row n = 0: SELL padding)
template <typename T>
__global__ __launch_bounds__(256) void k_bits_prep(const void *__restrict__ raw, int64_t n, int64_t rs, int64_t cs,
                                                   int64_t col0, int64_t mloc, int64_t n_wg, uint2 *__restrict__ bbits) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= n_wg * (n + 1)) return;
    const int64_t wg = idx / (n + 1), r = idx % (n + 1);
    uint32_t w[2] = {0, 0};
    if (r < n) {
        const T *src = reinterpret_cast<const T *>(raw) + r * rs + (col0 + wg * 64) * cs;
        if (wg * 64 + 64 <= mloc) {
            // a whole word group: sixteen loads in flight at a time (the loop below is load - wait - test per column)
#pragma unroll
            for (int a0 = 0; a0 < 64; a0 += 16) {
                T x[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) x[u] = src[(a0 + u) * cs];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (x[u] == static_cast<T>(1)) w[(a0 + u) >> 5] |= 1u << ((a0 + u) & 31);
            }
        } else {
            for (int a = 0; a < 64; ++a) {
                if (wg * 64 + a >= mloc) break;
                if (src[a * cs] == static_cast<T>(1)) w[a >> 5] |= 1u << (a & 31);
            }
        }
    }
    bbits[idx] = make_uint2(w[0], w[1]);
}

// the same for row-major (C order) matrices: a wave reads 64 consecutive columns of one row
// (coalesced) and the ballot of (x == 1) IS the word
template <typename T>
__global__ __launch_bounds__(256) void k_bits_prep_rowmajor(const void *__restrict__ raw, int64_t n, int64_t rs, int64_t cs,
                                                            int64_t col0, int64_t mloc, int64_t n_wg, uint2 *__restrict__ bbits) {
    const int64_t wg = blockIdx.x, r = static_cast<int64_t>(blockIdx.y) * 4 + (threadIdx.x >> 6);
    if (r > n) return;
    const int lane = threadIdx.x & 63;
    const int64_t j = wg * 64 + lane;
    bool one = false;
    if (r < n && j < mloc) one = reinterpret_cast<const T *>(raw)[r * rs + (col0 + j) * cs] == static_cast<T>(1);
    const unsigned long long w = __builtin_amdgcn_ballot_w64(one);
    if (lane == 0) bbits[wg * (n + 1) + r] = make_uint2(static_cast<uint32_t>(w), static_cast<uint32_t>(w >> 32));
}

static void launch_bits_prep(safe_ctx *ctx, const safe_attr *attr, int64_t col0, int64_t mloc, int64_t n_wg, uint2 *d_bits) {
    const int64_t n = attr->n;
    const bool f32 = attr->dtype == SAFE_DTYPE_F32;
    if (attr->col_stride == 1 && attr->row_stride != 1) {              // C order
        const dim3 grid(n_wg, ceil_div(n + 1, 4)), block(256);
        if (f32) hipLaunchKernelGGL(k_bits_prep_rowmajor<float>, grid, block, 0, ctx->stream, attr->raw, n, attr->row_stride,
                                    attr->col_stride, col0, mloc, n_wg, d_bits);
        else hipLaunchKernelGGL(k_bits_prep_rowmajor<double>, grid, block, 0, ctx->stream, attr->raw, n, attr->row_stride,
                                attr->col_stride, col0, mloc, n_wg, d_bits);
    } else {                                                            // Fortran order (consecutive rows adjacent) or general strides
        const dim3 grid(ceil_div(n_wg * (n + 1), 256)), block(256);
        if (f32) hipLaunchKernelGGL(k_bits_prep<float>, grid, block, 0, ctx->stream, attr->raw, n, attr->row_stride,
                                    attr->col_stride, col0, mloc, n_wg, d_bits);
        else hipLaunchKernelGGL(k_bits_prep<double>, grid, block, 0, ctx->stream, attr->raw, n, attr->row_stride,
                                attr->col_stride, col0, mloc, n_wg, d_bits);
    }
}

// Observed neighborhood counts of binary attributes, bit-sliced (safe.py:593-594 for the
// hypergeometric path; safe_extras.py:15 for 'sum' scores): one wave per (SELL slice, 64-attribute
// word group), member words gathered from the L2-resident bit matrix, vertical carry-save sums,
// bit-matrix transpose, 512 contiguous output bytes per lane.
// helpers of the hypergeometric kernels (K4, further down)
__device__ __forceinline__ double hyp_logpmf(const double *__restrict__ lf, int64_t t, int64_t pop, int64_t good,
                                             int64_t draws) {
    return (lf[good] - lf[t] - lf[good - t]) + (lf[pop - good] - lf[draws - t] - lf[pop - good - draws + t]) -
           (lf[pop] - lf[draws] - lf[pop - draws]);
}

// 1 / x for the term ratios of the tail recurrence: hardware reciprocal estimate + two Newton
// steps (a couple of ulp, far inside the 1e-6 relative parity bound) instead of the ~30-instruction
// IEEE division; no table loads inside the serial loop (they left the waves waiting 80 % of the time)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}


// TABLE = true: the hypergeometric epilogue is fused in -- instead of the count X the kernel writes
// p = tab[(nid[row] * xs + X) * n_kid + kid[col]] (k_hyp_table), -log10 p, the binarised value and the
// per-attribute enriched counts (safe.py:596-608, 468-472); the counts never reach memory.

template <bool TABLE>
__global__ __launch_bounds__(64) void k_counts_bits(const int32_t *__restrict__ sell_row,
                                                    const int64_t *__restrict__ slice_off,
                                                    const int32_t *__restrict__ slice_width,
                                                    const int32_t *__restrict__ sell_col, int64_t n,
                                                    const uint2 *__restrict__ bbits, int64_t mloc,
                                                    double *__restrict__ out, HypLookup hl) {
    const int64_t s = blockIdx.x, wg = blockIdx.y;
    const int lane = threadIdx.x;
    const int32_t row = sell_row[s * 64 + lane];
    const int32_t *cols = sell_col + slice_off[s] + lane;
    const int wdt = slice_width[s];
    const uint2 *T = bbits + wg * (n + 1);
    uint32_t s0[BT_LV], s1[BT_LV];
#pragma unroll
    for (int l = 0; l < BT_LV; ++l) s0[l] = s1[l] = 0;
    for (int t0 = 0; t0 < wdt; t0 += 8) {
        uint32_t x0[8], x1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint2 w = T[cols[(t0 + u) * 64]];
            x0[u] = w.x;
            x1[u] = w.y;
        }
        const uint32_t e0 = vadd8(s0, x0);
        const uint32_t e1 = vadd8(s1, x1);
        if (__builtin_amdgcn_ballot_w64((e0 | e1) != 0)) {
            vripple(s0, e0);
            vripple(s1, e1);
        }
    }
    // ---- epilogue: un-slice the counts (bit-matrix transpose), turn the 64 x 64 tile around through
    //      LDS so that a lane owns a COLUMN, and write row by row: 512 contiguous bytes per store
    __shared__ unsigned short tile[64][66];                             // [row in slice][column], padded rows
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t m[32];
#pragma unroll
        for (int l = 0; l < 32; ++l) m[l] = l < BT_LV ? (half ? s1[l < BT_LV ? l : 0] : s0[l < BT_LV ? l : 0]) : 0u;
        transpose32(m);
#pragma unroll
        for (int bit = 0; bit < 32; bit += 2)
            *reinterpret_cast<uint32_t *>(&tile[lane][half * 32 + bit]) = m[bit] | (m[bit + 1] << 16);
    }
    __syncthreads();
    const int64_t jc = wg * 64 + lane;                                  // this lane's column
    const bool col_ok = jc < mloc;
    if constexpr (!TABLE) {
        for (int r = 0; r < 64; ++r) {
            const int32_t rr = sell_row[s * 64 + r];                    // wave-uniform
            if (rr < 0 || !col_ok) continue;
            out[static_cast<int64_t>(rr) * mloc + jc] = static_cast<double>(tile[r][lane]);
        }
    } else {
        const int64_t kofs = col_ok ? static_cast<int64_t>(hl.kid[jc]) : 0;
        unsigned int hits = 0;
        // row ids and their neighborhood-size ids once per wave (lane r holds row r's), then four rows per trip: the four table
        // gathers are in flight together and their twelve stores follow back to back (the plain loop was a chain of three
        // dependent loads -- row id, size id, table entry -- in front of every row's stores)
        const int32_t my_nid = row >= 0 ? hl.nid[row] : 0;
        const int64_t slab_stride = hl.n_kid * hl.xs;
        for (int r0 = 0; r0 < 64; r0 += 4) {
            int32_t rr[4];
            double2 e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rr[u] = __shfl(row, r0 + u);
                const int32_t nid_u = __shfl(my_nid, r0 + u);
                const double2 *slab = hl.tab + static_cast<int64_t>(nid_u) * slab_stride;
                e[u] = slab[(rr[u] >= 0 && col_ok ? static_cast<int64_t>(tile[r0 + u][lane]) * hl.n_kid + kofs : 0)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (rr[u] < 0 || !col_ok) continue;
                const double p = e[u].x, nes = e[u].y;                  // p and -log10 p (safe.py:608), both from the table
                const bool hit = p < hl.p_cut;                          // safe.py:468-470 (nes_p_cut)
                const int64_t o = static_cast<int64_t>(rr[u]) * mloc + jc;
                hl.pvalues_pos[o] = p;
                hl.nes[o] = nes;
                hl.nes_binary[o] = hit ? 1.0 : 0.0;
                hits += hit;
            }
        }
        if (hits) atomicAdd(&hl.enriched[jc], hits);
    }
}

// double-double helpers of the table kernel (error-free transformations; the TU is built with
// -ffp-contract=off, every fma below is explicit)
struct dd_t {
    double hi, lo;
};
__device__ __forceinline__ dd_t dd_fast_two_sum(double a, double b) {
    const double s = a + b;
    return {s, b - (s - a)};
}
__device__ __forceinline__ dd_t dd_add(dd_t x, dd_t y) {
    const double s = x.hi + y.hi, bb = s - x.hi;
    const double e = ((x.hi - (s - bb)) + (y.hi - bb)) + (x.lo + y.lo);
    return dd_fast_two_sum(s, e);
}
__device__ __forceinline__ dd_t dd_mul_d(dd_t x, double d) {
    const double p = x.hi * d;
    const double e = fma(x.lo, d, fma(x.hi, d, -p));
    return dd_fast_two_sum(p, e);
}
__device__ __forceinline__ dd_t dd_div_d(dd_t x, double d) {
    const double r = fast_rcp(d), q1 = x.hi * r, p = q1 * d;
    const double rem = ((x.hi - p) - fma(q1, d, -p)) + x.lo;            // x - q1 * d, exactly enough
    return dd_fast_two_sum(q1, rem * r);
}
__device__ __forceinline__ double dd_ratio(dd_t a, dd_t b) {            // a / b rounded to double
    const double r = fast_rcp(b.hi), q1 = a.hi * r;
    const dd_t prod = dd_mul_d(b, q1);
    const double rem = ((a.hi - prod.hi) - prod.lo) + a.lo;
    return q1 + rem * r;
}

// tab[nid][x][kid] = (p, -log10 p) with p = P[H >= x] for H ~ Hypergeom(pop, K = kvals[kid], n = nvals[nid]),
// x = 0 .. xs-1, with the support rules of scipy's rv_discrete.sf (below the support 1, above it 0).  One
// thread per (n, K) pair.  The pmf is carried RELATIVE to its value at the mode (term recurrence outwards
// from the mode, so nothing underflows at the start) in double-double arithmetic; tails are summed from
// the top down (smallest terms first) and divided by the sum over the whole support, so no log-gamma
// rounding enters and p is the correctly rounded value except for ties of the last-but-~50th bit.  That
// matters for the binarisation: p values that are short rationals (9/180 = 0.05 exactly) sit ON the
// enrichment threshold, and SciPy returns them exactly.
__global__ __launch_bounds__(64) void k_hyp_table(const int32_t *__restrict__ nvals, int64_t n_nid,
                                                  const int32_t *__restrict__ kvals, int64_t n_kid, int64_t xs, int64_t pop,
                                                  const unsigned int *__restrict__ xmax, double2 *__restrict__ tab) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
    if (idx >= n_nid * n_kid) return;
    // xmax (split matrix-core form: the counts are known before the lookup) = the largest count of the call;
    // only x <= xmax is ever looked up, so the table stops there and the serial recurrences stop as soon as
    // the terms can no longer reach the last bit of a double-double sum (a few dozen steps past the mode
    // instead of the whole support: 5x fewer at 20 000 nodes / 1 % density).  xc = entries per column.
    const int64_t xc = xmax ? (static_cast<int64_t>(*xmax) + 1 < xs ? static_cast<int64_t>(*xmax) + 1 : xs) : xs;
    const int64_t draws = nvals[idx / n_kid], good = kvals[idx % n_kid];
    // layout [size id][x][count id]: consecutive lanes (count ids) touch consecutive entries, and the
    // entries of the small x that counts actually reach are one contiguous run per size (k_hyp_emit's LDS slab)
    double2 *t_base = tab + (idx / n_kid) * xs * n_kid + idx % n_kid;    // scratch first: (hi, lo) of the relative pmf
#define t_out(t) t_base[(t) * n_kid]
    const int64_t lo = draws - (pop - good) > 0 ? draws - (pop - good) : 0;
    const int64_t hi = good < draws ? good : draws;
    const double good_d = static_cast<double>(good), draws_d = static_cast<double>(draws);
    const double rest_d = static_cast<double>(pop) - good_d - draws_d;
    int64_t mode = static_cast<int64_t>(floor(static_cast<double>(good + 1) * static_cast<double>(draws + 1) / static_cast<double>(pop + 2)));
    mode = mode < lo ? lo : (mode > hi ? hi : mode);
    // terms at or beyond xc are only summed (`beyond`).  The pmf is unimodal and falls faster than
    // geometrically away from the mode, so once a term is below 1e-40 of a sum every needed quantity
    // contains (`beyond` <= every tail P[H >= x], x < xc, and <= total), the rest of that side adds
    // nothing at double-double precision (1e-32).
    dd_t beyond{0.0, 0.0}, total{0.0, 0.0}, term{1.0, 0.0};
    double td = static_cast<double>(mode);
    for (int64_t t = mode; t <= hi; ++t) {                              // upwards from the mode
        if (t < xc) t_out(t) = make_double2(term.hi, term.lo);
        else beyond = dd_add(beyond, term);
        total = dd_add(total, term);
        if (t >= xc && (term.hi == 0.0 || term.hi < beyond.hi * 1e-40)) break;
        term = dd_mul_d(dd_mul_d(term, good_d - td), draws_d - td);
        term = dd_div_d(dd_div_d(term, td + 1.0), rest_d + td + 1.0);
        td += 1.0;
    }
    term = dd_t{1.0, 0.0};
    td = static_cast<double>(mode);
    int64_t t_stop = lo;                                                // terms below t_stop are zero at this precision
    for (int64_t t = mode - 1; t >= lo; --t) {                          // downwards from the mode
        term = dd_mul_d(dd_mul_d(term, td), rest_d + td);
        term = dd_div_d(dd_div_d(term, good_d - td + 1.0), draws_d - td + 1.0);
        td -= 1.0;
        if (term.hi < total.hi * 1e-45) {                               // (and every tail that would contain it is >= the mode's term = 1)
            t_stop = t + 1;
            break;
        }
        if (t < xc) t_out(t) = make_double2(term.hi, term.lo);
        else beyond = dd_add(beyond, term);
        total = dd_add(total, term);
    }
    dd_t running = beyond;
    for (int64_t t = xc - 1; t >= 0; --t) {
        double p;
        if (t > hi) {
            p = 0.0;                                                    // sf(x - 1) with x - 1 >= top of the support
        } else if (t <= lo) {
            p = 1.0;                                                    // x - 1 below the support
        } else {
            if (t >= t_stop) running = dd_add(running, dd_t{t_out(t).x, t_out(t).y});
            p = dd_ratio(running, total);
            p = p > 1.0 ? 1.0 : p;
        }
        t_out(t) = make_double2(p, -log10(p));                          // safe.py:608
    }
#undef t_out
}

// --------------------------------------------------------------------------------------
// K5 (general f64 form, LDS resident): quantitative attributes and z-scores on networks of up
// to ~4500 nodes.  Same task structure as the bit-sliced kernel -- a workgroup owns 4 adjacent
// SELL slices x one column tile x a permutation sub-range; LDS holds the tile as 32-byte rows
// (sum: 4 attribute columns; z-score: B0, B0^2, not-NaN of ONE column) and the current
// permutation row (u16, double buffered) -- but the arithmetic is the reference's f64: a lane
// adds its members' rows in SELL order, compares with the observed score and keeps the two
// counters in registers; they leave as packed (#>= << 16 | #<=) atomics, [column][SELL pos].
// --------------------------------------------------------------------------------------
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) f64x2 *lds_d2_ptr;

template <bool Z, int NW>
__global__ __launch_bounds__(64 * NW) void k_permtest_lds(
    int64_t n, const uint16_t *__restrict__ cur16, int64_t stride16, const int32_t *__restrict__ sell_row,
    const int64_t *__restrict__ slice_off, const int32_t *__restrict__ slice_width,
    const uint16_t *__restrict__ sell_col2, int64_t n_slices, const double *__restrict__ tiles, int64_t n_tasks,
    const int4 *__restrict__ tasks, int64_t p_base, int64_t p_limit, unsigned int *__restrict__ queue, int64_t mloc,
    unsigned int *__restrict__ counts, int64_t n_pad, double *__restrict__ ns_out) {
    extern __shared__ unsigned int lds[];
    constexpr int BN = Z ? 1 : 4;
    constexpr int NT = 64 * NW;                                          // NW waves = NW adjacent slices per workgroup
    const int64_t t_words = 8 * (n + 1);                                 // tile: (n+1) rows x 32 B
    double *T = reinterpret_cast<double *>(lds);
    unsigned short *CUR = reinterpret_cast<unsigned short *>(lds + t_words);     // [2][stride16]
    unsigned int *slot_box = lds + t_words + stride16;
    const uint32_t t_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned int *)lds);
    const uint32_t cur_bytes0 = t_addr + static_cast<uint32_t>(t_words * 4);
    const uint32_t cur_bytes1 = cur_bytes0 + static_cast<uint32_t>(stride16 * 2);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int vec_per_row = static_cast<int>(stride16 / 8);

    // one pass over the lane's neighborhood: acc = sum of the members' 32-byte rows
    auto accumulate = [&](const uint16_t *cols2, int wdt, uint32_t cur_addr, bool ident, double (&acc)[4]) {
        acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
        for (int t0 = 0; t0 < wdt; t0 += 8) {
            uint32_t r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t c2 = cols2[(t0 + u) * 64];
                r[u] = ident ? (c2 << 4) : (static_cast<uint32_t>(*(lds_u16_ptr)(uintptr_t)(cur_addr + c2)) << 5);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f64x2 a = *(lds_d2_ptr)(uintptr_t)(t_addr + r[u]);
                const f64x2 b = *(lds_d2_ptr)(uintptr_t)(t_addr + r[u] + 16);
                acc[0] += a.x;
                acc[1] += a.y;
                acc[2] += b.x;
                if (!Z) acc[3] += b.y;
            }
        }
    };
    auto score_of = [](const double (&acc)[4], double (&sc)[BN]) {
        if (!Z) {
#pragma unroll
            for (int c = 0; c < BN; ++c) sc[c] = acc[c];
        } else {                                   // safe_extras.py:19-31
            const double cnt = acc[2];
            const double mean = acc[0] / cnt;
            const double exx = acc[1] / cnt;
            const double sd = sqrt(exx - mean * mean);
            double v = mean / sd;
            if (sd == 0.0) v = __longlong_as_double(0x7FF8000000000000ll);
            if (cnt < 3.0) v = __longlong_as_double(0x7FF8000000000000ll);
            sc[0] = v;
        }
    };

    for (;;) {
        if (threadIdx.x == 0) *slot_box = atomicAdd(queue, 1u);
        __syncthreads();
        const int64_t slot = *slot_box;
        __syncthreads();
        if (slot >= n_tasks) break;
        const int4 task = tasks[slot];
        const int tile = task.x, sg = task.y;
        const int64_t p_begin = p_base + task.z;
        const int64_t p_end = p_base + task.w < p_limit ? p_base + task.w : p_limit;
        if (p_end <= p_begin) continue;                                   // a launch shorter than the task grid's span

        const double *src = tiles + static_cast<int64_t>(tile) * (n + 1) * 4;
        for (int64_t i = threadIdx.x; i < (n + 1) * 4; i += NT) T[i] = src[i];
        if (p_end > p_begin)
            for (int v = threadIdx.x; v < vec_per_row; v += NT)
                reinterpret_cast<uint4_alias *>(CUR)[v] = reinterpret_cast<const uint4_alias *>(cur16 + p_begin * stride16)[v];

        const int64_t s = static_cast<int64_t>(sg) * NW + wave;
        const bool active = s < n_slices;
        const int32_t row = active ? sell_row[s * 64 + lane] : -1;
        const uint16_t *cols2 = sell_col2 + (active ? slice_off[s] : 0) + lane;
        const int wdt = active ? slice_width[s] : 0;
        __syncthreads();

        double acc[4], obs[BN];
        accumulate(cols2, wdt, 0u, true, acc);
        score_of(acc, obs);
        unsigned int cneg[BN], cpos[BN];
#pragma unroll
        for (int c = 0; c < BN; ++c) cneg[c] = cpos[c] = 0;

        for (int64_t p = p_begin; p < p_end; ++p) {
            const int64_t rel = p - p_begin;
            const uint32_t cur_base = (rel & 1) ? cur_bytes1 : cur_bytes0;
            uint4 nxt = make_uint4(0, 0, 0, 0);
            const bool fetch = (p + 1 < p_end) && (static_cast<int>(threadIdx.x) < vec_per_row);
            if (fetch) nxt = reinterpret_cast<const uint4_alias *>(cur16 + (p + 1) * stride16)[threadIdx.x];

            accumulate(cols2, wdt, cur_base, false, acc);
            double sc[BN];
            score_of(acc, sc);
#pragma unroll
            for (int c = 0; c < BN; ++c) {
                cneg[c] += sc[c] <= obs[c];                              // safe_extras.py:65
                cpos[c] += sc[c] >= obs[c];                              // safe_extras.py:66
            }
            if (vec_per_row > NT)
                for (int v = threadIdx.x + NT; v < vec_per_row; v += NT)
                    if (p + 1 < p_end)
                        reinterpret_cast<uint4_alias *>(CUR + ((rel + 1) & 1) * stride16)[v] =
                            reinterpret_cast<const uint4_alias *>(cur16 + (p + 1) * stride16)[v];
            if (fetch) reinterpret_cast<uint4_alias *>(CUR + ((rel + 1) & 1) * stride16)[threadIdx.x] = nxt;
            __syncthreads();
        }

        const int64_t spos = s * 64 + lane;
#pragma unroll
        for (int c = 0; c < BN; ++c) {
            const int64_t jc = static_cast<int64_t>(tile) * BN + c;
            if (active && jc < mloc) {
                const unsigned int v = (cpos[c] << 16) | cneg[c];
                if (v) atomicAdd(&counts[jc * n_pad + spos], v);
                if (row >= 0 && p_begin == 0) ns_out[static_cast<int64_t>(row) * mloc + jc] = obs[c];
            }
        }
        __syncthreads();
    }
}

// row-major tiles of 32 bytes per node for k_permtest_lds: sum -> 4 columns of B0; z-score ->
// (B0, B0^2 squared in B's dtype, not-NaN, 0) of one column; row n = zeros (SELL padding)
template <typename T, bool Z>
__global__ __launch_bounds__(256) void k_tile32_prep(const void *__restrict__ raw, int64_t n, int64_t rs, int64_t cs,
                                                     int64_t col0, int64_t mloc, int64_t n_tiles, double *__restrict__ out) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= n_tiles * (n + 1)) return;
    const int64_t tile = idx / (n + 1), r = idx % (n + 1);
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (r < n) {
        if (!Z) {
            for (int c = 0; c < 4; ++c) {
                const int64_t j = tile * 4 + c;
                if (j < mloc) {
                    const T x = reinterpret_cast<const T *>(raw)[r * rs + (col0 + j) * cs];
                    if (x == x) v[c] = static_cast<double>(x);
                }
            }
        } else if (tile < mloc) {
            const T x = reinterpret_cast<const T *>(raw)[r * rs + (col0 + tile) * cs];
            if (x == x) {
                v[0] = static_cast<double>(x);
                v[1] = static_cast<double>(static_cast<T>(x * x));
                v[2] = 1.0;
            }
        }
    }
    double *o = out + idx * 4;
    o[0] = v[0];
    o[1] = v[1];
    o[2] = v[2];
    o[3] = v[3];
}

__global__ void k_u32_to_f64(const unsigned int *__restrict__ in, double *__restrict__ out, int64_t count) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < count) out[i] = static_cast<double>(in[i]);
}

// neighborhood_size = A . nodes_not_nan (safe.py:587-588)
// one wave per row: the member list is read coalesced
__global__ __launch_bounds__(256) void k_nbr_size(const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                  const uint8_t *__restrict__ row_flags, int64_t n, double *__restrict__ out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    int c = 0;
    for (int32_t e = row_ptr[i] + lane; e < row_ptr[i + 1]; e += 64) c += row_flags[col[e]] != 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (lane == 0) out[i] = static_cast<double>(c);
}

// --------------------------------------------------------------------------------------
// K4: hypergeometric upper tail P[H >= X] = sf(X - 1) with the semantics of
// scipy.stats.hypergeom.sf as called at safe.py:596 (rv_discrete.sf wrapper: argument
// check -> NaN, below support -> 1, at/after the top of the support -> 0, result clipped
// to [0,1]).  pmf from a host-built log-factorial table, tail by the term recurrence,
// summed on the side of the mode that keeps the sum short (complemented when needed).
// --------------------------------------------------------------------------------------
__device__ double hyp_sf(const double *__restrict__ lf, double x_hits, double pop_d, double good_d, double draws_d) {
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    // _argcheck of scipy's hypergeom: integers, 0 <= good <= pop, 0 <= draws <= pop
    if (!(pop_d >= 0.0) || !(good_d >= 0.0) || !(draws_d >= 0.0) || good_d > pop_d || draws_d > pop_d ||
        pop_d != floor(pop_d) || good_d != floor(good_d) || draws_d != floor(draws_d))
        return qnan;
    const double k_d = x_hits - 1.0;
    if (k_d != k_d) return qnan;
    const int64_t pop = static_cast<int64_t>(pop_d), good = static_cast<int64_t>(good_d),
                  draws = static_cast<int64_t>(draws_d);
    const int64_t lo = draws - (pop - good) > 0 ? draws - (pop - good) : 0;
    const int64_t hi = good < draws ? good : draws;
    if (k_d < static_cast<double>(lo)) return 1.0;
    if (k_d >= static_cast<double>(hi)) return 0.0;
    // inside the support a non-integer k is NaN (SciPy 1.15's Boost tail, pinned by tests/golden/fdr.npz `hyp_nan`:
    // half-integer hit counts of a forced-hypergeometric call on non-0/1 data); outside it the rules above win
    if (k_d != floor(k_d)) return qnan;
    const int64_t k = static_cast<int64_t>(k_d);
    const double eps = 2.220446049250313e-16;
    const double mode = floor(static_cast<double>(good + 1) * static_cast<double>(draws + 1) / static_cast<double>(pop + 2));
    // the loops count in doubles (exact: integers below 2^53): 64-bit integer -> double
    // conversions would cost more than the recurrence itself
    const int lo_i = static_cast<int>(lo), hi_i = static_cast<int>(hi);
    const double rest_d = pop_d - good_d - draws_d;                     // may be negative; rest + t >= 0 inside the support
    double result;
    if (static_cast<double>(k) < mode) {
        // lower tail cdf(k) downwards from k, then complement
        int t = static_cast<int>(k);
        double td = static_cast<double>(t);
        double term = exp(hyp_logpmf(lf, t, pop, good, draws));
        double sum = term;
        while (t > lo_i && term > eps) {
            // pmf(t-1) / pmf(t)
            term = term * (td * (rest_d + td)) * fast_rcp((good_d - td + 1.0) * (draws_d - td + 1.0));
            sum += term;
            --t;
            td -= 1.0;
        }
        result = 1.0 - sum;
    } else {
        int t = static_cast<int>(k) + 1;
        double td = static_cast<double>(t);
        double term = exp(hyp_logpmf(lf, t, pop, good, draws));
        double sum = term;
        while (t < hi_i && term > eps * sum) {
            // pmf(t+1) / pmf(t)
            term = term * ((good_d - td) * (draws_d - td)) * fast_rcp((td + 1.0) * (rest_d + td + 1.0));
            sum += term;
            ++t;
            td += 1.0;
        }
        result = sum;
    }
    return result < 0.0 ? 0.0 : (result > 1.0 ? 1.0 : result);
}

__global__ __launch_bounds__(256) void k_hypergeom_tail(const double *__restrict__ hits, const double *__restrict__ nb_size,
                                                        const double *__restrict__ col_sum, int64_t col0, int64_t n,
                                                        int64_t mloc, double pop, const double *__restrict__ lf,
                                                        double p_cut, double *__restrict__ pvalues_pos,
                                                        double *__restrict__ nes_out, double *__restrict__ nes_binary,
                                                        unsigned int *__restrict__ enriched) {
    // 2-D: x over columns (coalesced), y over rows
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 64 + (threadIdx.x & 63);
    const int64_t i = static_cast<int64_t>(blockIdx.y) * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    bool hit = false;
    if (c < mloc) {
        const double p = hyp_sf(lf, hits[i * mloc + c], pop, col_sum[col0 + c], nb_size[i]);
        const double nes = -log10(p);                       // safe.py:608
        hit = p < p_cut;                                    // safe.py:468-470 (nes_p_cut)
        pvalues_pos[i * mloc + c] = p;
        nes_out[i * mloc + c] = nes;
        nes_binary[i * mloc + c] = hit ? 1.0 : 0.0;
    }
    if (hit) atomicAdd(&enriched[c], 1u);
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
struct Tiles {
    double *bt = nullptr;
    int64_t n_tiles = 0;
    int bn = 0;
    int planes = 1;
};

static int build_tiles(safe_ctx *ctx, safe_attr *attr, int64_t col0, int64_t col1, bool z, Tiles *tiles) {
    const int64_t n = attr->n, mloc = col1 - col0;
    const int bn = z ? 8 : 16;
    const int planes = z ? 3 : 1;
    tiles->bn = bn;
    tiles->planes = planes;
    tiles->n_tiles = ceil_div(mloc, bn);
    const int64_t total = tiles->n_tiles * (n + 1) * bn;
    SAFE_TRY(dev_alloc(&tiles->bt, static_cast<size_t>(total) * planes));
    const dim3 grid(ceil_div(total, 256)), block(256);
    const bool f32 = attr->dtype == SAFE_DTYPE_F32;
#define LAUNCH_PREP(T, BN, PL)                                                                                   \
    hipLaunchKernelGGL((k_tile_prep<T, BN, PL>), grid, block, 0, ctx->stream, attr->raw, n, attr->row_stride,   \
                       attr->col_stride, col0, mloc, tiles->n_tiles, tiles->bt)
    if (z) {
        if (f32) LAUNCH_PREP(float, 8, 3);
        else LAUNCH_PREP(double, 8, 3);
    } else {
        if (f32) LAUNCH_PREP(float, 16, 1);
        else LAUNCH_PREP(double, 16, 1);
    }
#undef LAUNCH_PREP
    SAFE_HIP_CHECK(hipGetLastError());
    return SAFE_OK;
}

static int check_cols(const safe_nbr *nbr, const safe_attr *attr, int64_t col0, int64_t col1, const char *who) {
    SAFE_REQUIRE(nbr && attr, "%s: NULL handle", who);
    SAFE_REQUIRE(nbr->n == attr->n, "%s: membership is %lld x %lld but the attribute matrix has %lld rows", who,
                 (long long)nbr->n, (long long)nbr->n, (long long)attr->n);
    SAFE_REQUIRE(0 <= col0 && col0 < col1 && col1 <= attr->m, "%s: column range [%lld,%lld) outside [0,%lld)", who,
                 (long long)col0, (long long)col1, (long long)attr->m);
    return SAFE_OK;
}

// launches the gather kernel; table == NULL / n_perm == 0 gives the observed score only
static int launch_gather(safe_ctx *ctx, safe_nbr *nbr, const Tiles &tiles, const int32_t *table, int64_t n_perm,
                         int64_t mloc, bool z, const PermOut &out) {
    ctx->last_kernel.name.clear();
    // slice groups: enough workgroups to fill the chip several times over
    int n_groups = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(nbr->n_slices, 4),
                                                                           ceil_div(ctx->num_cu * 8, tiles.n_tiles))));
    const int64_t blocks = ceil_div(tiles.n_tiles, 8) * 8 * n_groups;
    SAFE_HIP_CHECK(hipEventRecord(ctx->k0, ctx->stream));
    if (z)
        hipLaunchKernelGGL((k_permtest_gather<8, true>), dim3(blocks), dim3(256), 0, ctx->stream, nbr->sell_row,
                           nbr->slice_off, nbr->slice_width, nbr->sell_col, nbr->n_slices, nbr->n, tiles.bt,
                           tiles.n_tiles, n_groups, table, n_perm, mloc, out);
    else
        hipLaunchKernelGGL((k_permtest_gather<16, false>), dim3(blocks), dim3(256), 0, ctx->stream, nbr->sell_row,
                           nbr->slice_off, nbr->slice_width, nbr->sell_col, nbr->n_slices, nbr->n, tiles.bt,
                           tiles.n_tiles, n_groups, table, n_perm, mloc, out);
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(hipEventRecord(ctx->k1, ctx->stream));
    ctx->last_kernel.name = z ? "k_permtest_gather<8,true>" : "k_permtest_gather<16,false>";
    ctx->last_kernel.launches = -1;   // resolved lazily by safe_last_kernel_stats... see finish_kernel_timing
    return SAFE_OK;
}

static size_t scatter_lds_bytes(int64_t n) {
    return (2 * static_cast<size_t>(n) + 4) * sizeof(unsigned int) + ((static_cast<size_t>(n) + 1) & ~size_t(1)) * sizeof(unsigned short);
}

static int launch_scatter(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, safe_perms *perms, int64_t col0,
                          int64_t col1, const PermOut &out) {
    const int64_t n = nbr->n, mloc = col1 - col0;
    SAFE_TRY(nbr_build_transpose(nbr));
    SAFE_TRY(perms_build_inverse(perms));
    // attributes in descending support size: dynamic queue = longest-processing-time-first
    std::vector<int32_t> order(mloc);
    for (int64_t j = 0; j < mloc; ++j) order[j] = static_cast<int32_t>(j);
    const int32_t *sp = attr->h_sup_ptr.data() + col0;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return sp[a + 1] - sp[a] > sp[b + 1] - sp[b]; });
    int32_t *d_order = nullptr;
    unsigned int *d_queue = nullptr;
    SAFE_TRY(dev_alloc(&d_order, mloc));
    SAFE_TRY(dev_alloc(&d_queue, 1));
    SAFE_HIP_CHECK(hipMemcpyAsync(d_order, order.data(), mloc * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemsetAsync(d_queue, 0, sizeof(unsigned int), ctx->stream));
    const size_t lds_bytes = scatter_lds_bytes(n);
    const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds_bytes)));
    const int64_t blocks = std::min<int64_t>(mloc, static_cast<int64_t>(ctx->num_cu) * per_cu);
    constexpr int waves = 4;
#define LAUNCH_SCATTER(W)                                                                                          \
    do {                                                                                                           \
        SAFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_permtest_scatter<W>),                  \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes))); \
        SAFE_HIP_CHECK(hipEventRecord(ctx->k0, ctx->stream));                                                      \
        hipLaunchKernelGGL(k_permtest_scatter<W>, dim3(blocks), dim3(64 * W), lds_bytes, ctx->stream, n, perms->count, \
                           perms->inverse_t, perms->inv_stride, nbr->at_ptr, nbr->at_col, attr->sup_ptr, attr->sup_row, \
                           col0, d_order, mloc, d_queue, out);                                                                          \
    } while (0)
    static_assert(waves == 4, "k_permtest_scatter is launched with four waves");
    LAUNCH_SCATTER(4);
#undef LAUNCH_SCATTER
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(hipEventRecord(ctx->k1, ctx->stream));
    ctx->last_kernel.name = "k_permtest_scatter";
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));    // order (host vector) and temporaries
    (void)hipFree(d_order);
    (void)hipFree(d_queue);
    return SAFE_OK;
}

static size_t bits_lds_bytes(int64_t n, int64_t stride16) {
    const size_t t_words = 2 * ((static_cast<size_t>(n) + 2) & ~size_t(1));
    return (t_words + static_cast<size_t>(stride16) + 4) * sizeof(unsigned int);
}

// the pre-permuted form with sixteen-wave workgroups (k_permtest_bits_pre<8, 16, 2>): networks beyond the 16-bit LDS offsets whose
// word column still fits a CU's LDS (8 bytes per node: N <= 20 470) and whose doubled ids fit 16 bits
static size_t bits_pre_lds_bytes(int64_t n) { return (2 * ((static_cast<size_t>(n) + 2) & ~size_t(1)) + 4) * sizeof(unsigned int); }
// ... or with neighborhoods of 1024 .. 2047 members at any size (eleven levels of the vertical sums, which only this form has)
static bool bits_pre_wide_applicable(int64_t n, int64_t max_count) {
    const char *pe = getenv("SAFE_HIP_BITS_PRE");
    return ((n + 1) * 8 >= 65536 || max_count >= (1 << BT_LV)) && max_count < (2 << BT_LV) && n < 32768 &&
           bits_pre_lds_bytes(n) <= 160 * 1024 && !(pe && !strcmp(pe, "0"));
}

// ... and the half-word form (k_permtest_bits_pre32): the networks whose full word column no longer fits, up to the 2 * id lists' 32 767
static size_t bits_half_lds_bytes(int64_t n) { return (((static_cast<size_t>(n) + 4) & ~size_t(3)) + 4) * sizeof(unsigned int); }
static bool bits_pre_half_applicable(int64_t n, int64_t max_count) {
    const char *pe = getenv("SAFE_HIP_BITS_PRE");
    return bits_pre_lds_bytes(n) > 160 * 1024 && bits_half_lds_bytes(n) <= 160 * 1024 && n < 32768 && max_count < (2 << BT_LV) &&
           !(pe && !strcmp(pe, "0"));
}

enum PermPath { PATH_GATHER = 0, PATH_SCATTER = 1, PATH_BITS = 2 };

// Picks the kernel form.  The two integer forms need 'sum' scores of 0/1 data (exact in
// integers); scatter additionally wants sparse attributes (its work is nnz(A)*nnz(B)/N per
// permutation, the bit-sliced form's is nnz(A)*M/64 word-adds).
static PermPath choose_path(const safe_ctx *ctx, const safe_nbr *nbr, safe_attr *attr, int64_t n_perm, bool z) {
    const char *force = getenv("SAFE_HIP_FORCE_PATH");
    if (force && !strcmp(force, "gather")) return PATH_GATHER;
    if (z || n_perm < 1 || n_perm > 65535) return PATH_GATHER;
    if (safe_attr_prepare(attr) != SAFE_OK || attr->n_other != 0) return PATH_GATHER;
    if (nbr->n >= 65535) return PATH_GATHER;
    const bool bits_ok = nbr->sell_col2 != nullptr &&
                         ((nbr->max_count < (1 << BT_LV) && bits_lds_bytes(nbr->n, (nbr->n + 8) / 8 * 8) <= 160 * 1024) ||
                          bits_pre_wide_applicable(nbr->n, nbr->max_count) || bits_pre_half_applicable(nbr->n, nbr->max_count));
    const bool scatter_ok = nbr->max_count < SC_EPOCH && scatter_lds_bytes(nbr->n) <= 160 * 1024;
    if (force && !strcmp(force, "bits") && bits_ok) return PATH_BITS;
    if (force && !strcmp(force, "scatter") && scatter_ok) return attr_build_support(attr) == SAFE_OK ? PATH_SCATTER : PATH_GATHER;
    if (bits_ok) return PATH_BITS;
    if (scatter_ok && attr_build_support(attr) == SAFE_OK) return PATH_SCATTER;
    return PATH_GATHER;
}

// ev[2c], ev[2c+1] bracket launch c (recorded on alternating streams, all complete): sum of the durations, their count, and the
// union of the intervals -- consecutive launches overlap, so the sum exceeds the time the GPU spent on them
int kernel_stat_from_events(safe_ctx *ctx, hipEvent_t *ev, int64_t n_launch) {
    double busy = 0.0, covered_to = 0.0;
    for (int64_t c = 0; c < n_launch; ++c) {
        float ms = 0.f, t0 = 0.f;
        SAFE_HIP_CHECK(hipEventElapsedTime(&ms, ev[2 * c], ev[2 * c + 1]));
        if (c) SAFE_HIP_CHECK(hipEventElapsedTime(&t0, ev[0], ev[2 * c]));
        ctx->last_kernel.total_ms += ms;
        ctx->last_kernel.launches += 1;
        const double a = std::max<double>(t0, covered_to), b = static_cast<double>(t0) + ms;
        if (b > a) busy += b - a;
        covered_to = std::max(covered_to, b);
    }
    ctx->last_kernel.busy_ms = busy;
    return SAFE_OK;
}

// SAFE_HIP_BITS_DBG=128: what the waves of the last call's launches did (g_blk_trace), on stderr
static void blk_trace_dump(int n_launch) {
    unsigned int cnt = 0;
    if (hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(g_blk_trace_n), sizeof(cnt)) != hipSuccess) return;
    cnt = std::min<unsigned int>(cnt, BLK_TRACE_MAX);
    std::vector<unsigned long long> rec(static_cast<size_t>(cnt) * 8);
    if (cnt && hipMemcpyFromSymbol(rec.data(), HIP_SYMBOL(g_blk_trace), rec.size() * sizeof(unsigned long long)) != hipSuccess) return;
    const unsigned int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_blk_trace_n), &zero, sizeof(zero));
    if (!cnt) return;
    double busy = 0.0, work = 0.0;
    std::vector<double> dur;
    for (unsigned int i = 0; i < cnt; ++i) {
        const double d = static_cast<double>(rec[8 * i + 2] - rec[8 * i + 1]);
        busy += d;
        dur.push_back(d);
        work += std::max<double>(1.0, static_cast<double>(rec[8 * i + 3] >> 32)) * static_cast<double>(rec[8 * i + 3] & 0xFFFFFFFFu);
    }
    std::sort(dur.begin(), dur.end());
    fprintf(stderr, "[blk trace] %u wave-tasks over %d launches; wave-task duration min %.0f median %.0f p90 %.0f max %.0f kclk; "
            "%.3g block-iterations, %.0f clocks per block-iteration per wave (4 waves share a SIMD)\n",
            cnt, n_launch, dur.front() / 1e3, dur[dur.size() / 2] / 1e3, dur[dur.size() * 9 / 10] / 1e3, dur.back() / 1e3, work, busy / std::max(work, 1.0));
    double cls_busy[5] = {0, 0, 0, 0, 0}, cls_work[5] = {0, 0, 0, 0, 0};
    for (unsigned int i = 0; i < cnt; ++i) {
        const double nb = static_cast<double>(rec[8 * i + 3] >> 32), np = static_cast<double>(rec[8 * i + 3] & 0xFFFFFFFFu);
        const int c = nb <= 1 ? 0 : nb <= 7 ? 1 : nb <= 31 ? 2 : nb <= 63 ? 3 : 4;
        cls_busy[c] += static_cast<double>(rec[8 * i + 2] - rec[8 * i + 1]);
        cls_work[c] += std::max(nb, 1.0) * np;
    }
    const char *names[5] = {"1 block", "2-7 blocks", "8-31 blocks", "32-63 blocks", ">= 64 blocks"};
    double ph[5][3] = {}, cls_n[5] = {0, 0, 0, 0, 0};                 // per class: table load + barrier | counting | flush, kclk per wave-task
    for (unsigned int i = 0; i < cnt; ++i) {
        const double nb = static_cast<double>(rec[8 * i + 3] >> 32);
        const int c = nb <= 1 ? 0 : nb <= 7 ? 1 : nb <= 31 ? 2 : nb <= 63 ? 3 : 4;
        ph[c][0] += static_cast<double>(rec[8 * i + 4] - rec[8 * i + 1]);
        ph[c][1] += static_cast<double>(rec[8 * i + 5] - rec[8 * i + 4]);
        ph[c][2] += static_cast<double>(rec[8 * i + 2] - rec[8 * i + 5]);
        cls_n[c] += 1.0;
    }
    for (int c = 0; c < 5; ++c)
        if (cls_work[c] > 0)
            fprintf(stderr, "[blk trace]   slices of %-12s: %.3g block-iterations, %.0f clocks each; %.0f wave-tasks: table %.1f | counting %.1f | flush %.1f kclk each\n",
                    names[c], cls_work[c], cls_busy[c] / cls_work[c], cls_n[c], ph[c][0] / cls_n[c] / 1e3, ph[c][1] / cls_n[c] / 1e3, ph[c][2] / cls_n[c] / 1e3);
}

static int launch_bits(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, safe_perms *perms, int64_t col0, int64_t col1,
                       const PermOut &out) {
    const int64_t n = nbr->n, mloc = col1 - col0, n_wg = ceil_div(mloc, 64), P = perms->count;
    safe_trace("launch_bits: enter");
    uint2 *d_bits = nullptr;
    SAFE_TRY(ctx_scratch(ctx, 1, static_cast<size_t>(n_wg) * (n + 1) * sizeof(uint2), reinterpret_cast<void **>(&d_bits)));
    launch_bits_prep(ctx, attr, col0, mloc, n_wg, d_bits);
    safe_trace("launch_bits: prep launched");
    // The permutations are consumed in launches of `span` permutations so that the host's draw
    // stream for the next span overlaps this span's kernel.  Inside a launch, tasks = (word
    // group, group of 4 adjacent slices, permutation sub-range), sized to about equal cost
    // with several per workgroup slot, heaviest first for the dynamic queue.
    int64_t span = 1;
    // launches follow the stream's stages, one launch per stage (a launch over m stages after the start-up has fewer tails and
    // less per-task fixed cost -- kernel time per 1000 permutations 4.06 -> 3.94 / 3.89 ms at m = 2 / 3 in round 3 -- but the
    // step gets LONGER, 5.2 -> 5.5 / 5.8 ms: a merged launch waits for its last stage's draws)
    const int merge = 1;                               // (round 6, kernel-bound step: 2 / 3 stages per launch 3.13 -> 3.33-3.42 / 3.65 ms, kernels busy 2.54 -> 2.8)
    const std::vector<int64_t> starts = perm_launch_starts(perms, &span, merge);
    // Exchange overlap of the sharded step (safe_set_exchange_chunks): the LAST permutations -- the tail -- run as one launch per
    // COLUMN chunk over all of the tail instead of one launch per stage over all columns.  A chunk's counters are then final when
    // its launch ends, and its all-gather overlaps the launches of the later chunks (xc_events; sharding.ChunkedExchange).
    // Tasks keep their size (a chunk has 1/K of the word groups and more permutations), so the kernels' work is the same; the
    // permuted member lists of the tail are built stage by stage beside the earlier launches and shared by the chunks' launches.
    // MEASURED (configs[1] on one MI355X with a one-rank RCCL group, tools/probe/xchg_ab.sh; plain step 3.6 ms, kernels 2.55 ms):
    //   * seeded stream, tail = 0.4 of the permutations: step +0.63 ms.  The stage launches run in step with the host's draws (a
    //     stage every 0.3 ms, kernels 0.5 ms behind), and a tail's launches cannot start before the LAST draw: every stage
    //     moved into the tail waits for the end of the stream.
    //   * seeded, tail = the last stage (waits for nothing new): step +0.7 ms.  A chunk of one stage has 1/K of the tasks: with
    //     the stages' task size 288 tasks for 960 slots (each chunk launch as long as a whole stage), with tasks cut to fill
    //     the slots 4.5 x as many task set-ups (table reload, counter flush): the four chunk launches took 0.72 ms for the
    //     0.31 ms of one launch.
    //   * unseeded (tables generated on the device, nothing to wait for), tail 0.2 / 0.4 / 1.0: kernels +0.2 .. +0.7 ms, step
    //     +1.2 ms: the chunks' copy-out and all-gather kernels crawl on the 16 CUs the persistent workgroups leave free.
    // It hides at most the exchange of K - 1 chunks (~0.9 ms of 1.2 ms at 8 ranks) and never paid for itself at this size, so the
    // tail is OFF by default (SAFE_HIP_XCHG_TAIL=<fraction of the permutations> switches it on; 1e-9 = the last stage); the
    // chunked exchange itself (sharding.ChunkedExchange) then runs its collectives right after the kernels, chunk by chunk
    // beside the derivation of the previous chunk's matrices.
    const bool pre_w = bits_pre_wide_applicable(n, nbr->max_count);     // the sixteen-wave pre-permuted form (k_permtest_bits_pre<8, 16, 2, ..>)
    const bool pre_h = bits_pre_half_applicable(n, nbr->max_count);     // ... with 32-attribute half words (k_permtest_bits_pre32): tasks per HALF word group
    const int64_t n_wgt = pre_h ? 2 * n_wg : n_wg;
    const bool blk_expected = [&] {
        const char *pe = getenv("SAFE_HIP_BITS_PRE"), *ke = getenv("SAFE_HIP_BITS_KERNEL");
        return !pre_w && !pre_h && (n + 1) * 8 < 65536 && !(pe && !strcmp(pe, "0")) && nbr->sell_col2b != nullptr && !(ke && !strcmp(ke, "pre"));
    }();
    int64_t n_major = static_cast<int64_t>(starts.size()) - 1, p_split = P, xc_wpc = 0;
    int xc_k = 0;
    ctx->xc_made = 0;
    if (ctx->xc_want >= 2 && ctx->xc_cols >= 64 && ctx->xc_cols % 64 == 0 && blk_expected && n_major >= 3 &&
        static_cast<int64_t>(std::min<int>(ctx->xc_want, safe_ctx::XC_MAX)) * ctx->xc_cols >= mloc) {
        double frac = 0.0;                              // (off unless asked for: see the measurements above)
        if (const char *e = getenv("SAFE_HIP_XCHG_TAIL")) frac = std::min(1.0, std::max(0.0, atof(e)));
        int64_t c_split = 1;
        while (c_split < n_major - 1 && static_cast<double>(starts[c_split]) < (1.0 - frac) * static_cast<double>(P)) ++c_split;
        if (frac > 0.0) {
            n_major = c_split;
            p_split = starts[c_split];
            xc_k = std::min<int>(ctx->xc_want, safe_ctx::XC_MAX);
            xc_wpc = ctx->xc_cols / 64;
        }
    }
    const int64_t n_tail = xc_k, tail_span = P - p_split;
    const size_t lds_bytes = bits_lds_bytes(n, perms->stride16);
    const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds_bytes)));
    const int64_t slots = static_cast<int64_t>(ctx->num_cu) * per_cu;
    // waves (= adjacent slices of a task) per workgroup: 4; 16 when the word column and the permutation rows leave room for one
    // workgroup per CU only (k_permtest_bits<.., 16>: N > 8190, where neither the pre-permuted nor the blocked lists apply)
    const int wv = (n + 1) * 8 < 65536 && !pre_w && !pre_h ? 4 : 16;
    const int64_t n_sg = ceil_div(nbr->n_slices, wv);
    std::vector<int64_t> sg_blocks(n_sg, 0);
    int64_t blocks_per_perm = 0;
    for (int64_t s = 0; s < nbr->n_slices; ++s) sg_blocks[s / wv] = std::max<int64_t>(sg_blocks[s / wv], nbr->h_slice_width[s] / 8);
    for (int64_t g = 0; g < n_sg; ++g) blocks_per_perm += std::max<int64_t>(sg_blocks[g], 1);
    // five waves per SIMD (k_permtest_bits_blk<.., 5>) when T fits five times into a CU's LDS; its wide classes count with five
    // levels, so their tasks hold at most 31 permutations
    const size_t lds_T = (2 * ((static_cast<size_t>(n) + 2) & ~size_t(1)) + 4) * sizeof(unsigned int);
    (void)lds_T;
    const bool occ5 = false;                             // (five waves per SIMD: built and measured in round 3, slower; the kernel form stays for reference)
    int tasks_per_slot = 1;                              // queue depth per workgroup slot: every task reloads T and flushes its counters (64 wave-atomics
                                                         // + two 32 x 32 bit transposes per wave: 11 % of the kernel at depth 2, tools/bits_ablate.py dbg=2), so
                                                         // as few tasks as fill the chip once -- depth 1 vs 2: seeded step 3.52 -> 3.41 ms, 10 000 unseeded
                                                         // permutations 25.8 -> 24.4 ms (tools/exp_ab.sh; round 3 had chosen 2 while the host stream bound the step)
    const int64_t tasks_per_wg = std::max<int64_t>(1, ceil_div(tasks_per_slot * slots, n_wgt));
    // One task list per DISTINCT launch size: the stream's stages are 32, 96, 128 ... and short last ones, and a list cut for 128
    // permutations leaves a 32-permutation launch with half-empty and empty tasks (each still reloads T): a 32-permutation launch
    // took 207 us, 6.5 us per permutation against 3.2 in the long launches.
    const int64_t min_ppt = 16, target_min = 256;        // floors of a task's size: permutations, block-permutations
    const int64_t max_ppt = 0;                           // (a cap on a task's permutations -- long launches with short tasks -- measured no gain)
    // SHORT launches (the ramp at the head of the stream, the stage behind the last draw: nothing else runs beside them) cost
    // ~150 us + 2 us per permutation alone on the chip (10 / 16 / 32 / 64 / 128 permutations: 160 / 200 / 186 / 266 / 413 us,
    // tools/r6/short_launch.sh): their heaviest slice group's tasks are the critical path (16 permutations x 40-49 blocks x
    // 420-700 clocks) and every wave-task carries 13 + 30 + 30-60 kclk of table load, start-up and counter flush.  Cutting
    // their tasks finer (4 or 8 permutations per task at least) shortened a lone 16-permutation launch to 170 / 145 us but the
    // seeded step got LONGER (3.16-3.34 -> 3.30-3.38 ms: twice the wave-tasks, each with its fixed part) -- not kept.
    struct TaskCost { int4 t; int64_t cost; };
    auto build_tasks = [&](int64_t span_c, int64_t w_lo = 0, int64_t w_hi = -1) {
        if (w_hi < 0) w_hi = n_wgt;
        // (a column chunk of the tail has fewer word groups: its tasks are cut so that ITS launch fills the slots once too -- with
        // the stage launches' task size a chunk of the last stage had 288 tasks for 960 slots and lasted as long as a whole stage)
        const int64_t tpw = w_hi - w_lo == n_wgt ? tasks_per_wg : std::max<int64_t>(1, ceil_div(tasks_per_slot * slots, std::max<int64_t>(1, w_hi - w_lo)));
        const int64_t target = std::max<int64_t>(target_min, blocks_per_perm * span_c / tpw);    // block-permutations per task
        std::vector<TaskCost> tc;
        for (int64_t g = 0; g < n_sg; ++g) {
            const int64_t bl = std::max<int64_t>(sg_blocks[g], 1);
            int64_t ppt_cap = !occ5 ? 255 : (bl * 8 > 56 ? 31 : 63);            // counter levels of the task's class: 8, or 5 / 6
            if (max_ppt > 0) ppt_cap = std::min<int64_t>(ppt_cap, max_ppt);
            int64_t ppt = std::min<int64_t>(std::min<int64_t>(span_c, ppt_cap), std::max<int64_t>(min_ppt, target / bl));
            const int64_t chunks = ceil_div(span_c, ppt);
            ppt = ceil_div(span_c, chunks);
            for (int64_t c = 0; c < chunks; ++c) {
                const int64_t p0 = c * ppt, p1 = std::min<int64_t>(span_c, p0 + ppt);
                for (int64_t w = w_lo; w < w_hi; ++w)
                    if (!pre_h || (w >> 1) * 64 + (w & 1) * 32 < mloc)      // (a last half word without columns has no task)
                    tc.push_back({make_int4(static_cast<int>(w), static_cast<int>(g), static_cast<int>(p0), static_cast<int>(p1)),
                                  bl * (p1 - p0)});
            }
        }
        std::stable_sort(tc.begin(), tc.end(), [](const TaskCost &a, const TaskCost &b) { return a.cost > b.cost; });
        std::vector<int4> out(tc.size());
        for (size_t i = 0; i < tc.size(); ++i) out[i] = tc[i].t;
        return out;
    };
    // the blocked kernel's per-XCD queues: the tasks of one (slice group, permutation range) -- one per word group, adjacent
    // after the stable sort -- go to one queue, (group, range) pairs dealt round-robin in heaviest-first order
    const bool xcd_queues = true;
    auto split_queues = [&](const std::vector<int4> &list, int (&off)[9]) {
        std::vector<std::vector<int4>> q(8);
        int64_t pair = -1;
        int last_g = -1, last_p0 = -1;
        for (size_t i = 0; i < list.size(); ++i) {
            if (!xcd_queues) pair = static_cast<int64_t>(i);
            else if (list[i].y != last_g || list[i].z != last_p0) {
                ++pair;
                last_g = list[i].y;
                last_p0 = list[i].z;
            }
            q[pair & 7].push_back(list[i]);
        }
        std::vector<int4> out;
        out.reserve(list.size());
        off[0] = 0;
        for (int k = 0; k < 8; ++k) {
            out.insert(out.end(), q[k].begin(), q[k].end());
            off[k + 1] = static_cast<int>(out.size());
        }
        return out;
    };
    const int64_t n_launch = n_major + n_tail;           // stage launches over all columns, then the tail's column chunks
    // the lists only depend on the handle and on these numbers: the handle keeps the last plan
    std::vector<int64_t> plan_key = {n_wg, slots, tasks_per_slot, occ5 ? 1 : 0, xcd_queues ? 1 : 0, min_ppt, target_min, n_launch, max_ppt,
                                     n_major, n_tail, xc_wpc, wv, pre_h ? 1 : 0, mloc};
    plan_key.insert(plan_key.end(), starts.begin(), starts.end());
    BitsTaskPlan &plan = nbr->bits_plan;
    if (plan.key != plan_key) {
        plan = BitsTaskPlan{};
        plan.launch_list.assign(std::max<int64_t>(n_launch, 1), 0);
        for (int64_t k = 0; k < n_tail; ++k) {             // one list per column chunk (possibly empty on a rank with fewer columns)
            const int64_t w_lo = std::min<int64_t>(k * xc_wpc, n_wg), w_hi = k + 1 == n_tail ? n_wg : std::min<int64_t>((k + 1) * xc_wpc, n_wg);
            BitsQueues bq{};
            const std::vector<int4> one = w_hi > w_lo ? split_queues(build_tasks(tail_span, w_lo, w_hi), bq.off) : std::vector<int4>();
            plan.launch_list[n_major + k] = static_cast<int64_t>(plan.list_span.size());
            plan.list_queues.push_back(bq);
            plan.list_span.push_back(-1 - k);                // (never matches a stage's span)
            plan.list_first.push_back(static_cast<int64_t>(plan.tasks.size()));
            plan.list_count.push_back(static_cast<int64_t>(one.size()));
            plan.tasks.insert(plan.tasks.end(), one.begin(), one.end());
        }
        for (int64_t c = 0; c < n_major; ++c) {
            const int64_t span_c = starts[c + 1] - starts[c];
            size_t k = 0;
            while (k < plan.list_span.size() && plan.list_span[k] != span_c) ++k;
            if (k == plan.list_span.size()) {
                BitsQueues bq{};
                const std::vector<int4> one = split_queues(build_tasks(span_c), bq.off);
                plan.list_queues.push_back(bq);
                plan.list_span.push_back(span_c);
                plan.list_first.push_back(static_cast<int64_t>(plan.tasks.size()));
                plan.list_count.push_back(static_cast<int64_t>(one.size()));
                plan.tasks.insert(plan.tasks.end(), one.begin(), one.end());
            }
            plan.launch_list[c] = static_cast<int64_t>(k);
        }
        // (plan.key is set LAST, below: an allocation or a wait that fails on the way leaves no key, and the next call rebuilds
        // the plan instead of uploading from a missing or stale pinned copy)
        const size_t bytes = plan.tasks.size() * sizeof(int4);
        if (nbr->bits_plan_pinned_bytes < bytes) {
            SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));                 // (an earlier pass may still be reading the old copy)
            if (nbr->bits_plan_pinned) SAFE_HIP_CHECK(hipHostFree(nbr->bits_plan_pinned));
            nbr->bits_plan_pinned = nullptr;
            nbr->bits_plan_pinned_bytes = 0;
            g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
            SAFE_HIP_CHECK(hipHostMalloc(&nbr->bits_plan_pinned, bytes + 4096, hipHostMallocDefault));
            nbr->bits_plan_pinned_bytes = bytes + 4096;
        } else {
            SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
        }
        memcpy(nbr->bits_plan_pinned, plan.tasks.data(), bytes);
        plan.key = plan_key;
    }
    const std::vector<int4> &tasks = plan.tasks;
    const std::vector<int64_t> &list_first = plan.list_first, &list_count = plan.list_count, &launch_list = plan.launch_list;
    const std::vector<BitsQueues> &list_queues = plan.list_queues;
    safe_trace("launch_bits: tasks built");
    int4 *d_tasks = nullptr;
    unsigned int *d_queue = nullptr;
    unsigned int *d_gl = nullptr;
    {
        void *ws = nullptr;       // tasks + queue words in one scratch buffer
        SAFE_TRY(ctx_scratch(ctx, 3, tasks.size() * sizeof(int4) + (8 * n_launch + 4) * sizeof(unsigned int), &ws));
        d_tasks = static_cast<int4 *>(ws);
        d_queue = reinterpret_cast<unsigned int *>(d_tasks + tasks.size());
    }
    const int64_t n_pad = nbr->n_slices * 64;
    SAFE_TRY(ctx_scratch(ctx, 0, static_cast<size_t>(n_pad) * mloc * sizeof(unsigned int), reinterpret_cast<void **>(&d_gl)));
    SAFE_HIP_CHECK(hipMemcpyAsync(d_tasks, nbr->bits_plan_pinned, tasks.size() * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemsetAsync(d_queue, 0, 8 * n_launch * sizeof(unsigned int), ctx->stream));
    SAFE_HIP_CHECK(hipMemsetAsync(d_gl, 0, static_cast<size_t>(n_pad) * mloc * sizeof(unsigned int), ctx->stream));
    safe_trace("launch_bits: buffers ready");
    int4 *const d_task_lists = d_tasks;
    const bool wide = P >= 1024;
    const bool narrow = true;                         // a task counts at most 255 permutations (ppt above): 8 counter levels do
    const bool scaled = (n + 1) * 8 < 65536;
    const char *pre_env = getenv("SAFE_HIP_BITS_PRE");
    const bool pre = (scaled && !(pre_env && !strcmp(pre_env, "0"))) || pre_w || pre_h;      // pre-permuted member lists
    const int id_shift = pre_h ? 0 : pre_w ? 1 : 3;                       // the lists hold id << id_shift
    const int64_t entries_pad = (nbr->sell_entries + 1024 + 255) / 256 * 256;    // tail: the kernels fetch ids two blocks (2 x 512) ahead
    // consecutive launches run on NS = 2 streams: a launch is as long as its longest task, the next one fills the slots its short
    // tasks leave.  Three or four launches in flight measured WORSE (unseeded 1000-permutation step 3.01 -> 3.19 -> 3.43 ms,
    // round 4): the later launches' workgroups take slots from the long tasks of the first
    constexpr int NS = 2;
    const int diag_banks = 0;                          // (k_permute_cols' diagnostic address form: conflict-free gathers measured no gain, round 5)
    uint16_t *d_ids[4] = {nullptr, nullptr, nullptr, nullptr};
    if (pre)
        for (int b = 0; b < NS; ++b)
            SAFE_TRY(ctx_scratch(ctx, b < 2 ? 4 + b : 10 + b, static_cast<size_t>(span) * entries_pad * sizeof(uint16_t),
                                 reinterpret_cast<void **>(&d_ids[b])));
    uint16_t *d_ids_tail = nullptr;                   // the tail's permuted member lists: built once, read by every column chunk's launch
    if (n_tail)
        SAFE_TRY(ctx_scratch(ctx, 18, static_cast<size_t>(tail_span) * entries_pad * sizeof(uint16_t), reinterpret_cast<void **>(&d_ids_tail)));
    hipStream_t kstreams[4] = {ctx->stream, ctx->side_stream, ctx->more_streams[0], ctx->more_streams[1]};
    const size_t lds_pre = (2 * ((static_cast<size_t>(n) + 2) & ~size_t(1)) + 4) * sizeof(unsigned int);
    // blocked member lists (k_permtest_bits_blk) unless SAFE_HIP_BITS_KERNEL=pre; SAFE_HIP_BITS_DBG=<mask>: diagnostic builds
    const char *kern_env = getenv("SAFE_HIP_BITS_KERNEL");
    const bool blk = pre && !pre_w && !pre_h && nbr->sell_col2b != nullptr && !(kern_env && !strcmp(kern_env, "pre"));
    // SAFE_HIP_BITS_DBG: variants of the blocked kernel.  32 / 128 / 256 / 384 / 512 / 640 give correct results (A/B: one-stage carry
    // ripple, ..., 256 = no id stream = no hidden registers, 512 = no half-block gather pipeline); 1 / 2 / 4 / 8 / 16 / 64 skip work
    // (WRONG results) and exist only in a library built with `make DIAG=1`.  Anything else is refused.
    int dbg = 0;
    if (const char *e = getenv("SAFE_HIP_BITS_DBG")) dbg = atoi(e);
    if (dbg == 0 && !occ5 && !safe_hidden_regs_checked()) {
        // the build could not disassemble the stream kernels (no llvm-objdump): nothing vouches for the registers they hide from
        // the compiler, so the form without them runs (the same counts, ~5 % slower)
        static bool told = false;
        if (!told) fprintf(stderr, "safepy_amd: this library was built without the hidden-register check; running k_permtest_bits_blk without the id stream\n");
        told = true;
        dbg = 256;
    }
    const void *blk_fn = nullptr;
    switch (dbg) {
        case 0: blk_fn = occ5 ? reinterpret_cast<const void *>(k_permtest_bits_blk_plain<8, 0, 5>) : reinterpret_cast<const void *>(k_permtest_bits_blk<8, 0>); break;
        case 32: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 32>); break;
        case 128: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 128>); break;
        case 256: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk_plain<8, 256, 4>); break;
        case 384: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk_plain<8, 384, 4>); break;
        case 512: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 512>); break;
        case 640: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 640>); break;
#ifdef SAFE_HIP_DIAG
        case 1: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 1>); break;
        case 2: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 2>); break;
        case 4: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 4>); break;
        case 7: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 7>); break;
        case 8: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 8>); break;
        case 16: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 16>); break;
        case 64: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 64>); break;
        case 1024: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 1024>); break;
        case 2048: blk_fn = reinterpret_cast<const void *>(k_permtest_bits_blk<8, 2048>); break;
#endif
        default:
            SAFE_REQUIRE(false, "SAFE_HIP_BITS_DBG=%d is not a variant of this build (correct variants: 0 32 128 256 384 512 640; the "
                                "work-skipping ones 1 2 4 7 8 16 64 need a library built with make DIAG=1)", dbg);
    }
    if (dbg & (95 | 1024 | 2048)) safe_warn_diagnostic("SAFE_HIP_BITS_DBG");
    uint32_t *d_obs = nullptr;
    if (blk) {
        SAFE_HIP_CHECK(hipFuncSetAttribute(blk_fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_pre)));
        // observed sums of every (word group, slice), once: the compare operand of every task, and `ns`
        SAFE_TRY(ctx_scratch(ctx, 7, static_cast<size_t>(n_wg) * nbr->n_slices * 2 * BT_LV * 64 * sizeof(uint32_t),
                             reinterpret_cast<void **>(&d_obs)));
        const size_t lds_obs = lds_pre + (out.ns ? 4 * 64 * 66 * sizeof(uint16_t) : 0);            // + the waves' transposition tiles
        SAFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bits_observed), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds_obs)));
        hipLaunchKernelGGL(k_bits_observed, dim3(n_wg, ceil_div(nbr->n_slices, 4)), dim3(256), lds_obs, ctx->stream, n, nbr->sell_row,
                           nbr->slice_off, nbr->slice_width, nbr->sell_col2b, nbr->n_slices, d_bits, mloc, d_obs, out.ns);
        SAFE_HIP_CHECK(hipGetLastError());
    }
    const bool lv11 = nbr->max_count >= (1 << BT_LV);
    const void *pre_w_fn = pre_h ? (lv11 ? reinterpret_cast<const void *>(k_permtest_bits_pre32<8, 16, BT_LV + 1>)
                                         : reinterpret_cast<const void *>(k_permtest_bits_pre32<8, 16, BT_LV>))
                                 : (lv11 ? reinterpret_cast<const void *>(k_permtest_bits_pre<8, 16, 2, BT_LV + 1>)
                                         : reinterpret_cast<const void *>(k_permtest_bits_pre<8, 16, 2>));
    const size_t lds_pre_w = pre_h ? bits_half_lds_bytes(n) : lds_pre;
    if (pre_w || pre_h)
        SAFE_HIP_CHECK(hipFuncSetAttribute(pre_w_fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_pre_w)));
    else if (pre)
        SAFE_HIP_CHECK(hipFuncSetAttribute(narrow ? reinterpret_cast<const void *>(k_permtest_bits_pre<8>)
                                           : wide ? reinterpret_cast<const void *>(k_permtest_bits_pre<16>)
                                                  : reinterpret_cast<const void *>(k_permtest_bits_pre<10>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_pre)));
    const void *kfn = wide ? (scaled ? reinterpret_cast<const void *>(k_permtest_bits<16, true>)
                                     : reinterpret_cast<const void *>(k_permtest_bits<16, false>))
                           : (scaled ? reinterpret_cast<const void *>(k_permtest_bits<10, true>)
                                     : reinterpret_cast<const void *>(k_permtest_bits<10, false>));
    SAFE_REQUIRE(pre || lds_bytes <= 160 * 1024, "launch_bits: the word column and the permutation rows of %lld nodes do not fit a CU's LDS",
                 static_cast<long long>(n));
    if (!pre) SAFE_HIP_CHECK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
    if (wv == 16 && !pre)         // (a task holds at most 255 permutations -- ppt_cap above --: eight counter levels, whatever P is)
        SAFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_permtest_bits<8, false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds_bytes)));
    ctx->last_kernel.name = blk ? "k_permtest_bits_blk" : pre ? "k_permtest_bits_pre" : "k_permtest_bits";
    ctx->last_kernel.total_ms = 0.0;
    ctx->last_kernel.busy_ms = 0.0;
    ctx->last_kernel.launches = 0;
    hipEvent_t *ev = nullptr, *plain = nullptr;                   // pooled on the context
    SAFE_TRY(ctx_events(ctx, true, 2 * n_launch, &ev));
    SAFE_TRY(ctx_events(ctx, false, 12, &plain));                 // 0: inputs ready, 1..3: end join, 4..7: join before the tail, 8: tail lists
    SAFE_REQUIRE(!n_tail || blk, "launch_bits: column-chunked tail without the blocked kernel");
    for (int64_t k = 0; k < n_tail; ++k)
        if (!ctx->xc_events[k]) SAFE_HIP_CHECK(hipEventCreateWithFlags(&ctx->xc_events[k], hipEventDisableTiming));
    // consecutive spans alternate between two streams so the tail of one launch (a few long
    // tasks) overlaps the head of the next; both wait for the inputs prepared on ctx->stream
    hipEvent_t ready = plain[0];
    SAFE_HIP_CHECK(hipEventRecord(ready, ctx->stream));
    for (int b = 1; b < NS; ++b) SAFE_HIP_CHECK(hipStreamWaitEvent(kstreams[b], ready, 0));
    hipStream_t tail_ps = ctx->more_streams[1];
    int xc_rc = SAFE_OK;
    auto chunk_final = [&](int64_t c, hipStream_t ks) {          // behind chunk launch c on ks: the event that says its counters are final
        for (int b = 0; b < NS; ++b)
            if (kstreams[b] != ks && hipStreamWaitEvent(ks, plain[4 + b], 0) != hipSuccess) xc_rc = SAFE_E_HIP;
        if (hipEventRecord(ctx->xc_events[c - n_major], ks) != hipSuccess) xc_rc = SAFE_E_HIP;
    };
    for (int64_t c = 0; c < n_launch; ++c) {
        const bool tail = c >= n_major;                                     // a column chunk over the tail's permutations
        const int64_t p_base = tail ? p_split : starts[c], p_limit = tail ? P : starts[c + 1];
        const int64_t n_tasks = list_count[launch_list[c]];                 // this launch size's task list
        const int4 *d_tasks = d_task_lists + list_first[launch_list[c]];
        const int64_t blocks = std::min<int64_t>(n_tasks, slots);
        hipStream_t ks = kstreams[c % NS];
        if (tail && c == n_major) {
            // the tail's permuted member lists: one k_permute_cols per pipeline stage as its tables arrive, on a stream of their
            // own beside the stage launches (in one piece after the last draw they were 0.1 ms of idle GPU)
            const int64_t n_stage = static_cast<int64_t>(starts.size()) - 1;
            SAFE_HIP_CHECK(hipStreamWaitEvent(tail_ps, ready, 0));
            for (int64_t t = n_major; t < n_stage; ++t) {
                SAFE_TRY(perms_wait(perms, starts[t + 1], tail_ps));
                hipLaunchKernelGGL(k_permute_cols, dim3(ceil_div(entries_pad, 4096), starts[t + 1] - starts[t]), dim3(256),
                                   static_cast<size_t>(perms->stride16) * sizeof(uint16_t), tail_ps, perms->table16, perms->stride16,
                                   nbr->sell_col2b, nbr->sell_entries, entries_pad, starts[t], starts[t + 1] - starts[t],
                                   static_cast<uint32_t>(8 * n), d_ids_tail + (starts[t] - p_split) * entries_pad, diag_banks, 3);
                SAFE_HIP_CHECK(hipGetLastError());
            }
            SAFE_HIP_CHECK(hipEventRecord(plain[8], tail_ps));
            // a chunk's counters are final once ITS launch and every stage launch have ended: the chunk launches themselves do not
            // wait for the stage launches of the other stream (their tails overlap as everywhere else) -- the chunk's event does
            for (int b = 0; b < NS; ++b) {
                SAFE_HIP_CHECK(hipEventRecord(plain[4 + b], kstreams[b]));
                SAFE_HIP_CHECK(hipStreamWaitEvent(kstreams[b], plain[8], 0));
            }
        }
        if (!tail) SAFE_TRY(perms_wait(perms, p_limit, ks));            // host draws + table kernels for this span
        safe_trace("launch_bits: span tables enqueued");
        if (pre) {
            if (!tail)
                hipLaunchKernelGGL(k_permute_cols, dim3(ceil_div(entries_pad, 4096), p_limit - p_base), dim3(256),
                                   static_cast<size_t>(perms->stride16) * sizeof(uint16_t), ks, perms->table16,
                                   perms->stride16, blk ? nbr->sell_col2b : nbr->sell_col2, nbr->sell_entries, entries_pad, p_base,
                                   p_limit - p_base, static_cast<uint32_t>(static_cast<uint64_t>(n) << id_shift), d_ids[c % NS], diag_banks, id_shift);
            SAFE_HIP_CHECK(hipEventRecord(ev[2 * c], ks));
            // the kernel's workgroups are persistent and fill the register file (4 waves x 128 VGPRs per SIMD): on a CU they
            // hold, the table kernels of the next pipeline stage (aux stream: scan rounds, row emission) wait for a whole
            // workgroup to finish.  A few CUs are therefore left out of the grid (16 of 256):
            // the median step shrinks by ~3 % at 1000 permutations, ~6 % at 10 000 (round-3 sweep, CHANGELOG.md)
            const int spare = std::min(16, ctx->num_cu / 8);  // (8 until the kernels got faster than k_permute_cols on 8 CUs: 10 000-permutation step 27.1 -> 26.3 ms with 16)
            const int64_t blocks_pre = std::min<int64_t>(n_tasks, static_cast<int64_t>(std::max(1, ctx->num_cu - spare)) *
                                                         std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds_pre)));
            if (blk && tail && n_tasks == 0) {                           // (a rank with fewer columns: nothing in this chunk)
                SAFE_HIP_CHECK(hipEventRecord(ev[2 * c + 1], ks));
                chunk_final(c, ks);
                continue;
            }
            if (blk) {
                const uint16_t *ids_c = tail ? d_ids_tail : d_ids[c % NS];
                unsigned int *queue_c = d_queue + 8 * c;
                BitsQueues bq = list_queues[launch_list[c]];
                void *args[] = {(void *)&n, (void *)&ids_c, (void *)&entries_pad, (void *)&nbr->sell_row, (void *)&nbr->slice_off,
                                (void *)&nbr->slice_width, (void *)&d_obs, (void *)&nbr->n_slices, (void *)&d_bits, (void *)&bq,
                                (void *)&d_tasks, (void *)&p_base, (void *)&p_limit, (void *)&queue_c, (void *)&mloc, (void *)&d_gl,
                                (void *)&n_pad};
                const int64_t blocks_blk = std::min<int64_t>(n_tasks, static_cast<int64_t>(std::max(1, ctx->num_cu - spare)) *
                                                                          std::max<size_t>(1, std::min<size_t>(occ5 ? 5 : 4, (160 * 1024) / lds_pre)));   // 4 (5): the register file holds 16 (20) waves per CU
                SAFE_HIP_CHECK(hipLaunchKernel(blk_fn, dim3(blocks_blk), dim3(256), args, lds_pre, ks));
                if (tail) chunk_final(c, ks);
            } else if (pre_w || pre_h) {
                const uint16_t *ids_c = d_ids[c % NS];
                unsigned int *queue_c = d_queue + 8 * c;
                void *args[] = {(void *)&n, (void *)&ids_c, (void *)&entries_pad, (void *)&nbr->sell_row, (void *)&nbr->slice_off,
                                (void *)&nbr->slice_width, (void *)&nbr->sell_col2, (void *)&nbr->n_slices, (void *)&d_bits, (void *)&n_tasks,
                                (void *)&d_tasks, (void *)&p_base, (void *)&p_limit, (void *)&queue_c, (void *)&mloc, (void *)&d_gl,
                                (void *)&n_pad, (void *)&out.ns};
                SAFE_HIP_CHECK(hipLaunchKernel(pre_w_fn, dim3(std::min<int64_t>(n_tasks, std::max(1, ctx->num_cu - spare))), dim3(1024), args,
                                               lds_pre_w, ks));
            } else if (narrow)
                hipLaunchKernelGGL(k_permtest_bits_pre<8>, dim3(blocks_pre), dim3(256), lds_pre, ks, n, d_ids[c % NS], entries_pad,
                                   nbr->sell_row, nbr->slice_off, nbr->slice_width, nbr->sell_col2, nbr->n_slices, d_bits,
                                   n_tasks, d_tasks, p_base, p_limit, d_queue + 8 * c, mloc, d_gl, n_pad, out.ns);
            else if (wide)
                hipLaunchKernelGGL(k_permtest_bits_pre<16>, dim3(blocks_pre), dim3(256), lds_pre, ks, n, d_ids[c % NS], entries_pad,
                                   nbr->sell_row, nbr->slice_off, nbr->slice_width, nbr->sell_col2, nbr->n_slices, d_bits,
                                   n_tasks, d_tasks, p_base, p_limit, d_queue + 8 * c, mloc, d_gl, n_pad, out.ns);
            else
                hipLaunchKernelGGL(k_permtest_bits_pre<10>, dim3(blocks_pre), dim3(256), lds_pre, ks, n, d_ids[c % NS], entries_pad,
                                   nbr->sell_row, nbr->slice_off, nbr->slice_width, nbr->sell_col2, nbr->n_slices, d_bits,
                                   n_tasks, d_tasks, p_base, p_limit, d_queue + 8 * c, mloc, d_gl, n_pad, out.ns);
            SAFE_HIP_CHECK(hipGetLastError());
            SAFE_HIP_CHECK(hipEventRecord(ev[2 * c + 1], ks));
            continue;
        }
        SAFE_HIP_CHECK(hipEventRecord(ev[2 * c], ks));
#define LAUNCH_BITS(CLV, SC)                                                                                          \
        hipLaunchKernelGGL((k_permtest_bits<CLV, SC>), dim3(blocks), dim3(256), lds_bytes, ks, n, P,                  \
                           perms->table16, perms->stride16, nbr->sell_row, nbr->slice_off, nbr->slice_width,          \
                           nbr->sell_col2, nbr->n_slices, d_bits, n_tasks, d_tasks, p_base, p_limit, d_queue + 8 * c, mloc, \
                           d_gl, n_pad, out.ns)
        if (wv == 16)
            hipLaunchKernelGGL((k_permtest_bits<8, false, 16>), dim3(blocks), dim3(1024), lds_bytes, ks, n, P, perms->table16, perms->stride16,
                               nbr->sell_row, nbr->slice_off, nbr->slice_width, nbr->sell_col2, nbr->n_slices, d_bits, n_tasks, d_tasks,
                               p_base, p_limit, d_queue + 8 * c, mloc, d_gl, n_pad, out.ns);
        else if (wide && scaled) LAUNCH_BITS(16, true);
        else if (wide) LAUNCH_BITS(16, false);
        else if (scaled) LAUNCH_BITS(10, true);
        else LAUNCH_BITS(10, false);
#undef LAUNCH_BITS
        SAFE_HIP_CHECK(hipGetLastError());
        SAFE_HIP_CHECK(hipEventRecord(ev[2 * c + 1], ks));
    }
    ctx->packed_counts = d_gl;
    ctx->packed_n_pad = n_pad;
    ctx->packed_m = mloc;
    ctx->packed_perms = P;
    ctx->packed_layout = 0;
    SAFE_REQUIRE(xc_rc == SAFE_OK, "launch_bits: recording a column chunk's event failed");
    if (n_tail) {
        ctx->xc_made = static_cast<int>(n_tail);
        ctx->xc_tail_perms = tail_span;
        for (int64_t k = 0; k <= n_tail; ++k) ctx->xc_bounds[k] = std::min<int64_t>(k * ctx->xc_cols, mloc);
        ctx->xc_bounds[n_tail] = mloc;
        safe_trace("launch_bits: column chunks enqueued");
        if (ctx->xc_callback) ctx->xc_callback(ctx->xc_user);          // the caller's exchange of the chunks goes out behind these launches
    }
    for (int b = 1; b < NS; ++b) {
        SAFE_HIP_CHECK(hipEventRecord(plain[b], kstreams[b]));
        SAFE_HIP_CHECK(hipStreamWaitEvent(ctx->stream, plain[b], 0));
    }
    SAFE_TRY(enrich_finalize_counts(ctx, d_gl, n_pad, nbr->sell_row, mloc, P, out, nullptr));
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(hipEventRecord(ctx->k0, ctx->stream));
    SAFE_HIP_CHECK(hipEventRecord(ctx->k1, ctx->stream));
    safe_trace("launch_bits: all enqueued");
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));    // tasks (host vector) and temporaries
    safe_trace("launch_bits: synced");
    SAFE_TRY(kernel_stat_from_events(ctx, ev, n_launch));
    if (dbg & 128) blk_trace_dump(static_cast<int>(n_launch));
    return SAFE_OK;
}

// X = A . B0 for a binary attribute block through the bit-sliced count kernel; false if the
// block does not qualify (not binary, or neighborhoods of 1024+ members)
static bool counts_bits_applicable(const safe_nbr *nbr, safe_attr *attr) {
    const char *force = getenv("SAFE_HIP_FORCE_PATH");
    if (force && !strcmp(force, "gather")) return false;
    return safe_attr_prepare(attr) == SAFE_OK && attr->n_other == 0 && nbr->max_count < (1 << BT_LV);
}

static int launch_counts_bits(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, int64_t col0, int64_t col1, double *out_dev) {
    const int64_t n = nbr->n, mloc = col1 - col0, n_wg = ceil_div(mloc, 64);
    uint2 *d_bits = nullptr;
    SAFE_TRY(ctx_scratch(ctx, 1, static_cast<size_t>(n_wg) * (n + 1) * sizeof(uint2), reinterpret_cast<void **>(&d_bits)));
    launch_bits_prep(ctx, attr, col0, mloc, n_wg, d_bits);
    SAFE_HIP_CHECK(hipEventRecord(ctx->k0, ctx->stream));
    hipLaunchKernelGGL(k_counts_bits<false>, dim3(nbr->n_slices, n_wg), dim3(64), 0, ctx->stream, nbr->sell_row, nbr->slice_off,
                       nbr->slice_width, nbr->sell_col, n, d_bits, mloc, out_dev, HypLookup{});
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(hipEventRecord(ctx->k1, ctx->stream));
    ctx->last_kernel.name = "k_counts_bits";
    return SAFE_OK;
}

static size_t ldsf64_bytes(int64_t n, int64_t stride16) {
    return (8 * static_cast<size_t>(n + 1) + static_cast<size_t>(stride16) + 4) * sizeof(unsigned int);
}

static bool lds_f64_applicable(const safe_nbr *nbr, const safe_perms *perms) {
    const char *force = getenv("SAFE_HIP_FORCE_PATH");
    if (force && !strcmp(force, "gather")) return false;
    return nbr->sell_col2 != nullptr && perms->table16 != nullptr && perms->count >= 1 && perms->count <= 65535 &&
           ldsf64_bytes(nbr->n, perms->stride16) <= 160 * 1024;
}

// A NARROW block of quantitative columns (the reference's own Example 3 tests ONE attribute with 10 000 permutations,
// examples/Example_3_Scatterplot_annotation.ipynb:104,147): the matrix-core kernel pads the block to 32 columns and its launches
// are latency-bound (0.67 ms per 128 permutations whatever the width up to ~256 columns), the LDS-resident f64 kernel is bound
// by its 4-column tiles -- 2-2.6 x faster until n x columns ~ 2e5 (tools/probe/narrow_paths.py: 1586 x 1: 2.85 vs 5.59 ms per
// 2000 permutations; 3971 x 32: 2.55 vs 3.78; 3971 x 64: 4.16 vs 3.78).
static bool narrow_block_prefers_lds(const safe_nbr *nbr, const safe_perms *perms, int64_t mloc) {
    const char *force = getenv("SAFE_HIP_FORCE_PATH");
    if (force && !strcmp(force, "mfma")) return false;
    const char *knob = getenv("SAFE_HIP_NARROW_LDS");                  // =0: A/B, and the matrix-core tests at small sizes
    const bool off = knob && !strcmp(knob, "0");
    return !off && lds_f64_applicable(nbr, perms) && nbr->n * mloc <= 204800;
}

// general f64 permutation test with LDS-resident tiles (k_permtest_lds), pipelined over spans like launch_bits
static int launch_lds_f64(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, safe_perms *perms, int64_t col0, int64_t col1,
                          bool z, const PermOut &out_in) {
    const int64_t n = nbr->n, mloc = col1 - col0, P = perms->count;
    const int64_t n_tiles = z ? mloc : ceil_div(mloc, 4);
    PermOut out = out_in;
    double *d_tiles = nullptr, *d_ns = out.ns;
    SAFE_TRY(ctx_scratch(ctx, 1, static_cast<size_t>(n_tiles) * (n + 1) * 4 * sizeof(double), reinterpret_cast<void **>(&d_tiles)));
    if (!d_ns) SAFE_TRY(ctx_scratch(ctx, 2, static_cast<size_t>(n) * mloc * sizeof(double), reinterpret_cast<void **>(&d_ns)));
    {
        const dim3 grid(ceil_div(n_tiles * (n + 1), 256)), block(256);
        const bool f32 = attr->dtype == SAFE_DTYPE_F32;
#define PREP(T, ZZ) hipLaunchKernelGGL((k_tile32_prep<T, ZZ>), grid, block, 0, ctx->stream, attr->raw, n, attr->row_stride, \
                                        attr->col_stride, col0, mloc, n_tiles, d_tiles)
        if (z) { if (f32) PREP(float, true); else PREP(double, true); }
        else { if (f32) PREP(float, false); else PREP(double, false); }
#undef PREP
    }
    int64_t span = 1;
    // a narrow block (the reference's Example 3: ONE attribute, 10 000 permutations) gives a launch almost nothing to do: its 128
    // permutations take 0.25 ms of launch and tail latency whatever the width, so after the start-up stages a launch covers
    // eight pipeline stages (at the Example-3 shape 7.5 -> 2.0 ms unseeded, 11.6 -> 10.5 ms seeded, where the
    // host draws of 10 000 short shuffles are then the bound)
    const int merge = n * mloc <= 204800 ? 8 : 1;
    const std::vector<int64_t> starts = perm_launch_starts(perms, &span, merge);
    const size_t lds_bytes = ldsf64_bytes(n, perms->stride16);
    const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds_bytes)));
    const int64_t slots = static_cast<int64_t>(ctx->num_cu) * per_cu;
    const int NW = 16;                             // waves (= adjacent slices) per workgroup: LDS allows one workgroup per CU
    const int64_t n_sg = ceil_div(nbr->n_slices, NW);
    std::vector<int64_t> sg_blocks(n_sg, 0);
    int64_t blocks_per_perm = 0;
    for (int64_t s = 0; s < nbr->n_slices; ++s) sg_blocks[s / NW] = std::max<int64_t>(sg_blocks[s / NW], nbr->h_slice_width[s] / 8);
    for (int64_t g = 0; g < n_sg; ++g) blocks_per_perm += std::max<int64_t>(sg_blocks[g], 1);
    const int64_t tasks_per_tile = std::max<int64_t>(1, ceil_div(6 * slots, n_tiles));
    const int64_t target = std::max<int64_t>(256, blocks_per_perm * span / tasks_per_tile);
    struct TaskCost { int4 t; int64_t cost; };
    std::vector<TaskCost> tc;
    for (int64_t g = 0; g < n_sg; ++g) {
        const int64_t bl = std::max<int64_t>(sg_blocks[g], 1);
        int64_t ppt = std::min<int64_t>(span, std::max<int64_t>(16, target / bl));
        const int64_t chunks = ceil_div(span, ppt);
        ppt = ceil_div(span, chunks);
        for (int64_t c = 0; c < chunks; ++c) {
            const int64_t p0 = c * ppt, p1 = std::min<int64_t>(span, p0 + ppt);
            for (int64_t w = 0; w < n_tiles; ++w)
                tc.push_back({make_int4(static_cast<int>(w), static_cast<int>(g), static_cast<int>(p0), static_cast<int>(p1)),
                              bl * (p1 - p0)});
        }
    }
    std::stable_sort(tc.begin(), tc.end(), [](const TaskCost &a, const TaskCost &b) { return a.cost > b.cost; });
    std::vector<int4> tasks(tc.size());
    for (size_t i = 0; i < tc.size(); ++i) tasks[i] = tc[i].t;
    const int64_t n_launch = static_cast<int64_t>(starts.size()) - 1, n_pad = nbr->n_slices * 64;
    int4 *d_tasks = nullptr;
    unsigned int *d_queue = nullptr, *d_counts = nullptr;
    SAFE_TRY(dev_alloc(&d_tasks, tasks.size()));
    SAFE_TRY(dev_alloc(&d_queue, n_launch));
    SAFE_TRY(ctx_scratch(ctx, 0, static_cast<size_t>(n_pad) * mloc * sizeof(unsigned int), reinterpret_cast<void **>(&d_counts)));
    SAFE_HIP_CHECK(hipMemcpyAsync(d_tasks, tasks.data(), tasks.size() * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemsetAsync(d_queue, 0, n_launch * sizeof(unsigned int), ctx->stream));
    SAFE_HIP_CHECK(hipMemsetAsync(d_counts, 0, static_cast<size_t>(n_pad) * mloc * sizeof(unsigned int), ctx->stream));
    const int64_t blocks = std::min<int64_t>(static_cast<int64_t>(tasks.size()), slots);
    const int64_t n_tasks = static_cast<int64_t>(tasks.size());
#define LDS_DISPATCH(ACTION)                                      \
    do {                                                          \
        if (z) {                                                  \
            if (NW == 16) ACTION(true, 16);                       \
            else if (NW == 8) ACTION(true, 8);                    \
            else ACTION(true, 4);                                 \
        } else {                                                  \
            if (NW == 16) ACTION(false, 16);                      \
            else if (NW == 8) ACTION(false, 8);                   \
            else ACTION(false, 4);                                \
        }                                                         \
    } while (0)
#define LDS_SETATTR(ZZ, W)                                                                                            \
    SAFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_permtest_lds<ZZ, W>),                         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)))
    LDS_DISPATCH(LDS_SETATTR);
    ctx->last_kernel.name = "k_permtest_lds";
    ctx->last_kernel.total_ms = 0.0;
    ctx->last_kernel.busy_ms = 0.0;
    ctx->last_kernel.launches = 0;
    hipEvent_t *ev = nullptr, *plain = nullptr;                   // pooled on the context
    SAFE_TRY(ctx_events(ctx, true, 2 * n_launch, &ev));
    SAFE_TRY(ctx_events(ctx, false, 2, &plain));
    hipEvent_t ready = plain[0], side_done = plain[1];
    SAFE_HIP_CHECK(hipEventRecord(ready, ctx->stream));
    SAFE_HIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ready, 0));
    for (int64_t c = 0; c < n_launch; ++c) {
        const int64_t p_base = starts[c], p_limit = starts[c + 1];
        hipStream_t ks = (c & 1) ? ctx->side_stream : ctx->stream;
        SAFE_TRY(perms_wait(perms, p_limit, ks));
        SAFE_HIP_CHECK(hipEventRecord(ev[2 * c], ks));
#define LDS_LAUNCH(ZZ, W)                                                                                              \
    hipLaunchKernelGGL((k_permtest_lds<ZZ, W>), dim3(blocks), dim3(64 * W), lds_bytes, ks, n, perms->table16,          \
                       perms->stride16, nbr->sell_row, nbr->slice_off, nbr->slice_width, nbr->sell_col2, nbr->n_slices, \
                       d_tiles, n_tasks, d_tasks, p_base, p_limit, d_queue + c, mloc, d_counts, n_pad, d_ns)
        LDS_DISPATCH(LDS_LAUNCH);
        SAFE_HIP_CHECK(hipGetLastError());
        SAFE_HIP_CHECK(hipEventRecord(ev[2 * c + 1], ks));
    }
    SAFE_HIP_CHECK(hipEventRecord(side_done, ctx->side_stream));
    SAFE_HIP_CHECK(hipStreamWaitEvent(ctx->stream, side_done, 0));
    SAFE_TRY(enrich_finalize_counts(ctx, d_counts, n_pad, nbr->sell_row, mloc, P, out, d_ns));
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(hipEventRecord(ctx->k0, ctx->stream));
    SAFE_HIP_CHECK(hipEventRecord(ctx->k1, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    SAFE_TRY(kernel_stat_from_events(ctx, ev, n_launch));
    (void)hipFree(d_tasks);
    (void)hipFree(d_queue);
    return SAFE_OK;
}

// Hypergeometric path for binary attributes with the tail looked up instead of evaluated per
// element: P[H >= X] only depends on (X, K_j, n_i), and a matrix has few distinct neighborhood
// sizes n_i and annotation counts K_j (17 k distinct triples among 17 M elements at config 2,
// SURVEY C12).  k_hyp_table evaluates every (n, K) pair once for all X and every element is a lookup:
//   * small neighborhoods: in the epilogue of the bit-sliced count kernel (the counts never reach memory);
//   * large neighborhoods, split form (default): matrix-core counts -> packed u16 counts -> k_hyp_table cut
//     at the largest count -> k_hyp_emit streams p / NES / nes_binary out (mfma.hip);
//   * large neighborhoods, fused form (SAFE_HIP_HYP_SPLIT=0): in the epilogue of the matrix-core kernel.
// *fused = false (nothing launched) when the table would be too large or K is not an integer (scipy then
// returns NaN: the per-element kernel reproduces that).
static int hypergeom_fused(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, int64_t col0, int64_t col1, int64_t pop,
                           const double *d_size, double p_cut, double *p_dev, double *nes_dev,
                           double *nb_dev, unsigned int *d_enr, bool use_mfma, hipStream_t hs, MfmaCountsSplit *split,
                           bool *fused, double *enriched_f64 = nullptr, bool *enriched_done = nullptr) {
    // hs = the stream of the table's own work: ctx->stream, or the side stream while the first half of the
    // split matrix-core form (which needs nothing from here) runs on ctx->stream
    *fused = false;
    const int64_t n = nbr->n, mloc = col1 - col0;
    // pinned staging: [neighborhood sizes | column sums] coming back, then everything that goes up (ids, and the split form's row
    // lists).  Uploads from pageable vectors went through the runtime's staging path: one 80 KB copy took 0.38 ms and held the
    // table kernel back behind the count kernel's end.
    void *pinned = nullptr;
    const size_t down_bytes = static_cast<size_t>(n + mloc) * sizeof(double);
    const size_t up_bytes = static_cast<size_t>(2 * (n + mloc) + 8) * sizeof(int32_t) + static_cast<size_t>(n) * (sizeof(int4) + sizeof(int2));
    SAFE_TRY(ctx_pinned(ctx, down_bytes + up_bytes, &pinned));
    const double *h_size = static_cast<const double *>(pinned), *h_k = h_size + n;
    int32_t *up = reinterpret_cast<int32_t *>(static_cast<char *>(pinned) + down_bytes);
    SAFE_HIP_CHECK(hipMemcpyAsync(pinned, d_size, n * sizeof(double), hipMemcpyDeviceToHost, hs));
    SAFE_HIP_CHECK(hipMemcpyAsync(static_cast<double *>(pinned) + n, attr->col_sum + col0, mloc * sizeof(double), hipMemcpyDeviceToHost, hs));
    SAFE_HIP_CHECK(safe_stream_sync(hs));
    // distinct values -> dense ids (both are integers in [0, n] here, or we decline)
    std::vector<int32_t> id_of(n + 2, -1), nvals, kvals, nid(n), kid(mloc);
    int64_t max_n = 0, max_k = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double v = h_size[i];
        if (!(v >= 0.0) || v > static_cast<double>(pop) || v != std::floor(v)) return SAFE_OK;
        const int32_t iv = static_cast<int32_t>(v);
        if (id_of[iv] < 0) {
            id_of[iv] = static_cast<int32_t>(nvals.size());
            nvals.push_back(iv);
        }
        nid[i] = id_of[iv];
        max_n = std::max<int64_t>(max_n, iv);
    }
    std::fill(id_of.begin(), id_of.end(), -1);
    for (int64_t j = 0; j < mloc; ++j) {
        const double v = h_k[j];
        if (!(v >= 0.0) || v > static_cast<double>(pop) || v != std::floor(v)) return SAFE_OK;
        const int32_t iv = static_cast<int32_t>(v);
        if (id_of[iv] < 0) {
            id_of[iv] = static_cast<int32_t>(kvals.size());
            kvals.push_back(iv);
        }
        kid[j] = id_of[iv];
        max_k = std::max<int64_t>(max_k, iv);
    }
    const int64_t xs = std::min(max_n, max_k) + 1;                      // X <= min(K, n)
    const int64_t n_nid = static_cast<int64_t>(nvals.size()), n_kid = static_cast<int64_t>(kvals.size());
    const double table_bytes = static_cast<double>(n_nid) * n_kid * xs * sizeof(double2);
    const double direct_cost = static_cast<double>(n) * mloc;          // elements the per-element kernel would evaluate
    if (table_bytes > 512e6 || static_cast<double>(n_nid) * n_kid * 4.0 > direct_cost) return SAFE_OK;

    double2 *d_tab = nullptr;
    int32_t *d_ids = nullptr;
    SAFE_TRY(ctx_scratch(ctx, 2, static_cast<size_t>(n_nid) * n_kid * xs * sizeof(double2), reinterpret_cast<void **>(&d_tab)));
    SAFE_TRY(ctx_scratch(ctx, 6, static_cast<size_t>(n_nid + n_kid + n + mloc) * sizeof(int32_t), reinterpret_cast<void **>(&d_ids)));
    int32_t *d_nvals = d_ids, *d_kvals = d_nvals + n_nid, *d_nid = d_kvals + n_kid, *d_kid = d_nid + n;
    // d_ids = [nvals | kvals | nid | kid]: one copy out of the pinned block
    memcpy(up, nvals.data(), n_nid * sizeof(int32_t));
    memcpy(up + n_nid, kvals.data(), n_kid * sizeof(int32_t));
    memcpy(up + n_nid + n_kid, nid.data(), n * sizeof(int32_t));
    memcpy(up + n_nid + n_kid + n, kid.data(), mloc * sizeof(int32_t));
    const int64_t n_up = n_nid + n_kid + n + mloc;
    SAFE_HIP_CHECK(hipMemcpyAsync(d_ids, up, n_up * sizeof(int32_t), hipMemcpyHostToDevice, hs));
    hipEvent_t ids_done = nullptr;
    if (split) {
        // the second half's row lists travel on this stream too (staged behind the ids, 16-byte aligned)
        SAFE_TRY(mfma_counts_split_rows(ctx, nbr, split, nid.data(), hs, up + (n_up + 3) / 4 * 4));
        // the counts are under way on ctx->stream: the table follows them there, cut at their largest value
        SAFE_HIP_CHECK(hipEventCreateWithFlags(&ids_done, safe_event_flags(hipEventDisableTiming)));
        SAFE_HIP_CHECK(hipEventRecord(ids_done, hs));
        SAFE_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ids_done, 0));
    }
    hipLaunchKernelGGL(k_hyp_table, dim3(ceil_div(n_nid * n_kid, 64)), dim3(64), 0, split ? ctx->stream : hs, d_nvals, n_nid, d_kvals,
                       n_kid, xs, pop, split ? mfma_counts_split_xmax(split) : static_cast<const unsigned int *>(nullptr), d_tab);

    const int64_t n_wg = ceil_div(mloc, 64);
    HypLookup hl{};
    hl.nid = d_nid;
    hl.kid = d_kid;
    hl.tab = d_tab;
    hl.n_kid = n_kid;
    hl.xs = xs;
    hl.p_cut = p_cut;
    SAFE_TRY(ctx_scratch(ctx, 7, 64 * sizeof(double), reinterpret_cast<void **>(&hl.dummy)));
    hl.pvalues_pos = p_dev;
    hl.nes = nes_dev;
    hl.nes_binary = nb_dev;
    hl.enriched = d_enr;
    if (split) {
        int rc = mfma_counts_split_emit(ctx, nbr, split, hl);
        if (rc == SAFE_OK && enriched_f64) {                           // per-attribute counts right behind the emit kernel (no host round trip between)
            hipLaunchKernelGGL(k_u32_to_f64, dim3(ceil_div(mloc, 256)), dim3(256), 0, ctx->stream, d_enr, enriched_f64, mloc);
            if (hipGetLastError() != hipSuccess) rc = SAFE_E_HIP;
        }
        (void)safe_stream_sync(hs);                                    // the id vectors above are host memory
        (void)hipEventDestroy(ids_done);
        SAFE_TRY(rc);
    } else if (use_mfma) {
        SAFE_TRY(launch_mfma_counts(ctx, nbr, attr, col0, col1, hl));   // records its own timing events
    } else {
        uint2 *d_bits = nullptr;                                       // bit-packed attributes: only the bit-sliced form reads them
        SAFE_TRY(ctx_scratch(ctx, 1, static_cast<size_t>(n_wg) * (n + 1) * sizeof(uint2), reinterpret_cast<void **>(&d_bits)));
        launch_bits_prep(ctx, attr, col0, mloc, n_wg, d_bits);
        SAFE_HIP_CHECK(hipEventRecord(ctx->k0, ctx->stream));
        hipLaunchKernelGGL(k_counts_bits<true>, dim3(nbr->n_slices, n_wg), dim3(64), 0, ctx->stream, nbr->sell_row, nbr->slice_off,
                           nbr->slice_width, nbr->sell_col, n, d_bits, mloc, static_cast<double *>(nullptr), hl);
        SAFE_HIP_CHECK(hipGetLastError());
        SAFE_HIP_CHECK(hipEventRecord(ctx->k1, ctx->stream));
        ctx->last_kernel.name = "k_counts_bits<hypergeom>";
    }
    if (split && enriched_f64 && enriched_done) *enriched_done = true;
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));                 // the id vectors above are host memory
    *fused = true;
    return SAFE_OK;
}

// counts of a whole call (#<=, #>= as f64 [N, M], e.g. summed over the ranks of a permutation-axis split) ->
// p-values, NES, nes_binary, enriched counts (safe.py:528-554, 468-472); an element whose observed score is NaN has no test
__global__ __launch_bounds__(256) void k_counts_to_outputs(const double *__restrict__ counts_neg, const double *__restrict__ counts_pos,
                                                           const double *__restrict__ ns, int64_t total, int64_t m, int64_t n_perm,
                                                           PermOut out) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= total) return;
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    bool obs_nan = ns && ns[idx] != ns[idx];
    // counts are whole numbers in [0, num_permutations] (the NES table has P + 1 entries): anything else -- sums of ranks
    // that each ran the full P, negative or NaN counts -- is reported (flag behind the enriched counters), never looked up
    const double cn = counts_neg[idx], cp = counts_pos[idx], Pd = static_cast<double>(n_perm);
    if (!obs_nan && !(cn >= 0.0 && cn <= Pd && cp >= 0.0 && cp <= Pd)) {
        atomicOr(&out.enriched[m], 1u);
        obs_nan = true;
    }
    const unsigned int kn = obs_nan ? 0u : static_cast<unsigned int>(cn), kp = obs_nan ? 0u : static_cast<unsigned int>(cp);
    const double en = obs_nan ? qnan : out.nes_table[kn], ep = obs_nan ? qnan : out.nes_table[kp];
    double nes = ep - en;
    if (out.sign_mode == SAFE_SIGN_HIGHEST) nes = ep;
    if (out.sign_mode == SAFE_SIGN_LOWEST) nes = en;
    const bool hit = (nes == nes) && (fabs(nes) > out.nes_threshold);
    out.pvalues_neg[idx] = obs_nan ? qnan : static_cast<double>(kn) / static_cast<double>(n_perm);
    out.pvalues_pos[idx] = obs_nan ? qnan : static_cast<double>(kp) / static_cast<double>(n_perm);
    out.nes[idx] = nes;
    out.nes_binary[idx] = hit ? 1.0 : 0.0;
    if (hit) atomicAdd(&out.enriched[idx % m], 1u);
}

static int finish_kernel_timing(safe_ctx *ctx) {
    SAFE_HIP_CHECK(hipEventSynchronize(ctx->k1));
    if (ctx->last_kernel.name == "k_permtest_bits" || ctx->last_kernel.name == "k_permtest_bits_pre" || ctx->last_kernel.name == "k_permtest_bits_blk" ||
        ctx->last_kernel.name == "k_permtest_lds" || ctx->last_kernel.name == "k_permtest_mfma")
        return SAFE_OK;   // per-launch events already summed
    float ms = 0.f;
    SAFE_HIP_CHECK(hipEventElapsedTime(&ms, ctx->k0, ctx->k1));
    ctx->last_kernel.total_ms = ms;
    ctx->last_kernel.busy_ms = ms;
    ctx->last_kernel.launches = 1;
    return SAFE_OK;
}

extern "C" {

int safe_score(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, int score_type, int64_t col0, int64_t col1,
               double *out_dev) {
    SAFE_REQUIRE(ctx && out_dev, "safe_score: NULL argument");
    SAFE_TRY(check_cols(nbr, attr, col0, col1, "safe_score"));
    SAFE_REQUIRE(score_type == SAFE_SCORE_SUM || score_type == SAFE_SCORE_ZSCORE, "safe_score: bad score_type %d", score_type);
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const bool z = score_type == SAFE_SCORE_ZSCORE;
    if (!z && counts_bits_applicable(nbr, attr)) {
        const char *counts_env = getenv("SAFE_HIP_COUNTS");
        const bool dense_nbr = nbr->n >= 256 && nbr->nnz >= 128 * nbr->n;
        if (counts_env ? !strcmp(counts_env, "mfma") : dense_nbr) {      // 0/1 data, large neighborhoods: matrix cores
            HypLookup hl{};
            hl.pvalues_pos = out_dev;                                     // tab == NULL: plain counts
            SAFE_TRY(launch_mfma_counts(ctx, nbr, attr, col0, col1, hl));
        } else {
            SAFE_TRY(launch_counts_bits(ctx, nbr, attr, col0, col1, out_dev));
        }
        return finish_kernel_timing(ctx);
    }
    Tiles tiles;
    SAFE_TRY(build_tiles(ctx, attr, col0, col1, z, &tiles));
    PermOut out{};
    out.ns = out_dev;
    out.mode = 0;
    int rc = launch_gather(ctx, nbr, tiles, nullptr, 0, col1 - col0, z, out);
    if (rc == SAFE_OK) rc = finish_kernel_timing(ctx);
    (void)hipFree(tiles.bt);
    return rc;
}

int safe_permtest_counts(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, safe_perms *perms, int score_type,
                         int64_t col0, int64_t col1, double *ns_dev, double *counts_neg_dev, double *counts_pos_dev) {
    SAFE_REQUIRE(ctx && perms && counts_neg_dev && counts_pos_dev, "safe_permtest_counts: NULL argument");
    SAFE_TRY(check_cols(nbr, attr, col0, col1, "safe_permtest_counts"));
    SAFE_REQUIRE(perms->n == nbr->n, "safe_permtest_counts: permutation tables are for %lld rows, membership has %lld",
                 (long long)perms->n, (long long)nbr->n);
    SAFE_REQUIRE(score_type == SAFE_SCORE_SUM || score_type == SAFE_SCORE_ZSCORE, "safe_permtest_counts: bad score_type %d", score_type);
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    ctx->packed_layout = -1;
    const bool z = score_type == SAFE_SCORE_ZSCORE;
    PermOut out{};
    out.ns = ns_dev;
    out.counts_neg = counts_neg_dev;
    out.counts_pos = counts_pos_dev;
    out.mode = 1;
    const PermPath path = choose_path(ctx, nbr, attr, perms->count, z);
    if (path != PATH_GATHER) {
        SAFE_TRY(path == PATH_BITS ? launch_bits(ctx, nbr, attr, perms, col0, col1, out)
                                   : launch_scatter(ctx, nbr, attr, perms, col0, col1, out));
        return finish_kernel_timing(ctx);
    }
    if (mfma_applicable(ctx, nbr, attr, perms, z) && !narrow_block_prefers_lds(nbr, perms, col1 - col0)) {
        bool declined = false;
        SAFE_TRY(launch_mfma(ctx, nbr, attr, perms, col0, col1, z, out, &declined));
        if (!declined) return finish_kernel_timing(ctx);
    }
    if (lds_f64_applicable(nbr, perms)) {
        SAFE_TRY(launch_lds_f64(ctx, nbr, attr, perms, col0, col1, z, out));
        return finish_kernel_timing(ctx);
    }
    Tiles tiles;
    SAFE_TRY(build_tiles(ctx, attr, col0, col1, z, &tiles));
    int rc = perms_wait(perms, perms->count, ctx->stream);          // (rc from here on: the tile buffer is freed on every path)
    if (rc == SAFE_OK) rc = launch_gather(ctx, nbr, tiles, perms->table, perms->count, col1 - col0, z, out);
    if (rc == SAFE_OK) rc = finish_kernel_timing(ctx);
    (void)hipFree(tiles.bt);
    return rc;
}

int safe_randomization(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, safe_perms *perms, int score_type, int sign_mode,
                       double enrichment_threshold, const double *nes_table_host, int64_t col0, int64_t col1,
                       double *ns_dev, double *pvalues_neg_dev, double *pvalues_pos_dev, double *nes_dev,
                       double *nes_binary_dev, double *num_enriched_dev) {
    SAFE_REQUIRE(ctx && perms && pvalues_neg_dev && pvalues_pos_dev && nes_dev && nes_binary_dev && num_enriched_dev,
                 "safe_randomization: NULL argument");
    SAFE_TRY(check_cols(nbr, attr, col0, col1, "safe_randomization"));
    SAFE_REQUIRE(perms->n == nbr->n, "safe_randomization: permutation tables are for %lld rows, membership has %lld",
                 (long long)perms->n, (long long)nbr->n);
    SAFE_REQUIRE(perms->count >= 1, "safe_randomization: no permutations");
    SAFE_REQUIRE(score_type == SAFE_SCORE_SUM || score_type == SAFE_SCORE_ZSCORE, "safe_randomization: bad score_type %d", score_type);
    SAFE_REQUIRE(sign_mode >= SAFE_SIGN_HIGHEST && sign_mode <= SAFE_SIGN_BOTH, "safe_randomization: bad sign_mode %d", sign_mode);
    SAFE_REQUIRE(enrichment_threshold > 0.0 && enrichment_threshold < 1.0, "safe_randomization: enrichment_threshold must be in (0,1)");
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    ctx->packed_layout = -1;
    const bool z = score_type == SAFE_SCORE_ZSCORE;
    const int64_t mloc = col1 - col0, P = perms->count;
    std::vector<double> tab(P + 1);
    if (nes_table_host) {
        std::copy(nes_table_host, nes_table_host + P + 1, tab.begin());
    } else {
        tab[0] = -std::log10(1.0 / static_cast<double>(P));
        for (int64_t k = 1; k <= P; ++k) tab[k] = -std::log10(static_cast<double>(k) / static_cast<double>(P));
    }
    double *d_tab = nullptr;
    unsigned int *d_enr = nullptr;
    Tiles tiles;
    const PermPath path = choose_path(ctx, nbr, attr, P, z);
    bool mfma = path == PATH_GATHER && mfma_applicable(ctx, nbr, attr, perms, z) && !narrow_block_prefers_lds(nbr, perms, mloc);
    const bool lds64 = path == PATH_GATHER && lds_f64_applicable(nbr, perms);
    void *small = nullptr;                           // NES table f64 [P + 1] | enriched counters u32 [mloc + 16] (grow-only scratch)
    int rc = ctx_scratch(ctx, 10, static_cast<size_t>(P + 1) * sizeof(double) + static_cast<size_t>(mloc + 16) * sizeof(unsigned int), &small);
    if (rc == SAFE_OK) {
        d_tab = static_cast<double *>(small);
        d_enr = reinterpret_cast<unsigned int *>(d_tab + P + 1);
    }
    if (rc == SAFE_OK) {
        hipError_t e = hipMemcpyAsync(d_tab, tab.data(), (P + 1) * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_enr, 0, (mloc + 16) * sizeof(unsigned int), ctx->stream);
        if (e != hipSuccess) {
            safe_set_error("safe_randomization: %s", hipGetErrorString(e));
            rc = SAFE_E_HIP;
        }
    }
    if (rc == SAFE_OK) {
        PermOut out{};
        out.ns = ns_dev;
        out.pvalues_neg = pvalues_neg_dev;
        out.pvalues_pos = pvalues_pos_dev;
        out.nes = nes_dev;
        out.nes_binary = nes_binary_dev;
        out.enriched = d_enr;
        out.nes_table = d_tab;
        out.nes_threshold = -std::log10(enrichment_threshold);
        out.sign_mode = sign_mode;
        out.mode = 2;
        if (mfma) {
            bool declined = false;
            rc = launch_mfma(ctx, nbr, attr, perms, col0, col1, z, out, &declined);
            if (declined) mfma = false;
        }
        if (rc == SAFE_OK && !mfma) {
            if (path == PATH_GATHER && !lds64) rc = build_tiles(ctx, attr, col0, col1, z, &tiles);
            if (rc == SAFE_OK)
                rc = lds64                  ? launch_lds_f64(ctx, nbr, attr, perms, col0, col1, z, out)
                     : path == PATH_BITS    ? launch_bits(ctx, nbr, attr, perms, col0, col1, out)
                     : path == PATH_SCATTER ? launch_scatter(ctx, nbr, attr, perms, col0, col1, out)
                                            : (perms_wait(perms, P, ctx->stream) == SAFE_OK
                                                   ? launch_gather(ctx, nbr, tiles, perms->table, P, mloc, z, out)
                                                   : SAFE_E_HIP);
        }
    }
    if (rc == SAFE_OK) {
        hipLaunchKernelGGL(k_u32_to_f64, dim3(ceil_div(mloc, 256)), dim3(256), 0, ctx->stream, d_enr, num_enriched_dev, mloc);
        if (hipGetLastError() != hipSuccess) rc = SAFE_E_HIP;
    }
    if (rc == SAFE_OK) rc = finish_kernel_timing(ctx);
    if (rc == SAFE_OK && safe_stream_sync(ctx->stream) != hipSuccess) rc = SAFE_E_HIP;   // tab (host) + temporaries
    (void)hipFree(tiles.bt);
    return rc;
}

int safe_hypergeom(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, double enrichment_threshold, int64_t col0,
                   int64_t col1, double *pvalues_pos_dev, double *nes_dev, double *nes_binary_dev,
                   double *num_enriched_dev) {
    SAFE_REQUIRE(ctx && pvalues_pos_dev && nes_dev && nes_binary_dev && num_enriched_dev, "safe_hypergeom: NULL argument");
    SAFE_TRY(check_cols(nbr, attr, col0, col1, "safe_hypergeom"));
    SAFE_REQUIRE(enrichment_threshold > 0.0 && enrichment_threshold < 1.0, "safe_hypergeom: enrichment_threshold must be in (0,1)");
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    SAFE_TRY(safe_attr_prepare(attr));
    const int64_t n = nbr->n, mloc = col1 - col0;
    const int64_t pop = attr->n_rows_with_value;
    double *d_lf = nullptr, *d_hits = nullptr, *d_size = nullptr;
    unsigned int *d_enr = nullptr;
    Tiles tiles;
    const bool bits = counts_bits_applicable(nbr, attr);
    const char *table_env = getenv("SAFE_HIP_HYPER_TABLE");
    const bool table = bits && !(table_env && !strcmp(table_env, "0"));
    const char *counts_env = getenv("SAFE_HIP_COUNTS");
    const bool dense_nbr = nbr->n >= 256 && nbr->nnz >= 128 * nbr->n;       // matrix cores pay off for large neighborhoods
    const bool use_mfma = counts_env ? !strcmp(counts_env, "mfma") : dense_nbr;
    void *small = nullptr;                                                  // d_size f64 [n] | d_enr u32 [mloc + 64]
    int rc = ctx_scratch(ctx, 9, static_cast<size_t>(n) * sizeof(double) + static_cast<size_t>(mloc + 64) * sizeof(unsigned int), &small);
    if (rc == SAFE_OK) {
        d_size = static_cast<double *>(small);
        d_enr = reinterpret_cast<unsigned int *>(d_size + n);
        if (hipMemsetAsync(d_enr, 0, (mloc + 64) * sizeof(unsigned int), ctx->stream) != hipSuccess) {
            safe_set_error("safe_hypergeom: hipMemsetAsync failed");
            rc = SAFE_E_HIP;
        }
    }
    // Split matrix-core form: bit planes and counts start on ctx->stream right away; neighborhood sizes,
    // the distinct (n, K) ids and the table are prepared on the side stream meanwhile (hypergeom_fused)
    MfmaCountsSplit *split = nullptr;
    hipStream_t hs = ctx->stream;
    hipEvent_t ev = nullptr;
    if (rc == SAFE_OK && table && use_mfma && mfma_counts_split_applicable(nbr)) {
        hs = ctx->side_stream;
        if (hipEventCreateWithFlags(&ev, safe_event_flags(hipEventDisableTiming)) != hipSuccess || hipEventRecord(ev, ctx->stream) != hipSuccess ||
            hipStreamWaitEvent(hs, ev, 0) != hipSuccess) {
            safe_set_error("safe_hypergeom: stream fork failed");
            rc = SAFE_E_HIP;
        }
        if (rc == SAFE_OK) rc = mfma_counts_split_begin(ctx, nbr, attr, col0, col1, &split);
    }
    bool fused = false, enriched_done = false;
    if (rc == SAFE_OK) {
        hipLaunchKernelGGL(k_nbr_size, dim3(ceil_div(n, 4)), dim3(256), 0, hs, nbr->row_ptr, nbr->col, attr->row_flags, n, d_size);
        if (table) rc = hypergeom_fused(ctx, nbr, attr, col0, col1, pop, d_size, nes_p_cut(enrichment_threshold),
                                        pvalues_pos_dev, nes_dev, nes_binary_dev, d_enr, use_mfma, hs, split, &fused, num_enriched_dev,
                                        &enriched_done);
    }
    if (hs != ctx->stream) {                                                // join (the fallback below reads d_size)
        if (ev && hipEventRecord(ev, hs) == hipSuccess) (void)hipStreamWaitEvent(ctx->stream, ev, 0);
    }
    if (rc == SAFE_OK && !fused) {
        // per-element evaluation: counts through memory, pmf from a log-factorial table lf[k] = log(k!), k = 0..n
        std::vector<double> lf(n + 2);
        for (int64_t k = 0; k <= n + 1; ++k) lf[k] = std::lgamma(static_cast<double>(k) + 1.0);
        rc = dev_alloc(&d_lf, n + 2);
        if (rc == SAFE_OK) rc = ctx_scratch(ctx, 2, static_cast<size_t>(n) * mloc * sizeof(double), reinterpret_cast<void **>(&d_hits));
        if (rc == SAFE_OK && !bits) rc = build_tiles(ctx, attr, col0, col1, false, &tiles);
        if (rc == SAFE_OK && hipMemcpyAsync(d_lf, lf.data(), (n + 2) * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            safe_set_error("safe_hypergeom: hipMemcpyAsync failed");
            rc = SAFE_E_HIP;
        }
        PermOut out{};
        out.ns = d_hits;
        out.mode = 0;
        if (rc == SAFE_OK)
            rc = bits ? launch_counts_bits(ctx, nbr, attr, col0, col1, d_hits)
                      : launch_gather(ctx, nbr, tiles, nullptr, 0, mloc, false, out);   // X = A . B0 (safe.py:593-594)
        if (rc == SAFE_OK) {                                                // (no early return: the join and the frees below must run)
            hipError_t e = hipEventRecord(ctx->k0, ctx->stream);
            hipLaunchKernelGGL(k_hypergeom_tail, dim3(ceil_div(mloc, 64), ceil_div(n, 4)), dim3(256), 0, ctx->stream, d_hits,
                               d_size, attr->col_sum, col0, n, mloc, static_cast<double>(pop), d_lf,
                               nes_p_cut(enrichment_threshold), pvalues_pos_dev, nes_dev, nes_binary_dev, d_enr);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(ctx->k1, ctx->stream);
            if (e != hipSuccess) {
                safe_set_error("safe_hypergeom: %s", hipGetErrorString(e));
                rc = SAFE_E_HIP;
            }
            ctx->last_kernel.name = "k_hypergeom_tail";
        }
        if (rc == SAFE_OK && safe_stream_sync(ctx->stream) != hipSuccess) rc = SAFE_E_HIP;   // lf is host memory
    }
    if (rc == SAFE_OK && !enriched_done) {
        hipLaunchKernelGGL(k_u32_to_f64, dim3(ceil_div(mloc, 256)), dim3(256), 0, ctx->stream, d_enr, num_enriched_dev, mloc);
        if (hipGetLastError() != hipSuccess) rc = SAFE_E_HIP;
    }
    if (rc == SAFE_OK) rc = finish_kernel_timing(ctx);
    if (safe_stream_sync(ctx->stream) != hipSuccess && rc == SAFE_OK) rc = SAFE_E_HIP;
    if (hs != ctx->stream) (void)safe_stream_sync(hs);
    if (split) mfma_counts_split_free(split);
    if (ev) (void)hipEventDestroy(ev);
    (void)hipFree(d_lf);
    (void)hipFree(tiles.bt);
    return rc;
}

int safe_outputs_from_counts(safe_ctx *ctx, int64_t n, int64_t m, int64_t num_permutations, int sign_mode,
                             double enrichment_threshold, const double *nes_table_host, const double *counts_neg_dev,
                             const double *counts_pos_dev, const double *ns_dev, double *pvalues_neg_dev, double *pvalues_pos_dev,
                             double *nes_dev, double *nes_binary_dev, double *num_enriched_dev) {
    SAFE_REQUIRE(ctx && counts_neg_dev && counts_pos_dev && pvalues_neg_dev && pvalues_pos_dev && nes_dev && nes_binary_dev &&
                     num_enriched_dev,
                 "safe_outputs_from_counts: NULL argument");
    SAFE_REQUIRE(n >= 1 && m >= 1 && num_permutations >= 1, "safe_outputs_from_counts: bad sizes");
    SAFE_REQUIRE(sign_mode >= SAFE_SIGN_HIGHEST && sign_mode <= SAFE_SIGN_BOTH, "safe_outputs_from_counts: bad sign_mode %d", sign_mode);
    SAFE_REQUIRE(enrichment_threshold > 0.0 && enrichment_threshold < 1.0, "safe_outputs_from_counts: enrichment_threshold must be in (0,1)");
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t P = num_permutations;
    std::vector<double> tab(P + 1);
    if (nes_table_host) {
        std::copy(nes_table_host, nes_table_host + P + 1, tab.begin());
    } else {
        tab[0] = -std::log10(1.0 / static_cast<double>(P));
        for (int64_t k = 1; k <= P; ++k) tab[k] = -std::log10(static_cast<double>(k) / static_cast<double>(P));
    }
    void *small = nullptr;                           // NES table f64 [P + 1] | enriched counters u32 [m + 16]
    SAFE_TRY(ctx_scratch(ctx, 10, static_cast<size_t>(P + 1) * sizeof(double) + static_cast<size_t>(m + 16) * sizeof(unsigned int), &small));
    double *d_tab = static_cast<double *>(small);
    unsigned int *d_enr = reinterpret_cast<unsigned int *>(d_tab + P + 1);
    SAFE_HIP_CHECK(hipMemcpyAsync(d_tab, tab.data(), (P + 1) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemsetAsync(d_enr, 0, (m + 16) * sizeof(unsigned int), ctx->stream));
    PermOut out{};
    out.pvalues_neg = pvalues_neg_dev;
    out.pvalues_pos = pvalues_pos_dev;
    out.nes = nes_dev;
    out.nes_binary = nes_binary_dev;
    out.enriched = d_enr;
    out.nes_table = d_tab;
    out.nes_threshold = -std::log10(enrichment_threshold);
    out.sign_mode = sign_mode;
    out.mode = 2;
    hipLaunchKernelGGL(k_counts_to_outputs, dim3(ceil_div(n * m, 256)), dim3(256), 0, ctx->stream, counts_neg_dev, counts_pos_dev, ns_dev,
                       n * m, m, P, out);
    hipLaunchKernelGGL(k_u32_to_f64, dim3(ceil_div(m, 256)), dim3(256), 0, ctx->stream, d_enr, num_enriched_dev, m);
    SAFE_HIP_CHECK(hipGetLastError());
    unsigned int *bad = nullptr;
    SAFE_TRY(ctx_pinned(ctx, sizeof(unsigned int), reinterpret_cast<void **>(&bad)));
    SAFE_HIP_CHECK(hipMemcpyAsync(bad, d_enr + m, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));                 // tab is a host vector
    if (*bad) {
        safe_set_error("safe_outputs_from_counts: a count lies outside [0, num_permutations = %lld] (or is NaN where the observed "
                       "score is not): the counts of a permutation-axis split must add up to ONE run of num_permutations",
                       (long long)P);
        return SAFE_E_VALUE;
    }
    return SAFE_OK;
}

int safe_export_packed_counts(safe_ctx *ctx, uint32_t *dst_dev, int64_t capacity, int64_t *n_pad, int64_t *m,
                              int *layout) {
    SAFE_REQUIRE(ctx && n_pad && m && layout, "safe_export_packed_counts: NULL argument");
    *layout = ctx->packed_layout;
    *n_pad = ctx->packed_layout >= 0 ? ctx->packed_n_pad : 0;
    *m = ctx->packed_layout >= 0 ? ctx->packed_m : 0;
    if (!dst_dev || ctx->packed_layout < 0) return SAFE_OK;
    const int64_t count = ctx->packed_n_pad * ctx->packed_m;
    SAFE_REQUIRE(capacity >= count, "safe_export_packed_counts: buffer holds %lld counters, %lld needed", (long long)capacity,
                 (long long)count);
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    SAFE_HIP_CHECK(hipMemcpyAsync(dst_dev, ctx->packed_counts, static_cast<size_t>(count) * sizeof(uint32_t),
                                  hipMemcpyDeviceToDevice, ctx->stream));
    return SAFE_OK;
}

int safe_outputs_from_packed_counts(safe_ctx *ctx, safe_nbr *nbr, const uint32_t *counts_dev, int layout, int64_t n_pad, int64_t m,
                                    int64_t num_permutations, int sign_mode, double enrichment_threshold,
                                    const double *nes_table_host, double *pvalues_neg_dev, double *pvalues_pos_dev,
                                    double *nes_dev, double *nes_binary_dev) {
    SAFE_REQUIRE(ctx && nbr && counts_dev, "safe_outputs_from_packed_counts: NULL argument");
    SAFE_REQUIRE(pvalues_neg_dev || pvalues_pos_dev || nes_dev || nes_binary_dev, "safe_outputs_from_packed_counts: no output requested");
    SAFE_REQUIRE(layout == 0 || layout == 1, "safe_outputs_from_packed_counts: bad layout %d", layout);
    SAFE_REQUIRE(sign_mode >= SAFE_SIGN_HIGHEST && sign_mode <= SAFE_SIGN_BOTH, "safe_outputs_from_packed_counts: bad sign_mode %d", sign_mode);
    SAFE_REQUIRE(num_permutations >= 1 && num_permutations <= 65535 && m >= 1, "safe_outputs_from_packed_counts: bad sizes");
    SAFE_REQUIRE(enrichment_threshold > 0.0 || !nes_binary_dev, "safe_outputs_from_packed_counts: enrichment_threshold must be positive");
    const int32_t *rowmap = layout == 0 ? nbr->sell_row : nbr->bs_rowmap;
    const int64_t want_pad = layout == 0 ? nbr->n_slices * 64 : nbr->bs_groups * 256;
    SAFE_REQUIRE(rowmap && n_pad == want_pad, "safe_outputs_from_packed_counts: counters are for %lld positions, the membership has %lld",
                 (long long)n_pad, (long long)want_pad);
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t P = num_permutations;
    std::vector<double> tab(P + 1);
    if (nes_table_host) {
        std::copy(nes_table_host, nes_table_host + P + 1, tab.begin());
    } else {
        tab[0] = -std::log10(1.0 / static_cast<double>(P));
        for (int64_t k = 1; k <= P; ++k) tab[k] = -std::log10(static_cast<double>(k) / static_cast<double>(P));
    }
    double *d_tab = nullptr;
    SAFE_TRY(ctx_scratch(ctx, 7, (P + 1) * sizeof(double), reinterpret_cast<void **>(&d_tab)));
    SAFE_HIP_CHECK(hipMemcpyAsync(d_tab, tab.data(), (P + 1) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PermOut out{};
    out.pvalues_neg = pvalues_neg_dev;
    out.pvalues_pos = pvalues_pos_dev;
    out.nes = nes_dev;
    out.nes_binary = nes_binary_dev;
    out.nes_table = d_tab;
    out.nes_threshold = enrichment_threshold > 0.0 ? -std::log10(enrichment_threshold) : 0.0;
    out.sign_mode = sign_mode;
    out.mode = 4;
    int rc = enrich_finalize_counts(ctx, counts_dev, n_pad, rowmap, m, P, out, nullptr);
    if (rc == SAFE_OK && safe_stream_sync(ctx->stream) != hipSuccess) rc = SAFE_E_HIP;   // tab is a host vector
    return rc;
}

int safe_set_exchange_chunks(safe_ctx *ctx, int chunks, int64_t cols_per_chunk, void (*on_enqueued)(void *), void *user) {
    SAFE_REQUIRE(ctx, "safe_set_exchange_chunks: NULL context");
    SAFE_REQUIRE(chunks == 0 || (chunks >= 1 && chunks <= safe_ctx::XC_MAX && cols_per_chunk >= 64 && cols_per_chunk % 64 == 0),
                 "safe_set_exchange_chunks: %d chunks of %lld columns (1..%d chunks of a multiple of 64 columns, or 0 to switch off)", chunks,
                 (long long)cols_per_chunk, safe_ctx::XC_MAX);
    ctx->xc_want = chunks;
    ctx->xc_cols = chunks ? cols_per_chunk : 0;
    ctx->xc_callback = chunks ? on_enqueued : nullptr;
    ctx->xc_user = chunks ? user : nullptr;
    if (chunks) ctx->xc_made = 0;                     // (switching off keeps what the last call did: safe_packed_chunk_info)
    return SAFE_OK;
}

int safe_packed_chunk_info(safe_ctx *ctx, int *chunks, int64_t *bounds, int64_t *tail_permutations) {
    SAFE_REQUIRE(ctx && chunks, "safe_packed_chunk_info: NULL argument");
    const int made = ctx->packed_layout == 0 ? ctx->xc_made : 0;
    *chunks = made;
    if (bounds)
        for (int k = 0; k <= made; ++k) bounds[k] = ctx->xc_bounds[k];
    if (tail_permutations) *tail_permutations = made ? ctx->xc_tail_perms : 0;
    return SAFE_OK;
}

int safe_export_packed_chunk(safe_ctx *ctx, int chunk, uint32_t *dst_dev, int64_t capacity, void *stream) {
    SAFE_REQUIRE(ctx && dst_dev, "safe_export_packed_chunk: NULL argument");
    SAFE_REQUIRE(ctx->packed_layout >= 0 && ctx->xc_want >= 1 && chunk >= 0 && chunk < ctx->xc_want,
                 "safe_export_packed_chunk: chunk %d of %d armed, counters %s", chunk, ctx->xc_want, ctx->packed_layout >= 0 ? "present" : "absent");
    // the armed grid, whether or not the call ran its tail chunk by chunk (it did not: a kernel form without the tail, too few
    // stages -- then the call has ended, the counters are final and no event is waited for)
    const int64_t c0 = std::min<int64_t>(chunk * ctx->xc_cols, ctx->packed_m);
    const int64_t c1 = chunk + 1 == ctx->xc_want ? ctx->packed_m : std::min<int64_t>((chunk + 1) * ctx->xc_cols, ctx->packed_m);
    const int64_t cells = (c1 - c0) * ctx->packed_n_pad;
    SAFE_REQUIRE(capacity >= cells, "safe_export_packed_chunk: buffer holds %lld counters, %lld needed", (long long)capacity, (long long)cells);
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    if (ctx->xc_made > 0) {
        SAFE_REQUIRE(ctx->xc_made == ctx->xc_want && ctx->xc_bounds[chunk] == c0 && ctx->xc_bounds[chunk + 1] == c1,
                     "safe_export_packed_chunk: the call's chunks are not the armed ones");
        SAFE_HIP_CHECK(hipStreamWaitEvent(s, ctx->xc_events[chunk], 0));
    }
    if (cells)
        SAFE_HIP_CHECK(hipMemcpyAsync(dst_dev, ctx->packed_counts + c0 * ctx->packed_n_pad, static_cast<size_t>(cells) * sizeof(uint32_t),
                                      hipMemcpyDeviceToDevice, s));
    if (capacity > cells)                              // (a rank with fewer columns: zero counters in the padding columns)
        SAFE_HIP_CHECK(hipMemsetAsync(dst_dev + cells, 0, static_cast<size_t>(capacity - cells) * sizeof(uint32_t), s));
    return SAFE_OK;
}

// counters (#less << 16 | #greater), both <= 1023  ->  20-bit pairs, two outputs in five bytes, column by column
__global__ __launch_bounds__(256) void k_pack_counts20(const unsigned int *__restrict__ counts, int64_t n_pad, int64_t cols,
                                                       unsigned int *__restrict__ dst) {
    const int64_t half = n_pad / 2, idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= cols * half) return;
    const int64_t c = idx / half, i = idx % half;
    const uint2 v = *reinterpret_cast<const uint2 *>(counts + c * n_pad + 2 * i);
    const unsigned long long x0 = ((v.x >> 16) << 10) | (v.x & 0x3FFu), x1 = ((v.y >> 16) << 10) | (v.y & 0x3FFu);
    const unsigned long long x = x0 | (x1 << 20);
    unsigned int *col = dst + c * (n_pad / 8 * 5);
    col[i] = static_cast<unsigned int>(x);
    reinterpret_cast<unsigned char *>(col + half)[i] = static_cast<unsigned char>(x >> 32);
}

int safe_export_packed_chunk_narrow(safe_ctx *ctx, int chunk, uint32_t *dst_dev, int64_t capacity_words, void *stream) {
    SAFE_REQUIRE(ctx && dst_dev, "safe_export_packed_chunk_narrow: NULL argument");
    SAFE_REQUIRE(ctx->packed_layout >= 0 && ctx->xc_want >= 1 && chunk >= 0 && chunk < ctx->xc_want,
                 "safe_export_packed_chunk_narrow: chunk %d of %d armed, counters %s", chunk, ctx->xc_want, ctx->packed_layout >= 0 ? "present" : "absent");
    SAFE_REQUIRE(ctx->packed_perms >= 1 && ctx->packed_perms <= 1023, "safe_export_packed_chunk_narrow: %lld permutations do not fit 10-bit counters",
                 (long long)ctx->packed_perms);
    const int64_t c0 = std::min<int64_t>(chunk * ctx->xc_cols, ctx->packed_m);
    const int64_t c1 = chunk + 1 == ctx->xc_want ? ctx->packed_m : std::min<int64_t>((chunk + 1) * ctx->xc_cols, ctx->packed_m);
    const int64_t words = (c1 - c0) * (ctx->packed_n_pad / 8 * 5);
    SAFE_REQUIRE(capacity_words >= words, "safe_export_packed_chunk_narrow: buffer holds %lld words, %lld needed", (long long)capacity_words, (long long)words);
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    if (ctx->xc_made > 0) {
        SAFE_REQUIRE(ctx->xc_made == ctx->xc_want && ctx->xc_bounds[chunk] == c0 && ctx->xc_bounds[chunk + 1] == c1,
                     "safe_export_packed_chunk_narrow: the call's chunks are not the armed ones");
        SAFE_HIP_CHECK(hipStreamWaitEvent(s, ctx->xc_events[chunk], 0));
    }
    if (words) {
        const int64_t pairs = (c1 - c0) * (ctx->packed_n_pad / 2);
        hipLaunchKernelGGL(k_pack_counts20, dim3(static_cast<unsigned int>(ceil_div(pairs, 256))), dim3(256), 0, s,
                           ctx->packed_counts + c0 * ctx->packed_n_pad, ctx->packed_n_pad, c1 - c0, dst_dev);
        SAFE_HIP_CHECK(hipGetLastError());
    }
    if (capacity_words > words)                        // (a rank with fewer columns: zero counters in the padding columns)
        SAFE_HIP_CHECK(hipMemsetAsync(dst_dev + words, 0, static_cast<size_t>(capacity_words - words) * sizeof(uint32_t), s));
    return SAFE_OK;
}

int safe_outputs_from_packed_slabs(safe_ctx *ctx, safe_nbr *nbr, const uint32_t *slabs_dev, int layout, int64_t n_pad, int n_slabs,
                                   int64_t slab_stride, const int64_t *slab_cols, const int64_t *out_col0, int64_t m_total,
                                   int64_t num_permutations, int sign_mode, double enrichment_threshold, const double *nes_table_host,
                                   double *pvalues_neg_dev, double *pvalues_pos_dev, double *nes_dev, double *nes_binary_dev,
                                   void *stream) {
    SAFE_REQUIRE(ctx && nbr && slabs_dev && slab_cols && out_col0 && n_slabs >= 1, "safe_outputs_from_packed_slabs: NULL argument");
    SAFE_REQUIRE(pvalues_neg_dev || pvalues_pos_dev || nes_dev || nes_binary_dev, "safe_outputs_from_packed_slabs: no output requested");
    const bool pk20 = (layout & SAFE_PACKED_NARROW) != 0;   // slabs of 20-bit pairs (safe_export_packed_chunk_narrow)
    layout &= ~SAFE_PACKED_NARROW;
    SAFE_REQUIRE(layout == 0 || layout == 1, "safe_outputs_from_packed_slabs: bad layout %d", layout);
    SAFE_REQUIRE(!pk20 || num_permutations <= 1023, "safe_outputs_from_packed_slabs: 20-bit pairs hold at most 1023 permutations");
    SAFE_REQUIRE(sign_mode >= SAFE_SIGN_HIGHEST && sign_mode <= SAFE_SIGN_BOTH, "safe_outputs_from_packed_slabs: bad sign_mode %d", sign_mode);
    SAFE_REQUIRE(num_permutations >= 1 && num_permutations <= 65535 && m_total >= 1, "safe_outputs_from_packed_slabs: bad sizes");
    SAFE_REQUIRE(enrichment_threshold > 0.0 || !nes_binary_dev, "safe_outputs_from_packed_slabs: enrichment_threshold must be positive");
    const int32_t *rowmap = layout == 0 ? nbr->sell_row : nbr->bs_rowmap;
    const int64_t want_pad = layout == 0 ? nbr->n_slices * 64 : nbr->bs_groups * 256;
    SAFE_REQUIRE(rowmap && n_pad == want_pad, "safe_outputs_from_packed_slabs: counters are for %lld positions, the membership has %lld",
                 (long long)n_pad, (long long)want_pad);
    for (int r = 0; r < n_slabs; ++r)
        SAFE_REQUIRE(slab_cols[r] >= 0 && slab_cols[r] * (pk20 ? n_pad / 8 * 5 : n_pad) <= slab_stride && out_col0[r] >= 0 && out_col0[r] + slab_cols[r] <= m_total,
                     "safe_outputs_from_packed_slabs: slab %d (%lld columns at column %lld) does not fit", r, (long long)slab_cols[r],
                     (long long)out_col0[r]);
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    const int64_t P = num_permutations;
    std::vector<double> tab(P + 1);
    if (nes_table_host) {
        std::copy(nes_table_host, nes_table_host + P + 1, tab.begin());
    } else {
        tab[0] = -std::log10(1.0 / static_cast<double>(P));
        for (int64_t k = 1; k <= P; ++k) tab[k] = -std::log10(static_cast<double>(k) / static_cast<double>(P));
    }
    // the table stays on the device between calls (one call per column chunk of a step: no upload, no sync after the first)
    double *d_tab = nullptr;
    const bool grow = ctx->scratch_bytes[19] < (P + 1) * sizeof(double);
    SAFE_TRY(ctx_scratch(ctx, 19, (P + 1) * sizeof(double), reinterpret_cast<void **>(&d_tab)));
    if (grow || ctx->nes_tab_host != tab) {
        SAFE_HIP_CHECK(hipMemcpyAsync(d_tab, tab.data(), (P + 1) * sizeof(double), hipMemcpyHostToDevice, s));
        SAFE_HIP_CHECK(safe_stream_sync(s));                                                  // (tab is a host vector)
        ctx->nes_tab_host = tab;
    }
    for (int r = 0; r < n_slabs; ++r) {
        if (slab_cols[r] == 0) continue;
        PermOut out{};
        out.pvalues_neg = pvalues_neg_dev ? pvalues_neg_dev + out_col0[r] : nullptr;
        out.pvalues_pos = pvalues_pos_dev ? pvalues_pos_dev + out_col0[r] : nullptr;
        out.nes = nes_dev ? nes_dev + out_col0[r] : nullptr;
        out.nes_binary = nes_binary_dev ? nes_binary_dev + out_col0[r] : nullptr;
        out.nes_table = d_tab;
        out.nes_threshold = enrichment_threshold > 0.0 ? -std::log10(enrichment_threshold) : 0.0;
        out.sign_mode = sign_mode;
        out.mode = 4;
        out.ld = m_total;
        SAFE_TRY(enrich_finalize_counts(ctx, slabs_dev + static_cast<int64_t>(r) * slab_stride, n_pad, rowmap, slab_cols[r], P, out, nullptr, s, pk20));
    }
    return SAFE_OK;
}

int safe_randomization_plan(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, int64_t num_permutations, int score_type, int *packed_layout) {
    SAFE_REQUIRE(ctx && nbr && attr && packed_layout, "safe_randomization_plan: NULL argument");
    SAFE_REQUIRE(score_type == SAFE_SCORE_SUM || score_type == SAFE_SCORE_ZSCORE, "safe_randomization_plan: bad score_type %d", score_type);
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    // what safe_randomization would run for this block: only the bit-sliced form is predicted (layout 0); everything else -1
    *packed_layout = choose_path(ctx, nbr, attr, num_permutations, score_type == SAFE_SCORE_ZSCORE) == PATH_BITS ? 0 : -1;
    return SAFE_OK;
}

int safe_nes_from_packed_counts(safe_ctx *ctx, safe_nbr *nbr, const uint32_t *counts_dev, int layout, int64_t n_pad, int64_t m,
                                int64_t num_permutations, int sign_mode, const double *nes_table_host, double *nes_dev) {
    SAFE_REQUIRE(nes_dev, "safe_nes_from_packed_counts: NULL argument");
    return safe_outputs_from_packed_counts(ctx, nbr, counts_dev, layout, n_pad, m, num_permutations, sign_mode, 0.05, nes_table_host,
                                           nullptr, nullptr, nes_dev, nullptr);
}

}  // extern "C"
