// Neighborhood definition on gfx950: all-pairs Euclidean distance + threshold, bounded
// all-pairs shortest paths, and the derived CSR / SELL-64 membership forms.
//
// Replaces SAFE.define_neighborhoods (safepy/safe.py:369-430) and
// calculate_edge_lengths (safepy/safe_io.py:311-333).  Compiled with -ffp-contract=off:
// the reference arithmetic is scipy pdist's sqrt(dx*dx + dy*dy) with every operation
// rounded separately, and bit-exact masks need the same roundings.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "common.h"

// --------------------------------------------------------------------------------------
// Strict threshold without a square root.  sqrt is correctly rounded and monotone, so
//   sqrt(s) < nr   <=>   s < T,   T = min{ s : sqrt(s) >= nr }.
// T is found on the host by stepping from nr*nr to the exact boundary.
// --------------------------------------------------------------------------------------
static double squared_threshold(double nr) {
    if (!(nr > 0.0)) return 0.0;                 // D >= 0 is never < nr
    if (std::isinf(nr)) return nr;
    double c = nr * nr;
    if (std::isinf(c)) return c;
    while (c > 0.0 && std::sqrt(c) >= nr) c = std::nextafter(c, 0.0);
    while (std::sqrt(c) < nr) c = std::nextafter(c, INFINITY);
    return c;
}

// --------------------------------------------------------------------------------------
// K1a: distance + threshold -> bit matrix.  One lane owns one (row, 64-column word):
// the 64 lanes of a wave share the word index, so x[j], y[j] are LDS broadcasts.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_euclid_bits(const double *__restrict__ xy, int64_t n, double thr_sq,
                                                     uint64_t *__restrict__ bits, int64_t words) {
    __shared__ double sx[64], sy[64];
    const int64_t w = blockIdx.x;                                   // word (column block)
    const int64_t i = static_cast<int64_t>(blockIdx.y) * 256 + threadIdx.x;
    const int64_t j0 = w * 64;
    if (threadIdx.x < 64) {
        const int64_t j = j0 + threadIdx.x;
        sx[threadIdx.x] = j < n ? xy[2 * j] : 0.0;
        sy[threadIdx.x] = j < n ? xy[2 * j + 1] : 0.0;
    }
    __syncthreads();
    if (i >= n) return;
    const double xi = xy[2 * i], yi = xy[2 * i + 1];
    const int jn = static_cast<int>(n - j0 < 64 ? n - j0 : 64);
    uint64_t word = 0;
#pragma unroll 8
    for (int t = 0; t < 64; ++t) {
        const double dx = xi - sx[t];
        const double dy = yi - sy[t];
        const double s = dx * dx + dy * dy;                          // no FMA (-ffp-contract=off)
        word |= static_cast<uint64_t>((s < thr_sq) & (t < jn)) << t;
    }
    bits[i * words + w] = word;
}

// --------------------------------------------------------------------------------------
// K1b: the fused all-pairs kernel in the reference's own output layout: int64 [n,n]
// membership and/or f64 [n,n] distances.  HBM-write bound, so the kernel is shaped by what
// the memory system likes for pure stores (tools/ubench/store_ceiling.hip: 5.5 TB/s for
// independent 512 x 32 tiles, up to 6.5 TB/s when few waves per CU sweep one compact window --
// which a kernel with twelve f64 operations per store cannot keep fed): persistent workgroups,
// four per CU; workgroup b owns the 512-column chunk b % chunks_per_row of the rows
// b / chunks_per_row, + R, + 2R, ... -- at any moment they write R whole consecutive rows.  A lane keeps its two columns'
// coordinates in registers for all its rows (16 B per lane and row, 1 KiB per wave); the row's
// coordinates are wave-uniform (scalar loads).
// --------------------------------------------------------------------------------------
#define K1B_BATCH 8
// MODE bit 0: membership out, bit 1: distances out; VEC: 16-byte pair stores (even n)
template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void k_euclid_dense(const double *__restrict__ xy, int64_t n, double thr_sq, int chunks_per_row,
                                                      int rows_per_sweep, int64_t *__restrict__ mask, double *__restrict__ dist) {
    const int chunk = blockIdx.x % chunks_per_row;
    const int64_t j = (static_cast<int64_t>(chunk) * 256 + threadIdx.x) * 2;
    if (j >= n) return;
    const bool two = (j + 1 < n);
    const double xa = xy[2 * j], ya = xy[2 * j + 1];
    const double xb = two ? xy[2 * j + 2] : 0.0, yb = two ? xy[2 * j + 3] : 0.0;
    auto row = [&](int64_t i, double xi, double yi) __attribute__((always_inline)) {
        const double dxa = xi - xa, dya = yi - ya;
        const double dxb = xi - xb, dyb = yi - yb;
        const double sa = dxa * dxa + dya * dya;          // no FMA (-ffp-contract=off)
        const double sb = dxb * dxb + dyb * dyb;
        const int64_t o = i * n + j;
        if (MODE & 1) {
            const int64_t ma = sa < thr_sq, mb = sb < thr_sq;
            if (VEC) {
                *reinterpret_cast<longlong2 *>(mask + o) = make_longlong2(ma, mb);
            } else {
                mask[o] = ma;
                if (two) mask[o + 1] = mb;
            }
        }
        if (MODE & 2) {
            const double da = sqrt(sa), db = sqrt(sb);
            if (VEC) {
                *reinterpret_cast<double2 *>(dist + o) = make_double2(da, db);
            } else {
                dist[o] = da;
                if (two) dist[o + 1] = db;
            }
        }
    };
    // rows in batches of K1B_BATCH: all their (wave-uniform) coordinates are requested before the first is used, so the
    // scalar-load latency is paid once per batch and the stores of a batch go out back to back
    const int64_t step = static_cast<int64_t>(K1B_BATCH) * rows_per_sweep;
    int64_t i0 = blockIdx.x / chunks_per_row;
    for (; i0 + step - rows_per_sweep < n; i0 += step) {   // whole batches
        double xi[K1B_BATCH], yi[K1B_BATCH];
#pragma unroll
        for (int u = 0; u < K1B_BATCH; ++u) {
            const int64_t i = i0 + static_cast<int64_t>(u) * rows_per_sweep;
            xi[u] = xy[2 * i];
            yi[u] = xy[2 * i + 1];
        }
#pragma unroll
        for (int u = 0; u < K1B_BATCH; ++u) row(i0 + static_cast<int64_t>(u) * rows_per_sweep, xi[u], yi[u]);
    }
    for (; i0 < n; i0 += rows_per_sweep) row(i0, xy[2 * i0], xy[2 * i0 + 1]);
}

__global__ void k_edge_lengths(const double *__restrict__ xy, int64_t n_edges, const int32_t *__restrict__ eu,
                               const int32_t *__restrict__ ev, double *__restrict__ out) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int64_t u = eu[e], v = ev[e];
    const double dx = xy[2 * u] - xy[2 * v];
    const double dy = xy[2 * u + 1] - xy[2 * v + 1];
    out[e] = sqrt(dx * dx + dy * dy);
}

// --------------------------------------------------------------------------------------
// dense int64 <-> bit matrix
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dense_to_bits(const int64_t *__restrict__ a, int64_t n, int64_t words,
                                                       uint64_t *__restrict__ bits, int *__restrict__ bad) {
    // one wave per (row, word): lane t tests column 64*w + t, ballot packs the word
    const int64_t wave = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n * words) return;
    const int64_t i = wave / words, w = wave % words;
    const int64_t j = w * 64 + lane;
    int64_t v = 0;
    if (j < n) v = a[i * n + j];
    if (v != 0 && v != 1) atomicOr(bad, 1);
    const uint64_t word = __ballot(v != 0);
    if (lane == 0) bits[wave] = word;
}

__global__ __launch_bounds__(256) void k_bits_to_dense(const uint64_t *__restrict__ bits, int64_t n, int64_t words,
                                                       int64_t *__restrict__ out) {
    // lane writes two adjacent int64 (16 B); a wave covers 128 columns = two words
    const int64_t j = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 2;
    const int64_t i = blockIdx.y;
    if (j >= n) return;
    const uint64_t word = bits[i * words + (j >> 6)];
    const int64_t a = (word >> (j & 63)) & 1, b = (word >> ((j + 1) & 63)) & 1;   // j even => same word
    const int64_t o = i * n + j;
    if (j + 1 < n && (n & 1) == 0) {
        *reinterpret_cast<longlong2 *>(out + o) = make_longlong2(a, b);
    } else {
        out[o] = a;
        if (j + 1 < n) out[o + 1] = b;
    }
}

__global__ void k_row_popcount(const uint64_t *__restrict__ bits, int64_t n, int64_t words,
                               int32_t *__restrict__ count) {
    // one wave per row
    const int64_t row = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    int c = 0;
    for (int64_t w = lane; w < words; w += 64) c += __popcll(bits[row * words + w]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if (lane == 0) count[row] = c;
}

__global__ void k_fill_csr(const uint64_t *__restrict__ bits, int64_t n, int64_t words,
                           const int32_t *__restrict__ row_ptr, int32_t *__restrict__ col) {
    // one wave per row; lanes take words round-robin, an exclusive wave scan orders the output
    const int64_t row = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    int base = row_ptr[row];
    for (int64_t w0 = 0; w0 < words; w0 += 64) {
        const int64_t w = w0 + lane;
        uint64_t word = w < words ? bits[row * words + w] : 0;
        const int c = __popcll(word);
        int incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        int pos = base + incl - c;
        while (word) {
            const int b = __ffsll(static_cast<unsigned long long>(word)) - 1;
            col[pos++] = static_cast<int32_t>(w * 64 + b);
            word &= word - 1;
        }
        base += __shfl(incl, 63);
    }
}

__global__ void k_fill_sell(const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                            const int32_t *__restrict__ sell_row, const int64_t *__restrict__ slice_off,
                            const int32_t *__restrict__ slice_width, int64_t n_slices, int32_t pad,
                            int32_t *__restrict__ sell_col) {
    const int64_t s = blockIdx.x;
    const int lane = threadIdx.x;     // 64 threads
    if (s >= n_slices) return;
    const int32_t r = sell_row[s * 64 + lane];
    const int32_t beg = r >= 0 ? row_ptr[r] : 0;
    const int32_t cnt = r >= 0 ? row_ptr[r + 1] - beg : 0;
    const int64_t off = slice_off[s];
    const int32_t wdt = slice_width[s];
    for (int32_t t = 0; t < wdt; ++t) sell_col[off + static_cast<int64_t>(t) * 64 + lane] = t < cnt ? col[beg + t] : pad;
}

__global__ void k_sell_scale16(const int32_t *__restrict__ sell_col, int64_t entries, int64_t total, uint32_t pad2,
                               uint16_t *__restrict__ out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < total) out[i] = i < entries ? static_cast<uint16_t>(2 * sell_col[i]) : static_cast<uint16_t>(pad2);
}

// The same 2*id list in BLOCKED order: inside a slice, block b (8 members) of lane l is the 16 bytes at
// entry offset (b * 64 + l) * 8 -- a lane fetches a block with ONE 16-byte load (64 lanes: 1 KiB contiguous)
// instead of eight 2-byte loads (k_permtest_bits_blk).  The order of a lane's members is free: sums commute.
__global__ void k_sell_blocked16(const int32_t *__restrict__ sell_col, const int64_t *__restrict__ slice_off,
                                 const int32_t *__restrict__ slice_width, int64_t n_slices, int64_t entries, int64_t total,
                                 uint32_t pad2, uint16_t *__restrict__ out) {
    const int64_t s = blockIdx.x;
    if (s == n_slices) {                                       // the tail one prefetched block may touch
        for (int64_t i = entries + threadIdx.x; i < total; i += blockDim.x) out[i] = static_cast<uint16_t>(pad2);
        return;
    }
    const int64_t off = slice_off[s];
    const int64_t cnt = static_cast<int64_t>(slice_width[s]) * 64;
    for (int64_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        const int64_t t = i >> 6, lane = i & 63;
        out[off + ((t >> 3) * 64 + lane) * 8 + (t & 7)] = static_cast<uint16_t>(2 * sell_col[off + i]);
    }
}

// --------------------------------------------------------------------------------------
// K2: bounded all-pairs shortest paths.  One wave per source, label-correcting frontier
// relaxation to a fixpoint over the CSR adjacency.  cand = dist[v] + w is an f64 add in
// the same order Dijkstra performs it, a candidate is dropped iff cand > cutoff, and the
// kept minimum is independent of relaxation order (f64 add is monotone), so distances and
// the reached set equal networkx's bounded Dijkstra bit for bit.
// Non-negative doubles order like their bit patterns, so the per-wave distance array is
// updated with a 64-bit integer atomicMin.
// --------------------------------------------------------------------------------------
struct SpScratch {
    unsigned long long *dist;   // [workers][n]  f64 bits, init +inf
    int32_t *qa, *qb;           // [workers][n]  frontier queues
    int32_t *reached;           // [workers][n]
    int32_t *flag;              // [workers][n]  0 = not queued for next round
};

__global__ __launch_bounds__(64) void k_shortpath(int64_t n, const int32_t *__restrict__ adj_ptr,
                                                  const int32_t *__restrict__ adj_col,
                                                  const double *__restrict__ adj_w, double cutoff, SpScratch sc,
                                                  int64_t n_workers, uint64_t *__restrict__ bits, int64_t words,
                                                  double *__restrict__ dist_out) {
    const int64_t worker = blockIdx.x;
    const int lane = threadIdx.x;
    unsigned long long *dist = sc.dist + worker * n;
    int32_t *qcur = sc.qa + worker * n;
    int32_t *qnext = sc.qb + worker * n;
    int32_t *reached = sc.reached + worker * n;
    int32_t *flag = sc.flag + worker * n;
    __shared__ int s_nnext, s_nreached;
    const unsigned long long INF_BITS = 0x7FF0000000000000ull;

    for (int64_t src = worker; src < n; src += n_workers) {
        if (lane == 0) {
            atomicExch(&dist[src], 0ull);
            qcur[0] = static_cast<int32_t>(src);
            reached[0] = static_cast<int32_t>(src);
            s_nreached = 1;
            s_nnext = 0;
        }
        __syncthreads();
        int ncur = 1;
        while (ncur > 0) {
            for (int q = lane; q < ncur; q += 64) {
                const int32_t v = qcur[q];
                atomicExch(&flag[v], 0);     // dist/flag are only ever touched by L2 atomics (no stale L1 reads)
            }
            __syncthreads();
            for (int q = lane; q < ncur; q += 64) {
                const int32_t v = qcur[q];
                const double dv = __longlong_as_double(static_cast<long long>(
                    __hip_atomic_load(&dist[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
                const int32_t e1 = adj_ptr[v + 1];
                for (int32_t e = adj_ptr[v]; e < e1; ++e) {
                    const int32_t u = adj_col[e];
                    const double cand = dv + (adj_w ? adj_w[e] : 1.0);
                    if (cand > cutoff) continue;
                    const unsigned long long cb = static_cast<unsigned long long>(__double_as_longlong(cand));
                    const unsigned long long old = atomicMin(&dist[u], cb);
                    if (cb < old) {
                        if (old == INF_BITS) reached[atomicAdd(&s_nreached, 1)] = u;
                        if (atomicExch(&flag[u], 1) == 0) qnext[atomicAdd(&s_nnext, 1)] = u;
                    }
                }
            }
            __syncthreads();
            ncur = s_nnext;
            __syncthreads();
            if (lane == 0) s_nnext = 0;
            int32_t *t = qcur;
            qcur = qnext;
            qnext = t;
            __syncthreads();
        }
        const int nr = s_nreached;
        for (int q = lane; q < nr; q += 64) {
            const int32_t u = reached[q];
            atomicOr(reinterpret_cast<unsigned long long *>(&bits[src * words + (u >> 6)]), 1ull << (u & 63));
            const unsigned long long du = atomicExch(&dist[u], INF_BITS);
            if (dist_out) dist_out[src * n + u] = __longlong_as_double(static_cast<long long>(du));
            atomicExch(&flag[u], 0);
        }
        __syncthreads();
    }
}

__global__ void k_count_cols(const int32_t *__restrict__ col, int64_t nnz, int32_t *__restrict__ cnt) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e < nnz) atomicAdd(&cnt[col[e]], 1);
}

__global__ void k_fill_transpose(const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col, int64_t n,
                                 const int32_t *__restrict__ at_ptr, int32_t *__restrict__ cursor,
                                 int32_t *__restrict__ at_col) {
    // one wave per row
    const int64_t row = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    for (int32_t e = row_ptr[row] + lane; e < row_ptr[row + 1]; e += 64) {
        const int32_t k = col[e];
        at_col[at_ptr[k] + atomicAdd(&cursor[k], 1)] = static_cast<int32_t>(row);
    }
}

__global__ void k_fill_u64(unsigned long long *p, unsigned long long v, int64_t count) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        p[i] = v;
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
static void nbr_free(safe_nbr *nbr) {
    if (!nbr) return;
    (void)hipFree(nbr->bits);
    (void)hipFree(nbr->row_ptr);
    (void)hipFree(nbr->col);
    (void)hipFree(nbr->sell_row);
    (void)hipFree(nbr->sell_pos);
    (void)hipFree(nbr->slice_off);
    (void)hipFree(nbr->slice_width);
    (void)hipFree(nbr->sell_col);
    (void)hipFree(nbr->sell_col2);
    (void)hipFree(nbr->sell_col2b);
    (void)hipFree(nbr->dist);
    (void)hipFree(nbr->at_ptr);
    (void)hipFree(nbr->at_col);
    nbr_free_blocks(nbr);
    if (nbr->bits_plan_pinned) (void)hipHostFree(nbr->bits_plan_pinned);
    delete nbr;
}

static int nbr_new(safe_ctx *ctx, int64_t n, safe_nbr **out) {
    SAFE_REQUIRE(n >= 1 && n < (1ll << 31) - 64, "neighborhood size n=%lld out of range", (long long)n);
    safe_nbr *nbr = new safe_nbr();
    nbr->ctx = ctx;
    nbr->n = n;
    nbr->words = ceil_div(n, 64);
    int rc = dev_alloc(&nbr->bits, static_cast<size_t>(n) * nbr->words);
    if (rc != SAFE_OK) {
        nbr_free(nbr);
        return rc;
    }
    *out = nbr;
    return SAFE_OK;
}

int nbr_finalize_from_bits(safe_nbr *nbr) {
    safe_ctx *ctx = nbr->ctx;
    const int64_t n = nbr->n;
    int32_t *d_count = nullptr;
    SAFE_TRY(dev_alloc(&d_count, n));
    hipLaunchKernelGGL(k_row_popcount, dim3(ceil_div(n * 64, 256)), dim3(256), 0, ctx->stream, nbr->bits, n,
                       nbr->words, d_count);
    nbr->h_row_count.resize(n);
    SAFE_HIP_CHECK(hipMemcpyAsync(nbr->h_row_count.data(), d_count, n * sizeof(int32_t), hipMemcpyDeviceToHost,
                                  ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    (void)hipFree(d_count);

    std::vector<int32_t> row_ptr(n + 1, 0);
    int64_t nnz = 0, max_count = 0;
    for (int64_t i = 0; i < n; ++i) {
        nnz += nbr->h_row_count[i];
        max_count = std::max<int64_t>(max_count, nbr->h_row_count[i]);
        SAFE_REQUIRE(nnz < (1ll << 31), "membership has too many entries for int32 CSR offsets");
        row_ptr[i + 1] = static_cast<int32_t>(nnz);
    }
    nbr->nnz = nnz;
    nbr->max_count = max_count;

    // SELL-64: rows sorted by descending count (stable: ties keep node order)
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return nbr->h_row_count[a] > nbr->h_row_count[b];
    });
    nbr->n_slices = ceil_div(n, 64);
    std::vector<int32_t> sell_row(nbr->n_slices * 64, -1);
    std::copy(order.begin(), order.end(), sell_row.begin());
    std::vector<int32_t> sell_pos(n);
    for (int64_t t = 0; t < n; ++t) sell_pos[order[t]] = static_cast<int32_t>(t);
    nbr->h_slice_width.assign(nbr->n_slices, 0);
    nbr->h_slice_off.assign(nbr->n_slices + 1, 0);
    for (int64_t s = 0; s < nbr->n_slices; ++s) {
        int32_t wdt = nbr->h_row_count[order[s * 64]];              // first row of the slice is its largest
        wdt = (wdt + 7) & ~7;                                       // unroll granule of the gather loops (8-input carry-save blocks)
        nbr->h_slice_width[s] = wdt;
        nbr->h_slice_off[s + 1] = nbr->h_slice_off[s] + static_cast<int64_t>(wdt) * 64;
    }
    nbr->sell_entries = nbr->h_slice_off[nbr->n_slices];

    SAFE_TRY(dev_alloc(&nbr->row_ptr, n + 1));
    SAFE_TRY(dev_alloc(&nbr->col, nnz));
    SAFE_TRY(dev_alloc(&nbr->sell_row, nbr->n_slices * 64));
    SAFE_TRY(dev_alloc(&nbr->sell_pos, n));
    SAFE_HIP_CHECK(hipMemcpyAsync(nbr->sell_pos, sell_pos.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    SAFE_TRY(dev_alloc(&nbr->slice_off, nbr->n_slices + 1));
    SAFE_TRY(dev_alloc(&nbr->slice_width, nbr->n_slices));
    SAFE_TRY(dev_alloc(&nbr->sell_col, nbr->sell_entries));
    SAFE_HIP_CHECK(hipMemcpyAsync(nbr->row_ptr, row_ptr.data(), (n + 1) * sizeof(int32_t), hipMemcpyHostToDevice,
                                  ctx->stream));
    SAFE_HIP_CHECK(hipMemcpyAsync(nbr->sell_row, sell_row.data(), sell_row.size() * sizeof(int32_t),
                                  hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemcpyAsync(nbr->slice_off, nbr->h_slice_off.data(), (nbr->n_slices + 1) * sizeof(int64_t),
                                  hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemcpyAsync(nbr->slice_width, nbr->h_slice_width.data(), nbr->n_slices * sizeof(int32_t),
                                  hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_fill_csr, dim3(ceil_div(n * 64, 256)), dim3(256), 0, ctx->stream, nbr->bits, n, nbr->words,
                       nbr->row_ptr, nbr->col);
    hipLaunchKernelGGL(k_fill_sell, dim3(nbr->n_slices), dim3(64), 0, ctx->stream, nbr->row_ptr, nbr->col,
                       nbr->sell_row, nbr->slice_off, nbr->slice_width, nbr->n_slices, static_cast<int32_t>(n),
                       nbr->sell_col);
    if (n < 32768) {
        const int64_t total = nbr->sell_entries + 1024;         // tail: two blocks of 8 x 64 may be prefetched past the end
        SAFE_TRY(dev_alloc(&nbr->sell_col2, total));
        hipLaunchKernelGGL(k_sell_scale16, dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, nbr->sell_col,
                           nbr->sell_entries, total, static_cast<uint32_t>(2 * n), nbr->sell_col2);
        SAFE_TRY(dev_alloc(&nbr->sell_col2b, total));
        hipLaunchKernelGGL(k_sell_blocked16, dim3(nbr->n_slices + 1), dim3(256), 0, ctx->stream, nbr->sell_col, nbr->slice_off,
                           nbr->slice_width, nbr->n_slices, nbr->sell_entries, total, static_cast<uint32_t>(2 * n),
                           nbr->sell_col2b);
    }
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));   // host vectors above go out of scope
    return SAFE_OK;
}

int nbr_build_transpose(safe_nbr *nbr) {
    if (nbr->at_ptr) return SAFE_OK;
    safe_ctx *ctx = nbr->ctx;
    const int64_t n = nbr->n, nnz = nbr->nnz;
    int32_t *d_cnt = nullptr;
    SAFE_TRY(dev_alloc(&d_cnt, n));
    SAFE_HIP_CHECK(hipMemsetAsync(d_cnt, 0, n * sizeof(int32_t), ctx->stream));
    if (nnz) hipLaunchKernelGGL(k_count_cols, dim3(ceil_div(nnz, 256)), dim3(256), 0, ctx->stream, nbr->col, nnz, d_cnt);
    std::vector<int32_t> cnt(n), ptr(n + 1, 0);
    SAFE_HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    for (int64_t k = 0; k < n; ++k) ptr[k + 1] = ptr[k] + cnt[k];
    SAFE_TRY(dev_alloc(&nbr->at_ptr, n + 1));
    SAFE_TRY(dev_alloc(&nbr->at_col, nnz));
    SAFE_HIP_CHECK(hipMemcpyAsync(nbr->at_ptr, ptr.data(), (n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemsetAsync(d_cnt, 0, n * sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(k_fill_transpose, dim3(ceil_div(n * 64, 256)), dim3(256), 0, ctx->stream, nbr->row_ptr, nbr->col, n,
                       nbr->at_ptr, d_cnt, nbr->at_col);
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    (void)hipFree(d_cnt);
    return SAFE_OK;
}

extern "C" {

int safe_nbr_euclidean(safe_ctx *ctx, const double *xy_host, int64_t n, double nr, safe_nbr **out) {
    SAFE_REQUIRE(ctx && xy_host && out, "safe_nbr_euclidean: NULL argument");
    *out = nullptr;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    safe_nbr *nbr = nullptr;
    SAFE_TRY(nbr_new(ctx, n, &nbr));
    double *d_xy = nullptr;
    int rc = dev_alloc(&d_xy, 2 * n);
    if (rc != SAFE_OK) {
        nbr_free(nbr);
        return rc;
    }
    hipError_t e = hipMemcpyAsync(d_xy, xy_host, 2 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_euclid_bits, dim3(nbr->words, ceil_div(n, 256)), dim3(256), 0, ctx->stream, d_xy, n,
                           squared_threshold(nr), nbr->bits, nbr->words);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        safe_set_error("safe_nbr_euclidean: %s", hipGetErrorString(e));
        (void)hipFree(d_xy);
        nbr_free(nbr);
        return SAFE_E_HIP;
    }
    rc = nbr_finalize_from_bits(nbr);
    (void)hipFree(d_xy);
    if (rc != SAFE_OK) {
        nbr_free(nbr);
        return rc;
    }
    nbr->h_xy.assign(xy_host, xy_host + 2 * n);       // node order of the block-sparse form (mfma.hip)
    *out = nbr;
    return SAFE_OK;
}

int safe_nbr_block_count(const safe_nbr *nbr, int64_t *blocks) {
    SAFE_REQUIRE(nbr && blocks, "safe_nbr_block_count: NULL argument");
    *blocks = nbr->blocks_ready ? nbr->bs_blocks : 0;
    return SAFE_OK;
}

int safe_nbr_piece_count(const safe_nbr *nbr, int64_t *pieces) {
    SAFE_REQUIRE(nbr && pieces, "safe_nbr_piece_count: NULL argument");
    *pieces = nbr->blocks_ready ? nbr->bs_pieces : 0;
    return SAFE_OK;
}

int safe_nbr_set_layout(safe_nbr *nbr, const double *xy_host) {
    SAFE_REQUIRE(nbr && xy_host, "safe_nbr_set_layout: NULL argument");
    nbr->h_xy.assign(xy_host, xy_host + 2 * nbr->n);
    if (nbr->blocks_ready) nbr_free_blocks(nbr);      // rebuilt in the new order on next use
    return SAFE_OK;
}

int safe_euclidean_dense_dev(safe_ctx *ctx, const double *xy_dev, int64_t n, double nr, int64_t *mask_out_dev,
                             double *dist_out_dev) {
    SAFE_REQUIRE(ctx && xy_dev && n >= 1, "safe_euclidean_dense_dev: bad argument");
    SAFE_REQUIRE(mask_out_dev || dist_out_dev, "safe_euclidean_dense_dev: no output requested");
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t wgs = 4 * static_cast<int64_t>(ctx->num_cu);   // resident workgroups (16 waves per CU hide the scalar-load and f64 latency); they sweep whole rows together
    const int64_t chunks_per_row = ceil_div(n, 512);
    const int64_t rows_per_sweep = std::max<int64_t>(1, std::min<int64_t>(n, wgs / chunks_per_row));
    const int mode = (mask_out_dev ? 1 : 0) | (dist_out_dev ? 2 : 0);
    const bool vec = (n & 1) == 0;                         // 16-byte pair stores need even n (row starts stay 16-byte aligned)
    const dim3 grid(chunks_per_row * rows_per_sweep), block(256);
    const double thr_sq = squared_threshold(nr);
    const int cpr = static_cast<int>(chunks_per_row), rps = static_cast<int>(rows_per_sweep);
#define LAUNCH_K1B(M, V) hipLaunchKernelGGL((k_euclid_dense<M, V>), grid, block, 0, ctx->stream, xy_dev, n, thr_sq, cpr, rps, mask_out_dev, dist_out_dev)
    if (vec) {
        if (mode == 1) LAUNCH_K1B(1, true);
        else if (mode == 2) LAUNCH_K1B(2, true);
        else LAUNCH_K1B(3, true);
    } else {
        if (mode == 1) LAUNCH_K1B(1, false);
        else if (mode == 2) LAUNCH_K1B(2, false);
        else LAUNCH_K1B(3, false);
    }
#undef LAUNCH_K1B
    SAFE_HIP_CHECK(hipGetLastError());
    return SAFE_OK;
}

int safe_edge_lengths(safe_ctx *ctx, const double *xy_host, int64_t n, int64_t n_edges, const int32_t *edge_u,
                      const int32_t *edge_v, double *out_host) {
    SAFE_REQUIRE(ctx && xy_host && n >= 1 && n_edges >= 0, "safe_edge_lengths: bad argument");
    if (n_edges == 0) return SAFE_OK;
    SAFE_REQUIRE(edge_u && edge_v && out_host, "safe_edge_lengths: NULL argument");
    for (int64_t e = 0; e < n_edges; ++e)
        SAFE_REQUIRE(edge_u[e] >= 0 && edge_u[e] < n && edge_v[e] >= 0 && edge_v[e] < n,
                     "safe_edge_lengths: edge %lld has an end point outside [0,%lld)", (long long)e, (long long)n);
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    double *d_xy = nullptr, *d_out = nullptr;
    int32_t *d_u = nullptr, *d_v = nullptr;
    SAFE_TRY(dev_alloc(&d_xy, 2 * n));
    SAFE_TRY(dev_alloc(&d_out, n_edges));
    SAFE_TRY(dev_alloc(&d_u, n_edges));
    SAFE_TRY(dev_alloc(&d_v, n_edges));
    SAFE_HIP_CHECK(hipMemcpyAsync(d_xy, xy_host, 2 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemcpyAsync(d_u, edge_u, n_edges * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemcpyAsync(d_v, edge_v, n_edges * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_edge_lengths, dim3(ceil_div(n_edges, 256)), dim3(256), 0, ctx->stream, d_xy, n_edges, d_u,
                       d_v, d_out);
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(hipMemcpyAsync(out_host, d_out, n_edges * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    (void)hipFree(d_xy);
    (void)hipFree(d_out);
    (void)hipFree(d_u);
    (void)hipFree(d_v);
    return SAFE_OK;
}

int safe_nbr_shortpath(safe_ctx *ctx, int64_t n, int64_t n_edges, const int32_t *edge_u, const int32_t *edge_v,
                       const double *edge_w, double cutoff, int keep_distances, safe_nbr **out) {
    SAFE_REQUIRE(ctx && out && n >= 1 && n_edges >= 0, "safe_nbr_shortpath: bad argument");
    SAFE_REQUIRE(n_edges == 0 || (edge_u && edge_v), "safe_nbr_shortpath: NULL edge arrays");
    *out = nullptr;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    // undirected CSR adjacency on the host (both directions; self loops once)
    std::vector<int32_t> deg(n + 1, 0);
    for (int64_t e = 0; e < n_edges; ++e) {
        const int32_t u = edge_u[e], v = edge_v[e];
        SAFE_REQUIRE(u >= 0 && u < n && v >= 0 && v < n, "safe_nbr_shortpath: edge %lld has an end point outside [0,%lld)",
                     (long long)e, (long long)n);
        if (edge_w) SAFE_REQUIRE(edge_w[e] >= 0.0, "safe_nbr_shortpath: edge %lld has a negative or NaN weight", (long long)e);
        deg[u + 1]++;
        if (u != v) deg[v + 1]++;
    }
    for (int64_t i = 0; i < n; ++i) deg[i + 1] += deg[i];
    const int64_t n_adj = deg[n];
    std::vector<int32_t> adj_col(std::max<int64_t>(n_adj, 1));
    std::vector<double> adj_w(std::max<int64_t>(n_adj, 1));
    std::vector<int32_t> fill(deg.begin(), deg.end() - 1);
    for (int64_t e = 0; e < n_edges; ++e) {
        const int32_t u = edge_u[e], v = edge_v[e];
        const double w = edge_w ? edge_w[e] : 1.0;
        adj_col[fill[u]] = v;
        adj_w[fill[u]++] = w;
        if (u != v) {
            adj_col[fill[v]] = u;
            adj_w[fill[v]++] = w;
        }
    }
    safe_nbr *nbr = nullptr;
    SAFE_TRY(nbr_new(ctx, n, &nbr));
    const int64_t n_workers = std::min<int64_t>(n, static_cast<int64_t>(ctx->num_cu) * 8);
    int32_t *d_ptr = nullptr, *d_col = nullptr;
    double *d_w = nullptr;
    SpScratch sc{};
    int rc = SAFE_OK;
    do {
        if ((rc = dev_alloc(&d_ptr, n + 1)) != SAFE_OK) break;
        if ((rc = dev_alloc(&d_col, n_adj)) != SAFE_OK) break;
        if (edge_w && (rc = dev_alloc(&d_w, n_adj)) != SAFE_OK) break;
        if ((rc = dev_alloc(&sc.dist, n_workers * n)) != SAFE_OK) break;
        if ((rc = dev_alloc(&sc.qa, n_workers * n)) != SAFE_OK) break;
        if ((rc = dev_alloc(&sc.qb, n_workers * n)) != SAFE_OK) break;
        if ((rc = dev_alloc(&sc.reached, n_workers * n)) != SAFE_OK) break;
        if ((rc = dev_alloc(&sc.flag, n_workers * n)) != SAFE_OK) break;
        if (keep_distances && (rc = dev_alloc(&nbr->dist, n * n)) != SAFE_OK) break;
        hipError_t e = hipMemcpyAsync(d_ptr, deg.data(), (n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && n_adj)
            e = hipMemcpyAsync(d_col, adj_col.data(), n_adj * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && d_w && n_adj)
            e = hipMemcpyAsync(d_w, adj_w.data(), n_adj * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(nbr->bits, 0, n * nbr->words * sizeof(uint64_t), ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(sc.flag, 0, n_workers * n * sizeof(int32_t), ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_fill_u64, dim3(1024), dim3(256), 0, ctx->stream, sc.dist, 0x7FF0000000000000ull,
                               n_workers * n);
            if (nbr->dist)
                hipLaunchKernelGGL(k_fill_u64, dim3(2048), dim3(256), 0, ctx->stream,
                                   reinterpret_cast<unsigned long long *>(nbr->dist), 0x7FF0000000000000ull, n * n);
            hipLaunchKernelGGL(k_shortpath, dim3(n_workers), dim3(64), 0, ctx->stream, n, d_ptr, d_col, d_w, cutoff, sc,
                               n_workers, nbr->bits, nbr->words, nbr->dist);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = safe_stream_sync(ctx->stream);
        if (e != hipSuccess) {
            safe_set_error("safe_nbr_shortpath: %s", hipGetErrorString(e));
            rc = SAFE_E_HIP;
            break;
        }
        rc = nbr_finalize_from_bits(nbr);
    } while (0);
    (void)hipFree(d_ptr);
    (void)hipFree(d_col);
    (void)hipFree(d_w);
    (void)hipFree(sc.dist);
    (void)hipFree(sc.qa);
    (void)hipFree(sc.qb);
    (void)hipFree(sc.reached);
    (void)hipFree(sc.flag);
    if (rc != SAFE_OK) {
        nbr_free(nbr);
        return rc;
    }
    *out = nbr;
    return SAFE_OK;
}

int safe_nbr_from_dense_i64(safe_ctx *ctx, const int64_t *a_host, int64_t n, safe_nbr **out) {
    SAFE_REQUIRE(ctx && a_host && out, "safe_nbr_from_dense_i64: NULL argument");
    *out = nullptr;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    safe_nbr *nbr = nullptr;
    SAFE_TRY(nbr_new(ctx, n, &nbr));
    int64_t *d_a = nullptr;
    int *d_bad = nullptr;
    int rc = dev_alloc(&d_a, n * n);
    if (rc == SAFE_OK) rc = dev_alloc(&d_bad, 1);
    int bad = 0;
    if (rc == SAFE_OK) {
        hipError_t e = hipMemcpyAsync(d_a, a_host, n * n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_dense_to_bits, dim3(ceil_div(n * nbr->words * 64, 256)), dim3(256), 0, ctx->stream, d_a,
                               n, nbr->words, nbr->bits, d_bad);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = safe_stream_sync(ctx->stream);
        if (e != hipSuccess) {
            safe_set_error("safe_nbr_from_dense_i64: %s", hipGetErrorString(e));
            rc = SAFE_E_HIP;
        }
    }
    (void)hipFree(d_a);
    (void)hipFree(d_bad);
    if (rc == SAFE_OK && bad) {
        safe_set_error("safe_nbr_from_dense_i64: membership matrix has entries outside {0,1}");
        rc = SAFE_E_VALUE;
    }
    if (rc == SAFE_OK) rc = nbr_finalize_from_bits(nbr);
    if (rc != SAFE_OK) {
        nbr_free(nbr);
        return rc;
    }
    *out = nbr;
    return SAFE_OK;
}

int safe_nbr_destroy(safe_nbr *nbr) {
    if (!nbr) return SAFE_OK;
    (void)hipSetDevice(nbr->ctx->device);
    (void)safe_stream_sync(nbr->ctx->stream);
    nbr_free(nbr);
    return SAFE_OK;
}

int safe_nbr_info(const safe_nbr *nbr, int64_t *n, int64_t *nnz, int64_t *max_row_count) {
    SAFE_REQUIRE(nbr != nullptr, "safe_nbr_info: nbr is NULL");
    if (n) *n = nbr->n;
    if (nnz) *nnz = nbr->nnz;
    if (max_row_count) *max_row_count = nbr->max_count;
    return SAFE_OK;
}

int safe_nbr_to_dense_i64_dev(safe_nbr *nbr, int64_t *out_dev) {
    SAFE_REQUIRE(nbr && out_dev, "safe_nbr_to_dense_i64_dev: NULL argument");
    safe_ctx *ctx = nbr->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_bits_to_dense, dim3(ceil_div(nbr->n, 512), nbr->n), dim3(256), 0, ctx->stream, nbr->bits,
                       nbr->n, nbr->words, out_dev);
    SAFE_HIP_CHECK(hipGetLastError());
    return SAFE_OK;
}

int safe_nbr_to_dense_i64(safe_nbr *nbr, int64_t *out_host) {
    SAFE_REQUIRE(nbr && out_host, "safe_nbr_to_dense_i64: NULL argument");
    safe_ctx *ctx = nbr->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    int64_t *d = nullptr;
    SAFE_TRY(dev_alloc(&d, nbr->n * nbr->n));
    int rc = safe_nbr_to_dense_i64_dev(nbr, d);
    if (rc == SAFE_OK) {
        hipError_t e = hipMemcpyAsync(out_host, d, nbr->n * nbr->n * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = safe_stream_sync(ctx->stream);
        if (e != hipSuccess) {
            safe_set_error("safe_nbr_to_dense_i64: %s", hipGetErrorString(e));
            rc = SAFE_E_HIP;
        }
    }
    (void)hipFree(d);
    return rc;
}

int safe_nbr_row_counts(safe_nbr *nbr, int64_t *out_host) {
    SAFE_REQUIRE(nbr && out_host, "safe_nbr_row_counts: NULL argument");
    for (int64_t i = 0; i < nbr->n; ++i) out_host[i] = nbr->h_row_count[i];
    return SAFE_OK;
}

int safe_nbr_csr(safe_nbr *nbr, int32_t *row_ptr_host, int32_t *col_host) {
    SAFE_REQUIRE(nbr && row_ptr_host && col_host, "safe_nbr_csr: NULL argument");
    safe_ctx *ctx = nbr->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    SAFE_HIP_CHECK(hipMemcpyAsync(row_ptr_host, nbr->row_ptr, (nbr->n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost,
                                  ctx->stream));
    if (nbr->nnz)
        SAFE_HIP_CHECK(hipMemcpyAsync(col_host, nbr->col, nbr->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    return SAFE_OK;
}

int safe_nbr_distances(safe_nbr *nbr, double *out_host) {
    SAFE_REQUIRE(nbr && out_host, "safe_nbr_distances: NULL argument");
    SAFE_REQUIRE(nbr->dist != nullptr, "safe_nbr_distances: handle was built without keep_distances");
    safe_ctx *ctx = nbr->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    SAFE_HIP_CHECK(hipMemcpyAsync(out_host, nbr->dist, nbr->n * nbr->n * sizeof(double), hipMemcpyDeviceToHost,
                                  ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    return SAFE_OK;
}

}  // extern "C"
