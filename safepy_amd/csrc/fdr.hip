// multiple_testing=True: Benjamini-Hochberg adjustment of every ROW of the p-value matrices
// (across attributes), i.e. np.apply_along_axis(fdrcorrection, 1, pvalues)[:, 1, :] of
// safepy/safe.py:536-542 (randomization) and 599-605 (hypergeometric), followed by the NES /
// binarisation steps that depend on the adjusted values (safe.py:546-554, 608, 468-472).
//
// statsmodels' fdrcorrection(pvals) (method 'indep'), operation by operation:
//     order = argsort(pvals); ps = pvals[order]
//     raw = ps / (arange(1, n+1) / float(n))
//     corrected = minimum.accumulate(raw[::-1])[::-1];  corrected[corrected > 1] = 1
//     out[order] = corrected
// Ties need no care: tied entries end up with the same value whatever their order (the running
// minimum from the right passes through the group's last member).  NaN p-values sort last and
// np.minimum propagates them through the whole row, so a row with any NaN becomes all NaN.
//
// Randomization form (p = counts / num_permutations): no sort, the row's histogram over the counts (k_fdr_row_counts).
// Rows of up to 8192 attributes are adjusted by ONE kernel, one workgroup per row: block-wide radix sort of the
// row's p-values with their column ids, then the Benjamini-Hochberg pass on the registers that hold the
// sorted row (k_fdr_row_sort, instead of the library's segmented radix sort of the whole matrix + its temporaries).  Longer rows: hipCUB segmented radix sort,
// rows batched so a call stays below 2^31 keys, then the same BH pass (k_fdr_row).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"

namespace {

__global__ void k_fdr_cols(int32_t *__restrict__ idx, int64_t count, int64_t m) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < count) idx[i] = static_cast<int32_t>(i % m);
}

__global__ void k_fdr_offsets(int64_t *__restrict__ off, int64_t rows, int64_t m) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i <= rows) off[i] = i * m;
}

// one workgroup per row of the batch; ps / cols: the row sorted ascending with its column ids
__global__ __launch_bounds__(256) void k_fdr_row(const double *__restrict__ ps, const int32_t *__restrict__ cols, int64_t m,
                                                 double *__restrict__ out) {
    __shared__ double part[256];
    const int64_t row = blockIdx.x;
    const double *p = ps + row * m;
    const int32_t *c = cols + row * m;
    double *o = out + row * m;
    const int t = threadIdx.x;
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    const double last = p[m - 1];
    if (last != last) {                                             // a NaN anywhere poisons the whole row
        for (int64_t r = t; r < m; r += 256) o[r] = qnan;
        return;
    }
    const double n_d = static_cast<double>(m);
    const int64_t len = (m + 255) / 256, r0 = static_cast<int64_t>(t) * len, r1 = r0 + len < m ? r0 + len : m;
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    double mine = inf;
    for (int64_t r = r1 - 1; r >= r0; --r) mine = fmin(mine, p[r] / (static_cast<double>(r + 1) / n_d));
    part[t] = mine;
    __syncthreads();
    double carry = inf;                                             // minimum over everything right of this thread's chunk
    for (int u = t + 1; u < 256; ++u) carry = fmin(carry, part[u]);
    for (int64_t r = r1 - 1; r >= r0; --r) {
        carry = fmin(carry, p[r] / (static_cast<double>(r + 1) / n_d));
        o[c[r]] = carry > 1.0 ? 1.0 : carry;
    }
}

// One workgroup = one row: a block-wide radix sort (hipcub::BlockRadixSort: keys and column ids in registers, LDS
// for the exchanges) leaves thread t with ranks [t IPT, (t + 1) IPT) of the sorted row -- exactly the chunks the
// Benjamini-Hochberg pass wants.  Keys are the bit patterns of the p-values: non-negative doubles order like their
// bits and NaN (0x7FF8...) sorts behind every number, as np.argsort puts it; padding keys are all ones.
// (A bitonic network on an LDS copy of the row was tried first: 91 stages x 20 bytes per element = 15 MB of LDS
// traffic per row, 3.2 ms per 3971 x 4373 matrix.)
#ifndef FDR_RADIX_BITS
#define FDR_RADIX_BITS 8
#endif
template <int IPT>
__global__ __launch_bounds__(256) void k_fdr_row_sort(double *__restrict__ pvals, int64_t m) {
    using BlockSort = hipcub::BlockRadixSort<unsigned long long, 256, IPT, unsigned short, FDR_RADIX_BITS>;
    __shared__ typename BlockSort::TempStorage temp;
    __shared__ double part[256];
    __shared__ int has_nan;
    const int t = threadIdx.x;
    double *p = pvals + static_cast<int64_t>(blockIdx.x) * m;
    unsigned long long key[IPT];
    unsigned short idx[IPT];
    if (t == 0) has_nan = 0;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int e = t * IPT + i;
        key[i] = e < m ? static_cast<unsigned long long>(__double_as_longlong(p[e])) : ~0ull;
        idx[i] = static_cast<unsigned short>(e);
    }
    __syncthreads();
    BlockSort(temp).Sort(key, idx, 0, 64);
    // the same arithmetic, in the same order, as k_fdr_row; rank of key[i] = t * IPT + i
    const double n_d = static_cast<double>(m);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    const int r0 = t * IPT;
    double mine = inf;
    bool nan_here = false;
#pragma unroll
    for (int i = IPT - 1; i >= 0; --i) {
        if (r0 + i >= m) continue;
        const double v = __longlong_as_double(static_cast<long long>(key[i]));
        nan_here |= v != v;
        mine = fmin(mine, v / (static_cast<double>(r0 + i + 1) / n_d));
    }
    part[t] = mine;
    if (nan_here) has_nan = 1;
    __syncthreads();
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    if (has_nan) {                                                  // a NaN anywhere poisons the whole row
        for (int64_t r = t; r < m; r += 256) p[r] = qnan;
        return;
    }
    double carry = inf;                                             // minimum over everything right of this thread's chunk
    for (int u = t + 1; u < 256; ++u) carry = fmin(carry, part[u]);
#pragma unroll
    for (int i = IPT - 1; i >= 0; --i) {
        if (r0 + i >= m) continue;
        const double v = __longlong_as_double(static_cast<long long>(key[i]));
        carry = fmin(carry, v / (static_cast<double>(r0 + i + 1) / n_d));
        p[idx[i]] = carry > 1.0 ? 1.0 : carry;
    }
}

// The same adjustment WITHOUT a sort, for p-values that are counts / num_permutations (the randomization form: every entry is one of
// P + 1 values).  All members of a tie end up with the value of the tie's LAST sorted position (header above), so a row needs only
// its histogram over the counts: cum[c] = #entries with count <= c is that last position, raw[c] = (c / P) / (cum[c] / n) the
// value statsmodels computes there -- the same two divisions on the same doubles -- and adj[c] = min over the occupied c' >= c.
// One workgroup per row: LDS histogram, block scan, reverse running minimum, one table look-up per entry.  0.9 ms -> HBM speed
// per matrix at 3971 x 4373.  An entry that is not exactly c / P raises `flag` (the caller then sorts instead).
__global__ __launch_bounds__(256) void k_fdr_row_counts(double *__restrict__ pvals, int64_t m, int64_t n_perm, unsigned int *__restrict__ flag) {
    extern __shared__ unsigned char fdr_lds[];
    const int bins = static_cast<int>(n_perm) + 1;
    double *adj = reinterpret_cast<double *>(fdr_lds);                    // [bins]
    unsigned int *hist = reinterpret_cast<unsigned int *>(adj + bins);   // [bins]
    __shared__ unsigned int part_u[256];
    __shared__ double part_d[256];
    __shared__ int bad;
    const int t = threadIdx.x;
    double *p = pvals + static_cast<int64_t>(blockIdx.x) * m;
    const double P = static_cast<double>(n_perm), n_d = static_cast<double>(m);
    for (int c = t; c < bins; c += 256) hist[c] = 0u;
    if (t == 0) bad = 0;
    __syncthreads();
    int mine_bad = 0;                                                     // 1: a NaN (poisons the row), 2: not a count ratio
    for (int64_t e = t; e < m; e += 256) {
        const double v = p[e];
        if (v != v) {
            mine_bad |= 1;
            continue;
        }
        const double cf = rint(v * P);
        if (!(cf >= 0.0 && cf <= P) || cf / P != v) {
            mine_bad |= 2;
            continue;
        }
        atomicAdd(&hist[static_cast<int>(cf)], 1u);
    }
    if (mine_bad) atomicOr(&bad, mine_bad);
    __syncthreads();
    if (bad & 2) {
        if (t == 0) atomicOr(flag, 1u);
        return;
    }
    if (bad & 1) {                                                        // np.minimum propagates the NaN through the whole row
        const double qnan = __longlong_as_double(0x7FF8000000000000ll);
        for (int64_t e = t; e < m; e += 256) p[e] = qnan;
        return;
    }
    // cum over ascending counts: thread t owns the bins [b0, b1)
    const int per = (bins + 255) / 256, b0 = t * per < bins ? t * per : bins, b1 = b0 + per < bins ? b0 + per : bins;
    unsigned int local = 0;
    for (int c = b0; c < b1; ++c) local += hist[c];
    part_u[t] = local;
    __syncthreads();
    unsigned int before = 0;
    for (int u = 0; u < t; ++u) before += part_u[u];
    // raw values of the occupied bins (unoccupied: +inf), then the running minimum from the right
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    double mn = inf;
    {
        unsigned int cum = before;
        for (int c = b0; c < b1; ++c) {
            const unsigned int h = hist[c];
            cum += h;
            adj[c] = h ? (static_cast<double>(c) / P) / (static_cast<double>(cum) / n_d) : inf;
        }
        for (int c = b1 - 1; c >= b0; --c) mn = fmin(mn, adj[c]);
    }
    part_d[t] = mn;
    __syncthreads();
    double carry = inf;
    for (int u = t + 1; u < 256; ++u) carry = fmin(carry, part_d[u]);
    for (int c = b1 - 1; c >= b0; --c) {
        carry = fmin(carry, adj[c]);
        adj[c] = carry > 1.0 ? 1.0 : carry;
    }
    __syncthreads();
    for (int64_t e = t; e < m; e += 256) p[e] = adj[static_cast<int>(rint(p[e] * P))];
}

// read-only: is every non-NaN entry a count ratio c / P, c in [0, P]?  (what the histogram form needs; anything else is sorted)
__global__ __launch_bounds__(256) void k_fdr_check_ratio(const double *__restrict__ p, int64_t total, double P, unsigned int *__restrict__ flag) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    bool bad = false;
    if (i < total) {
        const double v = p[i];
        if (v == v) {
            const double cf = rint(v * P);
            bad = !(cf >= 0.0 && cf <= P) || cf / P != v;
        }
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// NES / binarisation from (adjusted) p-values; n_perm == 0: hypergeometric form nes = -log10(p_pos)
__global__ __launch_bounds__(256) void k_nes_from_pvalues(const double *__restrict__ p_neg, const double *__restrict__ p_pos,
                                                          int64_t n, int64_t m, double inv_perm, int sign_mode,
                                                          double nes_threshold, double p_cut, double *__restrict__ nes_out,
                                                          double *__restrict__ nes_binary, unsigned int *__restrict__ enriched) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 64 + (threadIdx.x & 63);
    const int64_t i0 = static_cast<int64_t>(blockIdx.y) * 64 + (threadIdx.x >> 6);
    if (c >= m) return;
    unsigned int hits = 0;
    for (int64_t i = i0; i < n && i < (static_cast<int64_t>(blockIdx.y) + 1) * 64; i += 4) {
        const int64_t o = i * m + c;
        double nes;
        bool hit;
        // where the NES is one logarithm, the binarisation is decided on p itself (nes_p_cut, common.h):
        // the device's log10 may round differently from the host libm the reference calls
        if (inv_perm == 0.0) {
            nes = -log10(p_pos[o]);                                  // safe.py:608
            hit = p_pos[o] < p_cut;                                  // safe.py:468-470
        } else {                                                     // safe.py:546-554
            const double pp = p_pos[o] == 0.0 ? inv_perm : p_pos[o], pn = p_neg[o] == 0.0 ? inv_perm : p_neg[o];
            const double ep = -log10(pp), en = -log10(pn);
            nes = sign_mode == SAFE_SIGN_HIGHEST ? ep : sign_mode == SAFE_SIGN_LOWEST ? en : ep - en;
            hit = sign_mode == SAFE_SIGN_HIGHEST ? pp < p_cut
                  : sign_mode == SAFE_SIGN_LOWEST ? pn < p_cut
                                                  : (nes == nes) && (fabs(nes) > nes_threshold);
        }
        nes_out[o] = nes;
        nes_binary[o] = hit ? 1.0 : 0.0;
        hits += hit;
    }
    if (hits) atomicAdd(&enriched[c], hits);
}

__global__ void k_fdr_u32_to_f64(const unsigned int *__restrict__ in, double *__restrict__ out, int64_t count) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < count) out[i] = static_cast<double>(in[i]);
}

int fdr_matrix(safe_ctx *ctx, double *p_dev, int64_t n, int64_t m, int64_t n_perm = 0) {
    const char *sort_env = getenv("SAFE_HIP_FDR_SORT");                   // "cub": library sort for every row length (tests); "sort": never the histogram form
    // randomization form: p-values are counts / n_perm -- no sort needed (k_fdr_row_counts); table + histogram must fit LDS
    if (n_perm > 0 && (n_perm + 1) * 12 <= 60 * 1024 && !sort_env) {
        unsigned int *d_flag = nullptr;
        SAFE_TRY(ctx_scratch(ctx, 10, sizeof(unsigned int), reinterpret_cast<void **>(&d_flag)));
        SAFE_HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(unsigned int), ctx->stream));
        const size_t lds = static_cast<size_t>(n_perm + 1) * 12;
        SAFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fdr_row_counts), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds)));
        hipLaunchKernelGGL(k_fdr_row_counts, dim3(n), dim3(256), lds, ctx->stream, p_dev, m, n_perm, d_flag);
        SAFE_HIP_CHECK(hipGetLastError());
        void *pinned = nullptr;
        SAFE_TRY(ctx_pinned(ctx, sizeof(unsigned int), &pinned));
        SAFE_HIP_CHECK(hipMemcpyAsync(pinned, d_flag, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
        if (*static_cast<unsigned int *>(pinned) == 0u) return SAFE_OK;
        // (cannot happen: safe_fdr_adjust validated the matrix before it chose this form)
        safe_set_error("safe_fdr_adjust: num_permutations = %lld but a p-value is not a multiple of 1 / num_permutations",
                       (long long)n_perm);
        return SAFE_E_VALUE;
    }
    if (m <= 8192 && !(sort_env && !strcmp(sort_env, "cub"))) {
        const int64_t ipt = ceil_div(m, 256);                                // items per thread: the sort pads the row to 256 * IPT
#define FDR_SORT(I) hipLaunchKernelGGL(k_fdr_row_sort<I>, dim3(n), dim3(256), 0, ctx->stream, p_dev, m)
        if (ipt <= 4) FDR_SORT(4);
        else if (ipt <= 8) FDR_SORT(8);
        else if (ipt <= 12) FDR_SORT(12);
        else if (ipt <= 16) FDR_SORT(16);
        else if (ipt <= 20) FDR_SORT(20);
        else if (ipt <= 24) FDR_SORT(24);
        else if (ipt <= 28) FDR_SORT(28);
        else FDR_SORT(32);
#undef FDR_SORT
        SAFE_HIP_CHECK(hipGetLastError());
        SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
        return SAFE_OK;
    }
    // rows per batch: at most 2^27 keys per library call (temporaries ~ 3 GB)
    const int64_t batch_rows = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t(1) << 27) / std::max<int64_t>(m, 1)));
    const int64_t batch_items = batch_rows * m;
    SAFE_REQUIRE(m < (int64_t(1) << 31) - 1 && batch_items < (int64_t(1) << 31), "safe_fdr_adjust: %lld attributes per row are too many",
                 (long long)m);
    double *keys_out = nullptr;
    int32_t *vals_in = nullptr, *vals_out = nullptr;
    int64_t *offsets = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    int rc = dev_alloc(&keys_out, batch_items);
    if (rc == SAFE_OK) rc = dev_alloc(&vals_in, batch_items);
    if (rc == SAFE_OK) rc = dev_alloc(&vals_out, batch_items);
    if (rc == SAFE_OK) rc = dev_alloc(&offsets, batch_rows + 1);
    hipError_t e = hipSuccess;
    if (rc == SAFE_OK) {
        hipLaunchKernelGGL(k_fdr_cols, dim3(ceil_div(batch_items, 256)), dim3(256), 0, ctx->stream, vals_in, batch_items, m);
        hipLaunchKernelGGL(k_fdr_offsets, dim3(ceil_div(batch_rows + 1, 256)), dim3(256), 0, ctx->stream, offsets, batch_rows, m);
        e = hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, temp_bytes, p_dev, keys_out, vals_in, vals_out,
                                                        static_cast<int>(batch_items), static_cast<int>(batch_rows), offsets,
                                                        offsets + 1, 0, 64, ctx->stream);
        g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
        if (e == hipSuccess) e = hipMalloc(&temp, std::max<size_t>(temp_bytes, 16));
    }
    for (int64_t r0 = 0; rc == SAFE_OK && e == hipSuccess && r0 < n; r0 += batch_rows) {
        const int64_t rows = std::min<int64_t>(batch_rows, n - r0);
        double *block = p_dev + r0 * m;
        e = hipcub::DeviceSegmentedRadixSort::SortPairs(temp, temp_bytes, block, keys_out, vals_in, vals_out,
                                                        static_cast<int>(rows * m), static_cast<int>(rows), offsets, offsets + 1, 0,
                                                        64, ctx->stream);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_fdr_row, dim3(rows), dim3(256), 0, ctx->stream, keys_out, vals_out, m, block);
        e = hipGetLastError();
    }
    if (rc == SAFE_OK && e == hipSuccess) e = safe_stream_sync(ctx->stream);
    if (e != hipSuccess) {
        safe_set_error("safe_fdr_adjust: %s", hipGetErrorString(e));
        rc = SAFE_E_HIP;
    }
    (void)hipFree(keys_out);
    (void)hipFree(vals_in);
    (void)hipFree(vals_out);
    (void)hipFree(offsets);
    (void)hipFree(temp);
    return rc;
}

}  // namespace

extern "C" int safe_fdr_adjust(safe_ctx *ctx, int64_t n, int64_t m, int64_t num_permutations, int sign_mode,
                               double enrichment_threshold, double *pvalues_neg_dev, double *pvalues_pos_dev, double *nes_dev,
                               double *nes_binary_dev, double *num_enriched_dev) {
    SAFE_REQUIRE(ctx && pvalues_pos_dev && nes_dev && nes_binary_dev && num_enriched_dev, "safe_fdr_adjust: NULL argument");
    SAFE_REQUIRE(n >= 1 && m >= 1 && num_permutations >= 0, "safe_fdr_adjust: bad sizes");
    SAFE_REQUIRE(num_permutations == 0 || pvalues_neg_dev, "safe_fdr_adjust: the randomization form needs pvalues_neg");
    SAFE_REQUIRE(sign_mode >= SAFE_SIGN_HIGHEST && sign_mode <= SAFE_SIGN_BOTH, "safe_fdr_adjust: bad sign_mode %d", sign_mode);
    SAFE_REQUIRE(enrichment_threshold > 0.0 && enrichment_threshold < 1.0, "safe_fdr_adjust: enrichment_threshold must be in (0,1)");
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    // The library's own empirical p-values are count ratios c / num_permutations and need no sort (k_fdr_row_counts).  A caller may
    // hand over other values with num_permutations > 0 (only the 0 -> 1 / P rule of the NES depends on it): both matrices are
    // checked READ-ONLY first, and anything that is not a count ratio goes through the sort like the hypergeometric form.
    int64_t hist_perm = 0;
    if (num_permutations > 0 && (num_permutations + 1) * 12 <= 60 * 1024 && !getenv("SAFE_HIP_FDR_SORT")) {
        unsigned int *d_flag = nullptr;
        SAFE_TRY(ctx_scratch(ctx, 10, sizeof(unsigned int), reinterpret_cast<void **>(&d_flag)));
        SAFE_HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(unsigned int), ctx->stream));
        const int64_t total = n * m;
        for (const double *mat : {static_cast<const double *>(pvalues_neg_dev), static_cast<const double *>(pvalues_pos_dev)})
            hipLaunchKernelGGL(k_fdr_check_ratio, dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, mat, total,
                               static_cast<double>(num_permutations), d_flag);
        SAFE_HIP_CHECK(hipGetLastError());
        void *pinned = nullptr;
        SAFE_TRY(ctx_pinned(ctx, sizeof(unsigned int), &pinned));
        SAFE_HIP_CHECK(hipMemcpyAsync(pinned, d_flag, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
        if (*static_cast<unsigned int *>(pinned) == 0u) hist_perm = num_permutations;
    }
    if (num_permutations > 0) SAFE_TRY(fdr_matrix(ctx, pvalues_neg_dev, n, m, hist_perm));
    SAFE_TRY(fdr_matrix(ctx, pvalues_pos_dev, n, m, hist_perm));
    unsigned int *d_enr = nullptr;
    SAFE_TRY(dev_alloc(&d_enr, m));
    SAFE_HIP_CHECK(hipMemsetAsync(d_enr, 0, m * sizeof(unsigned int), ctx->stream));
    hipLaunchKernelGGL(k_nes_from_pvalues, dim3(ceil_div(m, 64), ceil_div(n, 64)), dim3(256), 0, ctx->stream, pvalues_neg_dev,
                       pvalues_pos_dev, n, m, num_permutations > 0 ? 1.0 / static_cast<double>(num_permutations) : 0.0, sign_mode,
                       -std::log10(enrichment_threshold), nes_p_cut(enrichment_threshold), nes_dev, nes_binary_dev, d_enr);
    hipLaunchKernelGGL(k_fdr_u32_to_f64, dim3(ceil_div(m, 256)), dim3(256), 0, ctx->stream, d_enr, num_enriched_dev, m);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = safe_stream_sync(ctx->stream);
    (void)hipFree(d_enr);
    if (e != hipSuccess) {
        safe_set_error("safe_fdr_adjust: %s", hipGetErrorString(e));
        return SAFE_E_HIP;
    }
    return SAFE_OK;
}
