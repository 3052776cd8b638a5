// Attribute matrix handle: upload / borrow self.node2attribute and derive the
// whole-matrix facts compute_pvalues needs before it dispatches
// (safepy/safe.py:453-463, 574-583; safepy/safe_extras.py:51).
#include "common.h"

template <typename T>
__device__ __forceinline__ double load_attr(const void *raw, int64_t idx) {
    return static_cast<double>(reinterpret_cast<const T *>(raw)[idx]);
}

// Whole-matrix facts in one pass: per column NaN count, nansum, #values outside {0,1},
// #non-integers, max |v|; per row "has a value" bits.  One block per group of GC columns;
// the thread grid is laid out along the contiguous axis of the matrix so loads coalesce:
//   Fortran order (rs == 1): GC = 1,  256 threads walk the rows of one column
//   C order       (cs == 1): GC = 64, 64 x 4 threads: x = column, y = row lane
// Row bits are collected per block in an LDS bitmap (wave ballots, no atomics on the hot
// path) and OR-ed into the global bitmap once per block, skipping words already complete.
template <typename T, int GC>
__global__ __launch_bounds__(256) void k_attr_stats(const void *__restrict__ raw, int64_t n, int64_t m,
                                                    int64_t rs, int64_t cs, unsigned int *__restrict__ row_bits,
                                                    unsigned long long *__restrict__ acc /*[4 + 64 + 64]*/,
                                                    double *__restrict__ col_sum, unsigned long long *__restrict__ max_abs_bits,
                                                    unsigned int *__restrict__ col_nan, int64_t rows_per_block) {
    extern __shared__ unsigned int s_bits[];           // [ceil(n/32)]
    constexpr int RL = 256 / GC;                       // row lanes per column
    const int cx = GC == 1 ? 0 : (threadIdx.x & (GC - 1));
    const int ry = GC == 1 ? threadIdx.x : (threadIdx.x / GC);
    const int lane = threadIdx.x & 63;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * GC + cx;
    const int64_t n_words = (n + 31) / 32;
    __shared__ double s_sum[256];
    __shared__ double s_max[256];
    __shared__ unsigned int s_nan[256], s_other[256], s_nonint[256];
    // blockIdx.y: a chunk of rows_per_block rows (a multiple of 64), so that a matrix with few column
    // groups still fills the chip; per-column sums / NaN counts of the chunks meet in global atomics
    const int64_t r_begin = static_cast<int64_t>(blockIdx.y) * rows_per_block;
    const int64_t r_end = r_begin + rows_per_block < n ? r_begin + rows_per_block : n;
    const int64_t w_begin = r_begin >> 5, w_end = (r_end + 31) >> 5;
    for (int64_t w = w_begin + threadIdx.x; w < w_end; w += 256) s_bits[w] = 0;
    __syncthreads();
    unsigned int c_nan = 0, c_other = 0, c_nonint = 0;
    double mx = 0.0, sum = 0.0;
    // four rows per thread and trip, all four loads issued before the first is used (a trip is one memory latency, not four);
    // every thread runs the same trip count (ballots)
    constexpr int UN = 4;
    const int64_t n_round = r_begin + (r_end - r_begin + UN * RL - 1) / (UN * RL) * (UN * RL);
    if (GC == 1 && rs == 1) {
        // a contiguous column (Fortran order): FOUR consecutive rows per load (16 / 32 bytes; the column's start is only element-
        // aligned, hence the packed type), four loads in flight -- 4096 rows per trip: the configs[1] matrix (3971 rows) is one
        // trip of one memory latency instead of four (52 -> 2x us per pass of 69 MB).  Row bits: a nibble per load, OR-ed into the
        // workgroup's LDS bitmap (r_begin is a multiple of 64, so a nibble never straddles a word).
        struct __attribute__((packed, aligned(sizeof(T)))) Vec { T v[4]; };
        const T *colp = static_cast<const T *>(raw) + j * cs;
        for (int64_t i0 = r_begin + 4 * static_cast<int64_t>(threadIdx.x); i0 < r_end; i0 += UN * 4 * 256) {
            Vec x[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t i = i0 + static_cast<int64_t>(u) * 4 * 256;
                if (i + 4 <= r_end) {
                    x[u] = *reinterpret_cast<const Vec *>(colp + i);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[u].v[e] = i + e < r_end ? colp[i + e] : static_cast<T>(0);
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t i = i0 + static_cast<int64_t>(u) * 4 * 256;
                unsigned int nib = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (i + e >= r_end) continue;
                    const double v = static_cast<double>(x[u].v[e]);
                    if (v != v) {
                        ++c_nan;
                    } else {
                        nib |= 1u << e;
                        sum += v;
                        if (v != 0.0 && v != 1.0) ++c_other;
                        if (v != floor(v)) ++c_nonint;
                        const double a = fabs(v);
                        if (a > mx) mx = a;
                    }
                }
                if (nib) atomicOr(&s_bits[i >> 5], nib << (i & 31));
            }
        }
    } else
    for (int64_t i0 = r_begin + ry; i0 < n_round; i0 += UN * RL) {
        double v[UN];
        bool inb[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t i = i0 + u * RL;
            inb[u] = i < r_end && j < m;
            v[u] = inb[u] ? load_attr<T>(raw, i * rs + j * cs) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t i = i0 + u * RL;
            bool has = false;
            if (inb[u]) {
                if (v[u] != v[u]) {
                    ++c_nan;
                } else {
                    has = true;
                    sum += v[u];
                    if (v[u] != 0.0 && v[u] != 1.0) ++c_other;
                    if (v[u] != floor(v[u])) ++c_nonint;
                    const double a = fabs(v[u]);
                    if (a > mx) mx = a;
                }
            }
            const unsigned long long bal = __ballot(has);
            if (GC == 1) {
                // the wave holds 64 consecutive rows starting at i - lane: two bitmap words, owned by this wave
                const int64_t r0 = i - lane;
                if (lane == 0 && bal) {
                    if (r0 < r_end) s_bits[r0 >> 5] |= static_cast<unsigned int>(bal);
                    if (r0 + 32 < r_end) s_bits[(r0 >> 5) + 1] |= static_cast<unsigned int>(bal >> 32);
                }
            } else {
                // the wave holds ONE row (i) across 64 columns
                if (lane == 0 && bal && i < r_end) atomicOr(&s_bits[i >> 5], 1u << (i & 31));
            }
        }
    }
    if (GC == 1) {
        // one column per workgroup: the 256 partial results meet by wave shuffles, then four per column in LDS
        // (a single thread adding 256 x 5 LDS values took longer than the loads)
        for (int off = 32; off; off >>= 1) {
            sum += __shfl_down(sum, off);
            const double o = __shfl_down(mx, off);
            mx = o > mx ? o : mx;
            c_nan += __shfl_down(c_nan, off);
            c_other += __shfl_down(c_other, off);
            c_nonint += __shfl_down(c_nonint, off);
        }
    }
    const int slot = GC == 1 ? (lane == 0 ? static_cast<int>(threadIdx.x >> 6) : -1) : static_cast<int>(threadIdx.x);
    if (slot >= 0) {
        s_sum[slot] = sum;
        s_max[slot] = mx;
        s_nan[slot] = c_nan;
        s_other[slot] = c_other;
        s_nonint[slot] = c_nonint;
    }
    __syncthreads();
    for (int64_t w = w_begin + threadIdx.x; w < w_end; w += 256) {
        const unsigned int mine = s_bits[w];
        if (mine & ~row_bits[w]) atomicOr(&row_bits[w], mine);       // racy pre-check is benign
    }
    if (ry == 0 && j < m) {
        double total = 0.0, tmx = 0.0;
        unsigned long long t_nan = 0, t_other = 0, t_nonint = 0;
        for (int r = 0; r < (GC == 1 ? 4 : RL); ++r) {
            const int t = GC == 1 ? r : r * GC + cx;
            total += s_sum[t];
            if (s_max[t] > tmx) tmx = s_max[t];
            t_nan += s_nan[t];
            t_other += s_other[t];
            t_nonint += s_nonint[t];
        }
        if (gridDim.y == 1) col_sum[j] = total;
        else atomicAdd(&col_sum[j], total);               // integer-valued data (the consumers' case) adds exactly in any order
        if (t_nan) atomicAdd(&col_nan[j], static_cast<unsigned int>(t_nan));
        // sums over all columns: 64 slots each instead of one address for thousands of workgroups (k_fold_parts adds them up)
        if (t_other) atomicAdd(&acc[4 + (j & 63)], t_other);
        if (t_nonint) atomicAdd(&acc[68 + (j & 63)], t_nonint);
        // thousands of workgroups hit this one address: look first (a stale look only costs a redundant atomic)
        const unsigned long long mine = static_cast<unsigned long long>(__double_as_longlong(tmx));
        if (mine > __builtin_nontemporal_load(max_abs_bits)) atomicMax(max_abs_bits, mine);
    }
}

// One workgroup finishes the pass: acc[0] = sum of acc[4 .. 68) (#values outside {0,1}), acc[2] = sum of acc[68 .. 132)
// (#non-integers), acc[1] = largest NaN count of a column, row bitmap -> one byte per row; the four words are copied behind
// the flags (tail) so that ONE device-to-host copy brings everything the host needs.
__global__ __launch_bounds__(256) void k_stats_finish(unsigned long long *__restrict__ acc, const unsigned int *__restrict__ col_nan, int64_t m,
                                                      const unsigned int *__restrict__ bits, int64_t n, uint8_t *__restrict__ bytes,
                                                      unsigned long long *__restrict__ tail) {
    __shared__ unsigned int s_mx[4];
    __shared__ unsigned long long s_fold[2];
    unsigned int mx = 0;
    for (int64_t i = threadIdx.x; i < m; i += 256) mx = col_nan[i] > mx ? col_nan[i] : mx;
    for (int off = 32; off; off >>= 1) {
        const unsigned int o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
    if (threadIdx.x < 64) {
        unsigned long long a = acc[4 + threadIdx.x], b = acc[68 + threadIdx.x];
        for (int off = 32; off; off >>= 1) {
            a += __shfl_down(a, off);
            b += __shfl_down(b, off);
        }
        if (threadIdx.x == 0) s_fold[0] = a, s_fold[1] = b;
    }
    for (int64_t i = threadIdx.x; i < n; i += 256) bytes[i] = (bits[i >> 5] >> (i & 31)) & 1u;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int m01 = s_mx[0] > s_mx[1] ? s_mx[0] : s_mx[1], m23 = s_mx[2] > s_mx[3] ? s_mx[2] : s_mx[3];
        tail[0] = acc[0] = s_fold[0];
        tail[1] = acc[1] = m01 > m23 ? m01 : m23;
        tail[2] = acc[2] = s_fold[1];
        tail[3] = acc[3];
    }
}

// one wave per column: rows holding a 1, ascending, by ballot compaction
template <typename T>
__global__ __launch_bounds__(256) void k_fill_support(const void *__restrict__ raw, int64_t n, int64_t m, int64_t rs,
                                                      int64_t cs, const int32_t *__restrict__ sup_ptr,
                                                      int32_t *__restrict__ sup_row) {
    const int64_t j = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (j >= m) return;
    int32_t pos = sup_ptr[j];
    for (int64_t i0 = 0; i0 < n; i0 += 64) {
        const int64_t i = i0 + lane;
        const bool one = i < n && reinterpret_cast<const T *>(raw)[i * rs + j * cs] == static_cast<T>(1);
        const unsigned long long bal = __ballot(one);
        if (one) sup_row[pos + __popcll(bal & ((1ull << lane) - 1ull))] = static_cast<int32_t>(i);
        pos += __popcll(bal);
    }
}


// ---------------------------------------------------------------------------------------------
// read_attributes on the device (safepy/safe_io.py:386-390, 405-410, 426-429)
// ---------------------------------------------------------------------------------------------
// node2attribute.reindex(index=node_label_order, fill_value=...) followed by .values: output row i
// is table row row_map[i] (-1: not in the file -> fill_value, -2: masked duplicate -> NaN).  64 x 64
// tiles turned around through LDS so that the read runs along the table's contiguous axis and the
// write along the output's, whatever the two orders are.
template <typename T>
__global__ __launch_bounds__(256) void k_reindex_rows(const T *__restrict__ table, int64_t n_labels, int64_t m,
                                                      int64_t irs, int64_t ics, const int64_t *__restrict__ row_map,
                                                      T *__restrict__ out, int64_t n, int64_t ors, int64_t ocs, T fill) {
    __shared__ T tile[64][65];
    const int64_t i0 = static_cast<int64_t>(blockIdx.y) * 64, j0 = static_cast<int64_t>(blockIdx.x) * 64;
    const bool in_cols_contig = ics == 1, out_cols_contig = ocs == 1;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int t = threadIdx.x + 256 * k;
        const int r = in_cols_contig ? (t >> 6) : (t & 63), c = in_cols_contig ? (t & 63) : (t >> 6);
        const int64_t i = i0 + r, j = j0 + c;
        T v = fill;
        if (i < n && j < m) {
            const int64_t src = row_map[i];
            if (src >= 0) v = table[src * irs + j * ics];
            else if (src == -2) v = static_cast<T>(__builtin_nan(""));
        }
        tile[r][c] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int t = threadIdx.x + 256 * k;
        const int r = out_cols_contig ? (t >> 6) : (t & 63), c = out_cols_contig ? (t & 63) : (t >> 6);
        const int64_t i = i0 + r, j = j0 + c;
        if (i < n && j < m) out[i * ors + j * ocs] = tile[r][c];
    }
}

// the four value counts of the verbose log (safe_io.py:426-429) in one pass over the matrix
template <typename T>
__global__ __launch_bounds__(256) void k_value_census(const T *__restrict__ raw, int64_t count,
                                                      unsigned long long *__restrict__ acc /*[4]*/) {
    unsigned int c_nan = 0, c_zero = 0, c_pos = 0, c_neg = 0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < count; i += static_cast<int64_t>(gridDim.x) * 256) {
        const T v = raw[i];
        c_nan += v != v;
        c_zero += v == static_cast<T>(0);
        c_pos += v > static_cast<T>(0);
        c_neg += v < static_cast<T>(0);
    }
    unsigned int vals[4] = {c_nan, c_zero, c_pos, c_neg};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned int x = vals[q];
        for (int off = 32; off; off >>= 1) x += __shfl_down(x, off);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&acc[q], static_cast<unsigned long long>(x));
    }
}

// background='network' (safe.py:449-451): NaN -> 0 in place
template <typename T>
__global__ __launch_bounds__(256) void k_nan_to_zero(T *__restrict__ raw, int64_t count) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < count; i += static_cast<int64_t>(gridDim.x) * 256) {
        const T v = raw[i];
        if (v != v) raw[i] = static_cast<T>(0);
    }
}

int attr_build_support(safe_attr *attr) {
    if (attr->sup_ptr) return SAFE_OK;
    SAFE_TRY(safe_attr_prepare(attr));
    SAFE_REQUIRE(attr->n_other == 0, "attr_build_support: matrix is not binary");
    safe_ctx *ctx = attr->ctx;
    const int64_t n = attr->n, m = attr->m;
    std::vector<double> sums(m);
    SAFE_HIP_CHECK(hipMemcpyAsync(sums.data(), attr->col_sum, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    attr->h_sup_ptr.assign(m + 1, 0);
    int64_t total = 0;
    for (int64_t j = 0; j < m; ++j) {
        total += static_cast<int64_t>(sums[j]);
        SAFE_REQUIRE(total < (1ll << 31), "attr_build_support: too many ones for int32 offsets");
        attr->h_sup_ptr[j + 1] = static_cast<int32_t>(total);
    }
    attr->n_ones = total;
    SAFE_TRY(dev_alloc(&attr->sup_ptr, m + 1));
    SAFE_TRY(dev_alloc(&attr->sup_row, total));
    SAFE_HIP_CHECK(hipMemcpyAsync(attr->sup_ptr, attr->h_sup_ptr.data(), (m + 1) * sizeof(int32_t), hipMemcpyHostToDevice,
                                  ctx->stream));
    if (attr->dtype == SAFE_DTYPE_F32)
        hipLaunchKernelGGL(k_fill_support<float>, dim3(ceil_div(m * 64, 256)), dim3(256), 0, ctx->stream, attr->raw, n, m,
                           attr->row_stride, attr->col_stride, attr->sup_ptr, attr->sup_row);
    else
        hipLaunchKernelGGL(k_fill_support<double>, dim3(ceil_div(m * 64, 256)), dim3(256), 0, ctx->stream, attr->raw, n, m,
                           attr->row_stride, attr->col_stride, attr->sup_ptr, attr->sup_row);
    SAFE_HIP_CHECK(hipGetLastError());
    return SAFE_OK;
}

// device block of the row flags: one byte per row, then (8-aligned) four result words of the statistics pass
static size_t flags_block_bytes(int64_t n) { return static_cast<size_t>(ceil_div(n, 8)) * 8 + 32; }

int safe_attr_prepare(safe_attr *attr) {
    if (attr->stats_ready) return SAFE_OK;
    safe_ctx *ctx = attr->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t n = attr->n, m = attr->m;
    const size_t flag_bytes = flags_block_bytes(n), tail_off = flag_bytes - 32;
    // temporaries in one grow-only scratch block of the context (five hipMalloc / hipFree pairs per call cost more
    // than the kernel): accumulators u64 [4 + 2 x 64 partial sums] | row bitmap u32 [n_words] | NaN count per column u32 [m]
    const int64_t n_words = (n + 31) / 32;
    SAFE_REQUIRE(n_words * sizeof(unsigned int) <= 150 * 1024, "safe_attr_stats: too many rows for the LDS row bitmap");
    void *tmp = nullptr;
    const size_t tmp_bytes = 132 * sizeof(unsigned long long) + static_cast<size_t>(n_words + m) * sizeof(unsigned int);
    SAFE_TRY(ctx_scratch(ctx, 11, tmp_bytes, &tmp));
    unsigned long long *d_acc = static_cast<unsigned long long *>(tmp);
    unsigned int *d_rowbits = reinterpret_cast<unsigned int *>(d_acc + 132);
    unsigned int *d_colnan = d_rowbits + n_words;
    SAFE_HIP_CHECK(hipMemsetAsync(tmp, 0, tmp_bytes, ctx->stream));
    uint8_t *flags = nullptr;
    SAFE_TRY(ctx_block_alloc(ctx, flag_bytes, reinterpret_cast<void **>(&flags)));
    if (!attr->col_sum) SAFE_TRY(ctx_block_alloc(ctx, static_cast<size_t>(m) * sizeof(double), reinterpret_cast<void **>(&attr->col_sum)));
    const bool f32 = attr->dtype == SAFE_DTYPE_F32;
    const bool c_order = attr->col_stride == 1 && m > 1;
    if (c_order) SAFE_HIP_CHECK(hipMemsetAsync(attr->col_sum, 0, m * sizeof(double), ctx->stream));   // (several row chunks add up)
#define STATS(T, GC, RPB)                                                                                          \
    do {                                                                                                           \
        if (n_words * sizeof(unsigned int) > 32 * 1024)                                                            \
            SAFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_attr_stats<T, GC>),                \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,                         \
                                               static_cast<int>(n_words * sizeof(unsigned int))));                 \
        hipLaunchKernelGGL((k_attr_stats<T, GC>), dim3(ceil_div(m, GC), ceil_div(n, RPB)), dim3(256),              \
                           n_words * sizeof(unsigned int), ctx->stream, attr->raw, n, m, attr->row_stride,         \
                           attr->col_stride, d_rowbits, d_acc, attr->col_sum, d_acc + 3, d_colnan,                 \
                           static_cast<int64_t>(RPB));                                                             \
    } while (0)
    if (c_order) {
        // a C-order matrix has only m / 64 column groups: split the rows as well until ~2048 workgroups exist
        const int64_t groups = ceil_div(m, 64);
        const int64_t want_chunks = std::max<int64_t>(1, 2048 / groups);
        const int64_t rpb = std::max<int64_t>(256, ceil_div(ceil_div(n, want_chunks), 64) * 64);
        if (f32) STATS(float, 64, rpb);
        else STATS(double, 64, rpb);
    } else {
        if (f32) STATS(float, 1, n);
        else STATS(double, 1, n);
    }
#undef STATS
    hipLaunchKernelGGL(k_stats_finish, dim3(1), dim3(256), 0, ctx->stream, d_acc, d_colnan, m, d_rowbits, n, flags,
                       reinterpret_cast<unsigned long long *>(flags + tail_off));
    SAFE_HIP_CHECK(hipGetLastError());
    void *pinned = nullptr;                                   // flags and the four result words in one DMA copy
    SAFE_TRY(ctx_pinned(ctx, flag_bytes, &pinned));
    SAFE_HIP_CHECK(hipMemcpyAsync(pinned, flags, flag_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    const uint8_t *h_flags = static_cast<const uint8_t *>(pinned);
    unsigned long long h_acc[4];
    memcpy(h_acc, h_flags + tail_off, sizeof(h_acc));
    attr->n_other = static_cast<int64_t>(h_acc[0]);
    attr->max_nan_col = static_cast<int64_t>(h_acc[1]);
    attr->n_non_integer = static_cast<int64_t>(h_acc[2]);
    double mx;
    memcpy(&mx, &h_acc[3], sizeof(double));
    attr->max_abs = mx;
    if (!attr->flags_ready) {
        if (attr->row_flags) ctx_block_free(ctx, attr->row_flags, flag_bytes);
        attr->row_flags = flags;
        attr->flags_ready = true;
        int64_t cnt = 0;
        for (int64_t i = 0; i < n; ++i) cnt += h_flags[i] != 0;
        attr->n_rows_with_value = cnt;
        attr->h_row_flags.assign(h_flags, h_flags + n);
    } else {
        ctx_block_free(ctx, flags, flag_bytes);    // caller supplied global flags (sharded run): keep them
    }
    attr->stats_ready = true;
    return SAFE_OK;
}

static int attr_new(safe_ctx *ctx, int dtype, int64_t n, int64_t m, int64_t rs, int64_t cs, safe_attr **out) {
    SAFE_REQUIRE(ctx && out, "safe_attr_create: NULL argument");
    SAFE_REQUIRE(dtype == SAFE_DTYPE_F32 || dtype == SAFE_DTYPE_F64, "safe_attr_create: dtype must be f32 or f64 (u8: host form only)");
    SAFE_REQUIRE(n >= 1 && m >= 1, "safe_attr_create: empty matrix (%lld x %lld)", (long long)n, (long long)m);
    SAFE_REQUIRE((rs == m && cs == 1) || (rs == 1 && cs == n) || (m == 1 && cs >= 1 && rs == 1) || (n == 1 && cs == 1),
                 "safe_attr_create: matrix must be C- or Fortran-contiguous (strides %lld,%lld for %lld x %lld)",
                 (long long)rs, (long long)cs, (long long)n, (long long)m);
    safe_attr *a = new safe_attr();
    a->ctx = ctx;
    a->n = n;
    a->m = m;
    a->dtype = dtype;
    a->row_stride = rs;
    a->col_stride = cs;
    *out = a;
    return SAFE_OK;
}

extern "C" {

// SAFE_DTYPE_U8: one byte per value on the host and over the link, f32 on the device (same strides)
__global__ __launch_bounds__(256) void k_u8_to_f32(const uint8_t *__restrict__ src, int64_t count, float *__restrict__ dst) {
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 16;
    if (i + 16 <= count) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(dst + i + 4 * q) = make_float4(static_cast<float>(w[q] & 0xFFu), static_cast<float>((w[q] >> 8) & 0xFFu),
                                                                       static_cast<float>((w[q] >> 16) & 0xFFu), static_cast<float>(w[q] >> 24));
    } else {
        for (int64_t j = i; j < count; ++j) dst[j] = static_cast<float>(src[j]);
    }
}

int safe_attr_create_host(safe_ctx *ctx, const void *b_host, int dtype, int64_t n, int64_t m, int64_t row_stride,
                          int64_t col_stride, safe_attr **out) {
    SAFE_REQUIRE(b_host != nullptr, "safe_attr_create_host: b_host is NULL");
    const bool u8 = dtype == SAFE_DTYPE_U8;                 // (a 0/1 matrix as bytes: a quarter of the f32 upload; f32 from here on)
    safe_attr *a = nullptr;
    SAFE_TRY(attr_new(ctx, u8 ? SAFE_DTYPE_F32 : dtype, n, m, row_stride, col_stride, &a));
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t count = static_cast<size_t>(n) * m;
    const size_t bytes = count * (dtype == SAFE_DTYPE_F64 ? 8 : 4);
    uint8_t *d = nullptr;
    int rc = dev_alloc(&d, bytes);
    if (rc != SAFE_OK) {
        delete a;
        return rc;
    }
    hipError_t e = hipSuccess;
    if (u8) {
        void *staged = nullptr;                               // the bytes as they came (16-byte loads: padded)
        rc = ctx_scratch(ctx, 20, (count + 15) / 16 * 16, &staged);
        if (rc != SAFE_OK) {
            (void)hipFree(d);
            delete a;
            return rc;
        }
        e = hipMemcpyAsync(staged, b_host, count, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_u8_to_f32, dim3(static_cast<unsigned int>(ceil_div(static_cast<int64_t>(count), 256 * 16))), dim3(256), 0, ctx->stream,
                               static_cast<const uint8_t *>(staged), static_cast<int64_t>(count), reinterpret_cast<float *>(d));
            e = hipGetLastError();
        }
    } else {
        e = hipMemcpyAsync(d, b_host, bytes, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = safe_stream_sync(ctx->stream);
    if (e != hipSuccess) {
        safe_set_error("safe_attr_create_host: %s", hipGetErrorString(e));
        (void)hipFree(d);
        delete a;
        return SAFE_E_HIP;
    }
    a->raw = d;
    a->owns_raw = true;
    *out = a;
    return SAFE_OK;
}

int safe_attr_create_dev(safe_ctx *ctx, const void *b_dev, int dtype, int64_t n, int64_t m, int64_t row_stride,
                         int64_t col_stride, safe_attr **out) {
    SAFE_REQUIRE(b_dev != nullptr, "safe_attr_create_dev: b_dev is NULL");
    safe_attr *a = nullptr;
    SAFE_TRY(attr_new(ctx, dtype, n, m, row_stride, col_stride, &a));
    a->raw = b_dev;
    a->owns_raw = false;
    *out = a;
    return SAFE_OK;
}

int safe_attr_reindex(safe_ctx *ctx, const void *table_host, int dtype, int64_t n_labels, int64_t m, int64_t row_stride,
                      int64_t col_stride, const int64_t *row_map_host, int64_t n, double fill_value, int out_order,
                      void *out_host, safe_attr **out) {
    SAFE_REQUIRE(ctx && table_host && row_map_host && out, "safe_attr_reindex: NULL argument");
    SAFE_REQUIRE(dtype == SAFE_DTYPE_F32 || dtype == SAFE_DTYPE_F64, "safe_attr_reindex: dtype must be f32 or f64");
    SAFE_REQUIRE(n_labels >= 1 && m >= 1 && n >= 1, "safe_attr_reindex: empty input (%lld labels x %lld attributes -> %lld nodes)",
                 (long long)n_labels, (long long)m, (long long)n);
    SAFE_REQUIRE((row_stride == m && col_stride == 1) || (row_stride == 1 && col_stride == n_labels),
                 "safe_attr_reindex: table must be C- or Fortran-contiguous");
    SAFE_REQUIRE(out_order == 0 || out_order == 1, "safe_attr_reindex: out_order must be 0 (C) or 1 (Fortran)");
    for (int64_t i = 0; i < n; ++i)
        SAFE_REQUIRE(row_map_host[i] >= -2 && row_map_host[i] < n_labels, "safe_attr_reindex: row_map[%lld] = %lld out of range",
                     (long long)i, (long long)row_map_host[i]);
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t esz = dtype == SAFE_DTYPE_F32 ? 4 : 8;
    const size_t in_bytes = static_cast<size_t>(n_labels) * m * esz, out_bytes = static_cast<size_t>(n) * m * esz;
    uint8_t *d_in = nullptr, *d_out = nullptr;
    int64_t *d_map = nullptr;
    SAFE_TRY(dev_alloc(&d_in, in_bytes));
    int rc = dev_alloc(&d_out, out_bytes);
    if (rc == SAFE_OK) rc = dev_alloc(&d_map, n);
    const int64_t ors = out_order == 0 ? m : 1, ocs = out_order == 0 ? 1 : n;
    safe_attr *a = nullptr;
    if (rc == SAFE_OK) rc = attr_new(ctx, dtype, n, m, ors, ocs, &a);
    hipError_t e = hipSuccess;
    if (rc == SAFE_OK) {
        e = hipMemcpyAsync(d_in, table_host, in_bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_map, row_map_host, n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            const dim3 grid(static_cast<unsigned>(ceil_div(m, 64)), static_cast<unsigned>(ceil_div(n, 64)));
            if (dtype == SAFE_DTYPE_F32)
                hipLaunchKernelGGL(k_reindex_rows<float>, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const float *>(d_in),
                                   n_labels, m, row_stride, col_stride, d_map, reinterpret_cast<float *>(d_out), n, ors, ocs,
                                   static_cast<float>(fill_value));
            else
                hipLaunchKernelGGL(k_reindex_rows<double>, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const double *>(d_in),
                                   n_labels, m, row_stride, col_stride, d_map, reinterpret_cast<double *>(d_out), n, ors, ocs,
                                   fill_value);
            e = hipGetLastError();
        }
        if (e == hipSuccess && out_host) e = hipMemcpyAsync(out_host, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = safe_stream_sync(ctx->stream);
    }
    (void)hipFree(d_in);
    (void)hipFree(d_map);
    if (rc != SAFE_OK || e != hipSuccess) {
        if (e != hipSuccess) safe_set_error("safe_attr_reindex: %s", hipGetErrorString(e));
        (void)hipFree(d_out);
        delete a;
        return rc != SAFE_OK ? rc : SAFE_E_HIP;
    }
    a->raw = d_out;
    a->owns_raw = true;
    *out = a;
    return SAFE_OK;
}

int safe_attr_value_counts(safe_attr *attr, int64_t *n_nan, int64_t *n_zero, int64_t *n_positive, int64_t *n_negative) {
    SAFE_REQUIRE(attr != nullptr, "safe_attr_value_counts: attr is NULL");
    safe_ctx *ctx = attr->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    unsigned long long *d_acc = nullptr, h_acc[4];
    SAFE_TRY(dev_alloc(&d_acc, 4));
    SAFE_HIP_CHECK(hipMemsetAsync(d_acc, 0, sizeof(h_acc), ctx->stream));
    const int64_t count = attr->n * attr->m;
    const unsigned blocks = static_cast<unsigned>(std::min<int64_t>(ceil_div(count, 256), 8192));
    if (attr->dtype == SAFE_DTYPE_F32)
        hipLaunchKernelGGL(k_value_census<float>, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const float *>(attr->raw), count, d_acc);
    else
        hipLaunchKernelGGL(k_value_census<double>, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const double *>(attr->raw), count, d_acc);
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(hipMemcpyAsync(h_acc, d_acc, sizeof(h_acc), hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    (void)hipFree(d_acc);
    if (n_nan) *n_nan = static_cast<int64_t>(h_acc[0]);
    if (n_zero) *n_zero = static_cast<int64_t>(h_acc[1]);
    if (n_positive) *n_positive = static_cast<int64_t>(h_acc[2]);
    if (n_negative) *n_negative = static_cast<int64_t>(h_acc[3]);
    return SAFE_OK;
}

int safe_attr_nan_to_zero(safe_attr *attr) {
    SAFE_REQUIRE(attr != nullptr, "safe_attr_nan_to_zero: attr is NULL");
    safe_ctx *ctx = attr->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t count = attr->n * attr->m;
    const unsigned blocks = static_cast<unsigned>(std::min<int64_t>(ceil_div(count, 256), 8192));
    void *raw = const_cast<void *>(attr->raw);
    if (attr->dtype == SAFE_DTYPE_F32)
        hipLaunchKernelGGL(k_nan_to_zero<float>, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<float *>(raw), count);
    else
        hipLaunchKernelGGL(k_nan_to_zero<double>, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<double *>(raw), count);
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    // every derived fact is stale now
    attr->stats_ready = false;
    attr->flags_ready = false;
    (void)hipFree(attr->sup_ptr);
    (void)hipFree(attr->sup_row);
    attr->sup_ptr = nullptr;
    attr->sup_row = nullptr;
    attr->h_sup_ptr.clear();
    return SAFE_OK;
}

int safe_attr_download(safe_attr *attr, void *out_host) {
    SAFE_REQUIRE(attr && out_host, "safe_attr_download: NULL argument");
    safe_ctx *ctx = attr->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t bytes = static_cast<size_t>(attr->n) * attr->m * (attr->dtype == SAFE_DTYPE_F32 ? 4 : 8);
    SAFE_HIP_CHECK(hipMemcpyAsync(out_host, attr->raw, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    return SAFE_OK;
}

int safe_attr_destroy(safe_attr *attr) {
    if (!attr) return SAFE_OK;
    (void)hipSetDevice(attr->ctx->device);
    (void)safe_stream_sync(attr->ctx->stream);
    if (attr->owns_raw) (void)hipFree(const_cast<void *>(attr->raw));
    ctx_block_free(attr->ctx, attr->row_flags, flags_block_bytes(attr->n));
    ctx_block_free(attr->ctx, attr->col_sum, static_cast<size_t>(attr->m) * sizeof(double));
    (void)hipFree(attr->sup_ptr);
    (void)hipFree(attr->sup_row);
    delete attr;
    return SAFE_OK;
}

int safe_attr_stats(safe_attr *attr, int64_t *n_other, int64_t *max_nan_col, int64_t *n_rows_with_value,
                    int64_t *n_non_integer) {
    SAFE_REQUIRE(attr != nullptr, "safe_attr_stats: attr is NULL");
    SAFE_TRY(safe_attr_prepare(attr));
    if (n_other) *n_other = attr->n_other;
    if (max_nan_col) *max_nan_col = attr->max_nan_col;
    if (n_rows_with_value) *n_rows_with_value = attr->n_rows_with_value;
    if (n_non_integer) *n_non_integer = attr->n_non_integer;
    return SAFE_OK;
}

int safe_attr_row_flags(safe_attr *attr, uint8_t *out_host) {
    SAFE_REQUIRE(attr && out_host, "safe_attr_row_flags: NULL argument");
    SAFE_TRY(safe_attr_prepare(attr));
    if (static_cast<int64_t>(attr->h_row_flags.size()) == attr->n) {   // the host copy made by the statistics pass / set_row_flags
        memcpy(out_host, attr->h_row_flags.data(), attr->n);
        return SAFE_OK;
    }
    safe_ctx *ctx = attr->ctx;
    SAFE_HIP_CHECK(hipMemcpyAsync(out_host, attr->row_flags, attr->n, hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    return SAFE_OK;
}

int safe_attr_set_row_flags(safe_attr *attr, const uint8_t *flags_host) {
    SAFE_REQUIRE(attr && flags_host, "safe_attr_set_row_flags: NULL argument");
    safe_ctx *ctx = attr->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t flag_bytes = flags_block_bytes(attr->n);
    if (!attr->row_flags) SAFE_TRY(ctx_block_alloc(ctx, flag_bytes, reinterpret_cast<void **>(&attr->row_flags)));
    std::vector<uint8_t> tmp(flag_bytes, 0);
    int64_t cnt = 0;
    for (int64_t i = 0; i < attr->n; ++i) {
        tmp[i] = flags_host[i] ? 1 : 0;
        cnt += tmp[i];
    }
    SAFE_HIP_CHECK(hipMemcpyAsync(attr->row_flags, tmp.data(), flag_bytes, hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    attr->flags_ready = true;
    attr->n_rows_with_value = cnt;
    attr->h_row_flags.assign(tmp.begin(), tmp.begin() + attr->n);
    return SAFE_OK;
}

}  // extern "C"
