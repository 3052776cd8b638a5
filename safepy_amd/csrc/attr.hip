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
                                                    unsigned long long *__restrict__ acc /*[4]*/,
                                                    double *__restrict__ col_sum, unsigned long long *__restrict__ max_abs_bits) {
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
    for (int64_t w = threadIdx.x; w < n_words; w += 256) s_bits[w] = 0;
    __syncthreads();
    unsigned int c_nan = 0, c_other = 0, c_nonint = 0;
    double mx = 0.0, sum = 0.0;
    const int64_t n_round = (n + RL - 1) / RL * RL;    // every thread runs the same trip count (ballots)
    for (int64_t i = ry; i < n_round; i += RL) {
        bool has = false;
        if (i < n && j < m) {
            const double v = load_attr<T>(raw, i * rs + j * cs);
            if (v != v) {
                ++c_nan;
            } else {
                has = true;
                sum += v;
                if (v != 0.0 && v != 1.0) ++c_other;
                if (v != floor(v)) ++c_nonint;
                const double a = fabs(v);
                if (a > mx) mx = a;
            }
        }
        const unsigned long long bal = __ballot(has);
        if (GC == 1) {
            // the wave holds 64 consecutive rows starting at i - lane: two bitmap words, owned by this wave
            const int64_t r0 = i - lane;
            if (lane == 0 && bal) {
                if (r0 < n) s_bits[r0 >> 5] |= static_cast<unsigned int>(bal);
                if (r0 + 32 < n) s_bits[(r0 >> 5) + 1] |= static_cast<unsigned int>(bal >> 32);
            }
        } else {
            // the wave holds ONE row (i) across 64 columns
            if (lane == 0 && bal && i < n) atomicOr(&s_bits[i >> 5], 1u << (i & 31));
        }
    }
    s_sum[threadIdx.x] = sum;
    s_max[threadIdx.x] = mx;
    s_nan[threadIdx.x] = c_nan;
    s_other[threadIdx.x] = c_other;
    s_nonint[threadIdx.x] = c_nonint;
    __syncthreads();
    for (int64_t w = threadIdx.x; w < n_words; w += 256) {
        const unsigned int mine = s_bits[w];
        if (mine & ~row_bits[w]) atomicOr(&row_bits[w], mine);       // racy pre-check is benign
    }
    if (ry == 0 && j < m) {
        double total = 0.0, tmx = 0.0;
        unsigned long long t_nan = 0, t_other = 0, t_nonint = 0;
        for (int r = 0; r < RL; ++r) {
            const int t = GC == 1 ? r : r * GC + cx;
            total += s_sum[t];
            if (s_max[t] > tmx) tmx = s_max[t];
            t_nan += s_nan[t];
            t_other += s_other[t];
            t_nonint += s_nonint[t];
        }
        col_sum[j] = total;
        if (t_other) atomicAdd(&acc[0], t_other);
        atomicMax(&acc[1], t_nan);
        if (t_nonint) atomicAdd(&acc[2], t_nonint);
        atomicMax(max_abs_bits, static_cast<unsigned long long>(__double_as_longlong(tmx)));
    }
}

__global__ void k_bits_to_bytes(const unsigned int *__restrict__ bits, int64_t n, uint8_t *__restrict__ bytes) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) bytes[i] = (bits[i >> 5] >> (i & 31)) & 1u;
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

int attr_build_support(safe_attr *attr) {
    if (attr->sup_ptr) return SAFE_OK;
    SAFE_TRY(safe_attr_prepare(attr));
    SAFE_REQUIRE(attr->n_other == 0, "attr_build_support: matrix is not binary");
    safe_ctx *ctx = attr->ctx;
    const int64_t n = attr->n, m = attr->m;
    std::vector<double> sums(m);
    SAFE_HIP_CHECK(hipMemcpyAsync(sums.data(), attr->col_sum, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
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

int safe_attr_prepare(safe_attr *attr) {
    if (attr->stats_ready) return SAFE_OK;
    safe_ctx *ctx = attr->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t n = attr->n, m = attr->m;
    const size_t flag_bytes = static_cast<size_t>(ceil_div(n, 4)) * 4;
    unsigned long long *d_acc = nullptr;
    SAFE_TRY(dev_alloc(&d_acc, 4));
    SAFE_HIP_CHECK(hipMemsetAsync(d_acc, 0, 4 * sizeof(unsigned long long), ctx->stream));
    uint8_t *flags = nullptr;
    SAFE_TRY(dev_alloc(&flags, flag_bytes));
    const int64_t n_words = (n + 31) / 32;
    unsigned int *d_rowbits = nullptr;
    SAFE_TRY(dev_alloc(&d_rowbits, n_words));
    SAFE_HIP_CHECK(hipMemsetAsync(d_rowbits, 0, n_words * sizeof(unsigned int), ctx->stream));
    SAFE_REQUIRE(n_words * sizeof(unsigned int) <= 150 * 1024, "safe_attr_stats: too many rows for the LDS row bitmap");
    if (!attr->col_sum) SAFE_TRY(dev_alloc(&attr->col_sum, m));
    const bool f32 = attr->dtype == SAFE_DTYPE_F32;
    const bool c_order = attr->col_stride == 1 && m > 1;
#define STATS(T, GC)                                                                                               \
    do {                                                                                                           \
        if (n_words * sizeof(unsigned int) > 32 * 1024)                                                            \
            SAFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_attr_stats<T, GC>),                \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,                         \
                                               static_cast<int>(n_words * sizeof(unsigned int))));                 \
        hipLaunchKernelGGL((k_attr_stats<T, GC>), dim3(ceil_div(m, GC)), dim3(256), n_words * sizeof(unsigned int), \
                           ctx->stream, attr->raw, n, m, attr->row_stride, attr->col_stride, d_rowbits, d_acc,     \
                           attr->col_sum, d_acc + 3);                                                              \
    } while (0)
    if (c_order) {
        if (f32) STATS(float, 64);
        else STATS(double, 64);
    } else {
        if (f32) STATS(float, 1);
        else STATS(double, 1);
    }
#undef STATS
    hipLaunchKernelGGL(k_bits_to_bytes, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, d_rowbits, n, flags);
    SAFE_HIP_CHECK(hipGetLastError());
    unsigned long long h_acc[4];
    SAFE_HIP_CHECK(hipMemcpyAsync(h_acc, d_acc, sizeof(h_acc), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<uint8_t> h_flags(flag_bytes);
    SAFE_HIP_CHECK(hipMemcpyAsync(h_flags.data(), flags, flag_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    (void)hipFree(d_acc);
    (void)hipFree(d_rowbits);
    attr->n_other = static_cast<int64_t>(h_acc[0]);
    attr->max_nan_col = static_cast<int64_t>(h_acc[1]);
    attr->n_non_integer = static_cast<int64_t>(h_acc[2]);
    double mx;
    memcpy(&mx, &h_acc[3], sizeof(double));
    attr->max_abs = mx;
    if (!attr->flags_ready) {
        if (attr->row_flags) (void)hipFree(attr->row_flags);
        attr->row_flags = flags;
        attr->flags_ready = true;
        int64_t cnt = 0;
        for (int64_t i = 0; i < n; ++i) cnt += h_flags[i] != 0;
        attr->n_rows_with_value = cnt;
    } else {
        (void)hipFree(flags);    // caller supplied global flags (sharded run): keep them
    }
    attr->stats_ready = true;
    return SAFE_OK;
}

static int attr_new(safe_ctx *ctx, int dtype, int64_t n, int64_t m, int64_t rs, int64_t cs, safe_attr **out) {
    SAFE_REQUIRE(ctx && out, "safe_attr_create: NULL argument");
    SAFE_REQUIRE(dtype == SAFE_DTYPE_F32 || dtype == SAFE_DTYPE_F64, "safe_attr_create: dtype must be f32 or f64");
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

int safe_attr_create_host(safe_ctx *ctx, const void *b_host, int dtype, int64_t n, int64_t m, int64_t row_stride,
                          int64_t col_stride, safe_attr **out) {
    SAFE_REQUIRE(b_host != nullptr, "safe_attr_create_host: b_host is NULL");
    safe_attr *a = nullptr;
    SAFE_TRY(attr_new(ctx, dtype, n, m, row_stride, col_stride, &a));
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t bytes = static_cast<size_t>(n) * m * (dtype == SAFE_DTYPE_F32 ? 4 : 8);
    uint8_t *d = nullptr;
    int rc = dev_alloc(&d, bytes);
    if (rc != SAFE_OK) {
        delete a;
        return rc;
    }
    hipError_t e = hipMemcpyAsync(d, b_host, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
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

int safe_attr_destroy(safe_attr *attr) {
    if (!attr) return SAFE_OK;
    (void)hipSetDevice(attr->ctx->device);
    (void)hipStreamSynchronize(attr->ctx->stream);
    if (attr->owns_raw) (void)hipFree(const_cast<void *>(attr->raw));
    (void)hipFree(attr->row_flags);
    (void)hipFree(attr->col_sum);
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
    safe_ctx *ctx = attr->ctx;
    SAFE_HIP_CHECK(hipMemcpyAsync(out_host, attr->row_flags, attr->n, hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SAFE_OK;
}

int safe_attr_set_row_flags(safe_attr *attr, const uint8_t *flags_host) {
    SAFE_REQUIRE(attr && flags_host, "safe_attr_set_row_flags: NULL argument");
    safe_ctx *ctx = attr->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t flag_bytes = static_cast<size_t>(ceil_div(attr->n, 4)) * 4;
    if (!attr->row_flags) SAFE_TRY(dev_alloc(&attr->row_flags, flag_bytes));
    std::vector<uint8_t> tmp(flag_bytes, 0);
    int64_t cnt = 0;
    for (int64_t i = 0; i < attr->n; ++i) {
        tmp[i] = flags_host[i] ? 1 : 0;
        cnt += tmp[i];
    }
    SAFE_HIP_CHECK(hipMemcpyAsync(attr->row_flags, tmp.data(), flag_bytes, hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    attr->flags_ready = true;
    attr->n_rows_with_value = cnt;
    return SAFE_OK;
}

}  // extern "C"
