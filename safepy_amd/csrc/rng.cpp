// Host-side permutation stream of run_permutations (safepy/safe_extras.py:46-58):
// legacy NumPy RandomState = MT19937 seeded by init_genrand, np.random.permutation =
// Fisher-Yates from the top with masked-rejection bounded integers.
// The stream is inherently sequential (the number of draws a shuffle consumes depends on
// the rejections), so it runs on the host and the composed index tables are uploaded.
#include <algorithm>
#include <random>

#include "common.h"

namespace {

struct MT19937 {
    uint32_t mt[624];
    int pos;

    explicit MT19937(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + static_cast<uint32_t>(i);
        pos = 624;
    }

    void refill() {
        auto twist = [](uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); };
        int i = 0;
        for (; i < 624 - 397; ++i) mt[i] = mt[i + 397] ^ twist(mt[i], mt[i + 1]);
        for (; i < 623; ++i) mt[i] = mt[i + 397 - 624] ^ twist(mt[i], mt[i + 1]);
        mt[623] = mt[396] ^ twist(mt[623], mt[0]);
        pos = 0;
    }

    inline uint32_t next() {
        if (pos == 624) refill();
        uint32_t y = mt[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }

    // uniform integer in [0, top], legacy random_interval
    inline uint32_t interval(uint32_t top) {
        if (top == 0) return 0;
        uint32_t mask = top;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        uint32_t v;
        while ((v = next() & mask) > top) {
        }
        return v;
    }

    template <typename T>
    inline void shuffle(T *a, int64_t n) {
        for (int64_t i = n - 1; i > 0; --i) {
            const uint32_t j = interval(static_cast<uint32_t>(i));
            const T t = a[i];
            a[i] = a[j];
            a[j] = t;
        }
    }
};

uint32_t entropy_seed() {
    std::random_device rd;
    return rd();
}

}  // namespace

__global__ void k_invert_perms(const int32_t *__restrict__ table, int64_t stride, int64_t total, int64_t inv_stride,
                               uint16_t *__restrict__ inverse_t) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int64_t p = idx / stride, k = idx % stride;
    inverse_t[static_cast<int64_t>(table[idx]) * inv_stride + p] = static_cast<uint16_t>(k);
}

int perms_build_inverse(safe_perms *perms) {
    if (perms->inverse_t) return SAFE_OK;
    SAFE_REQUIRE(perms->n < 65535, "perms_build_inverse: n too large for 16-bit positions");
    safe_ctx *ctx = perms->ctx;
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

__global__ void k_table_u16(const int32_t *__restrict__ table, int64_t stride, int64_t count, int64_t stride16,
                            uint16_t *__restrict__ out) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= count * stride16) return;
    const int64_t p = idx / stride16, k = idx % stride16;
    out[idx] = k < stride ? static_cast<uint16_t>(table[p * stride + k]) : static_cast<uint16_t>(stride - 1);
}

int perms_build_table16(safe_perms *perms) {
    if (perms->table16) return SAFE_OK;
    SAFE_REQUIRE(perms->n < 65535, "perms_build_table16: n too large for 16-bit rows");
    safe_ctx *ctx = perms->ctx;
    const int64_t stride = perms->n + 1;
    perms->stride16 = (stride + 7) / 8 * 8;
    const int64_t total = std::max<int64_t>(perms->count, 1) * perms->stride16;
    SAFE_TRY(dev_alloc(&perms->table16, static_cast<size_t>(total)));
    if (perms->count)
        hipLaunchKernelGGL(k_table_u16, dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, perms->table, stride,
                           perms->count, perms->stride16, perms->table16);
    SAFE_HIP_CHECK(hipGetLastError());
    return SAFE_OK;
}

extern "C" {

int safe_rng_permutations_host(uint32_t seed, const int64_t *values, int64_t n_items, int64_t count, int64_t *out) {
    SAFE_REQUIRE(n_items >= 0 && count >= 0, "safe_rng_permutations_host: negative size");
    SAFE_REQUIRE(n_items == 0 || count == 0 || (values && out), "safe_rng_permutations_host: NULL argument");
    SAFE_REQUIRE(n_items < (1ll << 32), "safe_rng_permutations_host: n_items too large");
    MT19937 rng(seed);
    for (int64_t c = 0; c < count; ++c) {
        int64_t *dst = out + c * n_items;
        memcpy(dst, values, n_items * sizeof(int64_t));
        rng.shuffle(dst, n_items);
    }
    return SAFE_OK;
}

int safe_perms_create(safe_ctx *ctx, int64_t n, const uint8_t *movable_host, int64_t num_permutations, int has_seed,
                      uint32_t seed, safe_perms **out) {
    SAFE_REQUIRE(ctx && movable_host && out, "safe_perms_create: NULL argument");
    SAFE_REQUIRE(n >= 1 && n < (1ll << 31) - 1, "safe_perms_create: n out of range");
    SAFE_REQUIRE(num_permutations >= 0, "safe_perms_create: negative permutation count");
    *out = nullptr;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<int32_t> movable;
    movable.reserve(n);
    for (int64_t i = 0; i < n; ++i)
        if (movable_host[i]) movable.push_back(static_cast<int32_t>(i));
    const int64_t k = static_cast<int64_t>(movable.size());
    const int64_t stride = n + 1;
    safe_perms *p = new safe_perms();
    p->ctx = ctx;
    p->n = n;
    p->count = num_permutations;
    int rc = dev_alloc(&p->table, static_cast<size_t>(std::max<int64_t>(num_permutations, 1)) * stride);
    if (rc != SAFE_OK) {
        delete p;
        return rc;
    }
    // host staging in chunks so generation overlaps the uploads
    const int64_t chunk = 64;
    int32_t *h_buf[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
        e = hipHostMalloc(reinterpret_cast<void **>(&h_buf[b]), chunk * stride * sizeof(int32_t), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[b], hipEventDisableTiming);
    }
    if (e == hipSuccess) {
        MT19937 rng(has_seed ? seed : entropy_seed());
        std::vector<int32_t> cur(stride), draw(std::max<int64_t>(k, 1)), gathered(std::max<int64_t>(k, 1));
        for (int64_t i = 0; i < stride; ++i) cur[i] = static_cast<int32_t>(i);
        int64_t c = 0;
        for (int64_t p0 = 0; p0 < num_permutations && e == hipSuccess; p0 += chunk, ++c) {
            const int b = static_cast<int>(c & 1);
            if (c >= 2) e = hipEventSynchronize(done[b]);
            if (e != hipSuccess) break;
            const int64_t p1 = std::min(num_permutations, p0 + chunk);
            for (int64_t q = p0; q < p1; ++q) {
                // n2a[indx_vals,:] = n2a[np.random.permutation(indx_vals),:]  (safe_extras.py:58)
                memcpy(draw.data(), movable.data(), k * sizeof(int32_t));
                rng.shuffle(draw.data(), k);
                for (int64_t t = 0; t < k; ++t) gathered[t] = cur[draw[t]];
                for (int64_t t = 0; t < k; ++t) cur[movable[t]] = gathered[t];
                memcpy(h_buf[b] + (q - p0) * stride, cur.data(), stride * sizeof(int32_t));
            }
            e = hipMemcpyAsync(p->table + p0 * stride, h_buf[b], (p1 - p0) * stride * sizeof(int32_t),
                               hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(done[b], ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    for (int b = 0; b < 2; ++b) {
        if (h_buf[b]) (void)hipHostFree(h_buf[b]);
        if (done[b]) (void)hipEventDestroy(done[b]);
    }
    if (e != hipSuccess) {
        safe_set_error("safe_perms_create: %s", hipGetErrorString(e));
        (void)hipFree(p->table);
        delete p;
        return SAFE_E_HIP;
    }
    *out = p;
    return SAFE_OK;
}

int safe_perms_destroy(safe_perms *perms) {
    if (!perms) return SAFE_OK;
    (void)hipSetDevice(perms->ctx->device);
    (void)hipStreamSynchronize(perms->ctx->stream);
    (void)hipFree(perms->table);
    (void)hipFree(perms->inverse_t);
    (void)hipFree(perms->table16);
    delete perms;
    return SAFE_OK;
}

int safe_perms_read(safe_perms *perms, int64_t p0, int64_t p1, int32_t *out_host) {
    SAFE_REQUIRE(perms && out_host, "safe_perms_read: NULL argument");
    SAFE_REQUIRE(0 <= p0 && p0 <= p1 && p1 <= perms->count, "safe_perms_read: range [%lld,%lld) outside [0,%lld)",
                 (long long)p0, (long long)p1, (long long)perms->count);
    safe_ctx *ctx = perms->ctx;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t stride = perms->n + 1;
    SAFE_HIP_CHECK(hipMemcpy2DAsync(out_host, perms->n * sizeof(int32_t), perms->table + p0 * stride,
                                    stride * sizeof(int32_t), perms->n * sizeof(int32_t), p1 - p0,
                                    hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SAFE_OK;
}

}  // extern "C"
