// The two data-parallel pieces of the consumers of nes_binary:
//
//  * define_top_attributes (safepy/safe.py:631-656): for every candidate attribute, the connected
//    components of the subgraph induced by its enriched nodes (nx.subgraph + nx.connected_components,
//    one Python call per attribute in the reference).  Here: all attributes at once, min-label
//    hooking + pointer jumping over the edge list (Shiloach-Vishkin style); the final label of a
//    node is the smallest node id of its component.
//  * define_domains (safepy/safe.py:672-673): the condensed Jaccard distance vector between the
//    binarised enrichment profiles of the top attributes -- scipy's pdist(m, 'jaccard') inside
//    linkage(): d = #(x != y and (x != 0 or y != 0)) / #(x != 0 or y != 0), 0 when the denominator
//    is 0 -- from bit-packed columns with popcounts; the linkage itself stays SciPy's.
#include <algorithm>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void k_cc_init(const double *__restrict__ member, int64_t n, int64_t n_cols,
                                                 int32_t *__restrict__ parent) {
    // member: [n][n_cols] row-major (nes_binary[:, cols]); parent: [n_cols][n]
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= n * n_cols) return;
    const int64_t a = idx / n, v = idx % n;
    parent[idx] = member[v * n_cols + a] > 0.0 ? static_cast<int32_t>(v) : -1;
}

__global__ __launch_bounds__(256) void k_cc_hook(const int32_t *__restrict__ eu, const int32_t *__restrict__ ev, int64_t n_edges,
                                                 int64_t n, int32_t *__restrict__ parent, int *__restrict__ changed) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (e >= n_edges) return;
    int32_t *p = parent + static_cast<int64_t>(blockIdx.y) * n;
    const int32_t pu = p[eu[e]], pv = p[ev[e]];
    if (pu < 0 || pv < 0 || pu == pv) return;
    // hang the larger root under the smaller label (labels only ever decrease)
    if (pu < pv) atomicMin(&p[pv], pu);
    else atomicMin(&p[pu], pv);
    *changed = 1;
}

__global__ __launch_bounds__(256) void k_cc_compress(int64_t n, int64_t n_cols, int32_t *__restrict__ parent) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= n * n_cols) return;
    int32_t *p = parent + (idx / n) * n;
    int32_t l = parent[idx];
    if (l < 0) return;
    while (p[l] != l) l = p[l];                                          // roots are fixed points
    parent[idx] = l;
}

__global__ __launch_bounds__(256) void k_jaccard_pack(const double *__restrict__ x, int64_t m_top, int64_t n, int64_t words,
                                                      unsigned long long *__restrict__ bits) {
    // x: [m_top][n] row-major; one wave per 64 consecutive values of one row
    const int64_t w = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6), a = blockIdx.y;
    if (w >= words) return;
    const int64_t i = w * 64 + (threadIdx.x & 63);
    const bool on = i < n && x[a * n + i] != 0.0;
    const unsigned long long v = __builtin_amdgcn_ballot_w64(on);
    if ((threadIdx.x & 63) == 0) bits[a * words + w] = v;
}

__global__ __launch_bounds__(256) void k_jaccard_pairs(const unsigned long long *__restrict__ bits, int64_t m_top, int64_t words,
                                                       double *__restrict__ out) {
    const int64_t i = blockIdx.y, j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (j <= i || j >= m_top) return;
    const unsigned long long *a = bits + i * words, *b = bits + j * words;
    long long neq = 0, any = 0;
    for (int64_t w = 0; w < words; ++w) {
        const unsigned long long x = a[w], y = b[w];
        neq += __popcll(x ^ y);
        any += __popcll(x | y);
    }
    const int64_t k = i * (2 * m_top - i - 1) / 2 + (j - i - 1);       // scipy's condensed index
    out[k] = any == 0 ? 0.0 : static_cast<double>(neq) / static_cast<double>(any);
}

}  // namespace

extern "C" {

int safe_enriched_components(safe_ctx *ctx, int64_t n, int64_t n_edges, const int32_t *edge_u, const int32_t *edge_v,
                             const double *member_host, int64_t n_cols, int32_t *labels_host) {
    SAFE_REQUIRE(ctx && labels_host && (n_cols == 0 || member_host), "safe_enriched_components: NULL argument");
    SAFE_REQUIRE(n >= 1 && n < (1ll << 31) && n_edges >= 0 && n_cols >= 0, "safe_enriched_components: bad sizes");
    SAFE_REQUIRE(n_edges == 0 || (edge_u && edge_v), "safe_enriched_components: NULL edge list");
    if (n_cols == 0) return SAFE_OK;
    for (int64_t e = 0; e < n_edges; ++e)
        SAFE_REQUIRE(edge_u[e] >= 0 && edge_u[e] < n && edge_v[e] >= 0 && edge_v[e] < n, "safe_enriched_components: edge %lld out of range",
                     (long long)e);
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    double *d_member = nullptr;
    int32_t *d_parent = nullptr, *d_eu = nullptr, *d_ev = nullptr;
    int *d_changed = nullptr;
    int rc = dev_alloc(&d_member, static_cast<size_t>(n) * n_cols);
    if (rc == SAFE_OK) rc = dev_alloc(&d_parent, static_cast<size_t>(n) * n_cols);
    if (rc == SAFE_OK) rc = dev_alloc(&d_eu, std::max<int64_t>(n_edges, 1));
    if (rc == SAFE_OK) rc = dev_alloc(&d_ev, std::max<int64_t>(n_edges, 1));
    if (rc == SAFE_OK) rc = dev_alloc(&d_changed, 1);
    hipError_t e = hipSuccess;
    if (rc == SAFE_OK) {
        e = hipMemcpyAsync(d_member, member_host, static_cast<size_t>(n) * n_cols * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && n_edges)
            e = hipMemcpyAsync(d_eu, edge_u, n_edges * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && n_edges)
            e = hipMemcpyAsync(d_ev, edge_v, n_edges * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    }
    if (rc == SAFE_OK && e == hipSuccess) {
        const int64_t total = n * n_cols;
        hipLaunchKernelGGL(k_cc_init, dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, d_member, n, n_cols, d_parent);
        for (int round = 0; n_edges > 0 && round < 64; ++round) {         // O(log n) rounds in practice
            int changed = 0;
            e = hipMemsetAsync(d_changed, 0, sizeof(int), ctx->stream);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(k_cc_hook, dim3(ceil_div(n_edges, 256), n_cols), dim3(256), 0, ctx->stream, d_eu, d_ev, n_edges, n,
                               d_parent, d_changed);
            hipLaunchKernelGGL(k_cc_compress, dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, n, n_cols, d_parent);
            e = hipMemcpyAsync(&changed, d_changed, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = safe_stream_sync(ctx->stream);
            if (e != hipSuccess || !changed) break;
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(labels_host, d_parent, static_cast<size_t>(total) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = safe_stream_sync(ctx->stream);
    }
    if (rc == SAFE_OK && e != hipSuccess) {
        safe_set_error("safe_enriched_components: %s", hipGetErrorString(e));
        rc = SAFE_E_HIP;
    }
    (void)hipFree(d_member);
    (void)hipFree(d_parent);
    (void)hipFree(d_eu);
    (void)hipFree(d_ev);
    (void)hipFree(d_changed);
    return rc;
}

int safe_jaccard_condensed(safe_ctx *ctx, int64_t m_top, int64_t n, const double *x_host, double *out_host) {
    SAFE_REQUIRE(ctx && (m_top < 2 || (x_host && out_host)), "safe_jaccard_condensed: NULL argument");
    SAFE_REQUIRE(m_top >= 0 && n >= 1, "safe_jaccard_condensed: bad sizes");
    if (m_top < 2) return SAFE_OK;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t words = ceil_div(n, 64), pairs = m_top * (m_top - 1) / 2;
    double *d_x = nullptr, *d_out = nullptr;
    unsigned long long *d_bits = nullptr;
    int rc = dev_alloc(&d_x, static_cast<size_t>(m_top) * n);
    if (rc == SAFE_OK) rc = dev_alloc(&d_bits, static_cast<size_t>(m_top) * words);
    if (rc == SAFE_OK) rc = dev_alloc(&d_out, pairs);
    hipError_t e = hipSuccess;
    if (rc == SAFE_OK) {
        e = hipMemcpyAsync(d_x, x_host, static_cast<size_t>(m_top) * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_jaccard_pack, dim3(ceil_div(words, 4), m_top), dim3(256), 0, ctx->stream, d_x, m_top, n, words, d_bits);
            hipLaunchKernelGGL(k_jaccard_pairs, dim3(ceil_div(m_top, 256), m_top), dim3(256), 0, ctx->stream, d_bits, m_top, words, d_out);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out_host, d_out, pairs * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = safe_stream_sync(ctx->stream);
        if (e != hipSuccess) {
            safe_set_error("safe_jaccard_condensed: %s", hipGetErrorString(e));
            rc = SAFE_E_HIP;
        }
    }
    (void)hipFree(d_x);
    (void)hipFree(d_bits);
    (void)hipFree(d_out);
    return rc;
}

}  // extern "C"
