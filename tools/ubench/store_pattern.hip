// Store-pattern microbenchmark for the K4 epilogue (three f64 [N, M] outputs, N = 20000, M = 9984):
//   mode 0: the matrix-core epilogue's pattern -- a wave instruction writes 8 B per lane to 2 rows x 32 columns
//           (two 256-byte runs), a task = 256 permuted rows x 192 columns, 6 column tiles visited one after another
//   mode 1: the same bytes, 8 B per lane, 64 consecutive columns of ONE row per instruction (512-byte runs)
//   mode 2: 16 B per lane, 128 consecutive columns of one row per instruction (1 KiB runs)
// build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern ; run: ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#include <algorithm>

template <int MODE>
__global__ __launch_bounds__(512) void k_store(double *__restrict__ a, double *__restrict__ b, double *__restrict__ c, int64_t n,
                                               int64_t m, const int32_t *__restrict__ rowmap, int n_tasks, int n_ct,
                                               unsigned int *__restrict__ ctr) {
    __shared__ int slot_box;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lam = lane & 31, h = lane >> 5;
    for (;;) {
        if (tid == 0) slot_box = static_cast<int>(atomicAdd(ctr, 1u));
        __syncthreads();
        const int slot = slot_box;
        __syncthreads();
        if (slot >= n_tasks) break;
        const int g = slot / n_ct, ct = slot % n_ct;
        if (MODE == 0) {
            for (int s = 0; s < 6; ++s) {
                const int64_t col = (static_cast<int64_t>(ct) * 6 + s) * 32 + lam;
                for (int r = 0; r < 16; ++r) {
                    const int row = rowmap[g * 256 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
                    if (row < 0) continue;
                    const int64_t o = static_cast<int64_t>(row) * m + col;
                    a[o] = 1.0 + s;
                    b[o] = 2.0 + r;
                    c[o] = 3.0;
                }
            }
        } else if (MODE == 1) {
            for (int i = 0; i < 32; ++i) {
                const int row = rowmap[g * 256 + wave * 32 + i];
                if (row < 0) continue;
                for (int k = 0; k < 3; ++k) {
                    const int64_t o = static_cast<int64_t>(row) * m + static_cast<int64_t>(ct) * 192 + k * 64 + lane;
                    a[o] = 1.0 + k;
                    b[o] = 2.0 + i;
                    c[o] = 3.0;
                }
            }
        } else {
            for (int i = 0; i < 32; i += 2) {
                const int row0 = rowmap[g * 256 + wave * 32 + i], row1 = rowmap[g * 256 + wave * 32 + i + 1];
                if (row0 < 0 || row1 < 0) continue;
                const double2 v = make_double2(1.0 + i, 2.0);
                int64_t o = static_cast<int64_t>(row0) * m + static_cast<int64_t>(ct) * 192 + 2 * lane;
                *reinterpret_cast<double2 *>(a + o) = v;
                *reinterpret_cast<double2 *>(b + o) = v;
                *reinterpret_cast<double2 *>(c + o) = v;
                o = static_cast<int64_t>(h ? row1 : row0) * m + static_cast<int64_t>(ct) * 192 + 128 + 2 * lam;
                *reinterpret_cast<double2 *>(a + o) = v;
                *reinterpret_cast<double2 *>(b + o) = v;
                *reinterpret_cast<double2 *>(c + o) = v;
                o = static_cast<int64_t>(row1) * m + static_cast<int64_t>(ct) * 192 + 2 * lane;
                *reinterpret_cast<double2 *>(a + o) = v;
                *reinterpret_cast<double2 *>(b + o) = v;
                *reinterpret_cast<double2 *>(c + o) = v;
            }
        }
    }
}

// mode 3/4/5: the split form's k_hyp_emit pattern -- one block = 4 waves x 192 columns, UN rows per wave (no loop),
// a wave instruction writes 512 contiguous bytes of one row; NT = nontemporal stores
template <int UN, bool NT>
__global__ __launch_bounds__(256) void k_emit_like(double *__restrict__ a, double *__restrict__ b, double *__restrict__ c, int64_t m,
                                                   const int32_t *__restrict__ rowmap, int64_t n_grp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cc = lane & 31, hh = lane >> 5;
    const int64_t grp = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    if (grp >= n_grp) return;
    int row[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) row[i] = rowmap[static_cast<int64_t>(blockIdx.y) * UN + i];
#pragma unroll
    for (int i = 0; i < UN; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int64_t col = (grp * 6 + hh + 2 * j) * 32 + cc;
            const int64_t o = static_cast<int64_t>(row[i]) * m + (col < m ? col : 0);
            if (NT) {
                __builtin_nontemporal_store(1.0 + j, a + o);
                __builtin_nontemporal_store(2.0 + i, b + o);
                __builtin_nontemporal_store(3.0, c + o);
            } else {
                a[o] = 1.0 + j;
                b[o] = 2.0 + i;
                c[o] = 3.0;
            }
        }
}

// mode 7/8: mode 4 plus the emit kernel's reads (12 bytes per lane and row from a 0.4 GB count buffer, all requested
// before the first store); mode 8 reads the same 64 KB over and over (cache hits)
template <int UN, bool HOT>
__global__ __launch_bounds__(256) void k_emit_rw(double *__restrict__ a, double *__restrict__ b, double *__restrict__ c, int64_t m,
                                                 const int32_t *__restrict__ rowmap, int64_t n_grp, const unsigned int *__restrict__ cnt,
                                                 int64_t n_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cc = lane & 31, hh = lane >> 5;
    const int64_t grp = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    if (grp >= n_grp) return;
    int row[UN];
    uint3 w[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) row[i] = rowmap[static_cast<int64_t>(blockIdx.y) * UN + i];
#pragma unroll
    for (int i = 0; i < UN; ++i) {
        const int64_t u = HOT ? (row[i] & 127) : row[i];
        w[i] = *reinterpret_cast<const uint3 *>(cnt + ((HOT ? 0 : grp) * n_rows + u) * 96 + cc * 3);
    }
#pragma unroll
    for (int i = 0; i < UN; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int64_t col = (grp * 6 + hh + 2 * j) * 32 + cc;
            const int64_t o = static_cast<int64_t>(row[i]) * m + (col < m ? col : 0);
            const unsigned int x = j == 0 ? w[i].x : j == 1 ? w[i].y : w[i].z;
            a[o] = 1.0 + x;
            b[o] = 2.0 + i;
            c[o] = x > 100 ? 1.0 : 0.0;
        }
}

__global__ __launch_bounds__(256) void k_read_cnt(const uint4 *__restrict__ cnt, int64_t vecs, unsigned int *__restrict__ sink) {
    unsigned int acc = 0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < vecs; i += static_cast<int64_t>(gridDim.x) * 256) {
        const uint4 v = cnt[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0xdeadbeefu) *sink = acc;
}

__global__ void k_fill_cnt(unsigned int *__restrict__ cnt, int64_t words) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < words) cnt[i] = static_cast<unsigned int>(i & 15);
}

int main(int argc, char **argv) {
    const int64_t n = 20000, m = argc > 1 ? atoll(argv[1]) : 9984;
    const bool sorted_rows = argc > 2 && atoi(argv[2]);
    const int n_grp = (n + 255) / 256, n_ct = m / 192, n_tasks = n_grp * n_ct;
    double *a, *b, *c;
    hipMalloc(&a, n * m * 8);
    hipMalloc(&b, n * m * 8);
    hipMalloc(&c, n * m * 8);
    std::vector<int32_t> rowmap(n_grp * 256, -1);
    std::vector<int32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    for (int64_t i = 0; i < n; ++i) rowmap[i] = sorted_rows ? static_cast<int32_t>(i) : perm[i];
    printf("m = %lld, rows %s\n", (long long)m, sorted_rows ? "in order" : "shuffled");
    int32_t *d_map;
    unsigned int *d_ctr;
    hipMalloc(&d_map, rowmap.size() * 4);
    hipMalloc(&d_ctr, 4);
    hipMemcpy(d_map, rowmap.data(), rowmap.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int64_t n_g6 = (m + 191) / 192;
    unsigned int *d_cnt;
    hipMalloc(&d_cnt, n_g6 * n * 96 * 4);
    hipMemset(d_cnt, 0, n_g6 * n * 96 * 4);
    // chunked: the counts of a chunk of column groups are written right before the chunk is emitted (are they
    // still in the memory-side cache?); prints the emit time summed over the chunks
    for (int chunks : {1, 2, 4, 8, 16}) {
        const int64_t gpc = (n_g6 + chunks - 1) / chunks;
        float emit_ms = 0.f, fill_ms = 0.f;
        for (int ch = 0; ch < chunks; ++ch) {
            const int64_t g0 = ch * gpc, g1 = std::min<int64_t>(n_g6, g0 + gpc);
            if (g0 >= g1) break;
            const int64_t words = (g1 - g0) * n * 96;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_fill_cnt, dim3((words + 255) / 256), dim3(256), 0, 0, d_cnt + g0 * n * 96, words);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            fill_ms += ms;
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_emit_rw<4, false>), dim3((g1 - g0 + 3) / 4, n / 4), dim3(256), 0, 0, a + g0 * 192, b + g0 * 192, c + g0 * 192, m,
                               d_map, g1 - g0, d_cnt + g0 * n * 96, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            emit_ms += ms;
        }
        printf("chunks %2d: fill %.3f ms, emit %.3f ms\n", chunks, fill_ms, emit_ms);
    }
    {   // stores only on one stream, the 0.4 GB of reads by another kernel on a second stream, at the same time
        hipStream_t s1, s2;
        hipStreamCreate(&s1);
        hipStreamCreate(&s2);
        const int64_t vecs = n_g6 * n * 96 / 4;
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, s1);
            hipLaunchKernelGGL((k_emit_like<4, false>), dim3((n_g6 + 3) / 4, n / 4), dim3(256), 0, s1, a, b, c, m, d_map, n_g6);
            hipLaunchKernelGGL(k_read_cnt, dim3(64), dim3(256), 0, s2, reinterpret_cast<const uint4 *>(d_cnt), vecs, d_ctr);
            hipEventRecord(e1, s1);
            hipStreamSynchronize(s2);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("stores (stream 1) beside a 64-block read kernel (stream 2): stores took %.3f ms\n", ms);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, s2);
            hipLaunchKernelGGL(k_read_cnt, dim3(64), dim3(256), 0, s2, reinterpret_cast<const uint4 *>(d_cnt), vecs, d_ctr);
            hipEventRecord(e1, s2);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("read kernel alone: %.3f ms\n", ms);
        }
    }
    for (int mode = 0; mode < 9; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d_ctr, 0, 4);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_store<0>, dim3(256), dim3(512), 0, 0, a, b, c, n, m, d_map, n_tasks, n_ct, d_ctr);
            if (mode == 1) hipLaunchKernelGGL(k_store<1>, dim3(256), dim3(512), 0, 0, a, b, c, n, m, d_map, n_tasks, n_ct, d_ctr);
            if (mode == 2) hipLaunchKernelGGL(k_store<2>, dim3(256), dim3(512), 0, 0, a, b, c, n, m, d_map, n_tasks, n_ct, d_ctr);
            if (mode == 3) hipLaunchKernelGGL((k_emit_like<4, true>), dim3((n_g6 + 3) / 4, n / 4), dim3(256), 0, 0, a, b, c, m, d_map, n_g6);
            if (mode == 4) hipLaunchKernelGGL((k_emit_like<4, false>), dim3((n_g6 + 3) / 4, n / 4), dim3(256), 0, 0, a, b, c, m, d_map, n_g6);
            if (mode == 5) hipLaunchKernelGGL((k_emit_like<8, false>), dim3((n_g6 + 3) / 4, n / 8), dim3(256), 0, 0, a, b, c, m, d_map, n_g6);
            if (mode == 6) hipLaunchKernelGGL((k_emit_like<16, false>), dim3((n_g6 + 3) / 4, n / 16), dim3(256), 0, 0, a, b, c, m, d_map, n_g6);
            if (mode == 7) hipLaunchKernelGGL((k_emit_rw<4, false>), dim3((n_g6 + 3) / 4, n / 4), dim3(256), 0, 0, a, b, c, m, d_map, n_g6, d_cnt, n);
            if (mode == 8) hipLaunchKernelGGL((k_emit_rw<4, true>), dim3((n_g6 + 3) / 4, n / 4), dim3(256), 0, 0, a, b, c, m, d_map, n_g6, d_cnt, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d: %.3f ms, %.2f TB/s\n", mode, ms, 3.0 * n * m * 8 / ms * 1e-9);
        }
    }
    return 0;
}
