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
    for (int mode = 0; mode < 7; ++mode) {
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
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d: %.3f ms, %.2f TB/s\n", mode, ms, 3.0 * n * m * 8 / ms * 1e-9);
        }
    }
    return 0;
}
