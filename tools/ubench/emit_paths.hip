// Which path should the packed counts of k_hyp_emit take to the storing waves?  (K4, 20 000 x 9 984: three f64 outputs = 4.8 GB of
// stores, 0.38 GB of counts read; vmcnt is one in-order counter for loads and stores.)
// A workgroup = 8 waves = 8 column groups of 192, a task = R consecutive row positions (counts are read in row-position
// order, the stores go to shuffled rows as in the kernel), values come from a 45 KB table slab in LDS indexed by the count.
//   mode 0: no count reads (the floor of this shape)
//   mode 1: vector loads, all R rows requested before the first store (3 dwords per lane and row)
//   mode 2: vector loads, software-pipelined two rows ahead of the stores (the kernel's structure)
//   mode 3: scalar loads (s_load_dwordx16, counted by lgkmcnt, not vmcnt) + v_writelane distribution to the lanes
//   mode 4: scalar loads, values used wave-uniformly (what the scalar path costs without the distribution)
//   mode 5: mode 1 reading the same 48 KB over and over (cache hits: is it the HBM read/write mix?)
//   mode 7: mode 2 with all reads folded into a window of W MB (where does the penalty stop: L2 4 MB per XCD, memory-side cache 256 MB)
//   chunks: the counts of a chunk of column groups are written by a kernel right before the chunk is emitted
//   mode 6: one row PAIR per 3-dword load (lanes 0-31 row 2k, lanes 32-63 row 2k+1), halves exchanged in registers
// build: hipcc --offload-arch=gfx950 -O3 emit_paths.hip -o emit_paths ; run: ./emit_paths
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

typedef const unsigned int __attribute__((address_space(4))) *const_u32;
struct Chunk16 { unsigned int v[16]; };
typedef const Chunk16 __attribute__((address_space(4))) *const_chunk;

__device__ __forceinline__ unsigned int write_lane(unsigned int sval, int lane, unsigned int old) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(sval), "n"(lane));
    return old;
}

constexpr int SLAB = 2880;                                                   // double2 entries = 45 KB

template <int MODE, int R>
__global__ __launch_bounds__(512) void k_emit(double *__restrict__ a, double *__restrict__ b, double *__restrict__ c, int64_t m,
                                              const int32_t *__restrict__ rowmap, int64_t n_grp, const unsigned int *__restrict__ cnt,
                                              int64_t n_rows, const double2 *__restrict__ tab, int64_t window_rows = 0) {
    __shared__ double2 slab[SLAB];
    for (int e = threadIdx.x; e < SLAB; e += 512) slab[e] = tab[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cc = lane & 31, hh = lane >> 5;
    const int64_t grp = static_cast<int64_t>(blockIdx.x) * 8 + wave;
    if (grp >= n_grp) return;
    const int64_t pos0 = static_cast<int64_t>(blockIdx.y) * R;
    int64_t col[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        col[j] = (grp * 6 + hh + 2 * j) * 32 + cc;
        if (col[j] >= m) col[j] = 0;
    }
    const unsigned int kofs = lane % 45;
    auto emit = [&](int i, unsigned int x0, unsigned int x1, unsigned int x2) __attribute__((always_inline)) {
        const int64_t o = static_cast<int64_t>(rowmap[pos0 + i]) * m;
        const unsigned int x[3] = {x0, x1, x2};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double2 v = slab[(x[j] & 63u) * 45u + kofs];
            __builtin_nontemporal_store(v.x, a + o + col[j]);
            __builtin_nontemporal_store(v.y, b + o + col[j]);
            __builtin_nontemporal_store(v.x < 0.05 ? 1.0 : 0.0, c + o + col[j]);
        }
    };
    const unsigned int *src = cnt + ((MODE == 5 ? 0 : grp * n_rows) + (MODE == 5 ? (pos0 & 127) : pos0)) * 96;
    if (MODE == 7) src = cnt + ((grp * n_rows + pos0) % window_rows) * 96;       // the whole launch reads a window of this many rows
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < R; ++i) emit(i, lane + i, lane + 2 * i, lane ^ i);
    } else if (MODE == 1 || MODE == 5) {
        unsigned int w[R][3];
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) w[i][q] = src[i * 96 + cc * 3 + q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < R; ++i)
            emit(i, hh ? w[i][0] >> 16 : w[i][0] & 0xffffu, hh ? w[i][1] >> 16 : w[i][1] & 0xffffu, hh ? w[i][2] >> 16 : w[i][2] & 0xffffu);
    } else if (MODE == 2 || MODE == 7) {
        unsigned int w[2][2][3];
        auto load2 = [&](int i0, unsigned int (&ww)[2][3]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) ww[i][q] = src[min(i0 + i, R - 1) * 96 + cc * 3 + q];
        };
        load2(0, w[0]);
#pragma unroll
        for (int i0 = 0; i0 < R; i0 += 2) {
            const int cur = (i0 >> 1) & 1;
            load2(i0 + 2, w[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                emit(i0 + i, hh ? w[cur][i][0] >> 16 : w[cur][i][0] & 0xffffu, hh ? w[cur][i][1] >> 16 : w[cur][i][1] & 0xffffu,
                     hh ? w[cur][i][2] >> 16 : w[cur][i][2] & 0xffffu);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (MODE == 3 || MODE == 4) {
        // row layout for this path: dword q * 32 + c belongs to lane c's q-th pair of columns
#if defined(__HIP_DEVICE_COMPILE__)
        const_chunk sp = (const_chunk)(src);
#pragma unroll 2
        for (int i = 0; i < R; ++i) {
            unsigned int w[3] = {0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const Chunk16 d = sp[i * 6 + k];
                if (MODE == 3) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) w[k >> 1] = write_lane(d.v[t], (k & 1) * 16 + t, w[k >> 1]);
                } else {
                    unsigned int acc = 0;
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc ^= d.v[t];
                    w[k >> 1] += acc + lane;
                }
            }
            if (MODE == 3) {
#pragma unroll
                for (int q = 0; q < 3; ++q) w[q] = __builtin_amdgcn_ds_bpermute(cc * 4, w[q]);
            }
            emit(i, hh ? w[0] >> 16 : w[0] & 0xffffu, hh ? w[1] >> 16 : w[1] & 0xffffu, hh ? w[2] >> 16 : w[2] & 0xffffu);
        }
#endif
    } else if (MODE == 6) {
        static_assert(R % 2 == 0, "pairs");
        unsigned int w[R / 2][3];
#pragma unroll
        for (int p = 0; p < R / 2; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) w[p][q] = src[(2 * p + hh) * 96 + cc * 3 + q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < R / 2; ++p) {
            unsigned int x0[3], x1[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned int other = __builtin_amdgcn_ds_bpermute((lane ^ 32) * 4, w[p][q]);
                const unsigned int r0 = hh ? other : w[p][q], r1 = hh ? w[p][q] : other;      // dwords of row 2p / 2p + 1
                x0[q] = hh ? r0 >> 16 : r0 & 0xffffu;
                x1[q] = hh ? r1 >> 16 : r1 & 0xffffu;
            }
            emit(2 * p, x0[0], x0[1], x0[2]);
            emit(2 * p + 1, x1[0], x1[1], x1[2]);
        }
    }
}

// takes a CU whole (LDS) and spins: what is left of the emit rate when C CUs are busy with the matrix-core kernel?
__global__ __launch_bounds__(512) void k_hog(long long ticks, unsigned int *__restrict__ sink) {
    extern __shared__ unsigned int hog_lds[];
    const long long t0 = wall_clock64();
    unsigned int acc = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        for (int i = 0; i < 64; ++i) acc = acc * 1664525u + 1013904223u;
        hog_lds[threadIdx.x] = acc;
    }
    if (acc == 0x12345u) *sink = hog_lds[(threadIdx.x + 1) & 511];
}

__global__ void k_fill(unsigned int *__restrict__ cnt, int64_t words) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < words) cnt[i] = static_cast<unsigned int>((i * 2654435761u) >> 7);
}

template <int MODE, int R>
void run(const char *what, double *a, double *b, double *c, int64_t n, int64_t m, const int32_t *d_map, const unsigned int *d_cnt,
         const double2 *d_tab) {
    const int64_t n_g6 = (m + 191) / 192, tasks = n / R;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f;
    const int reps = 6;
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_emit<MODE, R>), dim3((n_g6 + 7) / 8, tasks), dim3(512), 0, 0, a, b, c, m, d_map, n_g6, d_cnt, n, d_tab);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = std::min(best, ms), sum += ms;
    }
    const double bytes = 3.0 * tasks * R * m * 8;
    printf("mode %d R %2d %-44s best %.3f ms  mean %.3f ms  (%.2f TB/s of stores at best; scaled to 20 000 rows: %.3f ms)\n", MODE, R, what, best,
           sum / (reps - 1), bytes / best * 1e-9, best * 20000.0 / (tasks * R));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main(int argc, char **argv) {
    const int64_t n = 20000, m = argc > 1 ? atoll(argv[1]) : 9984;
    double *a, *b, *c;
    hipMalloc(&a, n * m * 8);
    hipMalloc(&b, n * m * 8);
    hipMalloc(&c, n * m * 8);
    std::vector<int32_t> perm(n + 64);
    std::iota(perm.begin(), perm.begin() + n, 0);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.begin() + n, rng);
    int32_t *d_map;
    hipMalloc(&d_map, perm.size() * 4);
    hipMemcpy(d_map, perm.data(), perm.size() * 4, hipMemcpyHostToDevice);
    const int64_t n_g6 = (m + 191) / 192, words = n_g6 * (n + 64) * 96;
    unsigned int *d_cnt;
    hipMalloc(&d_cnt, words * 4);
    hipLaunchKernelGGL(k_fill, dim3((words + 255) / 256), dim3(256), 0, 0, d_cnt, words);
    std::vector<double2> tab(SLAB);
    for (int i = 0; i < SLAB; ++i) tab[i] = make_double2((i % 97) / 97.0, i * 0.25);
    double2 *d_tab;
    hipMalloc(&d_tab, SLAB * sizeof(double2));
    hipMemcpy(d_tab, tab.data(), SLAB * sizeof(double2), hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int64_t tasks = n / 24;
        for (int mb : {1, 4, 16, 32, 64, 128, 192, 256, 384}) {
            const int64_t window_rows = std::min<int64_t>(static_cast<int64_t>(mb) * (1 << 20) / 384, n_g6 * n);
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL((k_emit<7, 24>), dim3((n_g6 + 7) / 8, tasks), dim3(512), 0, 0, a, b, c, m, d_map, n_g6, d_cnt, n, d_tab, window_rows);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep) best = std::min(best, ms);
            }
            printf("mode 7: reads folded into %3d MB: best %.3f ms\n", mb, best);
        }
        {
            hipStream_t sa, sb;
            hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
            hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
            unsigned int *d_sink;
            hipMalloc(&d_sink, 4);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_hog), hipFuncAttributeMaxDynamicSharedMemorySize, 120 << 10);
            for (int hog : {0, 32, 64, 96, 128, 160, 192, 224}) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipDeviceSynchronize();
                    if (hog) hipLaunchKernelGGL(k_hog, dim3(hog), dim3(512), 120 << 10, sa, 200000ll, d_sink);      // 2 ms at 100 MHz
                    hipEventRecord(e0, sb);
                    hipLaunchKernelGGL((k_emit<2, 24>), dim3((n_g6 + 7) / 8, tasks), dim3(512), 0, sb, a, b, c, m, d_map, n_g6, d_cnt, n, d_tab, 0);
                    hipEventRecord(e1, sb);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep) best = std::min(best, ms);
                    hipDeviceSynchronize();
                }
                printf("emit (mode 2) beside %3d busy CUs: best %.3f ms\n", hog, best);
            }
        }
        for (int chunks : {1, 2, 4, 7, 13, 26, 52}) {
            const int64_t gpc = (n_g6 + chunks - 1) / chunks;
            float best = 1e9f, best_fill = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                float emit_ms = 0.f, fill_ms = 0.f;
                for (int ch = 0; ch < chunks; ++ch) {
                    const int64_t g0 = ch * gpc, g1 = std::min<int64_t>(n_g6, g0 + gpc);
                    if (g0 >= g1) break;
                    const int64_t cw = (g1 - g0) * n * 96;
                    float ms;
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k_fill, dim3((cw + 255) / 256), dim3(256), 0, 0, d_cnt + g0 * n * 96, cw);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                    fill_ms += ms;
                    hipEventRecord(e0);
                    hipLaunchKernelGGL((k_emit<2, 24>), dim3((g1 - g0 + 7) / 8, tasks), dim3(512), 0, 0, a + g0 * 192, b + g0 * 192, c + g0 * 192, m, d_map,
                                       g1 - g0, d_cnt + g0 * n * 96, n, d_tab, 0);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                    emit_ms += ms;
                }
                if (rep) best = std::min(best, emit_ms), best_fill = std::min(best_fill, fill_ms);
            }
            printf("chunks %2d (%.0f MB of counts each): fill %.3f ms, emit %.3f ms\n", chunks, gpc * n * 384.0 / 1e6, best_fill, best);
        }
    }
    for (int pass = 0; pass < 1; ++pass) {
        run<0, 24>("stores only", a, b, c, n, m, d_map, d_cnt, d_tab);
        run<0, 8>("stores only", a, b, c, n, m, d_map, d_cnt, d_tab);
        run<1, 8>("vector loads up front", a, b, c, n, m, d_map, d_cnt, d_tab);
        run<1, 16>("vector loads up front", a, b, c, n, m, d_map, d_cnt, d_tab);
        run<1, 24>("vector loads up front", a, b, c, n, m, d_map, d_cnt, d_tab);
        run<2, 24>("vector loads two rows ahead", a, b, c, n, m, d_map, d_cnt, d_tab);
        run<3, 24>("scalar loads + writelane", a, b, c, n, m, d_map, d_cnt, d_tab);
        run<4, 24>("scalar loads, uniform use", a, b, c, n, m, d_map, d_cnt, d_tab);
        run<5, 24>("vector loads up front, cache hits", a, b, c, n, m, d_map, d_cnt, d_tab);
        run<6, 24>("row-pair loads up front", a, b, c, n, m, d_map, d_cnt, d_tab);
        run<6, 32>("row-pair loads up front", a, b, c, n, m, d_map, d_cnt, d_tab);
        run<6, 48>("row-pair loads up front", a, b, c, n, m, d_map, d_cnt, d_tab);
    }
    return 0;
}
