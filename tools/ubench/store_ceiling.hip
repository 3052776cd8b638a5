// Pure streaming-store ceiling of the device: what fraction of the 8 TB/s HBM peak plain stores reach, by shape.
//   one array of 3.2 GB (K1's int64 [20000, 20000]) and three arrays of 1.6 GB (K4's f64 [20000, 10000] x 3),
//   16 B or 8 B per lane, default / nontemporal stores, linear sweep (a wave writes 1 KiB runs, workgroups walk
//   the array in order) with 256 / 1024 / 4096 workgroups.
// build: hipcc --offload-arch=gfx950 -O3 store_ceiling.hip -o store_ceiling ; run: ./store_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int BYTES, bool NT>
__global__ __launch_bounds__(256) void k_fill(void *__restrict__ p0, void *__restrict__ p1, void *__restrict__ p2, size_t bytes_each, int n_arr) {
    const size_t per_iter = static_cast<size_t>(gridDim.x) * 256 * BYTES;
    for (size_t off = (static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x) * BYTES; off < bytes_each; off += per_iter) {
        for (int a = 0; a < n_arr; ++a) {
            char *base = static_cast<char *>(a == 0 ? p0 : a == 1 ? p1 : p2) + off;
            if (BYTES == 16) {
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 v = {1.0, 2.0};
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(base));
                else *reinterpret_cast<d2 *>(base) = v;
            } else {
                if (NT) __builtin_nontemporal_store(3.0, reinterpret_cast<double *>(base));
                else *reinterpret_cast<double *>(base) = 3.0;
            }
        }
    }
}

// row-tiled like k_euclid_dense: block = 512 columns x ROWS rows of an [n, n] int64 matrix
template <int ROWS, bool NT>
__global__ __launch_bounds__(256) void k_tiles(long long *__restrict__ out, long long n) {
    const long long j = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 2;
    const long long i0 = static_cast<long long>(blockIdx.y) * ROWS;
    if (j >= n) return;
    typedef long long l2 __attribute__((ext_vector_type(2)));
    for (long long i = i0; i < i0 + ROWS && i < n; ++i) {
        l2 v = {i & 1, j & 1};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<l2 *>(out + i * n + j));
        else *reinterpret_cast<l2 *>(out + i * n + j) = v;
    }
}

// K4-like row sweep: a workgroup writes one whole row (m f64 columns) of three arrays, then the row `gridDim.x` further on;
// 8 B per lane, a wave covers 512 contiguous bytes per instruction (the emit kernel's lane layout)
template <bool NT>
__global__ __launch_bounds__(256) void k_rows3(double *__restrict__ a, double *__restrict__ b, double *__restrict__ c, long long n, long long m) {
    for (long long r = blockIdx.x; r < n; r += gridDim.x) {
        for (long long j = threadIdx.x; j < m; j += 256) {
            const long long o = r * m + j;
            if (NT) {
                __builtin_nontemporal_store(1.0, a + o);
                __builtin_nontemporal_store(2.0, b + o);
                __builtin_nontemporal_store(3.0, c + o);
            } else {
                a[o] = 1.0;
                b[o] = 2.0;
                c[o] = 3.0;
            }
        }
    }
}

template <typename F>
static double time_ms(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    const size_t big = 3200000000ull, each = 1600000000ull;
    void *p0, *p1, *p2;
    if (hipMalloc(&p0, big) != hipSuccess || hipMalloc(&p1, each) != hipSuccess || hipMalloc(&p2, each) != hipSuccess) return 1;
    for (int grid : {256, 1024, 4096, 16384}) {
        double t;
        t = time_ms([&] { hipLaunchKernelGGL((k_fill<16, false>), dim3(grid), dim3(256), 0, 0, p0, p1, p2, big, 1); });
        printf("1 x 3.2 GB, 16 B/lane, default, grid %5d: %.3f ms = %.2f TB/s\n", grid, t, big / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((k_fill<16, true>), dim3(grid), dim3(256), 0, 0, p0, p1, p2, big, 1); });
        printf("1 x 3.2 GB, 16 B/lane, nt     , grid %5d: %.3f ms = %.2f TB/s\n", grid, t, big / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((k_fill<8, true>), dim3(grid), dim3(256), 0, 0, p0, p1, p2, big, 1); });
        printf("1 x 3.2 GB,  8 B/lane, nt     , grid %5d: %.3f ms = %.2f TB/s\n", grid, t, big / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((k_fill<16, true>), dim3(grid), dim3(256), 0, 0, p0, p1, p2, each, 3); });
        printf("3 x 1.6 GB, 16 B/lane, nt     , grid %5d: %.3f ms = %.2f TB/s\n", grid, t, 3.0 * each / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((k_fill<8, true>), dim3(grid), dim3(256), 0, 0, p0, p1, p2, each, 3); });
        printf("3 x 1.6 GB,  8 B/lane, nt     , grid %5d: %.3f ms = %.2f TB/s\n", grid, t, 3.0 * each / t / 1e9);
    }
    for (int grid : {256, 512, 1024, 2048}) {
        const long long rn = 20000, rm = 10000;
        double tr = time_ms([&] { hipLaunchKernelGGL((k_rows3<true>), dim3(grid), dim3(256), 0, 0, (double *)p0, (double *)p1, (double *)p2, rn, rm); });
        printf("rows x 3 arrays (20000 x 10000 f64), 8 B/lane nt, grid %5d: %.3f ms = %.2f TB/s\n", grid, tr, 3.0 * 8 * rn * rm / tr / 1e9);
        tr = time_ms([&] { hipLaunchKernelGGL((k_rows3<false>), dim3(grid), dim3(256), 0, 0, (double *)p0, (double *)p1, (double *)p2, rn, rm); });
        printf("rows x 3 arrays (20000 x 10000 f64), 8 B/lane   , grid %5d: %.3f ms = %.2f TB/s\n", grid, tr, 3.0 * 8 * rn * rm / tr / 1e9);
    }
    const long long n = 20000;
    double t = time_ms([&] { hipLaunchKernelGGL((k_tiles<32, false>), dim3((n + 511) / 512, (n + 31) / 32), dim3(256), 0, 0, (long long *)p0, n); });
    printf("tiles 512 x 32 default: %.3f ms = %.2f TB/s\n", t, 8.0 * n * n / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((k_tiles<32, true>), dim3((n + 511) / 512, (n + 31) / 32), dim3(256), 0, 0, (long long *)p0, n); });
    printf("tiles 512 x 32 nt     : %.3f ms = %.2f TB/s\n", t, 8.0 * n * n / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((k_tiles<8, true>), dim3((n + 511) / 512, (n + 7) / 8), dim3(256), 0, 0, (long long *)p0, n); });
    printf("tiles 512 x  8 nt     : %.3f ms = %.2f TB/s\n", t, 8.0 * n * n / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((k_tiles<128, true>), dim3((n + 511) / 512, (n + 127) / 128), dim3(256), 0, 0, (long long *)p0, n); });
    printf("tiles 512 x 128 nt    : %.3f ms = %.2f TB/s\n", t, 8.0 * n * n / t / 1e9);
    return 0;
}
