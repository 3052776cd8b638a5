// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the permutation kernels
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports 1/2 of a wide coalesced read, "other access widths and WRITE_SIZE are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Four kernels over buffers far larger than
// the 256 MB Infinity Cache, each with a known byte count:
//   k_read16    16 B per lane, a wave reads 1 KiB contiguous            (the blocked member-id lists; slice rows of the MFMA form)
//   k_read8      8 B per lane                                          (the attribute bit words T)
//   k_read4 / k_read12   4 / 12 B per lane                              (counter reads; the packed u16 counts of k_hyp_emit)
//   k_gather192 random 192-byte rows, 16 B per lane                    (the matrix-core kernel's row gather)
//   k_atomic4   atomicAdd of 4 B per lane, a wave updates 256 B        (the <= / >= counter flush)
//   k_write16   16 B per lane stores                                   (the streaming outputs)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, tools/pmc_calib.sh); the ratio
// known bytes / reported bytes per kernel goes to profiles/pmc_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_read16(const uint4 *__restrict__ src, size_t n16, unsigned *__restrict__ out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read8(const uint2 *__restrict__ src, size_t n8, unsigned *__restrict__ out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint2 v = src[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read4(const unsigned *__restrict__ src, size_t n4, unsigned *__restrict__ out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc ^= src[i];
    if (acc == 0x12345678u) out[0] = acc;
}
struct u3 { unsigned x, y, z; };
__global__ __launch_bounds__(256) void k_read12(const u3 *__restrict__ src, size_t n12, unsigned *__restrict__ out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n12; i += (size_t)gridDim.x * 256) {
        const u3 v = src[i];
        acc ^= v.x ^ v.y ^ v.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// gathers of 192-byte rows, 16 B per lane, 12 lanes per row, rows in random order (the matrix-core kernel's slice rows)
__global__ __launch_bounds__(192) void k_gather192(const uint4 *__restrict__ src, size_t n_rows, unsigned *__restrict__ out) {
    unsigned acc = 0;
    const int part = threadIdx.x % 12, sub = threadIdx.x / 12;
    for (size_t i = (size_t)blockIdx.x * 16 + sub; i < n_rows; i += (size_t)gridDim.x * 16) {
        const size_t row = (i * 2654435761ull) % n_rows;
        const uint4 v = src[row * 12 + part];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_atomic4(unsigned *__restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) atomicAdd(dst + i, 1u + (unsigned)(i & 3));
}
__global__ __launch_bounds__(256) void k_write16(uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}

int main() {
    const size_t bytes = (size_t)2 << 30;                 // 2 GiB: 8x the Infinity Cache
    void *buf = nullptr, *out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 8;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_read16, dim3(grid), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, (unsigned *)out);
        hipLaunchKernelGGL(k_read8, dim3(grid), dim3(256), 0, 0, (const uint2 *)buf, bytes / 8, (unsigned *)out);
        hipLaunchKernelGGL(k_read4, dim3(grid), dim3(256), 0, 0, (const unsigned *)buf, bytes / 4, (unsigned *)out);
        hipLaunchKernelGGL(k_read12, dim3(grid), dim3(256), 0, 0, (const u3 *)buf, bytes / 12, (unsigned *)out);
        hipLaunchKernelGGL(k_gather192, dim3(grid), dim3(192), 0, 0, (const uint4 *)buf, bytes / 192, (unsigned *)out);
        hipLaunchKernelGGL(k_atomic4, dim3(grid), dim3(256), 0, 0, (unsigned *)buf, bytes / 4);
        hipLaunchKernelGGL(k_write16, dim3(grid), dim3(256), 0, 0, (uint4 *)buf, bytes / 16);
        hipDeviceSynchronize();
    }
    printf("bytes per kernel launch: %zu (read16, read8: read; atomic4: read + written; write16: written)\n", bytes);
    return 0;
}
