// integer VALU issue-rate microbenchmark: wave-instructions per cycle per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters) {
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7, e = a * 5, f = a + 11, g = a ^ 13, h = a + 17;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); b = __builtin_amdgcn_bitop3_b32(b, c, d, 0xE8);
            c = __builtin_amdgcn_bitop3_b32(c, d, e, 0x96); d = __builtin_amdgcn_bitop3_b32(d, e, f, 0xE8);
            e = __builtin_amdgcn_bitop3_b32(e, f, g, 0x96); f = __builtin_amdgcn_bitop3_b32(f, g, h, 0xE8);
            g = __builtin_amdgcn_bitop3_b32(g, h, a, 0x96); h = __builtin_amdgcn_bitop3_b32(h, a, b, 0xE8);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
__global__ __launch_bounds__(256) void kx(unsigned *out, int iters) {
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7, e = a * 5, f = a + 11, g = a ^ 13, h = a + 17;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a ^= b; b += c; c ^= d; d += e; e ^= f; f += g; g ^= h; h += a;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
int main() {
    unsigned *o; hipMalloc(&o, 256 * 2048 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc = 1; wpc <= 8; wpc *= 2) {   // workgroups (of 4 waves) per CU
        for (int which = 0; which < 2; ++which) {
            int blocks = 256 * wpc, iters = 20000;
            if (which) hipLaunchKernelGGL(kx, dim3(blocks), dim3(256), 0, 0, o, 10); else hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, o, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (which) hipLaunchKernelGGL(kx, dim3(blocks), dim3(256), 0, 0, o, iters); else hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double winstr = (double)blocks * 4 * iters * 16 * 8;
            printf("%s waves/SIMD=%d: %.3f ms, %.3g wave-instr/s, per SIMD per clk @2.4GHz: %.3f\n", which ? "xor/add" : "bitop3", wpc, ms, winstr / ms * 1e3, winstr / (ms * 1e-3) / (1024 * 2.4e9));
        }
    }
    return 0;
}
