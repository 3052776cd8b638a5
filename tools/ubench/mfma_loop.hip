// What limits the k_permtest_mfma inner loop?  8 waves per CU, each wave per k-step: 6 ds_read_b128 of
// B operands + 12 VALU (A expansion) + 6 x v_mfma_i32_32x32x32_i8; a barrier every 4 k-steps.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int MODE>   // bit0: LDS reads, bit1: A expansion, bit2: barrier per 4 k-steps, bit3: prefetch reads one k-step ahead
__global__ __launch_bounds__(512) void k(int *out, int iters, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 49920 / 4; i += 512) ((unsigned *)lds)[i] = i * 2654435761u;
    __syncthreads();
    v16i acc[6];
    for (int s = 0; s < 6; ++s) for (int r = 0; r < 16; ++r) acc[s][r] = 0;
    const unsigned char *base = lds + (lane >> 5) * 512 + (lane & 31) * 16;
    unsigned aw = seed + tid;
    v4i b[6], bn[6];
    for (int s = 0; s < 6; ++s) b[s] = *(const v4i *)(base + s * 1040);
    for (int it = 0; it < iters; ++it) {
        const unsigned char *buf = base + (it & 1) * 24960;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (MODE & 8) {
#pragma unroll
                for (int s = 0; s < 6; ++s) bn[s] = *(const v4i *)(buf + ((k + 1) & 3) * 6240 + s * 1040);
                __builtin_amdgcn_sched_barrier(0);
            } else if (MODE & 1) {
#pragma unroll
                for (int s = 0; s < 6; ++s) b[s] = *(const v4i *)(buf + k * 6240 + s * 1040);
            }
            v4i a;
            if (MODE & 2) {
                aw = aw * 1664525u + 1013904223u;
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = (int)(__umul24(__builtin_amdgcn_ubfe(aw, 4 * q + 16 * (lane >> 5), 4u), 0x204081u) & 0x01010101u);
            } else a = b[0];
#pragma unroll
            for (int s = 0; s < 6; ++s) acc[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[s], acc[s], 0, 0, 0);
            if (MODE & 8) {
#pragma unroll
                for (int s = 0; s < 6; ++s) b[s] = bn[s];
            }
        }
        if (MODE & 4) __syncthreads();
    }
    int x = 0;
    for (int s = 0; s < 6; ++s) for (int r = 0; r < 16; ++r) x ^= acc[s][r];
    out[blockIdx.x * 512 + tid] = x;
}
template <int MODE> void run(int *o, const char *what) {
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 49920);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 49920, 0, o, 10, 1u); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 49920, 0, o, iters, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = 256.0 * 8 * iters * 24;
    printf("%-60s %.2f ms  %.0f TOPS  %.1f clk@2.4GHz per MFMA per SIMD\n", what, ms, mf * 65536 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024));
}
int main() {
    int *o; hipMalloc(&o, 256 * 512 * 4);
    run<0>(o, "MFMA only");
    run<2>(o, "MFMA + A expansion");
    run<1>(o, "MFMA + LDS reads (just in time)");
    run<9>(o, "MFMA + LDS reads (one k-step ahead)");
    run<3>(o, "MFMA + LDS reads (jit) + A expansion");
    run<11>(o, "MFMA + LDS reads (ahead) + A expansion");
    run<7>(o, "MFMA + LDS jit + A + barrier/4 k-steps");
    run<15>(o, "MFMA + LDS ahead + A + barrier/4 k-steps");
    return 0;
}
