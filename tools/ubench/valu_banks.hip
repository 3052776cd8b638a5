// Does the source-register bank pattern of a three-operand VALU instruction change its issue rate on gfx950?
// Eight independent v_bitop3_b32 per group, destinations v0..v7, sources from v8..v31 chosen either from three different banks
// (register number mod 4) or all from one bank.  Prints wave-instructions per SIMD and microsecond at 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define G_DIFF \
    "v_bitop3_b32 v0, v8, v9, v10 bitop3:0x96\n v_bitop3_b32 v1, v11, v12, v13 bitop3:0xe8\n" \
    "v_bitop3_b32 v2, v14, v15, v16 bitop3:0x96\n v_bitop3_b32 v3, v17, v18, v19 bitop3:0xe8\n" \
    "v_bitop3_b32 v4, v20, v21, v22 bitop3:0x96\n v_bitop3_b32 v5, v23, v24, v25 bitop3:0xe8\n" \
    "v_bitop3_b32 v6, v26, v27, v28 bitop3:0x96\n v_bitop3_b32 v7, v29, v30, v31 bitop3:0xe8\n"
#define G_SAME \
    "v_bitop3_b32 v0, v8, v12, v16 bitop3:0x96\n v_bitop3_b32 v1, v9, v13, v17 bitop3:0xe8\n" \
    "v_bitop3_b32 v2, v10, v14, v18 bitop3:0x96\n v_bitop3_b32 v3, v11, v15, v19 bitop3:0xe8\n" \
    "v_bitop3_b32 v4, v20, v24, v28 bitop3:0x96\n v_bitop3_b32 v5, v21, v25, v29 bitop3:0xe8\n" \
    "v_bitop3_b32 v6, v22, v26, v30 bitop3:0x96\n v_bitop3_b32 v7, v23, v27, v31 bitop3:0xe8\n"
#define G_TWO \
    "v_bitop3_b32 v0, v8, v12, v9 bitop3:0x96\n v_bitop3_b32 v1, v9, v13, v10 bitop3:0xe8\n" \
    "v_bitop3_b32 v2, v10, v14, v11 bitop3:0x96\n v_bitop3_b32 v3, v11, v15, v16 bitop3:0xe8\n" \
    "v_bitop3_b32 v4, v20, v24, v21 bitop3:0x96\n v_bitop3_b32 v5, v21, v25, v22 bitop3:0xe8\n" \
    "v_bitop3_b32 v6, v22, v26, v23 bitop3:0x96\n v_bitop3_b32 v7, v23, v27, v28 bitop3:0xe8\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"
template <int W>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters) {
    for (int i = 0; i < iters; ++i) {
        if (W == 0) asm volatile(G_DIFF G_DIFF G_DIFF G_DIFF G_DIFF G_DIFF G_DIFF G_DIFF ::: CLOB);
        if (W == 1) asm volatile(G_SAME G_SAME G_SAME G_SAME G_SAME G_SAME G_SAME G_SAME ::: CLOB);
        if (W == 2) asm volatile(G_TWO G_TWO G_TWO G_TWO G_TWO G_TWO G_TWO G_TWO ::: CLOB);
    }
    if (iters < 0) out[threadIdx.x] = 1;
}
int main() {
    unsigned *o; hipMalloc(&o, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[3] = {"three banks", "one bank", "two banks"};
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int w = 0; w < 3; ++w) {
            const int blocks = 256 * wps, iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (w == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, o, iters);
                if (w == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, o, iters);
                if (w == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, o, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per_simd = (double)wps * iters * 64;
            printf("%-12s waves/SIMD=%d: %.3f ms, %.1f wave-instructions per SIMD and us\n", names[w], wps, ms, per_simd / (ms * 1e3));
        }
    return 0;
}
