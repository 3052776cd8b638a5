#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void ktr(unsigned short *out, int stride) {
    __shared__ unsigned short lds[8192];
    // byte at offset o encodes o (16-bit id per byte pair is not possible; use two passes: low and high)
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = 0;
    __syncthreads();
    unsigned char *b = (unsigned char *)lds;
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = threadIdx.x; i < 16384; i += 64) b[i] = pass ? (unsigned char)(i >> 8) : (unsigned char)(i & 255);
        __syncthreads();
        v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i *)(b + threadIdx.x * stride));
        ((v2i *)out)[pass * 64 + threadIdx.x] = r;
        __syncthreads();
    }
}
__global__ void kmfma(const signed char *A, const signed char *B, int *C) {
    // A [32][32] row-major (row, k); B [32][32] (k, col) row-major; C [32][32]
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    v4i a, b;
    signed char ta[16], tb[16];
    for (int q = 0; q < 16; ++q) { ta[q] = A[i * 32 + 16 * h + q]; tb[q] = B[(16 * h + q) * 32 + i]; }
    memcpy(&a, ta, 16); memcpy(&b, tb, 16);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * h; C[row * 32 + i] = c[r]; }
}
int main() {
    unsigned short *d; hipMalloc(&d, 2048);
    unsigned char h[2048];
    for (int stride : {8, 16, 64}) {
        hipLaunchKernelGGL(ktr, dim3(1), dim3(64), 0, 0, d, stride);
        hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
        printf("ds_read_tr8_b64, lane address = lane*%d: result bytes as source byte offsets\n", stride);
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int b = 0; b < 8; ++b) printf(" %5d", h[l * 8 + b] + 256 * h[512 + l * 8 + b]); printf("\n"); }
    }
    signed char A[1024], B[1024]; int C[1024], R[1024];
    srand(1); for (int i = 0; i < 1024; ++i) { A[i] = rand() % 255 - 127; B[i] = rand() % 255 - 127; }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 32 + j]; R[i * 32 + j] = s; }
    signed char *dA, *dB; int *dC; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kmfma, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(C, dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += C[i] != R[i];
    printf("mfma_i32_32x32x32_i8 layout check: %d mismatches of 1024\n", bad);
    return 0;
}
