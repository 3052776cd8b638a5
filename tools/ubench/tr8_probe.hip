// What ds_read_b64_tr_b8 delivers: every lane reads 8 bytes "transposed" within its 16-lane group.  LDS holds byte i = its own
// address (low byte in one run, high byte in a second), so the printed table says which LDS address each returned byte came from.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr8_probe.hip -o tools/ubench/tr8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned char *in, int *out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x;
    int addr = 0;
    if (mode == 0) addr = lane * 8;                                   // consecutive 8-byte pieces
    if (mode == 1) addr = (lane & 15) * 8 + (lane >> 4) * 512;        // 16-lane groups 512 bytes apart
    if (mode == 2) addr = (lane & 15) * 16 + (lane >> 4) * 1024;      // 16-byte row pitch inside a group
    if (mode == 3) addr = (lane & 7) * 16 + ((lane >> 3) & 1) * 8 + (lane >> 4) * 1024;
    v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i *)(lds + addr));
    out[2 * lane] = r[0];
    out[2 * lane + 1] = r[1];
}
int main() {
    std::vector<unsigned char> lo(8192), hi(8192);
    for (int i = 0; i < 8192; ++i) lo[i] = i & 0xFF, hi[i] = i >> 8;
    unsigned char *d_in;
    int *d_out;
    hipMalloc(&d_in, 8192);
    hipMalloc(&d_out, 128 * 4);
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<int> a(128), b(128);
        hipMemcpy(d_in, lo.data(), 8192, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out, mode);
        hipMemcpy(a.data(), d_out, 512, hipMemcpyDeviceToHost);
        hipMemcpy(d_in, hi.data(), 8192, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out, mode);
        hipMemcpy(b.data(), d_out, 512, hipMemcpyDeviceToHost);
        printf("mode %d: lane -> LDS addresses of its 8 returned bytes\n", mode);
        for (int lane = 0; lane < 64; ++lane) {
            printf("  lane %2d:", lane);
            for (int j = 0; j < 8; ++j) {
                const int l = (a[2 * lane + j / 4] >> (8 * (j % 4))) & 0xFF, h = (b[2 * lane + j / 4] >> (8 * (j % 4))) & 0xFF;
                printf(" %5d", h * 256 + l);
            }
            printf("\n");
        }
    }
    return 0;
}
