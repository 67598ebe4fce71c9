// Probe of ds_read_b64_tr_b16 (gfx950): which lane supplies which address, which lane receives what.
// LDS image L[row][col] = row * 64 + col % 64, rows of 160 16-bit elements (320 bytes, the GEMM's m-major plane stride).  hipcc --offload-arch=gfx950 tools/probe_tr16.hip -o build/probe_tr16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, int variant) {
    __shared__ __attribute__((aligned(16))) short L[64 * 160];
    for (int i = threadIdx.x; i < 64 * 160; i += 64) L[i] = (short)((i / 160) * 64 + (i % 160) % 64);
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, li = lane & 15;
    int q, p;
    if (variant == 0) { q = li >> 2; p = li & 3; }      // lane 4q+p: row q, columns 4p..4p+3
    else { q = li & 3; p = li >> 2; }                  // lane q+4p
    const short* a = L + (4 * g + q) * 160 + 4 * p;
    typedef s16x4 __attribute__((address_space(3))) * lp;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)a);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = v[i];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * sizeof(short));
    short h[256];
    for (int variant = 0; variant < 2; ++variant) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, variant);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("variant %d (lane: row,col of each of its 4 elements)\n", variant);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int i = 0; i < 4; ++i) printf(" (%d,%2d)", h[l * 4 + i] / 64, h[l * 4 + i] % 64);
            printf("\n");
        }
    }
    return 0;
}
