// ds_read_b64_tr_b16 with the block's base in the VGPR address vs in the instruction's offset field.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, int variant) {
    __shared__ __attribute__((aligned(16))) short L[80 * 160];
    for (int i = threadIdx.x; i < 80 * 160; i += 64) L[i] = (short)((i / 160) * 64 + (i % 160) % 64);
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    // rows 57.. (byte offset 57*320 = 18240 = the immediate)
    typedef s16x4 __attribute__((address_space(3))) * lp;
    const int rowbase = variant == 0 ? 0 : 57;
    const short* a = L + (rowbase + 4 * g + q) * 160 + 4 * p;
    s16x4 v;
    if (variant == 0) v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(L + (4 * g + q) * 160 + 4 * p));
    else v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(L + 57 * 160 + (4 * g + q) * 160 + 4 * p));
    (void)a;
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = v[i];
}
int main() {
    short* d; (void)hipMalloc(&d, 64 * 4 * sizeof(short));
    short h[256];
    for (int variant = 0; variant < 2; ++variant) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, variant);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("variant %d (%s): lane: (row - 57, col) of its 4 elements; expect (4g..4g+3, lane&15)\n", variant, variant ? "offset field" : "VGPR base");
        for (int l = 0; l < 64; l += 5) {
            printf("  lane %2d:", l);
            for (int i = 0; i < 4; ++i) printf(" (%d,%2d)", h[l * 4 + i] / 64 - (variant ? 57 : 0), h[l * 4 + i] % 64);
            printf("\n");
        }
    }
    return 0;
}
