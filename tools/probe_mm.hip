// Probe of the split GEMM's m-major staging path: store through split_stage_store<false>, read back through split_frag<false>.
#include "../tacotron2_amd/csrc/t2_gemm.hip"
#include <cstdio>
__global__ void probe(float* out, int what) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x;
    f32x4 reg[2];
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j, k = idx >> 5, r = (idx & 31) << 2;
        for (int e = 0; e < 4; ++e) reg[j][e] = what == 0 ? (float)k : (float)(r + e);
    }
    split_stage_store<false>(reg, smem + OP_BYTES, tid);
    __syncthreads();
    if (tid < 64) {
        const bf16x8 f = split_frag<false>(smem + OP_BYTES, 0, 32, tid);
        for (int e = 0; e < 8; ++e) out[tid * 8 + e] = (float)f[e];
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 64 * 8 * sizeof(float));
    float h[512];
    for (int what = 0; what < 2; ++what) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, what);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%s of each fragment element (expect k = 8*(lane>>5) + e, row = 32 + (lane & 31))\n", what == 0 ? "k" : "row");
        for (int l = 0; l < 64; l += 1) {
            printf("lane %2d:", l);
            for (int e = 0; e < 8; ++e) printf(" %3d", (int)h[l * 8 + e]);
            printf("\n");
        }
    }
    return 0;
}
