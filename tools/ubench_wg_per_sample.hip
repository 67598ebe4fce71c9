// Lower bound for a ONE-WORKGROUP-PER-SAMPLE attention backward: only the memory traffic of the two phases
// that such a kernel would fuse, with the arithmetic reduced to what touches every byte once:
//   phase 1 (dw):  dw[l] = memory[b][l][:] . dctx[b][:]                    reads L x Ef fp32 = 385 KB per sample (L2-resident:
//                                                                          the same encoder memory every frame)
//   phase 2 (ds):  dpmT[b][a][l] += f(th[t][b][a][l], dw[l])               reads the tanh stash of frame t (96 KB per sample,
//                                                                          from HBM: 2.7 GB per step) + read-modify-write of dpmT
// for B = 32 samples as 32 x NWG workgroups (NWG = 1: one 1024-thread workgroup per sample on 32 CUs; NWG = 8: the product's
// partition, 256 workgroups), timed as a chain of dependent launches like the frame loop.  Everything a real kernel adds
// (softmax backward, the two transposed convolutions, dq / dv / dU) comes on top.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_wg_per_sample.hip -o build/ubench_wg_per_sample && ./build/ubench_wg_per_sample
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NTH>
__global__ __launch_bounds__(NTH) void bwd_traffic(const float* memory, const float* dctx, const float* th, float* dpmT, float* dq,
                                                    int L, int Ef, int Ad, int L4, int nwg) {
    extern __shared__ float sm[];
    float* dctx_s = sm;          // [Ef]
    float* dw_s = sm + Ef;       // [L4]
    const int b = blockIdx.x / nwg, part = blockIdx.x % nwg, tid = threadIdx.x;
    for (int e = tid; e < Ef; e += NTH) dctx_s[e] = dctx[(long)b * Ef + e];
    for (int l = tid; l < L4; l += NTH) dw_s[l] = 0.f;
    __syncthreads();
    // phase 1: 8 lanes per position, every workgroup of the sample needs ALL positions' dw for the softmax backward, so with
    // nwg > 1 the product recomputes nothing here: positions are split over the sample's workgroups (written to LDS only for
    // the own share; the others' shares would come through memory - not modelled, this is a lower bound)
    const int rows_per = (L + nwg - 1) / nwg, l0 = part * rows_per, l1 = min(L, l0 + rows_per);
    const int sub = tid & 7;
    for (int l = l0 + (tid >> 3); l < l1; l += NTH / 8) {
        const f32x4* mp = reinterpret_cast<const f32x4*>(memory + ((long)b * L + l) * Ef);
        float acc = 0.f;
#pragma unroll 8
        for (int i = sub; i < Ef / 4; i += 8) {
            const f32x4 m = mp[i];
            const f32x4 d = *reinterpret_cast<const f32x4*>(dctx_s + 4 * i);
            acc += m[0] * d[0] + m[1] * d[1] + m[2] * d[2] + m[3] * d[3];
        }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
        if (sub == 0) dw_s[l] = acc;
    }
    __syncthreads();
    // phase 2: attention dims split over the sample's workgroups (as the product's ds kernel: Ad / nwg dims each)
    const int dims_per = Ad / nwg, a0 = part * dims_per;
    const int n4 = dims_per * L4 / 4;
    float sq = 0.f;
    for (int i = tid; i < n4; i += NTH) {
        const int a = a0 + i / (L4 / 4), l = 4 * (i % (L4 / 4));
        const long off = ((long)b * Ad + a) * L4 + l;
        const f32x4 t = *reinterpret_cast<const f32x4*>(th + off);
        f32x4 p = *reinterpret_cast<const f32x4*>(dpmT + off);
        const f32x4 w = *reinterpret_cast<const f32x4*>(dw_s + l);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float ds = w[k] * (1.f - t[k] * t[k]); p[k] += ds; sq += ds; }
        *reinterpret_cast<f32x4*>(dpmT + off) = p;
    }
    if (sq == 12345.678f) dq[blockIdx.x] = sq;     // keep the sum alive
}

int main() {
    const int B = 32, L = 188, Ef = 512, Ad = 128, L4 = 188, NFR = 128, ITER = 384;
    float *memory, *dctx, *th, *dpmT, *dq;
    CK(hipMalloc(&memory, (size_t)B * L * Ef * 4)); CK(hipMalloc(&dctx, (size_t)B * Ef * 4));
    CK(hipMalloc(&th, (size_t)NFR * B * Ad * L4 * 4)); CK(hipMalloc(&dpmT, (size_t)B * Ad * L4 * 4)); CK(hipMalloc(&dq, 4096));
    CK(hipMemset(memory, 0, (size_t)B * L * Ef * 4)); CK(hipMemset(dctx, 0, (size_t)B * Ef * 4));
    CK(hipMemset(th, 0, (size_t)NFR * B * Ad * L4 * 4)); CK(hipMemset(dpmT, 0, (size_t)B * Ad * L4 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = (Ef + L4) * 4;
    const double bytes = (double)B * (L * Ef * 4.0 + 3.0 * Ad * L4 * 4.0);
    printf("per launch: %.1f MB (memory %.1f MB from L2, tanh stash %.1f MB from HBM, dpmT read+write %.1f MB)\n", bytes / 1e6,
           B * L * Ef * 4 / 1e6, B * Ad * L4 * 4 / 1e6, 2.0 * B * Ad * L4 * 4 / 1e6);
    for (int nwg : {1, 2, 4, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int it = 0; it < ITER; ++it) {
                const float* tht = th + (size_t)(it % NFR) * B * Ad * L4;
                if (nwg <= 2) hipLaunchKernelGGL(bwd_traffic<1024>, dim3(B * nwg), dim3(1024), lds, 0, memory, dctx, tht, dpmT, dq, L, Ef, Ad, L4, nwg);
                else hipLaunchKernelGGL(bwd_traffic<512>, dim3(B * nwg), dim3(512), lds, 0, memory, dctx, tht, dpmT, dq, L, Ef, Ad, L4, nwg);
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1)
                printf("%d workgroup(s) per sample (%3d workgroups of %4d threads): %6.2f us per launch in a dependent chain  (%.2f TB/s)\n",
                       nwg, B * nwg, nwg <= 2 ? 1024 : 512, ms * 1e3 / ITER, bytes / (ms * 1e-3 / ITER) / 1e12);
        }
    }
    return 0;
}
