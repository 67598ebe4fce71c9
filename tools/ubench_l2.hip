// Micro-benchmark: per-CU vs aggregate load throughput for the frame-loop step kernels' access pattern on gfx950.
// Every workgroup streams `priv` KB of its own (HBM/Infinity-Cache resident) data and `shared` KB that all workgroups
// read (L2 resident after first touch), with all loads of a wave issued up front, like lstm_step_fwd_fast_kernel.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_l2.hip -o build/ubench_l2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 1.f; }

// each thread: np private float4 loads + ns shared float4 loads (strided by blockDim so a wave reads 1 KB contiguous)
template <int THREADS>
__global__ void __launch_bounds__(THREADS) stream_kernel(const float4* __restrict__ priv, const float4* __restrict__ shared, float* out, int np, int ns) {
  const float4* p = priv + (size_t)blockIdx.x * np * THREADS + threadIdx.x;
  const float4* s = shared + threadIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll 24
  for (int i = 0; i < np; ++i) { float4 v = p[(size_t)i * THREADS]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll 24
  for (int i = 0; i < ns; ++i) { float4 v = s[(size_t)i * THREADS]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  float r = acc.x + acc.y + acc.z + acc.w;
  if (r == 12345.678f) out[blockIdx.x * THREADS + threadIdx.x] = r;
}

template <int THREADS>
float run(int nwg, int priv_kb, int shared_kb, const float4* priv, const float4* shared, float* out, hipStream_t st, int iters) {
  int np = priv_kb * 1024 / 16 / THREADS, ns = shared_kb * 1024 / 16 / THREADS;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(stream_kernel<THREADS>, dim3(nwg), dim3(THREADS), 0, st, priv, shared, out, np, ns);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3f / iters;
}

int main() {
  const size_t PRIV = (size_t)1024 * 512 * 1024;  // up to 1024 WGs x 512 KB
  float4 *priv, *shared; float* out;
  CK(hipMalloc(&priv, PRIV)); CK(hipMalloc(&shared, 1 << 20)); CK(hipMalloc(&out, 1024 * 1024 * 4));
  CK(hipMemset(priv, 0, PRIV)); CK(hipMemset(shared, 0, 1 << 20));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, out);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  float base = ms;  // us per launch (1000 launches, ms*1e3/1000)
  printf("empty launch %.2f us\n", base);
  printf("%5s %5s %8s %8s | %9s %9s %12s %12s\n", "thr", "nwg", "privKB", "sharedKB", "us/launch", "us-base", "GB/s per WG", "TB/s total");
  int nwgs[] = {32, 64, 128, 256, 512, 1024};
  int cfgs[][2] = {{96, 0}, {0, 96}, {192, 0}, {0, 192}, {96, 192}, {48, 96}, {384, 0}, {0, 384}, {192, 192}};
  for (int thr : {256, 512, 1024})
    for (auto& c : cfgs)
      for (int nwg : nwgs) {
        float us = thr == 256 ? run<256>(nwg, c[0], c[1], priv, shared, out, st, 300)
                 : thr == 512 ? run<512>(nwg, c[0], c[1], priv, shared, out, st, 300)
                              : run<1024>(nwg, c[0], c[1], priv, shared, out, st, 300);
        float w = us - base; double bytes = (double)(c[0] + c[1]) * 1024;
        printf("%5d %5d %8d %8d | %9.2f %9.2f %12.1f %12.2f\n", thr, nwg, c[0], c[1], us, w, bytes / w * 1e-3, bytes * nwg / w * 1e-6);
      }
  return 0;
}
