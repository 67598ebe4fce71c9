// GPU-side cost of a dependent launch, with the host out of the picture: 2000 launches are enqueued behind a kernel that
// spins for ~30 ms, so the command processor finds every packet already waiting.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_gpu_launch.hip -o build/ubench_gpu_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin_kernel(long long cycles, float* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (out == nullptr) out[0] = 1.f;
}
template <int N> struct Args { float* out; long v[N]; };
template <int N> __global__ void k(Args<N> a) { if (threadIdx.x == 0) a.out[blockIdx.x] = (float)a.v[N - 1] + (float)a.v[0]; }
__global__ void k_chain(const float* in, float* out) {   // reads what the previous launch wrote
  out[blockIdx.x * 256 + threadIdx.x] = in[(blockIdx.x * 256 + threadIdx.x + 4096) & 65535] + 1.0f;
}

template <int N>
void run(float* out, hipStream_t st, int nwg) {
  Args<N> a; a.out = out; for (int i = 0; i < N; ++i) a.v[i] = i;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, 3000000LL, out);   // 100 MHz clock: 30 ms
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<N>, dim3(nwg), dim3(256), 0, st, a);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("kernarg %5zu bytes, %3d workgroups: %.3f us per dependent launch (GPU side)\n", sizeof(a), nwg, ms * 1e3 / iters);
}

int main() {
  float* out; CK(hipMalloc(&out, 65536 * 4 * 2)); CK(hipMemset(out, 0, 65536 * 4 * 2));
  hipStream_t st; CK(hipStreamCreate(&st));
  run<1>(out, st, 1); run<1>(out, st, 256); run<32>(out, st, 256); run<64>(out, st, 256); run<128>(out, st, 256);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, 3000000LL, out);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, st, out + (i & 1) * 65536, out + ((i + 1) & 1) * 65536);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("read-previous-output chain (256 KB per launch): %.3f us per dependent launch (GPU side)\n", ms * 1e3 / iters);
  return 0;
}
