// Does the kernel-argument size change the cost of a dependent launch?  (gfx950, HIP_FORCE_DEV_KERNARG default)
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_kernarg.hip -o build/ubench_kernarg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int N> struct Args { float* out; long v[N]; };

template <int N>
__global__ void k(Args<N> a) {
  // touch the last word so the whole block is fetched; one dependent store like a real kernel
  if (threadIdx.x == 0) a.out[blockIdx.x] = (float)a.v[N - 1] + (float)a.v[0];
}

__global__ void k_table(const Args<128>* tab, int i) {
  const Args<128>& a = tab[i];
  if (threadIdx.x == 0) a.out[blockIdx.x] = (float)a.v[127] + (float)a.v[0];
}

template <int N>
void run(float* out, hipStream_t st) {
  Args<N> a; a.out = out; for (int i = 0; i < N; ++i) a.v[i] = i;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0; const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<N>, dim3(256), dim3(256), 0, st, a);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("kernarg %5zu bytes: %.3f us per dependent launch\n", sizeof(a), ms * 1e3 / iters);
}

int main() {
  float* out; CK(hipMalloc(&out, 4096));
  hipStream_t st; CK(hipStreamCreate(&st));
  run<1>(out, st); run<8>(out, st); run<32>(out, st); run<64>(out, st); run<128>(out, st); run<256>(out, st);
  // parameter table resident in device memory, 16-byte kernarg
  Args<128>* tab; CK(hipMalloc(&tab, sizeof(Args<128>) * 64));
  Args<128> h; h.out = out; for (int i = 0; i < 128; ++i) h.v[i] = i;
  for (int i = 0; i < 64; ++i) CK(hipMemcpy(tab + i, &h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0; const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_table, dim3(256), dim3(256), 0, st, tab, i & 63);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("device-resident table (1032-byte entries), 16-byte kernarg: %.3f us per dependent launch\n", ms * 1e3 / iters);
  // Is a small pageable host->device hipMemcpyAsync stream-ordered WITHOUT draining the stream on the host side?
  {
    CK(hipStreamSynchronize(st));
    char* hbuf = (char*)malloc(128 * 1024);
    for (int i = 0; i < 128 * 1024; ++i) hbuf[i] = (char)i;
    char* dbuf; CK(hipMalloc(&dbuf, 128 * 1024));
    Args<1> a; a.out = out; a.v[0] = 1;
    for (size_t bytes : {4096, 65536, 131072}) {
      struct timespec t0, t1, t2;
      for (int i = 0; i < 3000; ++i) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, st, a);   // ~8 ms of queued work
      clock_gettime(CLOCK_MONOTONIC, &t0);
      CK(hipMemcpyAsync(dbuf, hbuf, bytes, hipMemcpyHostToDevice, st));
      clock_gettime(CLOCK_MONOTONIC, &t1);
      CK(hipStreamSynchronize(st));
      clock_gettime(CLOCK_MONOTONIC, &t2);
      printf("pageable H2D %6zu bytes behind 3000 queued kernels: call returned after %.1f us, stream drained %.1f us later\n", bytes,
             (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3, (t2.tv_sec - t1.tv_sec) * 1e6 + (t2.tv_nsec - t1.tv_nsec) * 1e-3);
    }
  }
  return 0;
}
