// Micro-benchmark: cost of a dependent kernel boundary vs. an in-kernel grid barrier on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_sync.hip -o build/ubench_sync ; run on the GPU box.
// Used to decide between per-phase launches and a persistent frame-loop kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 1.f; }

// every WG writes 64 floats, next launch reads all of them
__global__ void __launch_bounds__(256) exch_kernel(const float* __restrict__ in, float* __restrict__ out, int nwg) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nwg * 64; i += 256) s += in[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  float tot = red[0] + red[1] + red[2] + red[3];
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = tot * (1.0f / (64.f * nwg)) + 1.0f;
}

struct Bar { unsigned* cnt; unsigned* xcnt; unsigned* gen; unsigned* err; };

__device__ __forceinline__ bool spin_until(unsigned* p, unsigned target) {
  unsigned it = 0;
  while ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
    __builtin_amdgcn_s_sleep(1);
    if (++it > (1u << 22)) return false;
  }
  return true;
}

// flat barrier: one counter, everyone polls it
__device__ __forceinline__ bool grid_barrier_flat(Bar b, unsigned nwg, unsigned epoch) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __atomic_thread_fence(__ATOMIC_RELEASE);  // handled by scoped fence below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = spin_until(b.cnt, epoch * nwg);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!ok) *b.err = 1;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok;
}

// hierarchical: 8 per-XCD counters (WG id % 8 = XCD under round-robin dispatch), last arriver of each XCD bumps global,
// last global arriver publishes generation; everyone polls the generation word
__device__ __forceinline__ bool grid_barrier_hier(Bar b, unsigned nwg, unsigned epoch) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    unsigned x = blockIdx.x & 7u, per = nwg >> 3;
    unsigned v = __hip_atomic_fetch_add(b.xcnt + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v + 1 == epoch * per) {
      unsigned g = __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g + 1 == epoch * 8u) __hip_atomic_store(b.gen, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ok = spin_until(b.gen, epoch);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!ok) *b.err = 1;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok;
}

template <int MODE>
__global__ void __launch_bounds__(256) persistent_kernel(float* bufA, float* bufB, Bar b, int iters, int nwg, int payload) {
  __shared__ float red[4];
  float* in = bufA; float* out = bufB;
  for (int it = 0; it < iters; ++it) {
    if (payload) {
      float s = 0.f;
      for (int i = threadIdx.x; i < nwg * 64; i += 256) s += in[i];
      for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
      __syncthreads();
      float tot = red[0] + red[1] + red[2] + red[3];
      if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = tot * (1.0f / (64.f * nwg)) + 1.0f;
    }
    bool ok = MODE == 0 ? grid_barrier_flat(b, nwg, it + 1) : grid_barrier_hier(b, nwg, it + 1);
    if (!ok) return;
    float* t = in; in = out; out = t;
  }
}

int main(int argc, char** argv) {
  int nwg = 256, iters = 2000;
  if (argc > 1) iters = atoi(argv[1]);
  float *A, *B; unsigned* sync;
  CK(hipMalloc(&A, nwg * 64 * 4)); CK(hipMalloc(&B, nwg * 64 * 4)); CK(hipMalloc(&sync, 4096 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  std::vector<float> h(nwg * 64);
  auto reset = [&]() { CK(hipMemsetAsync(A, 0, nwg * 64 * 4, st)); CK(hipMemsetAsync(B, 0, nwg * 64 * 4, st)); CK(hipMemsetAsync(sync, 0, 4096 * 4, st)); };
  auto expect = [&](const char* name, float* res) {
    CK(hipMemcpy(h.data(), res, nwg * 64 * 4, hipMemcpyDeviceToHost));
    float mn = 1e30f, mx = -1e30f; for (float v : h) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    printf("   %s result min %.6f max %.6f (expect value after %d iterations of x -> x+1 averaged: %d)\n", name, mn, mx, iters, iters);
  };
  // 1. empty dependent launches
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(empty_kernel, dim3(nwg), dim3(256), 0, st, A);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("empty launches           : %.3f us per launch\n", ms * 1e3 / iters);
  // 2. exchange launches
  reset();
  for (int rep = 0; rep < 2; ++rep) {
    reset();
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(exch_kernel, dim3(nwg), dim3(256), 0, st, (i & 1) ? B : A, (i & 1) ? A : B, nwg);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("exchange launches        : %.3f us per launch\n", ms * 1e3 / iters);
  expect("launch", (iters & 1) ? B : A);
  // 3. graph of exchange launches
  {
    reset(); CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(exch_kernel, dim3(nwg), dim3(256), 0, st, (i & 1) ? B : A, (i & 1) ? A : B, nwg);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    reset();
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("exchange launches (graph): %.3f us per launch\n", ms * 1e3 / iters);
    expect("graph", (iters & 1) ? B : A);
  }
  // 4. persistent kernels
  for (int mode = 0; mode < 2; ++mode) for (int payload = 0; payload < 2; ++payload) {
    Bar b{sync, sync + 64, sync + 2048, sync + 3072};
    for (int rep = 0; rep < 2; ++rep) {
      reset();
      CK(hipEventRecord(e0, st));
      if (mode == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(nwg), dim3(256), 0, st, A, B, b, iters, nwg, payload);
      else hipLaunchKernelGGL(persistent_kernel<1>, dim3(nwg), dim3(256), 0, st, A, B, b, iters, nwg, payload);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned err; CK(hipMemcpy(&err, sync + 3072, 4, hipMemcpyDeviceToHost));
    printf("persistent %s barrier, payload %d: %.3f us per iteration (err %u)\n", mode ? "hier" : "flat", payload, ms * 1e3 / iters, err);
    if (payload) expect("persistent", (iters & 1) ? B : A);
  }
  return 0;
}
