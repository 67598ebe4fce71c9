// Micro-benchmark of the LSTM step kernel's main loop (gfx950): which of {weight stream, activation loads, MFMA chain,
// wave count, prefetch depth} sets the ~4.5 ns-per-k slope measured by tools/ubench_step.py.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_cell.hip -o build/ubench_cell
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 1.f; }

// WAVES waves per workgroup split K; per wave NCH chunks of 16 k; MT row tiles of 16 batch rows.
// XMODE 0: rows x 64 B pieces (real layout, ld = K), 1: contiguous 1 KB per wave-load, 2: no X loads (registers)
// WMODE 0: packed stream, 2: no W loads.  MMA 0/1.  DEPTH: loads issued ahead in units of chunks (NCH = all up front)
template <int WAVES, int NCH, int MT, int XMODE, int WMODE, int MMA, int DEPTH>
__global__ __launch_bounds__(WAVES * 64, 1) void cell_kernel(const float* __restrict__ W, const float* __restrict__ X, float* out, int K, int ld) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const float* wb = W + (long)blockIdx.x * (NCH * WAVES) * 256 + lane * 4;
  const float* xb[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) xb[m] = XMODE == 0 ? X + (long)(m * 16 + r) * ld + 4 * q : X + (long)m * 16 * K + lane * 4;
  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bw[NCH], ax[NCH][MT];
  auto load = [&](int j) {
    const int c = WAVES * j + w;
    if (WMODE == 0) bw[j] = *reinterpret_cast<const f32x4*>(wb + (long)c * 256);
    else bw[j] = (f32x4){1.f, 2.f, 3.f, (float)j};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (XMODE == 0) ax[j][m] = *reinterpret_cast<const f32x4*>(xb[m] + 16 * c);
      else if (XMODE == 1) ax[j][m] = *reinterpret_cast<const f32x4*>(xb[m] + 256 * c);
      else ax[j][m] = (f32x4){1.f, (float)m, 3.f, (float)j};
    }
  };
  auto mma = [&](int j) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (MMA) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[j][m][s], bw[j][s], acc[m], 0, 0, 0);
        else acc[m][s] += ax[j][m][s] * bw[j][s];
      }
  };
#pragma unroll
  for (int j = 0; j < DEPTH && j < NCH; ++j) load(j);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    if (j + DEPTH < NCH && (j % 8) == 0) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) if (j + DEPTH + jj < NCH) load(j + DEPTH + jj);
      __builtin_amdgcn_sched_barrier(0);
    }
    mma(j);
  }
  __shared__ float red[WAVES * MT * 256];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) red[((w * MT + m) * 16 + (q * 4 + g)) * 16 + r] = acc[m][g];
  __syncthreads();
  if (tid < MT * 256 / 4 * 4 && tid < 256 * MT / 2) {
    float s = 0.f;
    for (int ww = 0; ww < WAVES; ++ww) s += red[ww * MT * 256 + tid];
    out[(long)blockIdx.x * 256 + tid] = s;
  }
}

// Software-pipelined variant: chunk j+8 is loaded right before the MFMAs of chunk j (SCHED 1), or in groups of 8
// chunks as the product kernel did before (SCHED 0): all loads of the next group, then all MFMAs of the current one.
template <int NCH, int MT, int SCHED>
__global__ __launch_bounds__(256, 1) void cell_pipe_kernel(const float* __restrict__ W, const float* __restrict__ X, float* out, int K) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const float* wb = W + (long)blockIdx.x * (NCH * 4) * 256 + lane * 4;
  const float* xb[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) xb[m] = X + (long)m * 16 * K + lane * 4;
  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bw[NCH], ax[NCH][MT];
  auto load = [&](int j) {
    const int c = 4 * j + w;
    bw[j] = *reinterpret_cast<const f32x4*>(wb + (long)c * 256);
#pragma unroll
    for (int m = 0; m < MT; ++m) ax[j][m] = *reinterpret_cast<const f32x4*>(xb[m] + 256 * c);
  };
  auto mma = [&](int j) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[j][m][s], bw[j][s], acc[m], 0, 0, 0);
  };
#pragma unroll
  for (int j = 0; j < 8; ++j) load(j);
  if (SCHED == 0) {
#pragma unroll
    for (int g = 0; g < NCH / 8; ++g) {
      if (g + 1 < NCH / 8) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) load(8 * (g + 1) + jj);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) mma(8 * g + jj);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if (j + 8 < NCH) load(j + 8);
      __builtin_amdgcn_sched_barrier(0);
      mma(j);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __shared__ float red[4 * MT * 256];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) red[((w * MT + m) * 16 + (q * 4 + g)) * 16 + r] = acc[m][g];
  __syncthreads();
  if (tid < 128 * MT) {
    float s = 0.f;
    for (int ww = 0; ww < 4; ++ww) s += red[ww * MT * 256 + tid];
    out[(long)blockIdx.x * 256 + tid] = s;
  }
}

// Heterogeneous launch: workgroups [0, nE) run a VALU/latency-bound body shaped like the attention-energy kernel
// (64 KB of L2-resident loads, ~600 dependent-free FMAs per thread, one store), workgroups [nE, nE+nC) run the cell loop
// with 8 waves.  Do the two kinds overlap on a CU, or do they add up like two cells in one launch?
template <int NCH>
__global__ __launch_bounds__(512, 2) void hetero_kernel(const float* __restrict__ W, const float* __restrict__ X, const float* __restrict__ Q,
                                                        float* out, int K, int nE) {
  __shared__ float red[8 * 2 * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  if ((int)blockIdx.x < nE) {
    const float4* q = reinterpret_cast<const float4*>(Q) + (size_t)(blockIdx.x & 7) * 4096 + tid;
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = q[i * 512];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[0] += v[i].x; acc[1] += v[i].y; acc[2] += v[i].z; acc[3] += v[i].w; }
    float t0 = acc[0], t1 = acc[1], t2 = acc[2], t3 = acc[3];
#pragma unroll 16
    for (int k = 0; k < 160; ++k) {
      t0 = fmaf(t0, 1.0001f, acc[1]); t1 = fmaf(t1, 0.9999f, acc[2]); t2 = fmaf(t2, 1.0002f, acc[3]); t3 = fmaf(t3, 0.9998f, acc[0]);
    }
    red[tid] = t0 + t1 + t2 + t3;
    __syncthreads();
    if (tid < 64) out[(size_t)blockIdx.x * 64 + tid] = red[tid] + red[tid + 64];
    return;
  }
  const int bx = blockIdx.x - nE;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const float* wb = W + (long)bx * (NCH * 8) * 256 + lane * 4;
  const float* xb[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) xb[m] = X + (long)m * 16 * K + lane * 4;
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  f32x4 bw[NCH], ax[NCH][2];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = 8 * j + w;
    bw[j] = *reinterpret_cast<const f32x4*>(wb + (long)c * 256);
#pragma unroll
    for (int m = 0; m < 2; ++m) ax[j][m] = *reinterpret_cast<const f32x4*>(xb[m] + 256 * c);
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[j][m][s], bw[j][s], acc[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) red[((w * 2 + m) * 16 + (q * 4 + g)) * 16 + r] = acc[m][g];
  __syncthreads();
  if (tid < 256) {
    float s = 0.f;
    for (int ww = 0; ww < 8; ++ww) s += red[ww * 512 + tid];
    out[(size_t)nE * 64 + (long)bx * 256 + tid] = s;
  }
}

template <int NCH>
void run_hetero(const char* name, const float* W, const float* X, float* out, hipStream_t st, float base, int nE, int nC) {
  const int K = 8 * NCH * 16, iters = 500;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((hetero_kernel<NCH>), dim3(nE + nC), dim3(512), 0, st, W, X, X, out, K, nE);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  CK(hipGetLastError());
  printf("%-44s K=%4d nE=%3d nC=%3d : %6.2f us (%.2f over empty)\n", name, K, nE, nC, ms * 1e3 / iters, ms * 1e3 / iters - base);
  fflush(stdout);
}

template <int NCH, int MT, int SCHED>
void run_pipe(const char* name, const float* W, const float* X, float* out, hipStream_t st, float base, int nwg = 256) {
  const int K = 4 * NCH * 16, iters = 500;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((cell_pipe_kernel<NCH, MT, SCHED>), dim3(nwg), dim3(256), 0, st, W, X, out, K);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  CK(hipGetLastError());
  printf("%-44s K=%4d MT=%d nwg=%3d sched=%d : %6.2f us (%.2f over empty)\n", name, K, MT, nwg, SCHED, ms * 1e3 / iters, ms * 1e3 / iters - base);
  fflush(stdout);
}

template <int WAVES, int NCH, int MT, int XMODE, int WMODE, int MMA, int DEPTH>
void run(const char* name, const float* W, const float* X, float* out, hipStream_t st, float base, int ldx = 0) {
  const int K = WAVES * NCH * 16, iters = 500; const int ld = ldx ? ldx : K;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i)
      hipLaunchKernelGGL((cell_kernel<WAVES, NCH, MT, XMODE, WMODE, MMA, DEPTH>), dim3(256), dim3(WAVES * 64), 0, st, W, X, out, K, ld);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  CK(hipGetLastError());
  printf("%-58s K=%4d ld=%4d waves=%2d MT=%d depth=%2d : %6.2f us (%.2f over empty)\n", name, K, ld, WAVES, MT, DEPTH, ms * 1e3 / iters, ms * 1e3 / iters - base);
  fflush(stdout);
}

int main() {
  float *W, *X, *out;
  CK(hipMalloc(&W, (size_t)256 * 4096 * 16 * 4)); CK(hipMalloc(&X, 64 * 4096 * 4 * 4)); CK(hipMalloc(&out, 256 * 1024 * 4));
  CK(hipMemset(W, 0, (size_t)256 * 4096 * 16 * 4)); CK(hipMemset(X, 0, 64 * 4096 * 4 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, out);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  float base = ms;
  printf("empty launch %.2f us\n", base);
  run_pipe<24, 2, 0>("tiled, grouped (8-chunk phases)", W, X, out, st, base);
  run_pipe<24, 2, 1>("tiled, per-chunk software pipeline", W, X, out, st, base);
  run_pipe<16, 2, 0>("tiled, grouped", W, X, out, st, base);
  run_pipe<16, 2, 1>("tiled, per-chunk software pipeline", W, X, out, st, base);
  run_pipe<24, 2, 0>("grouped, 2 WGs per CU", W, X, out, st, base, 512);
  run_pipe<24, 2, 1>("per-chunk pipeline, 2 WGs per CU", W, X, out, st, base, 512);
  run_pipe<24, 1, 0>("tiled, grouped", W, X, out, st, base);
  run_pipe<24, 1, 1>("tiled, per-chunk software pipeline", W, X, out, st, base);
  run_hetero<8>("hetero: VALU-type only", W, X, out, st, base, 256, 0);
  run_hetero<8>("hetero: cell-type only (8 waves)", W, X, out, st, base, 0, 256);
  run_hetero<8>("hetero: both kinds in one launch", W, X, out, st, base, 256, 256);
  run_hetero<12>("hetero: cell-type only (8 waves)", W, X, out, st, base, 0, 256);
  run_hetero<12>("hetero: both kinds in one launch", W, X, out, st, base, 256, 256);
  // two-stream concurrency: the same chain of launches on one stream, and split over two streams
  {
    hipStream_t s2; CK(hipStreamCreate(&s2));
    const int iters = 400;
    auto chain = [&](hipStream_t q, int n, int nwg) {
      for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL((cell_kernel<4, 24, 2, 1, 0, 1, 16>), dim3(nwg), dim3(256), 0, q, W, X, out, 1536, 1536);
    };
    for (int nwg : {256, 128, 64}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, st)); chain(st, 2 * iters, nwg); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf("nwg %3d: 1 stream, %d launches: %.2f us per launch\n", nwg, 2 * iters, ms * 1e3 / (2 * iters));
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        hipEvent_t f0, f1, g1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1)); CK(hipEventCreate(&g1));
        CK(hipEventRecord(f0, st)); CK(hipStreamWaitEvent(s2, f0, 0));
        chain(st, iters, nwg); chain(s2, iters, nwg);
        CK(hipEventRecord(g1, s2)); CK(hipStreamWaitEvent(st, g1, 0)); CK(hipEventRecord(f1, st)); CK(hipEventSynchronize(f1));
        CK(hipEventElapsedTime(&ms, f0, f1));
      }
      printf("nwg %3d: 2 streams x %d launches: %.2f us per launch pair (perfect overlap = 1-stream time per launch)\n", nwg, iters, ms * 1e3 / iters);
    }
  }
  return 0;
}
