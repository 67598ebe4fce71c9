// Micro-benchmark: what does it cost to replace the kernel boundary between two per-frame phases by an arrival counter
// shared by the 8 workgroups of one sample?  (DESIGN.md section 4.1, next lever 1: energies -> context, dw -> ds.)
//   phase 1: every workgroup writes a 188-float partial;  phase 2: every workgroup reads the 8 partials of its group, adds
//   a 64 KB re-read of its own (L2-resident) slice - which shows whether the acquire throws the L2 contents away - and
//   writes a result.
//   A: two dependent launches per frame          B: one launch, groups of 8 consecutive workgroup ids (8 different XCDs)
//   C: one launch, groups of ids that are equal mod 8 (one XCD under round-robin dispatch)
//   D: like B, but the exchanged partials move with relaxed agent-scope atomic stores/loads and there is no cache-wide fence
// Spins are bounded (2^22 polls) and report through an error word, so a wrong count cannot hang the GPU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_group_sync.hip -o build/ubench_group_sync ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NWG = 256, NT = 512, L = 188, SLICE = 16384;   // 64 KB of floats per workgroup

struct P { float* part; const float* slice; float* out; unsigned* cnt; unsigned* err; unsigned epoch; int mode; };

// 32 groups of 8 workgroups.  mode 2: workgroup id = k * 8 + x with x = XCD (round-robin dispatch), k = 0..31 inside the XCD;
// group = x * 4 + k / 8, so all 8 members share x.  Other modes: group = id / 8 (members on 8 different XCDs).
__device__ __forceinline__ int group_of(int wg, int mode) { return mode == 2 ? (wg & 7) * 4 + ((wg >> 3) >> 3) : wg >> 3; }
__device__ __forceinline__ int member_wg(int grp, int j, int mode) {
  return mode == 2 ? (((grp & 3) * 8 + j) << 3) + (grp >> 2) : grp * 8 + j;
}

__device__ __forceinline__ void phase1(const P& p, int wg, float seed) {
  for (int l = threadIdx.x; l < L; l += NT) {
    if (p.mode >= 3) __hip_atomic_store(&p.part[wg * L + l], seed + l * 1e-3f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else p.part[wg * L + l] = seed + l * 1e-3f;
  }
}

__device__ __forceinline__ void phase2(const P& p, int wg) {
  const int grp = group_of(wg, p.mode);
  float s = 0.f;
  for (int l = threadIdx.x; l < L; l += NT)
    for (int j = 0; j < 8; ++j) {
      const float* q = &p.part[member_wg(grp, j, p.mode) * L + l];
      s += p.mode >= 3 ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
    }
  const float4* sl = reinterpret_cast<const float4*>(p.slice + (size_t)wg * SLICE);
  for (int i = threadIdx.x; i < SLICE / 4; i += NT) { const float4 v = sl[i]; s += v.x + v.y + v.z + v.w; }
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(&p.out[wg], s);
}

__global__ void __launch_bounds__(NT) k_phase1(P p) { phase1(p, blockIdx.x, (float)p.epoch); }
__global__ void __launch_bounds__(NT) k_phase2(P p) { phase2(p, blockIdx.x); }

__global__ void __launch_bounds__(NT) k_fused(P p) {
  const int wg = blockIdx.x;
  phase1(p, wg, (float)p.epoch);
  // modes 1, 2: plain stores/loads for the exchanged data, ordered by agent-scope release / acquire fences (L2 write-back and
  // invalidate).  mode 3: the exchanged data itself moves with relaxed agent-scope atomic stores / loads (they bypass the
  // non-coherent cache levels), every thread waits for its own stores (workgroup-scope fence + barrier), and only the
  // counter is an agent-scope atomic - no cache-wide operation.
  if (p.mode >= 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* c = p.cnt + group_of(wg, p.mode) * 32;                 // one counter per group, 128 bytes apart
    if (p.mode < 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned it = 0;
    while ((int)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 8u * p.epoch) < 0) {
      if (++it > (1u << 22)) { *p.err = 1; break; }
    }
    if (p.mode < 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (p.mode < 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  phase2(p, wg);
}

int main() {
  P p;
  float *part, *slice, *out; unsigned *cnt, *err;
  CK(hipMalloc(&part, NWG * L * 4)); CK(hipMalloc(&slice, (size_t)NWG * SLICE * 4)); CK(hipMalloc(&out, NWG * 4));
  CK(hipMalloc(&cnt, 32 * 32 * 4)); CK(hipMalloc(&err, 4));
  CK(hipMemset(slice, 0, (size_t)NWG * SLICE * 4)); CK(hipMemset(err, 0, 4));
  p.part = part; p.slice = slice; p.out = out; p.cnt = cnt; p.err = err;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 1000;
  const char* names[4] = {"A two dependent launches per frame", "B fused, agent fences, group over 8 XCDs",
                          "C fused, agent fences, group inside one XCD", "D fused, atomic data path, no cache-wide fence"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
      CK(hipMemset(cnt, 0, 32 * 32 * 4)); CK(hipMemset(out, 0, NWG * 4));
      p.mode = mode;
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 1; i <= N; ++i) {
        p.epoch = (unsigned)i;
        if (mode == 0) { hipLaunchKernelGGL(k_phase1, dim3(NWG), dim3(NT), 0, 0, p); hipLaunchKernelGGL(k_phase2, dim3(NWG), dim3(NT), 0, 0, p); }
        else hipLaunchKernelGGL(k_fused, dim3(NWG), dim3(NT), 0, 0, p);
      }
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      float h[NWG]; unsigned herr;
      CK(hipMemcpy(h, out, NWG * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      // expected: sum over epochs of 8 * sum_l (epoch + l * 1e-3)
      double exp = 0; for (int i = 1; i <= N; ++i) exp += 8.0 * (L * (double)i + 1e-3 * L * (L - 1) / 2.0);
      double worst = 0; for (int w = 0; w < NWG; ++w) { double r = fabs(h[w] - exp) / exp; if (r > worst) worst = r; }
      if (rep == 1) printf("%-48s : %6.2f us per frame   (result rel err %.1e, spin timeouts %u)\n", names[mode], ms * 1e3 / N, worst, herr);
    }
  return 0;
}
