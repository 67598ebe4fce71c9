// Shared host/device helpers for the gfx950 Tacotron 2 kernels.  CDNA4 only: 64-lane wavefronts,
// fp32-input MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32), no other targets.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <unordered_map>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/tacotron2_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define T2_WAVE 64

// Every C-ABI entry returns an int status (0 = ok) and never throws; see include/tacotron2_amd.h.
#define T2_CHECK_LAUNCH()                                                     \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) {                                              \
            t2_set_error(hipGetErrorString(e__), __FILE__, __LINE__);         \
            return T2_ERR_LAUNCH;                                             \
        }                                                                     \
    } while (0)

#define T2_REQUIRE(cond, msg)                                                 \
    do {                                                                      \
        if (!(cond)) {                                                        \
            t2_set_error(msg, __FILE__, __LINE__);                            \
            return T2_ERR_ARG;                                                \
        }                                                                     \
    } while (0)

#define T2_TRY(expr)                                                          \
    do {                                                                      \
        int rc__ = (expr);                                                    \
        if (rc__ != 0) return rc__;                                           \
    } while (0)

void t2_set_error(const char* msg, const char* file, int line);

// First statement of every kernel of the latency-bound frame chains (cell steps, attention kernels, decode linears): their
// waves win instruction-issue arbitration against the GEMM waves of the other stream that share the SIMD (default priority 0).
// 70.35 -> 69.47 ms per training step in one session (profiles/r02_ab_wave_priority.txt); no effect without a second stream.
#define T2_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)

// Event ring of the DIAGNOSTIC build (-DT2_STAMPS): the first thread of workgroup (0,0,0) of a stamped launch takes the next of 11
// entries of 8 words behind word 40 of the caller's 128-word stamp buffer (sequence counter: word 32) and records
//   [0] kind  [1] 100 MHz wall clock at entry  [2] ... at exit  [3] shader clock at entry  [4..6] phase stamps  [7] shader clock at exit
// so that consecutive launches of a dependent chain can be laid on one time axis (exit -> next entry gaps).  Nothing in the
// product build: every stamp is a branch that ends a basic block.
#ifdef T2_STAMPS
#define T2_RING_BEGIN(clk, cond, kind)                                                                              \
    unsigned long long* ring__ = nullptr;                                                                           \
    if ((clk) && (cond)) {                                                                                          \
        const unsigned long long seq__ = atomicAdd(&(clk)[32], 1ull);                                               \
        ring__ = (clk) + 40 + 8 * (seq__ % 11);                                                                     \
        ring__[0] = (kind); ring__[4] = 0; ring__[5] = 0; ring__[6] = 0;                                            \
        ring__[1] = __builtin_amdgcn_s_memrealtime(); ring__[3] = __builtin_amdgcn_s_memtime();                     \
    }
#define T2_RING(i) do { if (ring__) ring__[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define T2_RING_END() do { if (ring__) { ring__[7] = __builtin_amdgcn_s_memtime(); ring__[2] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define T2_RING_BEGIN(clk, cond, kind) do { } while (0)
#define T2_RING(i) do { } while (0)
#define T2_RING_END() do { } while (0)
#endif

// A/B knobs of kernel selection: compile-time constants in the product library.  Only the DIAGNOSTIC build
// (python -m tacotron2_amd.build --variant ab T2_AB_KNOBS, loaded through T2_LIB_PATH) reads them from the environment.
#ifdef T2_AB_KNOBS
#define T2_KNOB(env, dflt) (getenv(env) ? atoi(getenv(env)) : (dflt))
#else
#define T2_KNOB(env, dflt) (dflt)
#endif

static inline int t2_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool t2_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
// sigmoid / tanh on the hardware exp + rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).  Absolute error ~1e-7, far
// inside the fp32 parity budget, and ~6x fewer instructions than the branchy libm tanhf.
__device__ __forceinline__ float t2_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float t2_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

// Wave-wide reductions with DPP moves (quad_perm, row_ror, row_bcast15/31: ~50 cycles) instead of six dependent
// ds_bpermute shuffles through the LDS crossbar (~600 cycles); every lane receives the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float t2_dpp(float ident, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float t2_wave_sum(float v) {
    v += t2_dpp<0xB1, 0xf>(0.f, v);     // quad_perm [1,0,3,2]
    v += t2_dpp<0x4E, 0xf>(0.f, v);     // quad_perm [2,3,0,1]
    v += t2_dpp<0x124, 0xf>(0.f, v);    // row_ror:4
    v += t2_dpp<0x128, 0xf>(0.f, v);    // row_ror:8   -> every lane holds the sum of its row of 16
    v += t2_dpp<0x142, 0xa>(0.f, v);    // row_bcast15 into rows 1 and 3
    v += t2_dpp<0x143, 0xc>(0.f, v);    // row_bcast31 into rows 2 and 3 -> lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float t2_wave_max(float v) {
    const float ni = -__builtin_inff();
    v = fmaxf(v, t2_dpp<0xB1, 0xf>(ni, v));
    v = fmaxf(v, t2_dpp<0x4E, 0xf>(ni, v));
    v = fmaxf(v, t2_dpp<0x124, 0xf>(ni, v));
    v = fmaxf(v, t2_dpp<0x128, 0xf>(ni, v));
    v = fmaxf(v, t2_dpp<0x142, 0xa>(ni, v));
    v = fmaxf(v, t2_dpp<0x143, 0xc>(ni, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sums over aligned groups of 4 / 8 / 16 lanes (every lane of the group gets the result) and of 32 lanes (the UPPER 16
// lanes of each half-wave get the result)
__device__ __forceinline__ float t2_quad_sum(float v) {
    v += t2_dpp<0xB1, 0xf>(0.f, v);
    v += t2_dpp<0x4E, 0xf>(0.f, v);
    return v;
}
__device__ __forceinline__ float t2_oct_sum(float v) {
    v = t2_quad_sum(v);
    v += t2_dpp<0x141, 0xf>(0.f, v);    // row_half_mirror: the other quad of the aligned group of 8
    return v;
}
__device__ __forceinline__ float t2_row_sum(float v) {
    v = t2_quad_sum(v);
    v += t2_dpp<0x124, 0xf>(0.f, v);
    v += t2_dpp<0x128, 0xf>(0.f, v);
    return v;
}
__device__ __forceinline__ float t2_half_sum_hi(float v) {
    v = t2_row_sum(v);
    v += t2_dpp<0x142, 0xa>(0.f, v);    // lanes 16..31 / 48..63 now hold the sum of their 32 lanes
    return v;
}
__device__ __forceinline__ double t2_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum; `red` is >= (blockDim.x/64) floats of LDS.  All threads get the result.
__device__ __forceinline__ float t2_block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = t2_wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float t2_block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = t2_wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
    for (int i = 1; i < nw; ++i) s = fmaxf(s, red[i]);
    return s;
}

// Dynamic LDS above 64 KB (gfx950 has 160 KB per CU) must be enabled per kernel; the call is remembered per function.
template <typename K>
inline bool t2_allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return true;
    if (bytes > 160 * 1024) return false;
    static std::unordered_map<const void*, size_t> granted;     // per kernel function
    static std::mutex mu;
    const void* f = reinterpret_cast<const void*>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    size_t& g = granted[f];
    if (bytes > g) {
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
        g = bytes;
    }
    return true;
}

// Philox4x32-10 counter RNG (Salmon et al. 2011); one call -> 4 x 32 random bits.
__device__ __forceinline__ void t2_philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
