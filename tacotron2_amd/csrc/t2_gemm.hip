// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 64 cycles / instruction / SIMD).
//
// Tile: 128 x 128 x 32 per 256-thread workgroup (4 waves, one per SIMD, 2 x 2 waves of 64 x 64, each wave
// 2 x 2 MFMA tiles of 32 x 32 -> 64 accumulator registers).  LDS: two stages x (A tile + B tile), 73.7 KB, so
// two workgroups are resident per CU.  Global -> register -> LDS staging with the next tile's global loads in
// flight during the current tile's 64 MFMAs per wave.
//
// LDS images
//   k-major operand (element (r,k) contiguous in k):  tile[r][36]  (32 k + 4 pad).  A lane reads its fragment
//     as ONE ds_read_b128 = 4 consecutive k.  The K order inside a 32-deep tile is permuted consistently for A and B:
//     k-group g (8 k's), lane half h reads k = 8g+4h .. 8g+4h+3 and MFMA step s uses element s, i.e. the k pair
//     {8g+s, 8g+4+s}.  Row stride 36 dwords = 4*9 -> the 16-lane b128 groups hit 16 distinct 4-bank slots.
//   m-major operand (element (r,k) contiguous in r):  tile[k][132]; fragment reads are conflict-free ds_read_b32
//     (32 consecutive rows per lane half).
//
// blockIdx.x -> tile mapping is XCD-aware (8 XCDs, round-robin dispatch): the workgroups that land on one XCD
// form a contiguous run of a grouped (8 m-tiles wide) tile order, so their A/B panels share that XCD's L2.
#include "t2_common.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = BK + 4;    // k-major tile row stride
constexpr int LDM = BM + 4;    // m-major tile row stride
constexpr int TILE_FLOATS = BM * LDK;  // 4608 >= BK*LDM = 4224

struct GemmK {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    float alpha;
    const float* bias; const float* bias2;
    const float* mulmask; long ldmask;
    int relu, accumulate, splitk;
    long sA, sB, sC;
    int a_vec, b_vec;
    int ntm, ntn;
    int a_tap_len; long a_tap_stride;     // T2Gemm.a_tap_len / a_tap_stride (0: plain rows)
    float* stat_out; int stat_Lp, stat_L; // T2Gemm.stat_out: per-tile column statistics of the stored values (split kernel, plain store)
};

// physical element offset of logical k0 (a multiple of the tile depth, which divides a_tap_len) on A's K axis
__device__ __forceinline__ long a_koff(const GemmK& p, int k0) {
    return p.a_tap_len ? (long)(k0 / p.a_tap_len) * p.a_tap_stride + (k0 % p.a_tap_len) : (long)k0;
}

// Load 4 consecutive elements starting at p[0] with `nvalid` (<=4) of them in range.
__device__ __forceinline__ f32x4 load4(const float* p, int nvalid, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (nvalid >= 4 && vec) {
        v = *reinterpret_cast<const f32x4*>(p);
    } else {
        if (nvalid > 0) v[0] = p[0];
        if (nvalid > 1) v[1] = p[1];
        if (nvalid > 2) v[2] = p[2];
        if (nvalid > 3) v[3] = p[3];
    }
    return v;
}

// Stage one operand tile: global -> 4 x f32x4 registers.
// KMAJ: rows r in [r0, r0+128), k in [k0, k0+32): idx = tid + 256*j, r = idx>>3, kq = idx&7
// MMAJ: k rows kk in [k0,k0+32), r in [r0, r0+128): idx = tid + 256*j, kk = idx>>5, rq = idx&31
template <bool KMAJ>
__device__ __forceinline__ void stage_load(f32x4 (&reg)[4], const float* __restrict__ base, long ld, int r0, int k0,
                                           int R, int Kend, bool vec, int tid, long kphys = -1) {
    const long kp0 = kphys >= 0 ? kphys : (long)k0;     // where logical k0 sits on the operand's K axis (tap-strided A)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j;
        if (KMAJ) {
            const int r = r0 + (idx >> 3), k = k0 + ((idx & 7) << 2);
            int nv = (r < R) ? (Kend - k) : 0;
            nv = nv < 0 ? 0 : nv;
            reg[j] = load4(base + (long)r * ld + kp0 + ((idx & 7) << 2), nv, vec);
        } else {
            const int k = k0 + (idx >> 5), r = r0 + ((idx & 31) << 2);
            int nv = (k < Kend) ? (R - r) : 0;
            nv = nv < 0 ? 0 : nv;
            reg[j] = load4(base + (long)k * ld + r, nv, vec);
        }
    }
}

template <bool KMAJ>
__device__ __forceinline__ void stage_store(const f32x4 (&reg)[4], float* tile, int tid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j;
        if (KMAJ) {
            *reinterpret_cast<f32x4*>(tile + (idx >> 3) * LDK + ((idx & 7) << 2)) = reg[j];
        } else {
            *reinterpret_cast<f32x4*>(tile + (idx >> 5) * LDM + ((idx & 31) << 2)) = reg[j];
        }
    }
}

// Fragment for k-group g: 4 values (MFMA steps s=0..3) for tile row `row` (0..127), lane half h.
template <bool KMAJ>
__device__ __forceinline__ f32x4 frag(const float* tile, int row, int g, int h) {
    if (KMAJ) {
        return *reinterpret_cast<const f32x4*>(tile + row * LDK + 8 * g + 4 * h);
    } else {
        const float* p = tile + (8 * g + 4 * h) * LDM + row;
        f32x4 v;
        v[0] = p[0]; v[1] = p[LDM]; v[2] = p[2 * LDM]; v[3] = p[3 * LDM];
        return v;
    }
}

template <bool AK, bool BKM>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma(GemmK p) {
    __shared__ __attribute__((aligned(16))) float smem[4 * TILE_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // ---- XCD-aware tile mapping (bijective for any grid size) ----
    const int nblk = p.ntm * p.ntn;
    int id = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = id & 7, slot = id >> 3;
        id = slot + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q);
    }
    constexpr int GM = 8;
    const int per_group = GM * p.ntn;
    const int grp = id / per_group;
    const int gm0 = grp * GM;
    const int gsz = (p.ntm - gm0) < GM ? (p.ntm - gm0) : GM;
    const int within = id - grp * per_group;
    const int tm = gm0 + within % gsz, tn = within / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- batch / split-K ----
    const int kz = blockIdx.z % p.splitk, bz = blockIdx.z / p.splitk;
    const float* A = p.A + (long)bz * p.sA;
    const float* B = p.B + (long)bz * p.sB;
    float* C = p.C + (long)bz * p.sC;
    const int nkt = (p.K + BK - 1) / BK;
    const int per = (nkt + p.splitk - 1) / p.splitk;
    const int kt0 = kz * per;
    const int kt1 = (kt0 + per) < nkt ? (kt0 + per) : nkt;
    if (kt0 >= kt1) return;
    const int Kend = (kt1 * BK) < p.K ? (kt1 * BK) : p.K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[4], rb[4];
    stage_load<AK>(ra, A, p.lda, m0, kt0 * BK, p.M, Kend, p.a_vec, tid, AK ? a_koff(p, kt0 * BK) : -1);
    stage_load<BKM>(rb, B, p.ldb, n0, kt0 * BK, p.N, Kend, p.b_vec, tid);
    stage_store<AK>(ra, smem, tid);
    stage_store<BKM>(rb, smem + TILE_FLOATS, tid);
    __syncthreads();

    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = (kt + 1) < kt1;
        if (more) {
            stage_load<AK>(ra, A, p.lda, m0, (kt + 1) * BK, p.M, Kend, p.a_vec, tid, AK ? a_koff(p, (kt + 1) * BK) : -1);
            stage_load<BKM>(rb, B, p.ldb, n0, (kt + 1) * BK, p.N, Kend, p.b_vec, tid);
        }
        const float* As = smem + cur * 2 * TILE_FLOATS;
        const float* Bs = As + TILE_FLOATS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = frag<AK>(As, wm * 64 + i * 32 + li, g, lh);
                fb[i] = frag<BKM>(Bs, wn * 64 + i * 32 + li, g, lh);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        if (more) {
            float* An = smem + (cur ^ 1) * 2 * TILE_FLOATS;
            stage_store<AK>(ra, An, tid);
            stage_store<BKM>(rb, An + TILE_FLOATS, tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            if (col >= p.N) continue;
            float badd = 0.f;
            if (p.bias && kz == 0) badd += p.bias[col];
            if (p.bias2 && kz == 0) badd += p.bias2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= p.M) continue;
                float v = p.alpha * acc[i][j][r] + badd;
                float* cp = C + (long)row * p.ldc + col;
                if (p.accumulate == 2) {
                    atomicAdd(cp, v);
                } else {
                    if (p.accumulate == 1) v += *cp;
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.mulmask) v *= p.mulmask[(long)row * p.ldmask + col];
                    *cp = v;
                }
            }
        }
    }
}

// =====================================================================================================================
// fp32 GEMM on the bf16 matrix pipe by error-free operand splitting ("bf16x3 split, 6 products").
//
// gfx950 runs f32-input MFMA at the vector rate (64 FLOP/clk/SIMD); the bf16 forms run 16x faster.  Every fp32 value is the
// EXACT sum of three bf16 values: a = a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2) (each
// subtraction is exact in fp32 and 3 x 8 significand bits cover fp32's 24).  A product a*b is then the sum of 9 bf16
// products, each EXACT in the fp32 accumulator of v_mfma_f32_32x32x16_bf16; the three terms a2*b3, a3*b2, a3*b3 are below
// 2^-24 |a*b| (the size of one fp32 rounding) and are dropped, the other six are issued:
//     lo += a3*b1 + a1*b3 + a2*b2 + a2*b1 + a1*b2        hi += a1*b1        C = hi + lo
// The small terms have their own accumulator, so they are not rounded against the large partial sums.  The result has the
// accuracy of an fp32 GEMM (tests/test_gpu_kernels.py compares both kernels with float64) at 16/6 = 2.7x the MFMA rate.
//
// Tile 128 x 128 x 16 per 256-thread workgroup (2 x 2 waves of 64 x 64, each 2 x 2 MFMA tiles of 32 x 32 and TWO
// accumulator sets), two 36 KB LDS stages, two workgroups per CU = two waves per SIMD.  The split of the NEXT tile (VALU)
// and its LDS stores are interleaved instruction by instruction with the 24 MFMAs per wave of the current tile
// (sched_group_barrier), so a wave keeps the matrix pipe and the vector pipe busy at the same time, and the second wave
// of the SIMD fills its stalls (LDS latency after the barrier).  Operands are split once, on the way from the fp32 global
// tile into LDS (global -> registers two tiles ahead -> split -> three bf16 planes):
//   k-major operand: plane[row][16 k] with 48-byte rows; a lane's fragment (8 consecutive k) is ONE ds_read_b128, the
//     16 rows of a b128 lane group start 12 dwords apart: conflict-free.
//   m-major operand (dgrad B, wgrad A and B): plane[k][128 rows] with 320-byte rows, written as it is read from global
//     memory (4 consecutive rows of one k per lane: contiguous 8-byte stores) and read back TRANSPOSED with
//     ds_read_b64_tr_b16 (a 16-lane group reads 4 k-rows x 16 columns, lane i receives column i's 4 k values): the
//     four k-rows of a 32-lane half start 16 dwords apart: conflict-free.
// Both images deliver fragment element j = k offset j, so A and B agree on the k order inside an MFMA.
// =====================================================================================================================
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int SBK = 16;
constexpr int KM_ROW = 48, MM_ROW = 320;                   // bytes
constexpr int KM_PLANE = BM * KM_ROW, MM_PLANE = SBK * MM_ROW;   // 6144, 5120 bytes
constexpr int OP_BYTES = 3 * KM_PLANE;                     // one operand's three planes (the larger image)
constexpr int STAGE_BYTES = 2 * OP_BYTES;                  // 36864; two stages = 72 KB: two workgroups per CU

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// two floats -> one dword of two bf16 (round to nearest even: v_cvt_pk_bf16_f32), and the two bf16 back as floats
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// a = h + m + l exactly, h = bf16(a), m = bf16(a - h), l = bf16(a - h - m); four values at a time, packed results
// (per 4 values: 6 v_cvt_pk + 8 unpack + 8 v_sub = 22 VALU instructions)
struct Split3 { uint2 h, m, l; };
__device__ __forceinline__ Split3 split3(const f32x4 v) {
    Split3 s;
    s.h.x = pk_bf16(v[0], v[1]); s.h.y = pk_bf16(v[2], v[3]);
    const float r0 = v[0] - bf16_lo(s.h.x), r1 = v[1] - bf16_hi(s.h.x), r2 = v[2] - bf16_lo(s.h.y), r3 = v[3] - bf16_hi(s.h.y);   // exact
    s.m.x = pk_bf16(r0, r1); s.m.y = pk_bf16(r2, r3);
    const float q0 = r0 - bf16_lo(s.m.x), q1 = r1 - bf16_hi(s.m.x), q2 = r2 - bf16_lo(s.m.y), q3 = r3 - bf16_hi(s.m.y);           // exact
    s.l.x = pk_bf16(q0, q1); s.l.y = pk_bf16(q2, q3);      // exact: <= 8 significant bits are left
    return s;
}

// global -> 2 x f32x4 registers.  KMAJ: rows r0 + (idx>>2), k = k0 + 4*(idx&3); MMAJ: k = k0 + (idx>>5), rows r0 + 4*(idx&31).
// FULL (wave-uniform: the tile is interior and 16-byte loads are legal): plain vector loads, no guards.
template <bool KMAJ, bool FULL>
__device__ __forceinline__ void split_stage_load(f32x4 (&reg)[2], const float* __restrict__ base, long ld, int r0, int k0, int R,
                                                 int Kend, bool vec, int tid, long kphys = -1) {
    const long kp0 = kphys >= 0 ? kphys : (long)k0;     // where logical k0 sits on the operand's K axis (tap-strided A)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j;
        if (KMAJ) {
            const int r = r0 + (idx >> 2), k = k0 + ((idx & 3) << 2);
            const long kp = kp0 + ((idx & 3) << 2);
            if (FULL) { reg[j] = *reinterpret_cast<const f32x4*>(base + (long)r * ld + kp); continue; }
            int nv = (r < R) ? (Kend - k) : 0;
            nv = nv < 0 ? 0 : nv;
            reg[j] = load4(base + (long)r * ld + kp, nv, vec);
        } else {
            const int k = k0 + (idx >> 5), r = r0 + ((idx & 31) << 2);
            if (FULL) { reg[j] = *reinterpret_cast<const f32x4*>(base + (long)k * ld + r); continue; }
            int nv = (k < Kend) ? (R - r) : 0;
            nv = nv < 0 ? 0 : nv;
            reg[j] = load4(base + (long)k * ld + r, nv, vec);
        }
    }
}

// split register j of a staged tile and store its three planes (one "unit": ~22 VALU + 3 ds_write_b64)
// NP = number of planes kept: 3 (six products, fp32-exact operands), 2 (three products: torch's "high" = bf16x3), 1 (bf16)
template <bool KMAJ, int NP>
__device__ __forceinline__ void split_store_unit(const f32x4 v, char* op, int tid, int j) {
    const int idx = tid + 256 * j;
    const Split3 s = split3(v);          // (unused planes are dead code after inlining)
    const int off = KMAJ ? (idx >> 2) * KM_ROW + (idx & 3) * 8 : (idx >> 5) * MM_ROW + (idx & 31) * 8;
    constexpr int PL = KMAJ ? KM_PLANE : MM_PLANE;
    *reinterpret_cast<uint2*>(op + off) = s.h;
#if defined(T2_GEMM_ABL_STORE)          // diagnostic build (wrong results): one plane stored instead of three, the split itself kept
    if (NP >= 2 && (s.m.x ^ s.l.x) == 0x12345u) *reinterpret_cast<uint2*>(op + PL + off) = s.m;
#else
    if (NP >= 2) *reinterpret_cast<uint2*>(op + PL + off) = s.m;
    if (NP >= 3) *reinterpret_cast<uint2*>(op + 2 * PL + off) = s.l;
#endif
}

// fragment of plane `pl` for the 32-row MFMA tile starting at tile row r0: element j = k offset 8*(lane>>5) + j
template <bool KMAJ>
__device__ __forceinline__ bf16x8 split_frag(const char* op, int pl, int r0, int lane) {
    if (KMAJ) {
        return *reinterpret_cast<const bf16x8*>(op + pl * KM_PLANE + (r0 + (lane & 31)) * KM_ROW + (lane >> 5) * 16);
    } else {
        const int q = (lane & 15) >> 2, p4 = lane & 3, g2 = (lane >> 4) & 1, h = lane >> 5;
        const char* a0 = op + pl * MM_PLANE + (8 * h + q) * MM_ROW + (r0 + 16 * g2 + 4 * p4) * 2;
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
        const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0));
        const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0 + 4 * MM_ROW));
        // (whole-vector bit casts: element-wise bit_cast of the transposed read's result is miscompiled by ROCm 7.2's hipcc -
        // element 0 is splatted into all four - tools/probe_tr16.hip)
        return __builtin_shufflevector(__builtin_bit_cast(bf16x4, t0), __builtin_bit_cast(bf16x4, t1), 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// One 16-deep k-step of the six-product scheme for this wave's 2 x 2 MFMA tiles: 24 MFMAs in 8 groups of 3; after group g the
// caller's `unit(g)` (one split_store_unit of the NEXT tile) is issued, and sched_group_barriers pin the instruction order
// MFMA, 4 x VALU, MFMA, ... : with one wave per SIMD the conversion of the next tile runs in the issue gaps of this tile's
// MFMAs (a 32x32x16 MFMA occupies the matrix pipe for 32 cycles and the issue port for 8).
#define T2_SPLIT_MFMA(ACC, I, J, PA, PB) ACC[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[I][PA], fb[J][PB], ACC[I][J], 0, 0, 0)

template <bool AK, bool BKM, int NP>
__global__ __launch_bounds__(256, 2) void gemm_f32_split_bf16(GemmK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 * STAGE_BYTES
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // ---- XCD-aware tile mapping (as gemm_f32_mfma) ----
    const int nblk = p.ntm * p.ntn;
    int id = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = id & 7, slot = id >> 3;
        id = slot + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q);
    }
    constexpr int GM = 8;
    const int per_group = GM * p.ntn;
    const int grp = id / per_group;
    const int gm0 = grp * GM;
    const int gsz = (p.ntm - gm0) < GM ? (p.ntm - gm0) : GM;
    const int within = id - grp * per_group;
    const int tm = gm0 + within % gsz, tn = within / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int kz = blockIdx.z % p.splitk, bz = blockIdx.z / p.splitk;
    const float* A = p.A + (long)bz * p.sA;
    const float* B = p.B + (long)bz * p.sB;
    float* C = p.C + (long)bz * p.sC;
    const int nkt = (p.K + SBK - 1) / SBK;
    const int per = (nkt + p.splitk - 1) / p.splitk;
    const int kt0 = kz * per;
    const int kt1 = (kt0 + per) < nkt ? (kt0 + per) : nkt;
    if (kt0 >= kt1) return;
    const int Kend = (kt1 * SBK) < p.K ? (kt1 * SBK) : p.K;
    // interior tiles take guard-free vector loads (wave-uniform decision per k-tile)
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N) && p.a_vec && p.b_vec;

    f32x16 hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { hi[i][j][r] = 0.f; lo[i][j][r] = 0.f; }

    // Two register sets for the fp32 tiles in flight: while the set holding tile kt+1 is split and stored between this tile's
    // MFMAs, the other set receives tile kt+2 (requested at the top of the iteration: a whole tile of MFMAs to land).
    f32x4 ra[2], rb[2], na[2], nb[2];
    auto load_tile = [&](f32x4 (&xa)[2], f32x4 (&xb)[2], int kt) {
        const long ka = AK ? a_koff(p, kt * SBK) : -1;
        if (interior && (kt + 1) * SBK <= Kend) {
            split_stage_load<AK, true>(xa, A, p.lda, m0, kt * SBK, p.M, Kend, true, tid, ka);
            split_stage_load<BKM, true>(xb, B, p.ldb, n0, kt * SBK, p.N, Kend, true, tid);
        } else {
            split_stage_load<AK, false>(xa, A, p.lda, m0, kt * SBK, p.M, Kend, p.a_vec, tid, ka);
            split_stage_load<BKM, false>(xb, B, p.ldb, n0, kt * SBK, p.N, Kend, p.b_vec, tid);
        }
    };
    // prologue: tile kt0 -> LDS stage 0, tile kt0+1 -> register set (ra, rb)
    load_tile(ra, rb, kt0);
#pragma unroll
    for (int j = 0; j < 2; ++j) { split_store_unit<AK, NP>(ra[j], smem, tid, j); split_store_unit<BKM, NP>(rb[j], smem + OP_BYTES, tid, j); }
    if (kt0 + 1 < kt1) load_tile(ra, rb, kt0 + 1);
    __syncthreads();

    // one 16-deep tile: 24 MFMAs on LDS stage `cur`; (xa, xb) hold tile kt+1 and go to stage cur^1; (ya, yb) receive tile kt+2
    auto tile_step = [&](int kt, int cur, f32x4 (&xa)[2], f32x4 (&xb)[2], f32x4 (&ya)[2], f32x4 (&yb)[2]) {
        const char* As = smem + cur * STAGE_BYTES;
        const char* Bs = As + OP_BYTES;
        char* An = smem + (cur ^ 1) * STAGE_BYTES;
        if (kt + 2 < kt1) load_tile(ya, yb, kt + 2);
        bf16x8 fa[2][NP], fb[2][NP];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#if defined(T2_GEMM_ABL_FRAG)       // diagnostic build (wrong results): half of the fragment reads - what is the LDS read traffic worth?
                if (i == 1) { fa[1][pl] = fa[0][pl]; fb[1][pl] = fb[0][pl]; continue; }
#endif
                fa[i][pl] = split_frag<AK>(As, pl, wm * 64 + i * 32, lane);
                fb[i][pl] = split_frag<BKM>(Bs, pl, wn * 64 + i * 32, lane);
            }
        // small terms first into `lo`, the leading term into `hi`; the next tile's four split units in between
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int i = g >> 1, j = g & 1;
            if constexpr (NP == 3) { T2_SPLIT_MFMA(lo, i, j, 2, 0); T2_SPLIT_MFMA(lo, i, j, 0, 2); T2_SPLIT_MFMA(lo, i, j, 1, 1); }
            if constexpr (NP == 2) { T2_SPLIT_MFMA(lo, i, j, 1, 0); }
            // (unconditional: on the last tile the registers are stale and the stage they go to is never read)
            if (g < 2) split_store_unit<AK, NP>(xa[g], An, tid, g); else split_store_unit<BKM, NP>(xb[g - 2], An + OP_BYTES, tid, g - 2);
            if constexpr (NP == 3) { T2_SPLIT_MFMA(lo, i, j, 1, 0); T2_SPLIT_MFMA(lo, i, j, 0, 1); }
            if constexpr (NP == 2) { T2_SPLIT_MFMA(lo, i, j, 0, 1); }
            T2_SPLIT_MFMA(hi, i, j, 0, 0);
        }
        // order pin, per group of MFMAs + one split unit (NP = 3: 6 MFMAs, 22 VALU, 3 LDS stores): MFMA, a few VALU, MFMA, ...
        constexpr int NMF = NP == 3 ? 6 : NP == 2 ? 3 : 1, NVA = NP == 3 ? 4 : NP == 2 ? 5 : 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int m = 0; m < NMF; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NVA, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x200, NP, 0);
        }
        __syncthreads();     // stage cur^1 is complete, stage cur is free
    };
    for (int kt = kt0; kt < kt1; kt += 2) {
        tile_step(kt, 0, ra, rb, na, nb);
        if (kt + 1 < kt1) tile_step(kt + 1, 1, na, nb, ra, rb);
    }

    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            if (col >= p.N) continue;
            float badd = 0.f;
            if (p.bias && kz == 0) badd += p.bias[col];
            if (p.bias2 && kz == 0) badd += p.bias2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= p.M) continue;
                float v = p.alpha * (NP == 1 ? hi[i][j][r] : hi[i][j][r] + lo[i][j][r]) + badd;
                float* cp = C + (long)row * p.ldc + col;
                if (p.accumulate == 2) {
                    atomicAdd(cp, v);
                } else {
                    if (p.accumulate == 1) v += *cp;
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.mulmask) v *= p.mulmask[(long)row * p.ldmask + col];
                    *cp = v;
                }
            }
        }
    }
    // ---- optional: column statistics of this tile's stored values over its VALID rows (row % stat_Lp < stat_L: the rows of a
    // conv-as-GEMM output that are real positions, not the junk rows that straddle two samples), for the BatchNorm that follows:
    // stat_out[tile_m][0][col] = shift (the column's value at the tile's first valid row), [1] = sum (v - shift), [2] = sum (v - shift)^2.
    // The tile-local shift keeps the sums free of cancellation whatever the channel mean is; t2_bn_fwd merges the tiles in double
    // (Chan's update).  No atomics: every (tile, column) has one writer.
    if (p.stat_out) {
        __syncthreads();                                         // (the staging buffers are free: every wave is past its last read)
        float* sh = reinterpret_cast<float*>(smem);              // [128] shift, then [2 sums][2 wm][128]
        float* red = sh + 128;
        const int Lp = p.stat_Lp, Lv = p.stat_L;
        auto valid = [&](int row) { return row < p.M && (row % Lp) < Lv; };
        int rv = 0;                                              // first valid row of the tile (junk runs are Lp - L rows long)
        while (rv < 127 && !valid(m0 + rv)) ++rv;
        auto value = [&](int i, int j, int r) {
            const int col = n0 + wn * 64 + j * 32 + li;
            float badd = 0.f;
            if (col < p.N) { if (p.bias) badd += p.bias[col]; if (p.bias2) badd += p.bias2[col]; }
            return p.alpha * (NP == 1 ? hi[i][j][r] : hi[i][j][r] + lo[i][j][r]) + badd;
        };
        {   // owner of row rv: wave row wm = rv / 64, block i = (rv % 64) / 32, lane half lh = ((rv % 32) / 4) & 1, reg from the rest
            const int rr = rv & 31, own_wm = rv >> 6, own_i = (rv & 63) >> 5, own_lh = (rr >> 2) & 1, own_reg = (rr & 3) + 4 * (rr >> 3);
            if (wm == own_wm && lh == own_lh) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v = 0.f;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (i == own_i && r == own_reg) v = value(i, j, r);
                    sh[wn * 64 + j * 32 + li] = v;
                }
            }
        }
        __syncthreads();
        // validity of this lane's 32 rows, once: one division for the lane's first row, then offsets < 64 with a single wrap
        // (Lp >= 64; shorter padded rows take the plain modulo)
        unsigned vmask = 0u;
        {
            const int rbase = m0 + wm * 64 + 4 * lh, rem0 = rbase % Lp;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int off = i * 32 + (r & 3) + 8 * (r >> 2);
                    int rem = rem0 + off;
                    if (Lp >= 64) rem = rem >= Lp ? rem - Lp : rem; else rem %= Lp;
                    if (rbase + off < p.M && rem < Lv) vmask |= 1u << (i * 16 + r);
                }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float sft = sh[wn * 64 + j * 32 + li];
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dv = ((vmask >> (i * 16 + r)) & 1u) ? value(i, j, r) - sft : 0.f;
                    a += dv; q = fmaf(dv, dv, q);
                }
            a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);       // the two lane halves hold the other rows of the column
            if (lh == 0) { red[(0 * 2 + wm) * 128 + wn * 64 + j * 32 + li] = a; red[(1 * 2 + wm) * 128 + wn * 64 + j * 32 + li] = q; }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            float* o = p.stat_out + (long)tm * 3 * p.N + n0 + tid;
            o[0] = sh[tid];
            o[p.N] = red[(0 * 2 + 0) * 128 + tid] + red[(0 * 2 + 1) * 128 + tid];
            o[2 * (long)p.N] = red[(1 * 2 + 0) * 128 + tid] + red[(1 * 2 + 1) * 128 + tid];
        }
    }
}

}  // namespace

extern "C" int t2_gemm(const T2Gemm* g, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(g && g->A && g->B && g->C, "t2_gemm: null operand");
    T2_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0, "t2_gemm: empty dims");
    const int splitk = g->splitk > 1 ? g->splitk : 1;
    const int batch = g->batch > 1 ? g->batch : 1;
    T2_REQUIRE(splitk == 1 || g->accumulate == 2, "t2_gemm: splitk>1 requires accumulate=2 (atomic)");
    T2_REQUIRE(g->accumulate != 2 || (!g->relu && !g->mulmask), "t2_gemm: atomic epilogue cannot apply relu/mask");
    T2_REQUIRE(!(g->a_kmajor == 0 && g->b_kmajor == 1), "t2_gemm: (A m-major, B k-major) layout not instantiated");
    GemmK p;
    p.A = g->A; p.B = g->B; p.C = g->C;
    p.M = g->M; p.N = g->N; p.K = g->K;
    p.lda = g->lda; p.ldb = g->ldb; p.ldc = g->ldc;
    p.alpha = g->alpha;
    p.bias = g->bias; p.bias2 = g->bias2; p.mulmask = g->mulmask; p.ldmask = g->ldmask;
    p.relu = g->relu; p.accumulate = g->accumulate; p.splitk = splitk;
    p.sA = g->sA; p.sB = g->sB; p.sC = g->sC;
    p.a_vec = (g->lda % 4 == 0) && t2_aligned16(g->A) && (g->sA % 4 == 0);
    p.b_vec = (g->ldb % 4 == 0) && t2_aligned16(g->B) && (g->sB % 4 == 0);
    p.ntm = t2_cdiv(g->M, BM); p.ntn = t2_cdiv(g->N, BN);
    T2_REQUIRE(g->a_tap_len == 0 || (g->a_kmajor && g->a_tap_len % 32 == 0 && g->a_tap_len > 0 && g->K % g->a_tap_len == 0),
               "t2_gemm: a_tap_len needs a k-major A, a multiple of 32 and K a whole number of taps");
    p.a_tap_len = g->a_tap_len; p.a_tap_stride = (long)g->a_tap_stride;
    if (g->a_tap_len) p.a_vec = p.a_vec && (g->a_tap_stride % 4 == 0);
    T2_REQUIRE(!g->stat_out || (!g->native_fp32 && splitk == 1 && batch == 1 && g->accumulate == 0 && !g->relu && !g->mulmask &&
                                g->stat_Lp >= g->stat_L && g->stat_L >= 1 && g->stat_Lp - g->stat_L < 64),
               "t2_gemm: stat_out needs the split kernel, a plain store (no split-K, batch, accumulate, relu, mask) and 1 <= stat_L <= stat_Lp");
    p.stat_out = g->stat_out; p.stat_Lp = g->stat_Lp; p.stat_L = g->stat_L;
    dim3 grid(p.ntm * p.ntn, 1, batch * splitk), block(256);
    hipStream_t s = (hipStream_t)stream;
    // share_cu: 24 KB of (unused) dynamic LDS on top of the 73.7 KB static tile buffers -> a second workgroup no longer fits
    const size_t pad = g->share_cu ? 24 * 1024 : 0;
    if (!g->native_fp32) {
        const size_t lds = 2 * STAGE_BYTES + pad;    // 72 KB: two workgroups per CU (one with the share_cu padding)
        T2_REQUIRE(g->precision >= 0 && g->precision <= 2, "t2_gemm: precision must be 0 (highest), 1 (high) or 2 (medium)");
#define T2_GEMM_LAUNCH(NP)                                                                                                       \
        do {                                                                                                                     \
            T2_REQUIRE(t2_allow_lds(gemm_f32_split_bf16<true, true, NP>, lds) && t2_allow_lds(gemm_f32_split_bf16<true, false, NP>, lds) && \
                       t2_allow_lds(gemm_f32_split_bf16<false, false, NP>, lds), "t2_gemm: LDS budget");                          \
            if (g->a_kmajor && g->b_kmajor) hipLaunchKernelGGL((gemm_f32_split_bf16<true, true, NP>), grid, block, lds, s, p);    \
            else if (g->a_kmajor && !g->b_kmajor) hipLaunchKernelGGL((gemm_f32_split_bf16<true, false, NP>), grid, block, lds, s, p); \
            else hipLaunchKernelGGL((gemm_f32_split_bf16<false, false, NP>), grid, block, lds, s, p);                             \
        } while (0)
        if (g->precision == 0) T2_GEMM_LAUNCH(3);
        else if (g->precision == 1) T2_GEMM_LAUNCH(2);
        else T2_GEMM_LAUNCH(1);
#undef T2_GEMM_LAUNCH
        T2_CHECK_LAUNCH();
        return T2_OK;
    }
    if (g->a_kmajor && g->b_kmajor) hipLaunchKernelGGL((gemm_f32_mfma<true, true>), grid, block, pad, s, p);
    else if (g->a_kmajor && !g->b_kmajor) hipLaunchKernelGGL((gemm_f32_mfma<true, false>), grid, block, pad, s, p);
    else hipLaunchKernelGGL((gemm_f32_mfma<false, false>), grid, block, pad, s, p);
    T2_CHECK_LAUNCH();
    return T2_OK;
}
