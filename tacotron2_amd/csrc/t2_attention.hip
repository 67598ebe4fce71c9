// Location-sensitive attention step for gfx950 (replaces the ~14 ATen ops of model/attention.py:52-69 and
// the cumulative-weight update of model/decoder.py:78-90).
//
// Each sample's step is spread over many CUs so that the encoder memory (L x Ef fp32, ~330 KB/sample) is
// streamed by 16 workgroups instead of one (per-CU L2/HBM bandwidth is the limit for a single workgroup):
//
//  attn_energy_kernel   grid (B, Ad/16)   workgroup (b, j) owns 16 attention dims for ALL positions l:
//     q[a]      = Wq[a][:] . att_h[b][:]                           (wave-wide dot products, 16-byte loads)
//     loc[l][a] = sum_{c,k} U[a][c][k] * in[c][l+k-15]             (U = location_dense . location_conv folded
//                                                                   once per forward; sliding register window,
//                                                                   4 positions x 62 taps per work item)
//     th[l][a]  = tanh(q[a] + loc[l][a] + pmT[b][a][l]);  partial energy e_j[l] = sum_a v[a] * th[l][a]
//     (processed memory is kept transposed [b][a][l] so these reads are contiguous in l)
//  attn_context_kernel  grid (B, Ef/32)   workgroup (b, s): e[l] = sum_j e_j[l], mask l >= len -> -inf,
//     softmax over l (block reductions, LDS-staged), then its 32-column slice of
//     context[b][e] = sum_l w[l] * memory[b][l][e]  (128-byte coalesced rows), slice 0 also writes the new
//     weights (the alignments row) and cumulative weights.
#include "t2_common.hpp"
#include "t2_lstm_step.hpp"

int t2_lstm_step_fwd_launch(const T2LstmStep* steps, int n, hipStream_t st);
int t2_lstm_step_bwd_launch(const T2LstmBwdStep* steps, int n, hipStream_t st, unsigned long long* clk = nullptr);
void t2_lstm_fwd_advance(T2LstmStep& c, const T2LstmStride& inc);
void t2_lstm_bwd_advance(T2LstmBwdStep& c, const T2LstmBwdStride& inc);

namespace {

constexpr int KL = 31, KPAD = 15;

struct AttnK {
    int B, L, A, Ad, Ef;
    const float* att_h; long ldh;
    const float* Wq; const float* U; const float* v;
    const float* w_prev; long ldw; const float* cum_prev; long ldcum;
    const float* pmT; const float* memory; const int32_t* len;
    float* e_part; float* th_out;
    float* w_out; long ldwo; float* cum_out; long ldco;
    float* ctx_out; long ldctx; float* ctx_out2; long ldctx2;
    float* ctxt_out; int ctxt_col0; long ctxt_cs;
    unsigned long long* clk;   // diagnostic: shader-clock stamps of workgroup (0,0) (T2AttnStep.clk), or null
};
// In-kernel phase stamps exist in the DIAGNOSTIC build only (-DT2_STAMPS, tacotron2_amd/build.py --stamps): each one is a branch
// that ends a basic block, and the instruction scheduler does not move loads or MFMAs of the next phase across it.
#ifdef T2_STAMPS
#define T2_STAMP(p, cond, i) do { if ((p).clk && (cond)) (p).clk[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define T2_STAMP(p, cond, i) do { } while (0)
#endif

// Load discipline for these one-workgroup-per-CU kernels: every global load of a phase is ISSUED (unconditionally, from a
// clamped in-range address) before anything waits on one; out-of-range lanes are zeroed by a select afterwards.  A
// `load -> wait -> use` loop with a run-time trip count costs one full memory round trip (~1 us) per iteration.

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// Stage the haloed location inputs (w_prev, cum_prev) and this slice's folded filter rows into LDS (NTH threads), split
// into an ISSUE half (global loads into registers) and a COMMIT half (LDS writes) so that a kernel can put its other
// loads between the two: everything is in flight before anything is waited for.
template <int NTH>
struct StageRegs { float uv[1024 / NTH]; float iv[1024 / NTH]; };

// DO_INP / DO_U: which of the two images a kernel needs (the matrix-pipe ds kernel takes its filter operand from global memory)
template <int NTH, bool DO_INP = true, bool DO_U = true>
__device__ __forceinline__ void stage_issue(StageRegs<NTH>& r, const float* w_prev, long ldw, const float* cum_prev, long ldcum,
                                            const float* U, const float* dummy, int b, int j, int L, int Lp, int tid, int kpad = KPAD,
                                            int loff = 0 /* text position of image index kpad (position tiles of long texts) */) {
    constexpr int PER = 1024 / NTH;
    const float* wsrc = w_prev ? w_prev + (long)b * ldw : dummy;
    const float* csrc = cum_prev ? cum_prev + (long)b * ldcum : dummy;
    if (DO_U) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {   // Us[al][c][32]: taps 0..30 of channel c, tap 31 = 0 (rows padded for aligned 16-byte reads)
            const int idx = tid + NTH * i, al = idx >> 6, c = (idx >> 5) & 1, k = idx & 31;
            r.uv[i] = U[(long)(j * 16 + al) * 2 * KL + c * KL + imin(k, KL - 1)];
        }
    }
    if (DO_INP) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + NTH * i;
            const int c = idx >= Lp ? 1 : 0, l = idx - c * Lp - kpad + loff;
            const int lc = imin(imax(l, 0), L - 1);
            r.iv[i] = (c ? csrc : wsrc)[lc];
        }
    }
}

constexpr int ENT = 512;   // threads of the energy / ds kernels: two waves per SIMD double the VALU issue rate
constexpr int EMAXI = 2;   // 4-position work items per thread (32 threads per attention dim): 256 positions per pass of the ds kernel

// ---- exact fp32 products on the bf16 matrix pipe (the scheme of csrc/t2_gemm.hip): a = h + m + l with three bf16 terms, six of
//      the nine cross products issued, the small ones into their own accumulator ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
struct Split8 { bf16x8 h, m, l; };
__device__ __forceinline__ unsigned at_pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2v));
}
__device__ __forceinline__ Split8 t2_split8(const float (&v)[8]) {
    u32x4v h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = at_pk_bf16(v[2 * i], v[2 * i + 1]);
        const float r0 = v[2 * i] - __builtin_bit_cast(float, h[i] << 16), r1 = v[2 * i + 1] - __builtin_bit_cast(float, h[i] & 0xffff0000u);
        m[i] = at_pk_bf16(r0, r1);
        const float q0 = r0 - __builtin_bit_cast(float, m[i] << 16), q1 = r1 - __builtin_bit_cast(float, m[i] & 0xffff0000u);
        l[i] = at_pk_bf16(q0, q1);
    }
    Split8 s;
    s.h = __builtin_bit_cast(bf16x8, h); s.m = __builtin_bit_cast(bf16x8, m); s.l = __builtin_bit_cast(bf16x8, l);
    return s;
}
struct Split3 { uint2 h, m, l; };
__device__ __forceinline__ Split3 split3_attn(const f32x4 v) {      // four values -> packed (v0,v1), (v2,v3) of every plane
    Split3 s;
    s.h.x = at_pk_bf16(v[0], v[1]); s.h.y = at_pk_bf16(v[2], v[3]);
    const float r0 = v[0] - __builtin_bit_cast(float, s.h.x << 16), r1 = v[1] - __builtin_bit_cast(float, s.h.x & 0xffff0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, s.h.y << 16), r3 = v[3] - __builtin_bit_cast(float, s.h.y & 0xffff0000u);
    s.m.x = at_pk_bf16(r0, r1); s.m.y = at_pk_bf16(r2, r3);
    const float q0 = r0 - __builtin_bit_cast(float, s.m.x << 16), q1 = r1 - __builtin_bit_cast(float, s.m.x & 0xffff0000u);
    const float q2 = r2 - __builtin_bit_cast(float, s.m.y << 16), q3 = r3 - __builtin_bit_cast(float, s.m.y & 0xffff0000u);
    s.l.x = at_pk_bf16(q0, q1); s.l.y = at_pk_bf16(q2, q3);
    return s;
}
// hi/lo += A (x) B for one K = 32 block of v_mfma_f32_16x16x32_bf16 (lane l: A[row l&15][8(l>>4) + j], B[8(l>>4) + j][col l&15];
// C: col = l&15, row = 4(l>>4) + reg)
__device__ __forceinline__ void t2_mfma6(const Split8& a, const Split8& b, f32x4& hi, f32x4& lo) {
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, hi, 0, 0, 0);
}

// The location convolution of a (sample, 16-dim slice) is a small GEMM: loc[l][a] = sum_{(c,k)} IN[l][(c,k)] * U[a][(c,k)] with the
// Toeplitz operand IN[l][(c,k)] = in[c][l + k - 15] read straight from the haloed input rows in LDS - M = positions (tiles of 16),
// N = the slice's 16 dims, K = 2 x 32 taps - on v_mfma_f32_16x16x32_bf16 with exactly split operands (1.3 us of packed FMAs before).
// A lane ends up with 4 consecutive positions of ONE dim (C layout), the shape the tanh stash and the processed-memory rows are
// accessed in, and the sum over the slice's dims is a 16-lane DPP row reduction - no LDS round trip for the partial energies.
// Eight waves: two position tiles and two query dims per wave for L <= 256 (a 1024-thread build ended each kernel 0.3 us earlier and
// every launch 1.5 us later: sixteen waves per workgroup take that much longer to start and drain).  At most 128 VGPRs: two of
// these waves and one wave of a side-stream GEMM workgroup (up to 256 VGPRs) must fit one SIMD's register file, or the chain
// waits for GEMM tiles to finish (a 148-register build cost 2.3 ms per training step).
//
// Both MFMA operands are split into their three bf16 planes ONCE, by the thread that stages the value, on its way into LDS:
//   inputs   PX[plane][i] = the dword (x[i], x[i+1]) for EVERY i (each value is written as the low half of item i and the high
//            half of item i-1), so a lane's 8 consecutive taps at any offset o are the four dwords PX[o], PX[o+2], PX[o+4], PX[o+6];
//   filters  UX[plane][dim][c][tap] as bf16 with a row stride of 40 dwords: a lane's 8 taps are one 16-byte read, conflict-free
//            over the 16 dims of a lane group.
struct Bf3 { unsigned short h, m, l; };
__device__ __forceinline__ Bf3 t2_split1(float v) {
    Bf3 s;
    const unsigned hh = at_pk_bf16(v, 0.f) & 0xffffu;
    const float r = v - __builtin_bit_cast(float, hh << 16);
    const unsigned mm = at_pk_bf16(r, 0.f) & 0xffffu;
    const float t = r - __builtin_bit_cast(float, mm << 16);
    s.h = (unsigned short)hh; s.m = (unsigned short)mm; s.l = (unsigned short)(at_pk_bf16(t, 0.f) & 0xffffu);
    return s;
}

template <int NTH, bool DO_U = true>
__device__ __forceinline__ void stage_commit_split(const StageRegs<NTH>& r, unsigned* PX, unsigned* UX, const float* w_prev, long ldw,
                                                   const float* cum_prev, long ldcum, const float* dummy, int b, int L, int Lp, int tid,
                                                   int kpad = KPAD, int loff = 0) {
    constexpr int PER = 1024 / NTH;
    const float* wsrc = w_prev ? w_prev + (long)b * ldw : dummy;
    const float* csrc = cum_prev ? cum_prev + (long)b * ldcum : dummy;
    const bool wz = w_prev == nullptr, cz = cum_prev == nullptr;
    unsigned short* px = reinterpret_cast<unsigned short*>(PX);
    unsigned short* ux = reinterpret_cast<unsigned short*>(UX);
    auto put = [&](int idx, float v) {
        const Bf3 s = t2_split1(v);
        px[2 * idx] = s.h; px[2 * (2 * Lp + idx)] = s.m; px[2 * (4 * Lp + idx)] = s.l;
        if (idx > 0) { px[2 * idx - 1] = s.h; px[2 * (2 * Lp + idx) - 1] = s.m; px[2 * (4 * Lp + idx) - 1] = s.l; }
    };
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = tid + NTH * i;
        const int c = idx >= Lp ? 1 : 0, l = idx - c * Lp - kpad + loff;
        const bool ok = l >= 0 && l < L && !(c ? cz : wz);
        if (idx < 2 * Lp) put(idx, ok ? r.iv[i] : 0.f);
    }
    for (int base = 1024; base < 2 * Lp; base += 1024) {   // long texts (2*Lp > 1024): further rounds, load then store
        float iv[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = base + tid + NTH * i;
            const int c = idx >= Lp ? 1 : 0, l = idx - c * Lp - kpad + loff;
            const int lc = imin(imax(l, 0), L - 1);
            iv[i] = (c ? csrc : wsrc)[lc];
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = base + tid + NTH * i;
            const int c = idx >= Lp ? 1 : 0, l = idx - c * Lp - kpad + loff;
            const bool ok = l >= 0 && l < L && !(c ? cz : wz);
            if (idx < 2 * Lp) put(idx, ok ? iv[i] : 0.f);
        }
    }
    if (DO_U) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + NTH * i, al = idx >> 6, c = (idx >> 5) & 1, k = idx & 31;
            const Bf3 s = t2_split1(k < KL ? r.uv[i] : 0.f);
            const int o = al * 80 + c * 32 + k;
            ux[o] = s.h; ux[1280 + o] = s.m; ux[2560 + o] = s.l;
        }
    }
}

__device__ __forceinline__ void attn_energy_body(const AttnK& p, const int b, const int j, float* sm) {
    const int tid = threadIdx.x, lane = tid & 63;
    [[maybe_unused]] const bool stamp = b == 0 && j == 0 && tid == 0;
    T2_STAMP(p, stamp, 0);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int n = lane & 15, q = lane >> 4;                   // MFMA column = dim of the slice (also this lane's A row), row group
    const int a = j * 16 + n;
    const int L = p.L, NG = (L + 3) >> 2, L4 = 4 * NG, NT = (L + 15) >> 4, Lp = 4 * NG + 48;
    float* qs = sm;                                           // [16]
    unsigned* PX = reinterpret_cast<unsigned*>(qs + 16);      // [3][2*Lp]  haloed (w_prev, cum_prev) at index l + 15, neighbour pairs
    unsigned* UX = PX + 6 * Lp;                               // [3][16*40] folded location filter rows of this slice (tap 31 = 0)
    const long rowoff = ((long)b * p.Ad + a) * L;

    // ---- issue first: the haloed location inputs and filter rows (the only loads the convolution waits for) ----
    StageRegs<ENT> sr;
    stage_issue<ENT>(sr, p.w_prev, p.ldw, p.cum_prev, p.ldcum, p.U, p.pmT, b, j, L, Lp, tid);
    // ---- issue: processed-memory values of this wave's first two position tiles (mt = w, w + 8) + v ----
    float pmv[2][4];
    const float va = p.v[a];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int lg = imin(4 * (w + 8 * it) + q, NG - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) pmv[it][i] = p.pmT[rowoff + imin(4 * lg + i, L - 1)];
    }
    // ---- issue: query-projection operands of the first 1024 columns (2 dims per wave, 16-byte loads); consumed after the
    //      convolution, when the 64 KB of Wq rows have arrived ----
    const float* h = p.att_h + (long)b * p.ldh;
    const float* wq0 = p.Wq + (long)(j * 16 + w * 2) * p.A;
    f32x4 hv[4], wv[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = imin(lane * 4 + 256 * i, p.A - 4);
        hv[i] = *reinterpret_cast<const f32x4*>(h + k);
#pragma unroll
        for (int aa = 0; aa < 2; ++aa) wv[aa][i] = *reinterpret_cast<const f32x4*>(wq0 + (long)aa * p.A + k);
    }
    stage_commit_split<ENT>(sr, PX, UX, p.w_prev, p.ldw, p.cum_prev, p.ldcum, p.pmT, b, L, Lp, tid);
    __syncthreads();   // planes visible
    T2_STAMP(p, stamp, 1);
    // ---- fragments: filter rows of this lane's dim (B: taps 8q .. 8q+7 of channel c) and a tile's Toeplitz rows
    //      (A: IN[l = 16 mt + n][(c, 8q + jj)] = in[c][l + 8q + jj - 15] = padded index c*Lp + l + 8q + jj) ----
    Split8 bs[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int o = n * 40 + c * 16 + 4 * q;
        bs[c].h = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(UX + o));
        bs[c].m = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(UX + 640 + o));
        bs[c].l = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(UX + 1280 + o));
    }
    auto conv_tile = [&](int mt) -> f32x4 {
        Split8 as[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned* ap = PX + c * Lp + 16 * mt + n + 8 * q;
            as[c].h = __builtin_bit_cast(bf16x8, (u32x4v){ap[0], ap[2], ap[4], ap[6]});
            as[c].m = __builtin_bit_cast(bf16x8, (u32x4v){ap[2 * Lp], ap[2 * Lp + 2], ap[2 * Lp + 4], ap[2 * Lp + 6]});
            as[c].l = __builtin_bit_cast(bf16x8, (u32x4v){ap[4 * Lp], ap[4 * Lp + 2], ap[4 * Lp + 4], ap[4 * Lp + 6]});
        }
        f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, l0 = h0, l1 = h0, l2 = h0;     // four independent accumulators
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            l0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as[c].l, bs[c].h, l0, 0, 0, 0);
            l1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as[c].h, bs[c].l, l1, 0, 0, 0);
            l2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as[c].m, bs[c].m, l2, 0, 0, 0);
            l0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as[c].m, bs[c].h, l0, 0, 0, 0);
            l1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as[c].h, bs[c].m, l1, 0, 0, 0);
            h0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as[c].h, bs[c].h, h0, 0, 0, 0);
        }
        return h0 + ((l0 + l1) + l2);
    };
    // ---- the convolution of this wave's position tiles on the matrix pipe (independent of the query) ----
    f32x4 loc[2];
    loc[0] = w < NT ? conv_tile(w) : (f32x4){0.f, 0.f, 0.f, 0.f};
    loc[1] = w + 8 < NT ? conv_tile(w + 8) : (f32x4){0.f, 0.f, 0.f, 0.f};
    T2_STAMP(p, stamp, 4);
    // ---- query projection: dot products of the hoisted operands (+ the columns past 1024), wave sums -> qs ----
    {
        float qacc[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float okf = (lane * 4 + 256 * i) < p.A ? 1.f : 0.f;
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
                qacc[aa] += okf * (hv[i][0] * wv[aa][i][0] + hv[i][1] * wv[aa][i][1] + hv[i][2] * wv[aa][i][2] +
                                   hv[i][3] * wv[aa][i][3]);
        }
        for (int k0 = lane * 4 + 1024; k0 < p.A; k0 += 1024) {
            f32x4 hv2[4], wv2[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = imin(k0 + 256 * i, p.A - 4);
                hv2[i] = *reinterpret_cast<const f32x4*>(h + k);
#pragma unroll
                for (int aa = 0; aa < 2; ++aa) wv2[aa][i] = *reinterpret_cast<const f32x4*>(wq0 + (long)aa * p.A + k);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float okf = (k0 + 256 * i) < p.A ? 1.f : 0.f;
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
                    qacc[aa] += okf * (hv2[i][0] * wv2[aa][i][0] + hv2[i][1] * wv2[aa][i][1] + hv2[i][2] * wv2[aa][i][2] +
                                       hv2[i][3] * wv2[aa][i][3]);
            }
        }
        const float sq0 = t2_wave_sum(qacc[0]), sq1 = t2_wave_sum(qacc[1]);
        if (lane == 0) { qs[2 * w] = sq0; qs[2 * w + 1] = sq1; }
        T2_STAMP(p, stamp, 5);
    }
    __syncthreads();   // qs visible
    const float qa = qs[n];
    // this lane: dim n, positions 16 mt + 4q + i
    auto epilogue = [&](int mt, const f32x4 acc, const float (&pm4)[4]) {
        const int lg = 4 * mt + q;
        f32x4 th4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int l = 4 * lg + i;
            if (l < L) th4[i] = t2_tanh(qa + acc[i] + pm4[i]);
        }
        // tanh stash rows are padded to 4*NG floats: one aligned 16-byte store per lane
        if (p.th_out && lg < NG) *reinterpret_cast<f32x4*>(p.th_out + ((long)b * p.Ad + a) * L4 + 4 * lg) = th4;
        // Sum over the slice's 16 dims = the 16 lanes of this row group, for 4 positions at once: two exchange steps leave lane
        // (n & 3) = i with position i's sum over its quad, two rotations add the four quads (4 DPP adds instead of 16).
        const float e0 = va * th4[0], e1 = va * th4[1], e2 = va * th4[2], e3 = va * th4[3];
        const bool b0 = n & 1, b1 = n & 2;
        float x = b0 ? e1 : e0, y = b0 ? e0 : e1;          // keep the value of my parity, hand the other to my neighbour
        float z = b0 ? e3 : e2, u = b0 ? e2 : e3;
        x += t2_dpp<0xB1, 0xf>(0.f, y);                    // quad_perm [1,0,3,2]: lane gets its parity's value from the neighbour
        z += t2_dpp<0xB1, 0xf>(0.f, u);
        float s = b1 ? z : x, t = b1 ? x : z;              // lanes 0,1 keep positions 0,1; lanes 2,3 keep positions 2,3
        s += t2_dpp<0x4E, 0xf>(0.f, t);                    // quad_perm [2,3,0,1]
        s += t2_dpp<0x124, 0xf>(0.f, s);                   // row_ror:4
        s += t2_dpp<0x128, 0xf>(0.f, s);                   // row_ror:8 -> lane n holds position (n & 3)'s sum over all 16 dims
        const int lw = 4 * lg + n;
        if (n < 4 && lw < L) p.e_part[((long)b * (p.Ad >> 4) + j) * L + lw] = s;
    };
    if (w < NT) epilogue(w, loc[0], pmv[0]);
    if (w + 8 < NT) epilogue(w + 8, loc[1], pmv[1]);
    for (int mt = w + 16; mt < NT; mt += 8) {     // long texts (L > 256): further tiles, load then compute
        float pm4[4];
        const int lgc = imin(4 * mt + q, NG - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) pm4[i] = p.pmT[rowoff + imin(4 * lgc + i, L - 1)];
        epilogue(mt, conv_tile(mt), pm4);
    }
    T2_STAMP(p, stamp, 2);
    T2_STAMP(p, stamp, 3);
}

__global__ __launch_bounds__(ENT, 4) void attn_energy_kernel(AttnK p) {
    T2_CHAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    attn_energy_body(p, blockIdx.x, blockIdx.y, sm);
}


__device__ __forceinline__ void attn_context_body(const AttnK& p, const int b, const int es0, float* sm) {
    const int tid = threadIdx.x;
    [[maybe_unused]] const bool stamp = b == 0 && es0 == 0 && tid == 0;
    T2_STAMP(p, stamp, 8);
    const int L = p.L, NA = p.Ad >> 4;
    constexpr int NR = 24;
    const int WS = imax((L + 3) & ~3, 8 * NR);
    float* ws = sm;                       // [max(L rounded to 4, 8*NR)] softmax weights, zero past L
    float* red = ws + WS;                 // [8]
    float* part = red + 8;                // [8][32]
    const int el = tid & 31, lg = tid >> 5;
    if (L + tid < 8 * NR) ws[L + tid] = 0.f;   // zero weights for the register rows past L: the product loop is branch-free
    // ---- issue FIRST what the softmax waits for: the partial energies of position tid (first 8 slices) and the length;
    //      loads return in order, so the 24 memory rows below must not sit in front of them ----
    const int len = p.len[b];
    float ev0[8];
    {
        const int lc = imin(tid, L - 1);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ev0[jj] = p.e_part[((long)b * NA + imin(jj, NA - 1)) * L + lc];
    }
    __builtin_amdgcn_sched_barrier(0);   // keep these loads in front (the scheduler otherwise sinks them behind the rows below)
    // ---- issue: encoder-memory slice of the first 192 positions (independent of the softmax, consumed after it) ----
    const float* mp = p.memory + (long)b * L * p.Ef + es0 + el;
    float mv[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) mv[i] = mp[(long)imin(lg + 8 * i, L - 1) * p.Ef];
    float mx = -INFINITY;
    for (int l0 = 0; l0 < L; l0 += 256) {
        const int l = l0 + tid, lc = imin(l, L - 1);
        float ev[8];
        float e = 0.f;
        for (int j0 = 0; j0 < NA; j0 += 8) {
            if (l0 == 0 && j0 == 0) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) ev[jj] = ev0[jj];
            } else {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) ev[jj] = p.e_part[((long)b * NA + imin(j0 + jj, NA - 1)) * L + lc];
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) e += (j0 + jj) < NA ? ev[jj] : 0.f;
        }
        if (l < L) {
            if (l >= len) e = -INFINITY;
            ws[l] = e;
            mx = fmaxf(mx, e);
        }
    }
    float cprev[2] = {0.f, 0.f};   // previous cumulative weights for the (<= 2) positions this thread writes
    if (es0 == 0 && p.cum_prev) {
#pragma unroll
        for (int r = 0; r < 2; ++r) cprev[r] = p.cum_prev[(long)b * p.ldcum + imin(tid + 256 * r, L - 1)];
    }
    T2_STAMP(p, stamp, 9);
    // Softmax with ONE workgroup exchange: every wave exponentiates against its own maximum and publishes (max, sum); the global
    // maximum M and sum S follow from the four pairs, and a wave's weights are exp(e - m_w) * exp(m_w - M) / S (two block
    // reductions - four barriers - before).  A wave of masked positions only has m_w = -inf: its terms are zero.
    const float mw = t2_wave_max(mx);
    const float mws = mw == -INFINITY ? 0.f : mw;
    float sum = 0.f;
    for (int l = tid; l < L; l += 256) {
        const float pe = expf(ws[l] - mws);
        ws[l] = pe;
        sum += pe;
    }
    sum = t2_wave_sum(sum);
    if ((tid & 63) == 0) { red[tid >> 6] = mw; red[4 + (tid >> 6)] = sum; }
    __syncthreads();
    const float M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float S = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) S += red[4 + ww] * (red[ww] == -INFINITY ? 0.f : expf(red[ww] - M));
    const float scale = (mw == -INFINITY ? 0.f : expf(mw - M)) / S;
    T2_STAMP(p, stamp, 10);
    for (int l = tid, r = 0; l < L; l += 256, ++r) {
        const float wv = ws[l] * scale;
        ws[l] = wv;
        if (es0 == 0) {
            p.w_out[(long)b * p.ldwo + l] = wv;
            if (p.cum_out) {
                const float cp = r < 2 ? cprev[r] : (p.cum_prev ? p.cum_prev[(long)b * p.ldcum + l] : 0.f);
                p.cum_out[(long)b * p.ldco + l] = cp + wv;
            }
        }
    }
    __syncthreads();
    T2_STAMP(p, stamp, 11);
    // all 24 weights are read from LDS before the first use (a guarded `load -> wait -> fma` chain costs 24 LDS latencies)
    float wv24[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) wv24[i] = ws[lg + 8 * i];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) acc = fmaf(wv24[i], mv[i], acc);
#pragma unroll 8
    for (int l = lg + 8 * NR; l < L; l += 8) acc = fmaf(ws[l], mp[(long)l * p.Ef], acc);
    part[lg * 32 + el] = acc;
    T2_STAMP(p, stamp, 12);
    __syncthreads();
    if (tid < 32) {
        float s2 = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s2 += part[g * 32 + tid];
        p.ctx_out[(long)b * p.ldctx + es0 + tid] = s2;
        if (p.ctx_out2) p.ctx_out2[(long)b * p.ldctx2 + es0 + tid] = s2;
        if (p.ctxt_out) { const int col = p.ctxt_col0 + es0 + tid; p.ctxt_out[(long)(col >> 4) * p.ctxt_cs + b * 16 + (col & 15)] = s2; }
    }
    T2_STAMP(p, stamp, 13);
}

__global__ __launch_bounds__(256, 1) void attn_context_kernel(AttnK p) {
    T2_CHAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    attn_context_body(p, blockIdx.x, blockIdx.y * 32, sm);
}

// U[a][c][k] = sum_f Wd[a][f] * Wc[f][c][k]
__global__ void fold_location_kernel(const float* Wd, const float* Wc, float* U, int Ad, int F, int CK) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Ad * CK) return;
    const int a = idx / CK, ck = idx - a * CK;
    float s = 0.f;
    for (int f = 0; f < F; ++f) s = fmaf(Wd[a * F + f], Wc[f * CK + ck], s);
    U[idx] = s;
}

int check_attn(const T2AttnStep& s) {
    T2_REQUIRE(s.B >= 1 && s.L >= 1, "attention: need B >= 1 and L >= 1");
    T2_REQUIRE(s.Kl == KL, "attention: location kernel size must be 31 (model/decoder.py:36)");
    T2_REQUIRE(s.Ad % 16 == 0 && s.Ef % 32 == 0 && s.A % 4 == 0, "attention: need Ad%16==0, Ef%32==0, A%4==0");
    T2_REQUIRE(s.ldh % 4 == 0 && t2_aligned16(s.att_h) && t2_aligned16(s.Wq), "attention: att_h/Wq alignment");
    T2_REQUIRE(s.e_part && s.w_out && s.ctx_out && s.pmT && s.memory && s.len && s.U && s.v, "attention: null operand");
    T2_REQUIRE(!s.th_out || t2_aligned16(s.th_out), "attention: th_out must be 16-byte aligned (rows are padded to 4 floats)");
    return T2_OK;
}

void to_ak(const T2AttnStep& s, AttnK& k) {
    k.B = s.B; k.L = s.L; k.A = s.A; k.Ad = s.Ad; k.Ef = s.Ef;
    k.att_h = s.att_h; k.ldh = s.ldh; k.Wq = s.Wq; k.U = s.U; k.v = s.v;
    k.w_prev = s.w_prev; k.ldw = s.ldw; k.cum_prev = s.cum_prev; k.ldcum = s.ldcum;
    k.pmT = s.pmT; k.memory = s.memory; k.len = s.len; k.e_part = s.e_part; k.th_out = s.th_out;
    k.w_out = s.w_out; k.ldwo = s.ldwo; k.cum_out = s.cum_out; k.ldco = s.ldco;
    k.ctx_out = s.ctx_out; k.ldctx = s.ldctx; k.ctx_out2 = s.ctx_out2; k.ldctx2 = s.ldctx2;
    k.ctxt_out = s.ctxt_out; k.ctxt_col0 = s.ctxt_col0; k.ctxt_cs = (long)((s.B + 15) / 16 * 16) * 16;
    k.clk = (unsigned long long*)s.clk;
}

int launch_attn(const T2AttnStep& s, hipStream_t st) {
    AttnK k;
    to_ak(s, k);
    const int NG = (s.L + 3) >> 2, Lp = 4 * NG + 48;
    const size_t sm_e = (size_t)(16 + 6 * Lp + 3 * 640) * sizeof(float);
    const int wsn = ((s.L + 3) & ~3) > 192 ? ((s.L + 3) & ~3) : 192;
    const size_t sm_c = (size_t)(wsn + 8 + 256) * sizeof(float);
    // (no fixed length limit: what bounds a text is the 160 KB of LDS - 24 bytes per position in the energies kernel)
    T2_REQUIRE(t2_allow_lds(attn_energy_kernel, sm_e) && t2_allow_lds(attn_context_kernel, sm_c),
               "attention: the text is too long for the LDS images of the attention kernels");
    hipLaunchKernelGGL(attn_energy_kernel, dim3(s.B, s.Ad / 16), dim3(ENT), sm_e, st, k);
    hipLaunchKernelGGL(attn_context_kernel, dim3(s.B, s.Ef / 32), dim3(256), sm_c, st, k);
    T2_CHECK_LAUNCH();
    return T2_OK;
}

}  // namespace

int t2_attn_step_launch(const T2AttnStep* s, hipStream_t st) {
    T2_TRY(check_attn(*s));
    return launch_attn(*s, st);
}

extern "C" int t2_attn_fold_location(const float* Wd, const float* Wc, float* U, int Ad, int F, int Kl, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(Wd && Wc && U && Ad > 0 && F > 0 && Kl > 0, "t2_attn_fold_location: bad arguments");
    const int n = Ad * 2 * Kl;
    hipLaunchKernelGGL(fold_location_kernel, dim3(t2_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, Wd, Wc, U, Ad, F,
                       2 * Kl);
    T2_CHECK_LAUNCH();
    return T2_OK;
}

extern "C" int t2_attn_step_fwd(const T2AttnStep* s, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(s != nullptr, "t2_attn_step_fwd: null");
    return t2_attn_step_launch(s, (hipStream_t)stream);
}

// Teacher-forced attention chain over frames [t_begin, t_end): per frame  attention-LSTMCell -> energies -> softmax/context.
extern "C" int t2_attn_seq_fwd(const T2AttnSeq* a, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(a != nullptr, "t2_attn_seq_fwd: null");
    hipStream_t st = (hipStream_t)stream;
    const int B = a->B, L = a->L, T = a->T, A = a->A, Ef = a->Ef, Ad = a->Ad;
    const long ldx = A + Ef;
    const int tb = (a->t_begin == 0 && a->t_end == 0) ? 0 : a->t_begin, te = (a->t_begin == 0 && a->t_end == 0) ? T : a->t_end;
    T2_REQUIRE(tb >= 0 && te <= T && tb <= te, "t2_attn_seq_fwd: bad frame range");
    T2_REQUIRE(!a->xdec_t || (a->wpacked && (A + Ef) % 16 == 0 && A % 16 == 0), "t2_attn_seq_fwd: xdec_t needs wpacked and A, Ef multiples of 16");
    for (int t = tb; t < te; ++t) {
        T2LstmStep s;
        memset(&s, 0, sizeof(s));
        s.B = B; s.H = A;
        const float* slot = a->xdec + (long)t * B * ldx;
        float* slot1 = a->xdec + (long)(t + 1) * B * ldx;
        if (a->wpacked) {   // fast path: the xdec row [att_h | ctx] is ONE K = A+Ef segment against the packed stream
            s.nseg = 1;
            s.seg[0].x = slot; s.seg[0].ldx = ldx; s.seg[0].w = a->W_hh; s.seg[0].ldw = A; s.seg[0].K = A + Ef;
        } else {
            s.nseg = 2;
            s.seg[0].x = slot; s.seg[0].ldx = ldx; s.seg[0].w = a->W_hh; s.seg[0].ldw = A; s.seg[0].K = A;
            s.seg[1].x = slot + A; s.seg[1].ldx = ldx; s.seg[1].w = a->W_ih_ctx; s.seg[1].ldw = a->ld_wih; s.seg[1].K = Ef;
        }
        s.wpacked = a->wpacked;
        const long xts = (long)((A + Ef) / 16) * ((B + 15) / 16 * 16) * 16;   // one x16-tiled slot
        if (a->xdec_t) { s.xt = a->xdec_t + (long)t * xts; s.ht_out = a->xdec_t + (long)(t + 1) * xts; s.ht_col0 = 0; }
        s.pre = a->pre + (long)t * B * 4 * A; s.ldpre = 4 * A;
        s.c_prev = a->att_c + (long)t * B * A; s.ldc_prev = A;
        if (a->att_drop) { s.drop = a->att_drop + (long)t * B * A; s.lddrop = A; }
        s.h_out = slot1; s.ldh = ldx;
        s.c_out = a->att_c + (long)(t + 1) * B * A; s.ldc_out = A;
        if (a->gates) { s.gates_out = a->gates + (long)t * B * 4 * A; s.ldg = 4 * A; }
        T2_TRY(t2_lstm_step_fwd_launch(&s, 1, st));

        T2AttnStep q;
        memset(&q, 0, sizeof(q));
        q.B = B; q.L = L; q.A = A; q.Ad = Ad; q.Ef = Ef; q.Kl = a->Kl;
        q.att_h = slot1; q.ldh = ldx; q.Wq = a->Wq; q.U = a->U; q.v = a->v;
        if (t > 0) { q.w_prev = a->align + (long)(t - 1) * L; q.ldw = (long)T * L; }
        q.cum_prev = a->cum + (long)t * B * L; q.ldcum = L;
        q.pmT = a->pmT; q.memory = a->memory; q.len = a->len; q.e_part = a->e_part;
        if (a->th) q.th_out = a->th + (long)t * B * Ad * ((L + 3) & ~3);
        q.w_out = a->align + (long)t * L; q.ldwo = (long)T * L;
        q.cum_out = a->cum + (long)(t + 1) * B * L; q.ldco = L;
        q.ctx_out = slot1 + A; q.ldctx = ldx;
        if (a->xproj_ctx) { q.ctx_out2 = a->xproj_ctx + (long)t * B * a->ld_xproj; q.ldctx2 = a->ld_xproj; }
        if (a->xdec_t) { q.ctxt_out = a->xdec_t + (long)(t + 1) * xts; q.ctxt_col0 = A; }
        q.clk = a->clk;
        if (t == tb) T2_TRY(check_attn(q));
        T2_TRY(launch_attn(q, st));
    }
    return T2_OK;
}

// =================================================================================================
// Backward through one attention frame (autograd of the kernels above), again spread over many CUs.
//
//  attn_bwd_dw_kernel   grid (B, ceil(L/32)): grad w.r.t. the attention weights and energies.
//     dw[l]  = sum_e dctx[e] * memory[l][e]  (its 32 positions, 128-byte rows, 8 lanes per row)
//     dwx[l] = location-path gradient of w_t  = d_in_{t+1}[0][l] + G_t[l],  G_t = d_in_{t+1}[1] + G_{t+1}
//              (cumulative weights: cum_t = cum_{t-1} + w_t feeds every later frame)
//     softmax backward needs sigma = sum_l w[l]*(dw[l]+dwx[l]) = dctx.context_t + sum_l w[l]*dwx[l],
//     which every workgroup recomputes locally - no cross-workgroup reduction.
//     de[l] = w[l] * (dw[l] + dwx[l] - sigma)        (masked positions have w = 0 -> de = 0)
//  attn_bwd_ds_mfma_kernel / attn_bwd_ds_tiled_kernel   grid (B, Ad/16): workgroup (b, j) owns 16 attention dims for all l.
//     ds[l][a] = de[l] * v[a] * (1 - th^2);  dpmT += ds;  dq[a] = sum_l ds;  dv[a] += sum_l de[l]*th[l][a]
//     dU[a][c][k] += sum_l ds[l][a] * in[c][l+k-15]                   (per-sample partial, summed after the loop)
//     d_in partial [c][l'] = sum_{a in slice,k} ds[l'+15-k][a] * U[a][c][k]  (summed over slices by the next frame)
struct AttnBwdK {
    int B, L, Ad, Ef;
    const float* dctx; long lddctx;
    const float* ctx; long ldctx;
    const float* w; long ldw;
    const float* memory;
    const float* din_part; const float* G_in; float* G_out;
    float* de;
    const float* th; const float* v; const float* U;
    const float* w_prev; long ldwp; const float* cum_prev; long ldcp;
    float* dpmT; float* dq; long lddq; float* dv_part; float* dU_part; float* din_part_out;
    unsigned long long* clk;   // diagnostic stamps (T2AttnSeqBwd.clk) or null
    const unsigned* bd;        // fragment-ready bf16 planes of the d_in filter operand (attn_bwd_prep_kernel)
};

namespace {

__global__ __launch_bounds__(256, 1) void attn_bwd_dw_kernel(AttnBwdK p) {
    T2_CHAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, l0 = blockIdx.y * 32, tid = threadIdx.x;
    const int L = p.L, Ef = p.Ef, NA = p.Ad >> 4;
    constexpr int NEV = 20;                // memory-row items per thread held in registers (covers Ef <= 640)
    const int DS = imax(Ef, 32 * NEV);
    float* dctx_s = sm;                    // [max(Ef, 640)] zero past Ef: the product loop below is branch-free
    float* dwx_s = dctx_s + DS;            // [L rounded]
    float* red = dwx_s + ((L + 3) & ~3);   // [8]
    [[maybe_unused]] const bool stamp = b == 0 && blockIdx.y == 0 && tid == 0;
    T2_STAMP(p, stamp, 16);
    T2_RING_BEGIN(p.clk, stamp, 3);
    const int l = l0 + (tid >> 3), sub = tid & 7;
    // ---- issue FIRST the small loads that depend on the previous launches (upstream context gradient, location-path
    //      partials of frame t+1, weights): loads return in order, so the 80 KB of memory rows must not sit in front ----
    float dv0[4], cv0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = imin(tid + 256 * i, Ef - 1);
        dv0[i] = p.dctx[(long)b * p.lddctx + e];
        cv0[i] = p.ctx[(long)b * p.ldctx + e];
    }
    const int lc0 = imin(tid, L - 1);
    float a0[8], a1[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) { a0[jj] = 0.f; a1[jj] = 0.f; }
    if (p.din_part) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const long o = (((long)b * NA + imin(jj, NA - 1)) * 2) * L + lc0;
            a0[jj] = p.din_part[o];
            a1[jj] = p.din_part[o + L];
        }
    }
    const float gin0 = p.G_in ? p.G_in[(long)b * L + lc0] : 0.f;
    const float wl0 = p.w[(long)b * p.ldw + lc0];
    const float wme = p.w[(long)b * p.ldw + imin(l, L - 1)];
    __builtin_amdgcn_sched_barrier(0);
    // ---- issue: this thread's share of its memory row (8 lanes per position, 16 B each, stride 128 B) ----
    const float* mp = p.memory + ((long)b * L + imin(l, L - 1)) * Ef;
    f32x4 mv[NEV];
#pragma unroll
    for (int i = 0; i < NEV; ++i) mv[i] = *reinterpret_cast<const f32x4*>(mp + imin(sub * 4 + 32 * i, Ef - 4));
    for (int e = Ef + tid; e < DS; e += 256) dctx_s[e] = 0.f;
    // ---- dctx -> LDS, dctx . ctx ----
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i;
        if (e < Ef) { dctx_s[e] = dv0[i]; part = fmaf(dv0[i], cv0[i], part); }
    }
    for (int e0 = 1024; e0 < Ef; e0 += 1024) {
        float dv[4], cv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = imin(e0 + tid + 256 * i, Ef - 1);
            dv[i] = p.dctx[(long)b * p.lddctx + e];
            cv[i] = p.ctx[(long)b * p.ldctx + e];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + tid + 256 * i;
            if (e < Ef) { dctx_s[e] = dv[i]; part = fmaf(dv[i], cv[i], part); }
        }
    }
    // ---- location-path gradient of every position (needed for sigma), partials of frame t+1 ----
    for (int ll0 = 0; ll0 < L; ll0 += 256) {
        const int ll = ll0 + tid, lc = imin(ll, L - 1);
        float g0 = 0.f, g1 = 0.f;
        if (p.din_part) {
            for (int j0 = 0; j0 < NA; j0 += 8) {
                if (ll0 > 0 || j0 > 0) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const long o = (((long)b * NA + imin(j0 + jj, NA - 1)) * 2) * L + lc;
                        a0[jj] = p.din_part[o];
                        a1[jj] = p.din_part[o + L];
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    if (j0 + jj < NA) { g0 += a0[jj]; g1 += a1[jj]; }
            }
        }
        const float gin = ll0 == 0 ? gin0 : (p.G_in ? p.G_in[(long)b * L + lc] : 0.f);
        const float wl = ll0 == 0 ? wl0 : p.w[(long)b * p.ldw + lc];
        if (ll < L) {
            const float Gn = g1 + gin;
            const float dx = g0 + Gn;
            dwx_s[ll] = dx;
            if (blockIdx.y == 0) p.G_out[(long)b * L + ll] = Gn;
            part = fmaf(wl, dx, part);
        }
    }
    T2_STAMP(p, stamp, 17);
    const float sigma = t2_block_sum(part, red);
    T2_STAMP(p, stamp, 18);
    // dw[l] = memory[l] . dctx: all LDS reads of a half are issued before their FMAs (no guard: dctx_s is zero-padded)
    float acc = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 dq4[NEV / 2];
#pragma unroll
        for (int i = 0; i < NEV / 2; ++i) dq4[i] = *reinterpret_cast<const f32x4*>(dctx_s + sub * 4 + 32 * (h * (NEV / 2) + i));
#pragma unroll
        for (int i = 0; i < NEV / 2; ++i) {
            const f32x4 m = mv[h * (NEV / 2) + i];
            acc += m[0] * dq4[i][0] + m[1] * dq4[i][1] + m[2] * dq4[i][2] + m[3] * dq4[i][3];
        }
    }
    if (l < L) {
        for (int e = sub * 4 + 32 * NEV; e < Ef; e += 32) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(mp + e);
            acc += m[0] * dctx_s[e] + m[1] * dctx_s[e + 1] + m[2] * dctx_s[e + 2] + m[3] * dctx_s[e + 3];
        }
    }
    acc = t2_oct_sum(acc);
    if (sub == 0 && l < L) p.de[(long)b * L + l] = wme * (acc + dwx_s[l] - sigma);
    T2_STAMP(p, stamp, 19);
    T2_RING_END();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The per-slice kernel: both correlations on the bf16 matrix pipe (exactly split operands, six products - the scheme of the energies
// kernel).  Phases A / B: ds, dpmT, dq, dv (32 threads per attention dim), then
//   dU   (phase C)  dU[a][(c,k)] += sum_l ds[a][l] * IN[l][(c,k)],  IN[l][(c,k)] = in[c][l + k - 15]: M = the slice's 16 dims,
//                   N = 4 tiles of 16 (c,k) columns, K = positions; 24 (tile, k-step) pairs over 8 waves, the two K halves of a tile
//                   meet in LDS;
//   d_in (phase D)  with l' = 8 m + s:  d_in[c][8m + s] = sum_{a,k'} ds[a][8m + 15 - k'] * U[a][c][k' + s]  - the shift s of the
//                   output position moves into the FILTER operand, so that N = (c, s) fills all 16 MFMA columns (as a plain
//                   Toeplitz GEMM the two channels would use 2 of 16): M = positions / 8, K = 16 dims x 40 shifted taps.  The
//                   filter operand depends only on U: attn_bwd_prep_kernel lays it out once per call as fragment-ready bf16 planes
//                   (p.bd), loaded straight from L2 at kernel entry; the sum over the slice's dims happens in K (no 16-way LDS
//                   reduction), the 8 waves' K shares are summed through LDS.
// ds lives in LDS only as its three bf16 planes, plain bf16 rows DX[plane][dim][x] with ds[l] at x = 25 + l (halo 25: the d_in
// fragments - 8 consecutive elements from x = 8 (m + rb + 1) - and the dU fragments - from x = 32 ks + 8 q - are then aligned
// 16-byte items, one ds_read_b128 per plane, rows an odd number of items apart), written by the threads that compute it; the haloed
// inputs as neighbour pairs with their own halo of 40 (stage_commit_split), so that in[c][l + k - 15] sits at index x + k.
// (1.4 + 1.8 us of packed-FMA loops in rounds 1-2; that kernel was the only path above 252 positions until round 4.)
//
// One pass covers 252 positions.  Longer texts (TILED) are walked in position tiles INSIDE the launch: tile i owns the positions
// [216 i, 216 (i + 1)) - its ds, dpmT, dq, dv, dU terms - and runs the same code on the window that reaches 16 positions further on
// either side, with the energy gradients outside the owned range set to zero: the d_in of a window is then exactly the contribution
// of the tile's ds to the positions it can reach through the 31-tap filter, and the windows' d_in are added up in an LDS image of the
// whole text (8 bytes per position).  dq, dv and the dU accumulators simply run on across the tiles.
constexpr int DSH = 25;      // halo of the ds planes
constexpr int DS_ONE = 252;  // positions of one pass
constexpr int DS_TI = 216;   // positions a tile owns (a multiple of 8: window origins stay aligned for the 16-byte stash reads)
constexpr int DS_MARGIN = 16;

__global__ void attn_bwd_prep_kernel(const float* U, unsigned* bd, int Ad) {
    // bd[((j*20 + ks)*3 + plane)*64 + lane] (16 bytes each): lane (n = (c, s), q) holds B[(a, r = 8 rb + jj)][(c, s)] = U[a][c][32 - r + s]
    // for block 4 ks + q = 5 a + rb (zero outside the 31 taps)
    const int j = blockIdx.x, ks = blockIdx.y, lane = threadIdx.x, n = lane & 15, q = lane >> 4;
    const int blk = 4 * ks + q, al = blk / 5, rb = blk - 5 * al, c = n >> 3, sft = n & 7;
    float v[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int k = 32 - (8 * rb + jj) + sft;
        v[jj] = (k >= 0 && k < KL) ? U[((long)(j * 16 + al) * 2 + c) * KL + k] : 0.f;
    }
    const Split8 sp = t2_split8(v);
    u32x4v* o = reinterpret_cast<u32x4v*>(bd) + ((long)(j * 20 + ks) * 3) * 64 + lane;
    o[0] = __builtin_bit_cast(u32x4v, sp.h); o[64] = __builtin_bit_cast(u32x4v, sp.m); o[128] = __builtin_bit_cast(u32x4v, sp.l);
}

struct DsDims { int NG, L4, M8, MT, KS, S16, LpI; };
__host__ __device__ inline DsDims ds_dims(int L) {
    DsDims d;
    d.NG = (L + 3) >> 2; d.L4 = 4 * d.NG; d.M8 = (L + 7) >> 3; d.MT = (d.M8 + 15) >> 4; d.KS = (L + 31) >> 5;
    int g = d.M8 + 5;                                   // d_in fragments reach item m + rb + 1 <= M8 + 4
    if (4 * d.KS + 4 > g) g = 4 * d.KS + 4;             // dU fragments: items 4 ks + q, ks <= KS
    if ((DSH + d.L4 + 7) / 8 + 1 > g) g = (DSH + d.L4 + 7) / 8 + 1;
    d.S16 = g | 1;                                      // odd row stride (in 16-byte items): the 16 dims of a lane group hit 16 slots
    d.LpI = 32 * d.KS + 96;
    return d;
}

template <bool TILED>
__device__ __forceinline__ void attn_bwd_ds_mfma_body(const AttnBwdK& p, float* sm) {
    const int b = blockIdx.x, j = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    [[maybe_unused]] const bool stamp = b == 0 && j == 0 && tid == 0;
    T2_STAMP(p, stamp, 24);
    T2_RING_BEGIN(p.clk, stamp, 4);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int al = tid >> 5, sub = tid & 31, a = j * 16 + al;
    const int n = lane & 15, q = lane >> 4;
    const int Lg = p.L, Lg4 = (Lg + 3) & ~3;                 // the whole text
    // LDS layout: sized for the longest window of the launch (one pass: the text itself)
    const DsDims dm = ds_dims(TILED ? imin(Lg, DS_TI + 2 * DS_MARGIN) : Lg);
    const int S16 = dm.S16, LpI = dm.LpI;
    u32x4v* DX = reinterpret_cast<u32x4v*>(sm);              // [3][16][S16] items of 8 bf16
    unsigned* PX = reinterpret_cast<unsigned*>(DX + 48 * S16);   // [3][2*LpI] neighbour pairs, in[c][l'] at index l' + 40
    float* des = reinterpret_cast<float*>(PX + 6 * LpI);     // [4 + L4 + 4]: de[l] at index l + 1 (zero at 0, past L and outside the owned range)
    float* redC = des + dm.L4 + 8;                           // [4][256]   phase C: second K half of each (c,k) tile
    float* dinacc = redC + 4 * 256;                          // TILED: [2][Lg4] d_in of this slice over the whole text
    float* red = reinterpret_cast<float*>(DX);               // [8][MT][256] phase D: the waves' K shares (aliases DX after the MFMAs)
    const long rowoff = ((long)b * p.Ad + a) * Lg;
    const float va = p.v[a];
    const float* th_base = p.th + ((long)b * p.Ad + a) * Lg4;
    // phase C result ownership of waves 0..3: tile nt = w, lane holds dims 4q + r, column n -> (c, k)
    const int c_nt = w & 3, c_c = c_nt >> 1, c_k = 16 * (c_nt & 1) + n;
    float sq = 0.f, sv = 0.f;                                // dq, dv of this thread's positions (all tiles)
    f32x4 ch0 = {0.f, 0.f, 0.f, 0.f}, cl0 = ch0, cl1 = ch0, cl2 = ch0;     // dU accumulators (all tiles)
    u32x4v bdv[3][3];
    float dv_old = 0.f, dU_old[4] = {0.f, 0.f, 0.f, 0.f};
    const int ntile = TILED ? (Lg + DS_TI - 1) / DS_TI : 1;
    if (TILED) {
        for (int i = tid; i < 2 * Lg4; i += ENT) dinacc[i] = 0.f;
    }
    for (int tile = 0; tile < ntile; ++tile) {
        // owned positions [a0, b0), window [v0, v1) in text coordinates; everything below runs in window coordinates l = 0 .. L-1
        const int a0 = TILED ? tile * DS_TI : 0, b0 = TILED ? imin(a0 + DS_TI, Lg) : Lg;
        const int v0 = TILED ? imax(a0 - DS_MARGIN, 0) : 0, v1 = TILED ? imin(b0 + DS_MARGIN, Lg) : Lg;
        const int L = v1 - v0, ia = a0 - v0, ib = b0 - v0;
        const DsDims dd = TILED ? ds_dims(L) : dm;
        const int L4 = dd.L4, M8 = dd.M8, MT = dd.MT, KS = dd.KS;
        // position groups of phase A: group g = positions 4g - 1 .. 4g + 2, i.e. elements x = 24 + 4g .. 27 + 4g of a ds row: one aligned
        // 8-byte store per plane (the tanh stash rows are read from 4 g - 1: dword-aligned 16-byte global loads)
        const int NGA = (L + 4) >> 2;                        // groups 0 .. NGA - 1 cover positions -1 .. L - 1 (<= 64 for L <= 252)
        // ---- issue: the energy gradients first (the only operand that depends on the previous launch), tanh stash / old dpmT,
        //      location inputs, old accumulator values, the d_in filter fragments ----
        float dev[TILED ? 1 : 2];
#pragma unroll
        for (int i = 0; i < (TILED ? 1 : 2); ++i) dev[i] = p.de[(long)b * Lg + v0 + imin(tid + ENT * i, L - 1)];
        float thv[EMAXI][4], dpv[EMAXI][4];
        const float* th_row = th_base + v0;                  // (v0 is a multiple of 8: 16-byte items stay aligned)
#pragma unroll
        for (int it = 0; it < EMAXI; ++it) {
            const int g = imin(sub + 32 * it, NGA - 1);
            // positions 4g - 1 .. 4g + 2: one scalar + the aligned quad of positions 4g .. 4g + 3 (clamped; unused elements are guarded)
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(th_row + imin(4 * g, L4 - 4));
            thv[it][0] = th_row[imax(4 * g - 1, 0)];
            thv[it][1] = t4[0]; thv[it][2] = t4[1]; thv[it][3] = t4[2];
        }
#pragma unroll
        for (int it = 0; it < EMAXI; ++it) {                 // old dpmT values of this thread's positions (previous frame's launch: L2)
            const int g = imin(sub + 32 * it, NGA - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) dpv[it][i] = p.dpmT[rowoff + v0 + imin(imax(4 * g - 1 + i, 0), L - 1)];
        }
        if (tile == 0) dv_old = p.dv_part[(long)b * p.Ad + a];
        StageRegs<ENT> sr;
        stage_issue<ENT, true, false>(sr, p.w_prev, p.ldwp, p.cum_prev, p.ldcp, p.U, p.dpmT, b, j, Lg, LpI, tid, 40, v0);
        if (tile == 0) {       // this wave's d_in filter fragments (k-steps w, w + 8, w + 16 < 20): from L2, independent of the chain
            const u32x4v* bdp = reinterpret_cast<const u32x4v*>(p.bd) + (long)j * 20 * 3 * 64 + lane;
#pragma unroll
            for (int ki = 0; ki < 3; ++ki) {
                const int ks = imin(w + 8 * ki, 19);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bdv[ki][pl] = bdp[(ks * 3 + pl) * 64];
            }
        }
        for (int i = tid; i < 48 * S16; i += ENT) DX[i] = (u32x4v){0u, 0u, 0u, 0u};     // halo and tail of the ds planes
        stage_commit_split<ENT, false>(sr, PX, nullptr, p.w_prev, p.ldwp, p.cum_prev, p.ldcp, p.dpmT, b, Lg, LpI, tid, 40, v0);
#pragma unroll
        for (int i = 0; i < (TILED ? 1 : 2); ++i) {
            const int l = tid + ENT * i;
            if (l < dm.L4 + 7) des[l + 1] = (l >= ia && l < ib) ? dev[i] : 0.f;
        }
        if (tid == 0) des[0] = 0.f;
        __syncthreads();
        T2_STAMP(p, stamp, 25);

        // ---- phase A: ds -> its bf16 planes, dpmT accumulation; phase B sums in registers ----
        {
            uint2* dx64 = reinterpret_cast<uint2*>(DX);
#pragma unroll
            for (int it = 0; it < EMAXI; ++it) {
                const int g = sub + 32 * it;
                if (g >= NGA) continue;
                f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
                const f32x4 de4 = *reinterpret_cast<const f32x4*>(des + 4 * g);      // de[4g - 1 .. 4g + 2]
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int l = 4 * g - 1 + i;
                    if (l >= ia && l < ib) {
                        const float th = thv[it][i];
                        d4[i] = de4[i] * va * (1.f - th * th);
                        sv += de4[i] * th;
                        p.dpmT[rowoff + v0 + l] = dpv[it][i] + d4[i];
                    }
                }
                sq += (d4[0] + d4[1]) + (d4[2] + d4[3]);
                const Split3 s3 = split3_attn(d4);     // packed (d0,d1), (d2,d3) of every plane: elements x = 24 + 4g .. 27 + 4g
                const int o = al * S16 * 2 + 6 + g;    // 8-byte items: row base + (24 + 4 g) / 4
                dx64[o] = s3.h; dx64[16 * S16 * 2 + o] = s3.m; dx64[32 * S16 * 2 + o] = s3.l;
            }
        }
        if (!TILED) {       // one pass: the sums are complete - out before the MFMA phases (the cell-backward launch waits for dq)
            sq = t2_half_sum_hi(sq); sv = t2_half_sum_hi(sv);   // totals of the dim's 32 lanes land in its upper 16 lanes
            if (sub == 31) {
                p.dq[(long)b * p.lddq + a] = sq;
                p.dv_part[(long)b * p.Ad + a] = dv_old + sv;
            }
        }
        __syncthreads();
        T2_STAMP(p, stamp, 26);
        if (tile == 0 && w < 4 && c_k < KL) {      // old accumulator values: consumed after the last tile's MFMA phases
#pragma unroll
            for (int r = 0; r < 4; ++r) dU_old[r] = p.dU_part[(((long)b * p.Ad + j * 16 + 4 * q + r) * 2 + c_c) * KL + c_k];
        }

        // ---- phase C: dU tile c_nt, k-steps (w >> 2), +2, +4, ...: A[a = n][x = 32 ks + 8 q + jj] (ds[l = x - 25]),
        //      B[x][(c,k)] = in[c][l + k - 15] = input index x + k ----
        for (int ks = w >> 2; ks < KS + 1; ks += 2) {            // x runs to 25 + L - 1 < 32 (KS + 1)
            const int it = n * S16 + 4 * ks + q;
            Split8 fa, fb;
            fa.h = __builtin_bit_cast(bf16x8, DX[it]); fa.m = __builtin_bit_cast(bf16x8, DX[16 * S16 + it]); fa.l = __builtin_bit_cast(bf16x8, DX[32 * S16 + it]);
            const unsigned* bp = PX + c_c * LpI + 32 * ks + 8 * q + c_k;
            fb.h = __builtin_bit_cast(bf16x8, (u32x4v){bp[0], bp[2], bp[4], bp[6]});
            fb.m = __builtin_bit_cast(bf16x8, (u32x4v){bp[2 * LpI], bp[2 * LpI + 2], bp[2 * LpI + 4], bp[2 * LpI + 6]});
            fb.l = __builtin_bit_cast(bf16x8, (u32x4v){bp[4 * LpI], bp[4 * LpI + 2], bp[4 * LpI + 4], bp[4 * LpI + 6]});
            cl0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.l, fb.h, cl0, 0, 0, 0);
            cl1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.h, fb.l, cl1, 0, 0, 0);
            cl2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.m, fb.m, cl2, 0, 0, 0);
            cl0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.m, fb.h, cl0, 0, 0, 0);
            cl1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.h, fb.m, cl1, 0, 0, 0);
            ch0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.h, fb.h, ch0, 0, 0, 0);
        }
        if (!TILED && w >= 4) {      // one pass: the dU sums are complete - second K half of every (c,k) tile to its partner wave
            const f32x4 cC1 = ch0 + ((cl0 + cl1) + cl2);
#pragma unroll
            for (int r = 0; r < 4; ++r) redC[c_nt * 256 + (4 * q + r) * 16 + n] = cC1[r];
        }
        T2_STAMP(p, stamp, 28);
        // ---- phase D: d_in, this wave's K share (k-steps w, w + 8, w + 16) for the MT row tiles:
        //      A[m][(a, r = 8 rb + jj)] = ds[a][8 m + r - 17] = element x = 8 (m + rb + 1) + jj of dim a ----
        f32x4 cD[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, l0 = h0, l1 = h0, l2 = h0;
            if (mt < MT) {
                const int m = imin(16 * mt + n, M8 - 1);
#pragma unroll
                for (int ki = 0; ki < 3; ++ki) {
                    const int ks = w + 8 * ki;
                    if (ks < 20) {
                        const int blk = 4 * ks + q, ad = blk / 5, rb = blk - 5 * ad;
                        const int it = ad * S16 + m + rb + 1;
                        Split8 fa, fb;
                        fa.h = __builtin_bit_cast(bf16x8, DX[it]); fa.m = __builtin_bit_cast(bf16x8, DX[16 * S16 + it]); fa.l = __builtin_bit_cast(bf16x8, DX[32 * S16 + it]);
                        fb.h = __builtin_bit_cast(bf16x8, bdv[ki][0]); fb.m = __builtin_bit_cast(bf16x8, bdv[ki][1]); fb.l = __builtin_bit_cast(bf16x8, bdv[ki][2]);
                        l0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.l, fb.h, l0, 0, 0, 0);
                        l1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.h, fb.l, l1, 0, 0, 0);
                        l2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.m, fb.m, l2, 0, 0, 0);
                        l0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.m, fb.h, l0, 0, 0, 0);
                        l1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.h, fb.m, l1, 0, 0, 0);
                        h0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.h, fb.h, h0, 0, 0, 0);
                    }
                }
            }
            cD[mt] = h0 + ((l0 + l1) + l2);
        }
        __syncthreads();      // every wave has read its last ds fragment: the planes' memory becomes the reduction buffer
        T2_STAMP(p, stamp, 27);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (mt < MT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((w * MT + mt) * 16 + 4 * q + r) * 16 + n] = cD[mt][r];
            }
        }
        if (!TILED && w < 4 && c_k < KL) {      // dU: first K half (registers) + second (redC, visible since the barrier above)
            const f32x4 cC0 = ch0 + ((cl0 + cl1) + cl2);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p.dU_part[(((long)b * p.Ad + j * 16 + 4 * q + r) * 2 + c_c) * KL + c_k] = dU_old[r] + cC0[r] + redC[c_nt * 256 + (4 * q + r) * 16 + n];
        }
        __syncthreads();
        T2_STAMP(p, stamp, 29);
        for (int o = tid; o < MT * 256; o += ENT) {
            const int mt = o >> 8, ml = (o >> 4) & 15, nn = o & 15;
            float s2 = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) s2 += red[((ww * MT + mt) * 16 + ml) * 16 + nn];
            const int m = 16 * mt + ml, l = 8 * m + (nn & 7);
            if (m < M8 && l < L) {
                if (TILED) dinacc[(nn >> 3) * Lg4 + v0 + l] += s2;      // (one thread per (channel, position) and tile; tiles are sequential)
                else p.din_part_out[(((long)b * (p.Ad >> 4) + j) * 2 + (nn >> 3)) * Lg + l] = s2;
            }
        }
        if (TILED) __syncthreads();     // `red` (the ds planes' memory) is rewritten by the next tile
    }
    // ---- totals over all positions: dq, dv (32 lanes per dim), dU (two K halves of every (c,k) tile) ----
    if (TILED) {
        sq = t2_half_sum_hi(sq); sv = t2_half_sum_hi(sv);
        if (sub == 31) {
            p.dq[(long)b * p.lddq + a] = sq;
            p.dv_part[(long)b * p.Ad + a] = dv_old + sv;
        }
    }
    if (TILED) {
        const f32x4 cC = ch0 + ((cl0 + cl1) + cl2);
        if (w >= 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) redC[c_nt * 256 + (4 * q + r) * 16 + n] = cC[r];
        }
        __syncthreads();
        if (w < 4 && c_k < KL) {      // first K half (registers) + second (redC)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p.dU_part[(((long)b * p.Ad + j * 16 + 4 * q + r) * 2 + c_c) * KL + c_k] = dU_old[r] + cC[r] + redC[c_nt * 256 + (4 * q + r) * 16 + n];
        }
    }
    if (TILED) {
        for (int i = tid; i < 2 * Lg; i += ENT) {
            const int c = i >= Lg ? 1 : 0, l = i - c * Lg;
            p.din_part_out[(((long)b * (p.Ad >> 4) + j) * 2 + c) * Lg + l] = dinacc[c * Lg4 + l];
        }
    }
    T2_STAMP(p, stamp, 30);
    T2_RING_END();
}

// one pass (L <= 252): at most 128 VGPRs, so that two of its waves and a side-stream GEMM wave share a SIMD (section 4.5 of DESIGN.md)
__global__ __launch_bounds__(ENT, 4) void attn_bwd_ds_mfma_kernel(AttnBwdK p) {
    T2_CHAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    attn_bwd_ds_mfma_body<false>(p, sm);
}
// position tiles (L > 252)
__global__ __launch_bounds__(ENT, 2) void attn_bwd_ds_tiled_kernel(AttnBwdK p) {
    T2_CHAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    attn_bwd_ds_mfma_body<true>(p, sm);
}

// dynamic LDS of the per-slice kernel (floats: ds planes | input planes | de | phase-C exchange [| d_in image of the whole text])
size_t ds_mfma_lds(int L) {
    const bool tiled = L > DS_ONE;
    const DsDims dd = ds_dims(tiled ? (L < DS_TI + 2 * DS_MARGIN ? L : DS_TI + 2 * DS_MARGIN) : L);
    size_t f = (size_t)48 * dd.S16 * 4 + 6 * dd.LpI + dd.L4 + 8 + 4 * 256;
    if (tiled) f += (size_t)2 * ((L + 3) & ~3);
    return f * sizeof(float);
}

}  // namespace

extern "C" int t2_attn_seq_bwd(const T2AttnSeqBwd* a, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(a != nullptr, "t2_attn_seq_bwd: null");
    T2_REQUIRE(a->Kl == KL && a->Ad % 16 == 0 && a->Ef % 32 == 0, "t2_attn_seq_bwd: unsupported dims");
    T2_REQUIRE(a->L >= 1, "t2_attn_seq_bwd: need L >= 1");
    T2_REQUIRE(a->ws_bd && a->th, "t2_attn_seq_bwd: the filter workspace ws_bd and the forward's tanh stash th are required");
    hipStream_t st = (hipStream_t)stream;
    const int B = a->B, L = a->L, T = a->T, A = a->A, Ef = a->Ef, Ad = a->Ad, NA = Ad / 16;
    const long ldx = A + Ef;
    const size_t sm_dw = (size_t)((Ef > 640 ? Ef : 640) + ((L + 3) & ~3) + 8) * sizeof(float);
    const bool tiled = L > DS_ONE;
    const size_t sm_dsm = ds_mfma_lds(L);
    T2_REQUIRE(t2_allow_lds(attn_bwd_dw_kernel, sm_dw) && (tiled ? t2_allow_lds(attn_bwd_ds_tiled_kernel, sm_dsm) : t2_allow_lds(attn_bwd_ds_mfma_kernel, sm_dsm)),
               "t2_attn_seq_bwd: the text is too long for the LDS images of the attention backward kernels");
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(NA, 20), dim3(64), 0, st, a->U, reinterpret_cast<unsigned*>(a->ws_bd), Ad);
    T2_REQUIRE(a->wtp_ctx && a->wtp_h, "t2_attn_seq_bwd: packed weight streams (t2_lstm_pack_bwd) are required");
    // Z[s][b] = [ dgates_s (4A) | dq_{s-1} (Ad) ], s = 0..T; slot T's dgates part is zero-filled by the caller, so the
    // backward step of frame t always reads ONE contiguous row Z[t+1] (no special case for the last frame).
    const long ldz = 4 * A + Ad;
    float* Z = a->dgates;
    T2_REQUIRE(a->wtp_q && a->dh_rec, "t2_attn_seq_bwd: wtp_q / dh_rec required");
    T2_REQUIRE(!a->dgates_t || A % 16 == 0, "t2_attn_seq_bwd: dgates_t needs A % 16 == 0");
    const int thi = (a->t_hi == 0 && a->t_lo == 0) ? T : a->t_hi, tlo = (a->t_hi == 0 && a->t_lo == 0) ? 0 : a->t_lo;
    T2_REQUIRE(tlo >= 0 && thi <= T && tlo <= thi, "t2_attn_seq_bwd: bad frame range");
    T2LstmBwdStep s2[2];
    for (int t = thi - 1; t >= tlo; --t) {
        const bool last = (t == T - 1);
        const float* zrow = Z + (long)(t + 1) * B * ldz;
        // (1) ONE launch for both products of dgates_{t+1}: total gradient w.r.t. context_t  and the raw recurrent
        //     gradient dgates_{t+1}.W_hh of att_h_t (192 workgroups instead of 64 + 128 in two dependent launches)
        memset(s2, 0, sizeof(s2));
        T2LstmBwdStep& s = s2[0];
        s.B = B; s.H = A; s.N4 = 4 * A; s.dg_next = zrow; s.lddg = ldz; s.W = a->W_ih_ctx; s.ldw = a->ld_wih;
        s.ncols = Ef; s.epi = 0; s.wtpacked = a->wtp_ctx;
        s.ext1 = a->dctx_ext1 + (long)t * B * a->ld_dc1; s.ldx1 = a->ld_dc1;
        s.ext2 = a->dctx_ext2 + (long)t * B * a->ld_dc2; s.ldx2 = a->ld_dc2;
        s.dx_out = a->dctx_tot + (long)t * B * Ef; s.lddx = Ef;
        const long zts = (long)(4 * A / 16) * ((B + 15) / 16 * 16) * 16;   // one x16-tiled dgates slot
        if (a->dgates_t) s.dgt_next = a->dgates_t + (long)(t + 1) * zts;
        T2LstmBwdStep& r = s2[1];
        r.B = B; r.H = A; r.N4 = 4 * A; r.dg_next = zrow; r.lddg = ldz; r.W = a->W_hh; r.ldw = A;
        r.ncols = A; r.epi = 0; r.wtpacked = a->wtp_h;
        r.ext1 = a->dh_ext + (long)t * B * a->ld_dh; r.ldx1 = a->ld_dh;
        r.dx_out = a->dh_rec; r.lddx = A;
        if (a->dgates_t) r.dgt_next = a->dgates_t + (long)(t + 1) * zts;
        T2_TRY(t2_lstm_step_bwd_launch(s2, 2, st, (unsigned long long*)a->clk));
        // (2),(3) attention backward
        AttnBwdK k;
        memset(&k, 0, sizeof(k));
        k.B = B; k.L = L; k.Ad = Ad; k.Ef = Ef;
        k.dctx = a->dctx_tot + (long)t * B * Ef; k.lddctx = Ef;
        k.ctx = a->xdec + (long)(t + 1) * B * ldx + A; k.ldctx = ldx;
        k.w = a->align + (long)t * L; k.ldw = (long)T * L;
        k.memory = a->memory;
        k.din_part = last ? nullptr : a->din_part;
        k.G_in = last ? nullptr : a->G + (long)((t + 1) & 1) * B * L;
        k.G_out = a->G + (long)(t & 1) * B * L;
        k.de = a->de;
        k.th = a->th + (long)t * B * Ad * ((L + 3) & ~3); k.v = a->v; k.U = a->U;
        if (t > 0) { k.w_prev = a->align + (long)(t - 1) * L; k.ldwp = (long)T * L; }
        k.cum_prev = a->cum + (long)t * B * L; k.ldcp = L;
        k.dpmT = a->dpmT; k.dq = Z + (long)(t + 1) * B * ldz + 4 * A; k.lddq = ldz;
        k.dv_part = a->dv_part; k.dU_part = a->dU_part; k.din_part_out = a->din_part;
        k.clk = (unsigned long long*)a->clk;
        hipLaunchKernelGGL(attn_bwd_dw_kernel, dim3(B, t2_cdiv(L, 32)), dim3(256), sm_dw, st, k);
        k.bd = reinterpret_cast<const unsigned*>(a->ws_bd);
        if (tiled) hipLaunchKernelGGL(attn_bwd_ds_tiled_kernel, dim3(B, NA), dim3(ENT), sm_dsm, st, k);
        else hipLaunchKernelGGL(attn_bwd_ds_mfma_kernel, dim3(B, NA), dim3(ENT), sm_dsm, st, k);
        // (4) attention-LSTM cell backward: dh = (dh_ext + dgates_{t+1}.W_hh) + dq_t.Wq  (short K = Ad product + pointwise)
        T2LstmBwdStep c;
        memset(&c, 0, sizeof(c));
        c.B = B; c.H = A; c.N4 = Ad; c.dg_next = Z + (long)(t + 1) * B * ldz + 4 * A; c.lddg = ldz; c.W = a->Wq; c.ldw = A;
        c.ncols = A; c.epi = 1; c.wtpacked = a->wtp_q;
        c.ext1 = a->dh_rec; c.ldx1 = A;
        if (a->att_drop) { c.drop = a->att_drop + (long)t * B * A; c.lddrop = A; }
        c.gates = a->gates + (long)t * B * 4 * A; c.ldgs = 4 * A;
        c.c_prev = a->att_c + (long)t * B * A; c.ldcp = A;
        c.c_cur = a->att_c + (long)(t + 1) * B * A; c.ldcc = A;
        c.dc = a->dc; c.lddc = A;
        c.dg_out = Z + (long)t * B * ldz; c.ldgo = ldz;
        if (a->dgates_t) c.dgt_out = a->dgates_t + (long)t * zts;
        T2_TRY(t2_lstm_step_bwd_launch(&c, 1, st, (unsigned long long*)a->clk));
    }
    T2_CHECK_LAUNCH();
    return T2_OK;
}
