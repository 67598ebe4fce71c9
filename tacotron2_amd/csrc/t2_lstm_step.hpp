// Kernel-side operand block of one LSTM cell step, its host conversion, and the device body of a step that is
// CO-SCHEDULED inside another kernel's launch (shared by t2_lstm.hip and t2_attention.hip; no RDC needed).
#pragma once
#include "t2_common.hpp"

struct Seg { const float* x; long ldx; const float* w; long ldw; int K; };
struct LstmK {
    int B, H, nseg;
    Seg seg[3];
    const float* wpacked;               // optional: [H/4][NT][64 lanes][4] lane-contiguous weight stream
    const float* pre; long ldpre;
    const float* bias1; const float* bias2;
    const float* c_prev; long ldc_prev;
    const float* drop; long lddrop;
    float* h_out; long ldh;
    float* h_out2; long ldh2;
    float* c_out; long ldc_out;
    float* gates_out; long ldg;
    const int32_t* len; int t;
    const float* xt; long xt_cs;        // x16-tiled input (chunk stride Bp*16 floats) or null
    float* ht_out; int ht_col0;
};
struct LstmK2 { LstmK s[2]; };

inline int t2_lstm_check_step(const T2LstmStep& s) {
    T2_REQUIRE(s.B >= 1 && s.H >= 4 && s.H % 4 == 0, "lstm step: need B >= 1 and H % 4 == 0");
    T2_REQUIRE(s.nseg >= 0 && s.nseg <= 3, "lstm step: 0..3 input segments");
    for (int i = 0; i < s.nseg; ++i) {
        T2_REQUIRE(s.seg[i].K % 16 == 0 && s.seg[i].K > 0, "lstm step: segment K must be a multiple of 16");
        T2_REQUIRE(s.seg[i].ldx % 4 == 0 && t2_aligned16(s.seg[i].x), "lstm step: segment x must be 16-byte aligned");
        T2_REQUIRE(s.wpacked || (s.seg[i].ldw % 4 == 0 && t2_aligned16(s.seg[i].w)),
                   "lstm step: segment weights must be 16-byte aligned");
    }
    T2_REQUIRE(s.h_out != nullptr, "lstm step: h_out required");
    T2_REQUIRE((!s.xt && !s.ht_out) || (s.wpacked && s.nseg == 1), "lstm step: x16-tiled operands need the packed single-segment path");
    T2_REQUIRE(!s.xt || t2_aligned16(s.xt), "lstm step: xt must be 16-byte aligned");
    T2_REQUIRE(!s.ht_out || s.ht_col0 >= 0, "lstm step: ht_col0 must be >= 0");
    return T2_OK;
}

inline void t2_lstm_to_k(const T2LstmStep& s, LstmK& k, int b0, int bn) {
    k.B = bn; k.H = s.H; k.nseg = s.nseg;
    for (int i = 0; i < 3; ++i) {
        k.seg[i].x = (i < s.nseg && s.seg[i].x) ? s.seg[i].x + (long)b0 * s.seg[i].ldx : nullptr;
        k.seg[i].ldx = s.seg[i].ldx; k.seg[i].w = s.seg[i].w; k.seg[i].ldw = s.seg[i].ldw; k.seg[i].K = s.seg[i].K;
    }
    k.wpacked = s.wpacked;
    k.pre = s.pre ? s.pre + (long)b0 * s.ldpre : nullptr; k.ldpre = s.ldpre;
    k.bias1 = s.bias1; k.bias2 = s.bias2;
    k.c_prev = s.c_prev ? s.c_prev + (long)b0 * s.ldc_prev : nullptr; k.ldc_prev = s.ldc_prev;
    k.drop = s.drop ? s.drop + (long)b0 * s.lddrop : nullptr; k.lddrop = s.lddrop;
    k.h_out = s.h_out + (long)b0 * s.ldh; k.ldh = s.ldh;
    k.h_out2 = s.h_out2 ? s.h_out2 + (long)b0 * s.ldh2 : nullptr; k.ldh2 = s.ldh2;
    k.c_out = s.c_out ? s.c_out + (long)b0 * s.ldc_out : nullptr; k.ldc_out = s.ldc_out;
    k.gates_out = s.gates_out ? s.gates_out + (long)b0 * s.ldg : nullptr; k.ldg = s.ldg;
    k.len = s.len ? s.len + b0 : nullptr; k.t = s.t;
    k.xt_cs = (long)((s.B + 15) / 16 * 16) * 16;
    k.xt = s.xt ? s.xt + (long)b0 * 16 : nullptr;
    k.ht_out = s.ht_out ? s.ht_out + (long)b0 * 16 : nullptr; k.ht_col0 = s.ht_col0;
}


// ------------------------------------------------------------------------------------------------------------------
// Device body of one LSTM cell step on the packed path (256 threads: 4 waves split K; see t2_lstm.hip for the layout).
// It is a function, not a kernel, so that a step can also be CO-SCHEDULED inside another kernel's launch
// (t2_attention.hip: workgroups of the attention-context kernel and workgroups of an independent cell step share one
// launch).  Why: every dependent launch costs ~2.7 us + a memory round trip, kernels of two streams do not overlap at
// this size, and two cells in one launch take the sum of their times (same per-CU miss queue and MFMA pipe) - but a
// cell step next to a latency-bound attention kernel uses otherwise idle pipes (tools/ubench_cell.hip).
template <int MT, int U>
__device__ __forceinline__ void t2_lstm_fwd_fast_body(const LstmK& p, const int bx, float* red /* [4*MT*256] */) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int u0 = bx * 4;
    const int H = p.H;
    const int NT = p.seg[0].K >> 4, NTpad = (NT + 15) & ~15, G = NTpad / (4 * U);   // host guarantees NTpad % (4U) == 0
    const float* wb = p.wpacked + (long)bx * NTpad * 256 + lane * 4;
    // x16-tiled input: chunk c, tile m is one contiguous 1 KB block (8 full cache lines per wave-load); row-major
    // input: 16 rows x 64 B per wave-load (16 half lines - 3-4x slower through the vector memory pipe)
    const float* xb[MT];
    const long xcs = p.xt ? p.xt_cs : 16;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m * 16 + r;
        xb[m] = p.xt ? p.xt + (long)row * 16 + 4 * q : p.seg[0].x + (long)(row < p.B ? row : 0) * p.seg[0].ldx + 4 * q;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Epilogue operands (hoisted: issued now, consumed after the GEMM, so they never add a memory round trip)
    const int eb = tid >> 2, euu = tid & 3, eu = u0 + euu;
    const long ebc = eb < p.B ? eb : p.B - 1;
    float e_pre[4] = {0.f, 0.f, 0.f, 0.f}, e_b1[4] = {0.f, 0.f, 0.f, 0.f}, e_b2[4] = {0.f, 0.f, 0.f, 0.f};
    float e_cp = 0.f, e_drop = 1.f;
    int e_len = 0x7fffffff;
    if (p.pre) {
#pragma unroll
        for (int g = 0; g < 4; ++g) e_pre[g] = p.pre[ebc * p.ldpre + g * H + eu];
    }
    if (p.bias1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) e_b1[g] = p.bias1[g * H + eu];
    }
    if (p.bias2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) e_b2[g] = p.bias2[g * H + eu];
    }
    if (p.c_prev) e_cp = p.c_prev[ebc * p.ldc_prev + eu];
    if (p.drop) e_drop = p.drop[ebc * p.lddrop + eu];
    if (p.len) e_len = p.len[ebc];
    auto load_chunk = [&](int g, int j, f32x4& bw, f32x4 (&ax)[MT]) {
        const int c = 4 * U * g + 4 * j + w;
        const int cx = c < NT ? c : NT - 1;     // padding chunks: any finite activations x the zero weight chunk
        bw = *reinterpret_cast<const f32x4*>(wb + (long)c * 256);
#pragma unroll
        for (int m = 0; m < MT; ++m) ax[m] = *reinterpret_cast<const f32x4*>(xb[m] + xcs * cx);
    };
    auto mma_chunk = [&](const f32x4& bw, const f32x4 (&ax)[MT]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[m][s], bw[s], acc[m], 0, 0, 0);
    };
    // Software pipeline at chunk granularity: the loads of chunk j of group g+1 are issued right before the MFMAs of
    // chunk j of group g, so the vector-memory pipe and the MFMA pipe are busy at the same time (a wave that issues a
    // whole group of loads and then a whole group of MFMAs alternates between the two: 6.2 -> 5.3 us per step at
    // K = 1536, tools/ubench_cell.hip).  ~U*(1+MT) loads stay in flight; sched_barrier pins the order, the loads are
    // unconditional so the compiler's counted vmcnt waits stay exact.
    auto pipe_group = [&](int gl, f32x4 (&bwL)[U], f32x4 (&axL)[U][MT], const f32x4 (&bwM)[U], const f32x4 (&axM)[U][MT]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            load_chunk(gl, j, bwL[j], axL[j]);
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk(bwM[j], axM[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        f32x4 bwA[U], bwB[U], axA[U][MT], axB[U][MT];
#pragma unroll
        for (int j = 0; j < U; ++j) load_chunk(0, j, bwA[j], axA[j]);
        int g = 0;
        for (; g + 2 < G; g += 2) {
            pipe_group(g + 1, bwB, axB, bwA, axA);
            pipe_group(g + 2, bwA, axA, bwB, axB);
        }
        if (g + 1 < G) {
            pipe_group(g + 1, bwB, axB, bwA, axA);
#pragma unroll
            for (int j = 0; j < U; ++j) mma_chunk(bwB[j], axB[j]);
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) mma_chunk(bwA[j], axA[j]);
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) red[((w * MT + m) * 16 + (q * 4 + g)) * 16 + r] = acc[m][g];
    __syncthreads();
    if (tid < MT * 64) {
        const int b = eb, uu = euu;
        if (b < p.B) {
            const int u = eu;
            float gsum[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) s += red[((ww * MT + (b >> 4)) * 16 + (b & 15)) * 16 + g * 4 + uu];
                gsum[g] = s + e_pre[g] + e_b1[g] + e_b2[g];
            }
            const bool active = p.t < e_len;
            float gi = t2_sigmoid(gsum[0]), gf = t2_sigmoid(gsum[1]), gg = t2_tanh(gsum[2]), go = t2_sigmoid(gsum[3]);
            float cn = gf * e_cp + gi * gg;
            float hn = go * t2_tanh(cn) * e_drop;
            if (!active) { hn = 0.f; cn = 0.f; gi = gf = gg = go = 0.f; }
            p.h_out[(long)b * p.ldh + u] = hn;
            if (p.h_out2) p.h_out2[(long)b * p.ldh2 + u] = hn;
            if (p.ht_out) { const int col = p.ht_col0 + u; p.ht_out[(long)(col >> 4) * p.xt_cs + b * 16 + (col & 15)] = hn; }
            if (p.c_out) p.c_out[(long)b * p.ldc_out + u] = cn;
            if (p.gates_out) {
                float* go_ = p.gates_out + (long)b * p.ldg + u;
                go_[0] = gi; go_[H] = gf; go_[2 * H] = gg; go_[3 * H] = go;
            }
        }
    }
}
