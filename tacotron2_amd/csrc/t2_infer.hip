// Autoregressive decoding (model/tacotron2.py:262-325, teacher_forcing=False): per frame
//   prenet(prev mel) -> attention-LSTMCell -> attention -> decoder-LSTMCell -> mel/stop projection -> stop logic
// as 8 dependent launches with NO host synchronisation inside the loop: `done`, `lengths`, the all-done flag and the
// emitted frame count live on the device (the reference syncs `done.all()` every frame, model/tacotron2.py:321).
//
// State rows are laid out so that each LSTM cell reads ONE contiguous activation segment against its packed weight
// stream (t2_lstm_pack_fwd):   xatt[b] = [att_h | ctx | prenet_out]     xdec[b] = [att_h | ctx | dec_h]
// with two ping-pong slots each; xproj[b] = [dec_h | ctx] feeds the projection.
#include "t2_common.hpp"

int t2_lstm_step_fwd_launch(const T2LstmStep* steps, int n, hipStream_t st);
int t2_attn_step_launch(const T2AttnStep* s, hipStream_t st);

namespace {

struct LinK {
    int B, N, K;
    const float* x; long ldx;
    const float* w; long ldw;      // [N][K] row-major
    const float* bias;
    const float* mask; long ldmask;
    int relu;
    float* out; long ldo;
};

// out[b][n] = act(sum_k x[b][k] w[n][k] + bias[n]) * mask[b][n]; M = batch rows on the MFMA M axis (<= 64), one
// 16-column tile per workgroup, K split over the 4 waves (K % 16 == 0), all loads issued before the first MFMA wait.
template <int MT>
__global__ __launch_bounds__(256, 1) void linear_rows_kernel(LinK p) {
    __shared__ float red[4 * MT * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int nrow = (n0 + r) < p.N ? (n0 + r) : p.N - 1;
    const float* wb = p.w + (long)nrow * p.ldw + 4 * q;
    const float* xb[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m * 16 + r;
        xb[m] = p.x + (long)(row < p.B ? row : 0) * p.ldx + 4 * q;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int NT = p.K >> 4;
    constexpr int U = 4;
    for (int c0 = w; c0 < NT; c0 += 4 * U) {
        f32x4 bw[U], ax[U][MT];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int c = c0 + 4 * j < NT ? c0 + 4 * j : NT - 1;
            bw[j] = *reinterpret_cast<const f32x4*>(wb + 16 * c);
#pragma unroll
            for (int m = 0; m < MT; ++m) ax[j][m] = *reinterpret_cast<const f32x4*>(xb[m] + 16 * c);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const float okf = (c0 + 4 * j) < NT ? 1.f : 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[j][m][s], okf * bw[j][s], acc[m], 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) red[((w * MT + m) * 16 + (q * 4 + g)) * 16 + r] = acc[m][g];
    __syncthreads();
    for (int o = tid; o < MT * 256; o += 256) {
        const int b = o >> 4, nl = o & 15, n = n0 + nl;
        if (b < p.B && n < p.N) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) s += red[((ww * MT + (b >> 4)) * 16 + (b & 15)) * 16 + nl];
            if (p.bias) s += p.bias[n];
            if (p.relu) s = fmaxf(s, 0.f);
            if (p.mask) s *= p.mask[(long)b * p.ldmask + n];
            p.out[(long)b * p.ldo + n] = s;
        }
    }
}

int launch_linear(const LinK& k, hipStream_t st) {
    T2_REQUIRE(k.K % 16 == 0 && k.ldx % 4 == 0 && k.ldw % 4 == 0 && t2_aligned16(k.x) && t2_aligned16(k.w),
               "linear rows: K % 16 == 0 and 16-byte aligned operands required");
    T2_REQUIRE(k.B >= 1 && k.B <= 64, "linear rows: 1 <= B <= 64");
    dim3 grid(t2_cdiv(k.N, 16)), block(256);
    if (k.B <= 16) hipLaunchKernelGGL((linear_rows_kernel<1>), grid, block, 0, st, k);
    else if (k.B <= 32) hipLaunchKernelGGL((linear_rows_kernel<2>), grid, block, 0, st, k);
    else hipLaunchKernelGGL((linear_rows_kernel<4>), grid, block, 0, st, k);
    T2_CHECK_LAUNCH();
    return T2_OK;
}

// model/tacotron2.py:319-322: done[gate < 0] = True; lengths[gate >= 0] += 1; if done.all(): break  (the frame that
// completes `done` is still emitted).  state = {all_done, n_frames}; nothing changes once all_done is set.
__global__ void stop_kernel(const float* proj, long ldp, int M, int B, int t, int32_t* done, int64_t* lengths, int32_t* state) {
    __shared__ int notdone;
    if (threadIdx.x == 0) notdone = 0;
    __syncthreads();
    if (state[0]) return;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float g = proj[(long)b * ldp + M];
        if (g < 0.f) done[b] = 1; else lengths[b] += 1;
        if (!done[b]) atomicAdd(&notdone, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        state[1] = t + 1;
        if (notdone == 0) state[0] = 1;
    }
}

}  // namespace

extern "C" int t2_linear_rows(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* mask,
                              int64_t ldmask, int relu, float* out, int64_t ldo, int B, int N, int K, void* stream) {
    T2_REQUIRE(x && w && out, "t2_linear_rows: null operand");
    hipStream_t st = (hipStream_t)stream;
    for (int b0 = 0; b0 < B; b0 += 64) {
        LinK k;
        k.B = (B - b0) < 64 ? (B - b0) : 64; k.N = N; k.K = K;
        k.x = x + (long)b0 * ldx; k.ldx = ldx; k.w = w; k.ldw = ldw; k.bias = bias;
        k.mask = mask ? mask + (long)b0 * ldmask : nullptr; k.ldmask = ldmask; k.relu = relu;
        k.out = out + (long)b0 * ldo; k.ldo = ldo;
        T2_TRY(launch_linear(k, st));
    }
    return T2_OK;
}

extern "C" int t2_decoder_infer(const T2Infer* a, int t0, int t1, void* stream) {
    T2_REQUIRE(a != nullptr && t0 >= 0 && t1 >= t0, "t2_decoder_infer: bad arguments");
    T2_REQUIRE(a->B <= 64, "t2_decoder_infer: B <= 64 per call (split larger batches)");
    hipStream_t st = (hipStream_t)stream;
    const int B = a->B, L = a->L, A = a->A, D = a->D, Ef = a->Ef, Ad = a->Ad, P = a->P, M = a->M;
    const long lda = A + Ef + P, ldd = A + Ef + D, ldp = D + Ef, ldo = a->ld_proj;
    for (int t = t0; t < t1; ++t) {
        float* xa_cur = a->xatt + (long)(t & 1) * B * lda;
        float* xa_nxt = a->xatt + (long)((t + 1) & 1) * B * lda;
        float* xd_cur = a->xdec + (long)(t & 1) * B * ldd;
        float* xd_nxt = a->xdec + (long)((t + 1) & 1) * B * ldd;
        // prenet on the previous frame (zeros for t = 0), AlwaysDropout masks (model/modules.py)
        const float* prev = t == 0 ? a->zero_frame : a->proj + (long)(t - 1) * B * ldo;
        const long ldprev = t == 0 ? 0 : ldo;
        const float* m1 = a->prenet_mask ? a->prenet_mask + ((long)t * 2 + 0) * B * P : nullptr;
        const float* m2 = a->prenet_mask ? a->prenet_mask + ((long)t * 2 + 1) * B * P : nullptr;
        T2_TRY(t2_linear_rows(prev, ldprev, a->W_pre1, M, nullptr, m1, P, 1, a->p1, P, B, P, M, stream));
        T2_TRY(t2_linear_rows(a->p1, P, a->W_pre2, P, nullptr, m2, P, 1, xa_cur + A + Ef, lda, B, P, P, stream));
        // attention LSTM cell
        T2LstmStep s;
        memset(&s, 0, sizeof(s));
        s.B = B; s.H = A; s.nseg = 1; s.wpacked = a->wp_att;
        s.seg[0].x = xa_cur; s.seg[0].ldx = lda; s.seg[0].K = (int)lda;
        s.bias1 = a->b_att_ih; s.bias2 = a->b_att_hh;
        s.c_prev = a->att_c + (long)(t & 1) * B * A; s.ldc_prev = A;
        s.h_out = xa_nxt; s.ldh = lda; s.h_out2 = xd_cur; s.ldh2 = ldd;
        s.c_out = a->att_c + (long)((t + 1) & 1) * B * A; s.ldc_out = A;
        T2_TRY(t2_lstm_step_fwd_launch(&s, 1, st));
        // attention
        T2AttnStep q;
        memset(&q, 0, sizeof(q));
        q.B = B; q.L = L; q.A = A; q.Ad = Ad; q.Ef = Ef; q.Kl = a->Kl;
        q.att_h = xa_nxt; q.ldh = lda; q.Wq = a->Wq; q.U = a->U; q.v = a->v;
        if (t > 0) { q.w_prev = a->align + (long)(t - 1) * L; q.ldw = (long)a->Tcap * L; }
        q.cum_prev = a->cum + (long)(t & 1) * B * L; q.ldcum = L;
        q.pmT = a->pmT; q.memory = a->memory; q.len = a->len; q.e_part = a->e_part;
        q.w_out = a->align + (long)t * L; q.ldwo = (long)a->Tcap * L;
        q.cum_out = a->cum + (long)((t + 1) & 1) * B * L; q.ldco = L;
        q.ctx_out = xa_nxt + A; q.ldctx = lda; q.ctx_out2 = xd_cur + A; q.ldctx2 = ldd;
        T2_TRY(t2_attn_step_launch(&q, st));
        // decoder LSTM cell: [att_h | ctx | dec_h_prev]
        T2LstmStep d;
        memset(&d, 0, sizeof(d));
        d.B = B; d.H = D; d.nseg = 1; d.wpacked = a->wp_dec;
        d.seg[0].x = xd_cur; d.seg[0].ldx = ldd; d.seg[0].K = (int)ldd;
        d.bias1 = a->b_dec_ih; d.bias2 = a->b_dec_hh;
        d.c_prev = a->dec_c + (long)(t & 1) * B * D; d.ldc_prev = D;
        d.h_out = xd_nxt + A + Ef; d.ldh = ldd; d.h_out2 = a->xproj; d.ldh2 = ldp;
        d.c_out = a->dec_c + (long)((t + 1) & 1) * B * D; d.ldc_out = D;
        T2_TRY(t2_lstm_step_fwd_launch(&d, 1, st));
        // xproj = [dec_h | ctx]: ctx copied by a strided device copy (B x Ef floats)
        if (hipMemcpy2DAsync(a->xproj + D, ldp * sizeof(float), xd_cur + A, ldd * sizeof(float), Ef * sizeof(float), B,
                             hipMemcpyDeviceToDevice, st) != hipSuccess) {
            t2_set_error("hipMemcpy2DAsync failed", __FILE__, __LINE__);
            return T2_ERR_LAUNCH;
        }
        // mel + stop projection, then the device-side stop logic
        float* out = a->proj + (long)t * B * ldo;
        T2_TRY(t2_linear_rows(a->xproj, ldp, a->W_proj, ldp, a->b_proj, nullptr, 0, 0, out, ldo, B, M + 1, (int)ldp, stream));
        hipLaunchKernelGGL(stop_kernel, dim3(1), dim3(256), 0, st, out, ldo, M, B, t, a->done, a->lengths, a->state);
    }
    T2_CHECK_LAUNCH();
    return T2_OK;
}
