// Autoregressive decoding (model/tacotron2.py:262-325, teacher_forcing=False): per frame
//   prenet(prev mel) -> attention-LSTMCell -> attention -> decoder-LSTMCell -> mel/stop projection -> stop logic
// as 6 dependent launches with NO host synchronisation inside the loop: the per-group done flags and the emitted frame
// count live on the device (the reference syncs `done.all()` every frame, model/tacotron2.py:321).
//
//   L1  combined linear on xproj_{t-1} = [dec_h | ctx] (K = D+Ef), N = P + M + 1 output columns:
//         columns [0, P)      p1_t = relu((W_pre1 . W_mel) x + W_pre1 . b_mel) * mask1   - the first prenet layer folded
//                             onto the mel projection (two linear maps with nothing in between: pure re-association,
//                             like the location filter fold), so the prenet does not wait for a separate projection launch
//         columns [P, P+M+1)  mel_{t-1}, stop logit_{t-1}                                 - the frame's outputs
//       every sum runs in a fixed order inside one workgroup: the outputs (and the stop decision taken from them) are
//       bit-reproducible (round 1 added K-slice partials with fp32 atomics)
//   L2  second prenet layer -> tiled state; one extra workgroup runs the stop logic of frame t-1
//   L3  attention LSTM cell   L4/L5 attention energies / softmax + context   L6 decoder LSTM cell
// Frame 0 has no L1/L2: the prenet of the all-zero start frame is zero whatever the masks are (no biases, ReLU).
//
// Recurrent state lives in ONE x16-tiled buffer (layout of T2LstmStep.xt) with two ping-pong slots and the column order
//   xs[slot] = [ prenet_out (P) | att_h (A) | ctx (Ef) | dec_h (D) ]
// so that both LSTM cells read ONE contiguous chunk range against their packed weight streams (t2_lstm_pack_fwd):
//   attention cell of frame t : slot t&1,     chunks [0, (P+A+Ef)/16)    = [prenet_t | att_h_{t-1} | ctx_{t-1}]
//   decoder  cell of frame t : slot (t+1)&1, chunks [P/16, (P+A+Ef+D)/16) = [att_h_t  | ctx_t       | dec_h_{t-1}]
// Writers: prenet -> slot t&1; attention cell / context kernel -> slot (t+1)&1; decoder cell -> slot t&1 (read next frame).
// Row-major copies exist only where another kernel needs rows: att_h [B][A] (query projection) and xproj = [dec_h | ctx].
#include "t2_common.hpp"

int t2_lstm_step_fwd_launch(const T2LstmStep* steps, int n, hipStream_t st);
int t2_attn_step_launch(const T2AttnStep* s, hipStream_t st);

namespace {

struct LinK {
    int B, N, K;
    const float* x; long ldx;
    const float* w; long ldw;      // [N][K] row-major
    const float* bias;
    const float* rowterm; long ldrt;   // optional per-row term added before the activation: rowterm[b][n]
    const float* mask; long ldmask;
    int relu;
    int split_n;                   // columns n < split_n: act/mask epilogue into `out`; columns n >= split_n: plain linear into out2
    float* out; long ldo;          // may be null when only the tiled copy is wanted
    float* out2; long ldo2;        // column n lands at out2[b*ldo2 + n - split_n]
    float* out_t; int out_col0; long out_cs;   // optional x16-tiled copy of the `out` columns (chunk stride out_cs floats)
    // optional x16-tiled operands (one 16 x 16 MFMA operand tile = ONE contiguous 1 KB block, as in the LSTM step kernels;
    // row-major rows make every wave-load 16 half cache lines, 3-4x slower through the vector memory pipe):
    const float* wt; long wt_cs;               // weights [K/16][Npad][16], chunk stride wt_cs = Npad*16 floats
    const float* xt0; int nch0;                // activations, chunks [0, nch0): tiled [.][Bp][16] from xt0 ...
    const float* xt1; long xt_cs;              // ... chunks [nch0, K/16): from xt1; chunk stride xt_cs = Bp*16 floats
    // optional stop logic of the PREVIOUS frame, run by one extra workgroup next to the linear (saves one dependent launch per frame)
    const float* stop_proj; long stop_ldp; int stop_M, stop_t;
    int32_t* stop_done; int32_t* stop_state;
};

// model/tacotron2.py:319-322: done[gate < 0] = True; if done.all(): break.  The device keeps the sticky per-utterance flags
// and state = {every utterance of this group has stopped, frames emitted}; the exact break frame and the `lengths` count
// (every emitted frame whose stop logit is >= 0, Appendix C.4) are derived from the stored logits afterwards (stop_scan),
// so a group that is decoded a few frames past the break - the host looks only every `check_every` frames, and with several
// groups every group runs until ALL have stopped, as the reference's single loop does - still reports the reference's values.
__device__ __forceinline__ void stop_logic(const float* proj, long ldp, int M, int B, int t, int32_t* done, int32_t* state,
                                           int* notdone /* LDS */) {
    if (threadIdx.x == 0) *notdone = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        if (proj[(long)b * ldp + M] < 0.f) done[b] = 1;
        if (!done[b]) atomicAdd(notdone, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        state[1] = t + 1;
        if (*notdone == 0) state[0] = 1;
    }
}

// out[b][n] = act(sum_k x[b][k] w[n][k] + bias[n] + rowterm[b][n]) * mask[b][n]; batch rows on the MFMA M axis (MT tiles of
// 16 rows per workgroup, blockIdx.y selects the row block), one 16-column tile per workgroup, K split over the 4 waves
// (K % 16 == 0), all loads of a group issued before the first MFMA wait; partial tiles are summed in fixed wave order.
// NW waves split K (4, or 8 for the long reduction of the combined linear: all of a wave's loads are then ONE round in flight)
template <int MT, int NW>
__global__ __launch_bounds__(64 * NW, 1) void linear_rows_kernel(LinK p) {
    T2_CHAIN_PRIO();
    __shared__ float red[NW * MT * 256];
    __shared__ int notdone;
    if (p.stop_proj && blockIdx.x == gridDim.x - 1) {   // one extra workgroup: the stop logic runs next to the linear, not in front
        if (blockIdx.y == 0) stop_logic(p.stop_proj, p.stop_ldp, p.stop_M, p.B, p.stop_t, p.stop_done, p.stop_state, &notdone);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16, b0 = blockIdx.y * (MT * 16);
    const int nrow = (n0 + r) < p.N ? (n0 + r) : p.N - 1;
    const bool tiled = p.wt != nullptr;
    // tiled: chunk c of the weights at wb + c*wcs, of the activations at (c < nch0 ? xb : xb1) + c'*xcs
    const float* wb = tiled ? p.wt + (long)nrow * 16 + 4 * q : p.w + (long)nrow * p.ldw + 4 * q;
    const long wcs = tiled ? p.wt_cs : 16;
    const float* xb[MT];
    const float* xb1[MT];
    const long xcs = tiled ? p.xt_cs : 16;
    const int nch0 = tiled ? p.nch0 : (p.K >> 4);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = b0 + m * 16 + r;
        xb[m] = tiled ? p.xt0 + (long)row * 16 + 4 * q : p.x + (long)(row < p.B ? row : 0) * p.ldx + 4 * q;
        xb1[m] = tiled ? p.xt1 + (long)row * 16 + 4 * q : xb[m];
    }
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int NT = p.K >> 4;
    // 12 chunks per wave and round: every load of a K <= 768 slice - for the K = 1536 combined linear two rounds - is in
    // flight before the first MFMA waits (these launches are a handful of workgroups deep: latency, not bandwidth; with 4
    // chunks per round the K = 1536 launch took six dependent round trips, 8.7 us)
    constexpr int U = 12;
    for (int c0 = w; c0 < NT; c0 += NW * U) {
        f32x4 bw[U], ax[U][MT];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int c = c0 + NW * j < NT ? c0 + NW * j : NT - 1;
#ifdef T2_NT_WEIGHTS_MT1     // diagnostic build: the linear's weights as a non-temporal stream (A/B at B <= 16: one row block reads them once per frame)
            if (MT == 1) bw[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wb + wcs * c));
            else
#endif
            bw[j] = *reinterpret_cast<const f32x4*>(wb + wcs * c);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                ax[j][m] = *reinterpret_cast<const f32x4*>(c < nch0 ? xb[m] + xcs * c : xb1[m] + xcs * (c - nch0));
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const float okf = (c0 + NW * j) < NT ? 1.f : 0.f;
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
    for (int o = tid; o < MT * 256; o += 64 * NW) {
        const int bl = o >> 4, nl = o & 15, n = n0 + nl, b = b0 + bl;
        if (b < p.B && n < p.N) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) s += red[((ww * MT + (bl >> 4)) * 16 + (bl & 15)) * 16 + nl];
            if (p.bias) s += p.bias[n];
            if (p.rowterm) s += p.rowterm[(long)b * p.ldrt + n];
            if (n >= p.split_n) { p.out2[(long)b * p.ldo2 + (n - p.split_n)] = s; continue; }
            if (p.relu) s = fmaxf(s, 0.f);
            if (p.mask) s *= p.mask[(long)b * p.ldmask + n];
            if (p.out) p.out[(long)b * p.ldo + n] = s;
            if (p.out_t) { const int col = p.out_col0 + n; p.out_t[(long)(col >> 4) * p.out_cs + b * 16 + (col & 15)] = s; }
        }
    }
}

int launch_linear(const LinK& k, hipStream_t st) {
    T2_REQUIRE(k.K % 16 == 0 && (k.wt || (k.ldx % 4 == 0 && k.ldw % 4 == 0 && t2_aligned16(k.x) && t2_aligned16(k.w))),
               "linear rows: K % 16 == 0 and 16-byte aligned operands required");
    T2_REQUIRE(!k.wt || (k.xt0 && k.xt1 && t2_aligned16(k.wt) && t2_aligned16(k.xt0) && t2_aligned16(k.xt1) && k.nch0 >= 0 &&
                         k.nch0 <= (k.K >> 4)), "linear rows: tiled operands");
    T2_REQUIRE(k.B >= 1 && k.B <= 64, "linear rows: 1 <= B <= 64");
    T2_REQUIRE(k.split_n >= k.N || k.out2 != nullptr, "linear rows: out2 required for columns >= split_n");
    T2_REQUIRE(k.out || k.out_t || k.split_n <= 0, "linear rows: no destination for the activated columns");
    // one 16-row tile per workgroup, row blocks over blockIdx.y: these launches have few column tiles (16-22), so the rows
    // are what spreads them over the chip; each workgroup streams its 16 weight rows (re-read from L2 by the other row
    // blocks) and 16 activation rows
    const int ny = t2_cdiv(k.B, 16);
    dim3 grid(t2_cdiv(k.N, 16) + (k.stop_proj ? 1 : 0), ny);
    static const int nw8_min_k = T2_KNOB("T2_LINEAR_NW8_MIN_K", 1024);
    if (k.K >= nw8_min_k) hipLaunchKernelGGL((linear_rows_kernel<1, 8>), grid, dim3(512), 0, st, k);
    else hipLaunchKernelGGL((linear_rows_kernel<1, 4>), grid, dim3(256), 0, st, k);
    T2_CHECK_LAUNCH();
    return T2_OK;
}

__global__ void stop_kernel(const float* proj, long ldp, int M, int B, int t, int32_t* done, int32_t* state) {
    __shared__ int notdone;
    stop_logic(proj, ldp, M, B, t, done, state, &notdone);
}

// Exact break frame and lengths from the stored stop logits of ALL utterances (several groups: proj_g [n][B_g][ldp]):
// first[b] = first frame with logit < 0; the loop breaks after frame n* = max_b first[b] (or runs to `nframes` if some
// utterance never stops); lengths[b] = #{t <= n* : logit[t][b] >= 0}; out2 = {frames emitted, 0}.  One workgroup.
struct StopScan { const float* proj[64]; int Bg[64]; int ng; long ldp; int M; int nframes; int64_t* lengths; int32_t* out2; };
__global__ void stop_scan_kernel(StopScan p) {
    __shared__ int nstar;
    if (threadIdx.x == 0) nstar = 0;
    __syncthreads();
    int Btot = 0;
    for (int g = 0; g < p.ng; ++g) Btot += p.Bg[g];
    for (int bb = threadIdx.x; bb < Btot; bb += blockDim.x) {
        int g = 0, b = bb;
        while (b >= p.Bg[g]) { b -= p.Bg[g]; ++g; }
        int first = p.nframes - 1;     // never stops: the loop runs to the cap
        for (int t = 0; t < p.nframes; ++t)
            if (p.proj[g][((long)t * p.Bg[g] + b) * p.ldp + p.M] < 0.f) { first = t; break; }
        atomicMax(&nstar, first);
    }
    __syncthreads();
    const int n = nstar + 1;
    for (int bb = threadIdx.x; bb < Btot; bb += blockDim.x) {
        int g = 0, b = bb;
        while (b >= p.Bg[g]) { b -= p.Bg[g]; ++g; }
        long cnt = 0;
        for (int t = 0; t < n; ++t) cnt += p.proj[g][((long)t * p.Bg[g] + b) * p.ldp + p.M] >= 0.f ? 1 : 0;
        p.lengths[bb] = cnt;
    }
    if (threadIdx.x == 0) { p.out2[0] = n; p.out2[1] = 0; }
}

}  // namespace

extern "C" int t2_linear_rows(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* mask,
                              int64_t ldmask, int relu, float* out, int64_t ldo, int B, int N, int K, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(x && w && out, "t2_linear_rows: null operand");
    hipStream_t st = (hipStream_t)stream;
    for (int b0 = 0; b0 < B; b0 += 64) {
        LinK k;
        memset(&k, 0, sizeof(k));
        k.B = (B - b0) < 64 ? (B - b0) : 64; k.N = N; k.K = K;
        k.x = x + (long)b0 * ldx; k.ldx = ldx; k.w = w; k.ldw = ldw; k.bias = bias;
        k.mask = mask ? mask + (long)b0 * ldmask : nullptr; k.ldmask = ldmask; k.relu = relu;
        k.out = out + (long)b0 * ldo; k.ldo = ldo; k.split_n = N;
        T2_TRY(launch_linear(k, st));
    }
    return T2_OK;
}

extern "C" int t2_stop_scan(const T2StopScan* s, int64_t* lengths, int32_t* out2, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(s && lengths && out2 && s->ngroups >= 1 && s->ngroups <= 64 && s->nframes >= 1, "t2_stop_scan: bad arguments");
    StopScan p;
    memset(&p, 0, sizeof(p));
    for (int g = 0; g < s->ngroups; ++g) {
        T2_REQUIRE(s->proj[g] && s->Bg[g] >= 1, "t2_stop_scan: bad group");
        p.proj[g] = s->proj[g]; p.Bg[g] = s->Bg[g];
    }
    p.ng = s->ngroups; p.ldp = s->ld_proj; p.M = s->M; p.nframes = s->nframes; p.lengths = lengths; p.out2 = out2;
    hipLaunchKernelGGL(stop_scan_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p);
    T2_CHECK_LAUNCH();
    return T2_OK;
}

extern "C" int t2_decoder_infer(const T2Infer* a, int t0, int t1, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(a != nullptr && t0 >= 0 && t1 >= t0, "t2_decoder_infer: bad arguments");
    T2_REQUIRE(a->B <= 64, "t2_decoder_infer: B <= 64 per call (split larger batches)");
    hipStream_t st = (hipStream_t)stream;
    const int B = a->B, L = a->L, A = a->A, D = a->D, Ef = a->Ef, Ad = a->Ad, P = a->P, M = a->M;
    T2_REQUIRE(P % 16 == 0 && A % 16 == 0 && Ef % 16 == 0 && D % 16 == 0, "t2_decoder_infer: P, A, Ef, D must be multiples of 16");
    T2_REQUIRE(a->W_comb && a->W_pre2 && a->proj && a->xs, "t2_decoder_infer: null operand");
    const long ldp = D + Ef, ldo = a->ld_proj;
    const int Bp = (B + 15) / 16 * 16;
    const long cs = (long)Bp * 16;                               // chunk stride of the tiled state
    const long slot = (long)((P + A + Ef + D) / 16) * cs;        // one tiled slot
    // L1 of frame t: [p1_t | mel_{t-1} | stop logit_{t-1}] from xproj_{t-1}
    auto combined = [&](int t, bool with_mask) -> int {
        LinK k;
        memset(&k, 0, sizeof(k));
        k.B = B; k.N = P + M + 1; k.K = (int)ldp; k.x = a->xproj; k.ldx = ldp; k.w = a->W_comb; k.ldw = ldp; k.bias = a->b_comb;
        if (a->W_comb_t) {
            // tiled operands: weights [ (Ef + D)/16 ][Npad][16] in the K order [ctx | dec_h]; ctx_{t-1} sits in slot t&1 of the tiled
            // state (written by the context kernel of frame t-1), dec_h_{t-1} in slot (t+1)&1 (written by its decoder cell)
            const int Npad = (k.N + 15) / 16 * 16;
            k.wt = a->W_comb_t; k.wt_cs = (long)Npad * 16;
            k.xt0 = a->xs + (long)(t & 1) * slot + (long)((P + A) / 16) * cs; k.nch0 = Ef / 16;
            k.xt1 = a->xs + (long)((t + 1) & 1) * slot + (long)((P + A + Ef) / 16) * cs; k.xt_cs = cs;
        }
        k.rowterm = a->row_comb; k.ldrt = P + M + 1;
        k.mask = (with_mask && a->prenet_mask) ? a->prenet_mask + ((long)t * 2 + 0) * B * P : nullptr;
        k.ldmask = P; k.relu = 1; k.split_n = P;
        k.out = a->p1; k.ldo = P; k.out2 = a->proj + (long)(t - 1) * B * ldo; k.ldo2 = ldo;
        if (a->p1_t) { k.out_t = a->p1_t; k.out_col0 = 0; k.out_cs = cs; }
        return launch_linear(k, st);
    };
    for (int t = t0; t < t1; ++t) {
        float* xs_cur = a->xs + (long)(t & 1) * slot;
        float* xs_nxt = a->xs + (long)((t + 1) & 1) * slot;
        if (t > 0) {
            T2_TRY(combined(t, true));
            LinK k;     // second prenet layer -> tiled state; one extra workgroup: stop logic of frame t-1 (frames of this call only)
            memset(&k, 0, sizeof(k));
            k.B = B; k.N = P; k.K = P; k.x = a->p1; k.ldx = P; k.w = a->W_pre2; k.ldw = P;
            if (a->W_pre2_t && a->p1_t) { k.wt = a->W_pre2_t; k.wt_cs = (long)P * 16; k.xt0 = a->p1_t; k.nch0 = P / 16; k.xt1 = a->p1_t; k.xt_cs = cs; }
            k.mask = a->prenet_mask ? a->prenet_mask + ((long)t * 2 + 1) * B * P : nullptr;
            k.ldmask = P; k.relu = 1; k.split_n = P;
            k.out_t = xs_cur; k.out_col0 = 0; k.out_cs = cs;
            if (t > t0) {
                k.stop_proj = a->proj + (long)(t - 1) * B * ldo; k.stop_ldp = ldo; k.stop_M = M; k.stop_t = t - 1;
                k.stop_done = a->done; k.stop_state = a->state;
            }
            T2_TRY(launch_linear(k, st));
        }
        // attention LSTM cell: [prenet_t | att_h_{t-1} | ctx_{t-1}]  (frame 0: the prenet columns of the zero-filled state)
        T2LstmStep s;
        memset(&s, 0, sizeof(s));
        s.B = B; s.H = A; s.nseg = 1; s.wpacked = a->wp_att;
        s.seg[0].x = a->att_h; s.seg[0].ldx = A; s.seg[0].K = P + A + Ef;   // (row-major operand unused: xt is set)
        s.xt = xs_cur;
        s.bias1 = a->b_att_ih; s.bias2 = a->b_att_hh;
        s.c_prev = a->att_c + (long)(t & 1) * B * A; s.ldc_prev = A;
        s.h_out = a->att_h; s.ldh = A;
        s.ht_out = xs_nxt; s.ht_col0 = P;
        s.c_out = a->att_c + (long)((t + 1) & 1) * B * A; s.ldc_out = A;
        T2_TRY(t2_lstm_step_fwd_launch(&s, 1, st));
        // attention
        T2AttnStep q;
        memset(&q, 0, sizeof(q));
        q.B = B; q.L = L; q.A = A; q.Ad = Ad; q.Ef = Ef; q.Kl = a->Kl;
        q.att_h = a->att_h; q.ldh = A; q.Wq = a->Wq; q.U = a->U; q.v = a->v;
        if (t > 0) { q.w_prev = a->align + (long)(t - 1) * L; q.ldw = (long)a->Tcap * L; }
        q.cum_prev = a->cum + (long)(t & 1) * B * L; q.ldcum = L;
        q.pmT = a->pmT; q.memory = a->memory; q.len = a->len; q.e_part = a->e_part;
        q.w_out = a->align + (long)t * L; q.ldwo = (long)a->Tcap * L;
        q.cum_out = a->cum + (long)((t + 1) & 1) * B * L; q.ldco = L;
        q.ctx_out = a->xproj + D; q.ldctx = ldp;
        q.ctxt_out = xs_nxt; q.ctxt_col0 = P + A;
        T2_TRY(t2_attn_step_launch(&q, st));
        // decoder LSTM cell: [att_h_t | ctx_t | dec_h_{t-1}]
        T2LstmStep d;
        memset(&d, 0, sizeof(d));
        d.B = B; d.H = D; d.nseg = 1; d.wpacked = a->wp_dec;
        d.seg[0].x = a->xproj; d.seg[0].ldx = ldp; d.seg[0].K = A + Ef + D;    // (row-major operand unused: xt is set)
        d.xt = xs_nxt + (long)(P / 16) * cs;
        d.bias1 = a->b_dec_ih; d.bias2 = a->b_dec_hh;
        d.pre = a->dec_pre; d.ldpre = 4 * D;
        d.c_prev = a->dec_c + (long)(t & 1) * B * D; d.ldc_prev = D;
        d.h_out = a->xproj; d.ldh = ldp;
        d.ht_out = xs_cur; d.ht_col0 = P + A + Ef;
        d.c_out = a->dec_c + (long)((t + 1) & 1) * B * D; d.ldc_out = D;
        T2_TRY(t2_lstm_step_fwd_launch(&d, 1, st));
    }
    if (t1 > t0) {
        // the last frame of the call: its outputs (mel, stop logit) and stop logic, so that the host can look at the flags.  The
        // next call's first L1 recomputes the same row bit for bit (and the p1 scratch written here, with its mask).
        T2_TRY(combined(t1, false));
        hipLaunchKernelGGL(stop_kernel, dim3(1), dim3(256), 0, st, a->proj + (long)(t1 - 1) * B * ldo, ldo, M, B, t1 - 1, a->done,
                           a->state);
    }
    T2_CHECK_LAUNCH();
    return T2_OK;
}
