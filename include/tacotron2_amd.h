/* tacotron2_amd.h - C ABI of libtacotron2_amd.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * Tacotron 2 hot path of mattm458/tacotron2.
 *
 * The reference has NO native/FFI boundary: its hot path is nn.Module composition
 * (model/encoder.py, model/attention.py, model/decoder.py, model/postnet.py, model/tacotron2.py) and every
 * device kernel is an implicit ATen call.  This header is therefore the boundary a maintainer would bind
 * (ctypes) from those modules' forward()s; each entry cites the reference lines whose
 * ATen sequence it replaces.
 *
 * Conventions
 *  - plain C: raw DEVICE pointers, ints, floats; no torch / C++ types.  `stream` is a hipStream_t passed as
 *    void* (NULL = default stream).  Kernels are enqueued on it; nothing here synchronises or allocates
 *    (workspaces are passed in), so every entry is safe inside stream capture and re-entrant per stream.
 *  - every function returns 0 on success, a T2_ERR_* code otherwise, and never throws; t2_last_error()
 *    returns a human-readable message for the calling thread.
 *  - all floating point is fp32 (north_star), row-major, channel-last: (B,L,C) / (B,T,C); sequence stashes
 *    used by the frame loop are time-major [t][b][...].  LSTM gate order is PyTorch's i,f,g,o and weight
 *    matrices are [out][in] exactly as in the reference state_dict.
 *  - dropout is always an explicit *scale mask* tensor (0 or 1/(1-p)); NULL = identity.  Masks come from
 *    t2_philox_mask in production or from the caller in parity tests (SURVEY.md section 7 "Dropout RNG parity").
 */
#ifndef TACOTRON2_AMD_H
#define TACOTRON2_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define T2_OK 0
#define T2_ERR_ARG 1     /* bad argument / unsupported shape */
#define T2_ERR_LAUNCH 2  /* HIP launch or runtime error */
#define T2_ERR_RESIDENCY 3  /* a persistent launch would not be fully co-resident on this device (caller: use the per-step launches) */

const char* t2_last_error(void);
int t2_version(void);
int t2_sizeof(const char* struct_name); /* sizeof of an ABI struct by name, -1 if unknown */
/* diagnostic: enable/disable in-kernel clock stamps of the packed LSTM step kernel and read back 8 words: [0],[1] =
 * (s_memtime, s_memrealtime) at kernel entry of workgroup 0, [6],[7] = the same at its exit; [2..5] unused */
int t2_debug_clock(int enable, uint64_t* out8);

/* ------------------------------------------------------------------------------------------------
 * Generic fp32 MFMA GEMM:  C[M,N] (op)= alpha * A[M,K] x B[K,N]  (+ bias[n] + bias2[n]) (relu) (* mask[m,n])
 * Used for every dense contraction of the path: nn.Linear / nn.Conv1d (as GEMM over overlapping
 * channel-last rows) forward, dgrad and wgrad (model/tacotron2.py:85-92,229; model/encoder.py:33-39;
 * model/postnet.py:9-15; model/decoder.py:26-51 hoisted input projections).
 *   a_kmajor=1: A element (m,k) at A[m*lda + k];  a_kmajor=0: at A[k*lda + m]
 *   b_kmajor=1: B element (k,n) at B[n*ldb + k];  b_kmajor=0: at B[k*ldb + n]
 *   accumulate: 0 store, 1 C += result (plain), 2 C += result (fp32 atomics; required when splitk > 1)
 *   batch > 1: blockIdx.z strides A,B,C by sA,sB,sC elements.
 */
typedef struct {
    const float* A; const float* B; float* C;
    int M, N, K;
    int64_t lda, ldb, ldc;
    int a_kmajor, b_kmajor;
    float alpha;
    const float* bias; const float* bias2;
    const float* mulmask; int64_t ldmask;
    int relu;
    int accumulate;
    int splitk;
    int batch; int64_t sA, sB, sC;
    int share_cu;                    /* hint: 1 = keep ONE workgroup per CU (extra dynamic LDS), leaving LDS for the small kernels
                                        of a concurrent stream; two 73 KB workgroups per CU otherwise lock them out */
    int native_fp32;                 /* 0 (default): fp32 result on the bf16 matrix pipe by error-free 3-way operand splitting
                                        (6 exact bf16 products per element pair, two fp32 accumulators; csrc/t2_gemm.hip);
                                        1: v_mfma_f32_32x32x2_f32 (f32-input MFMA, 1/16 of the bf16 rate) */
    int a_tap_len; int64_t a_tap_stride;   /* optional (a_kmajor = 1): the K axis of A is a_tap_len-long blocks that are a_tap_stride
                                        elements apart: element (m, k) at A[m*lda + (k / a_tap_len)*a_tap_stride + k % a_tap_len].  A DILATED
                                        Conv1d(k, d) on channel-last rows is then ONE GEMM (tap block j = the Ci channels of row
                                        m + j*d: a_tap_len = Ci, a_tap_stride = d*Ci, lda = Ci) instead of k accumulating ones
                                        (model/hifi_gan.py:60-87).  a_tap_len must be a multiple of 32; 0 = plain rows */
    int precision;                   /* the reference's training.float32_matmul_precision (run/train.py:170), split kernel only:
                                        0 = "highest" (default): six bf16 products per element pair, fp32-exact operands;
                                        1 = "high": three products (a1b1 + a1b2 + a2b1, "bf16x3" in torch's terms, ~16
                                            significand bits); 2 = "medium": one bf16 product */
    float* stat_out; int stat_Lp, stat_L;  /* optional (split kernel, plain store): per-tile column statistics of the stored values for
                                        the BatchNorm that follows a conv-as-GEMM (model/encoder.py:33-41, model/postnet.py:9-16):
                                        stat_out[ceil(M/128)][3][N] = {shift, sum (v - shift), sum (v - shift)^2} over the tile's rows
                                        with row % stat_Lp < stat_L (real positions of the padded row layout); shift = the column's
                                        value at the tile's first such row.  t2_bn_fwd(tile_stats = ...) merges them (no atomics, no
                                        extra pass over the activations).  NULL = off */
} T2Gemm;
int t2_gemm(const T2Gemm* g, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One LSTM-cell step (nn.LSTMCell / one time step of nn.LSTM), M = batch rows on the MFMA M axis:
 *   gates[b][:] = pre[b][:] + bias1 + bias2 + sum_s x_s[b][0:K_s] . W_s[row][0:K_s]      (i,f,g,o blocks of H)
 *   c' = f*c + i*g ; h' = o*tanh(c') ; h' *= drop[b][:]         (the dropped h is the carried h,
 *                                                                model/decoder.py:75,101,114,118)
 * Replaces model/decoder.py:70-75 (att_rnn), :94-101 (lstm) and one step of model/encoder.py:64.
 * `len`/`t` (optional): rows with t >= len[b] are inactive: h' = c' = 0 (packed-sequence semantics,
 * model/encoder.py:61-65).
 */
typedef struct { const float* x; int64_t ldx; const float* w; int64_t ldw; int K; } T2Seg;
typedef struct {
    int B, H;
    int nseg; T2Seg seg[3];
    const float* wpacked;            /* optional lane-contiguous copy of the segment weights (t2_lstm_pack_fwd) */
    const float* pre; int64_t ldpre;
    const float* bias1; const float* bias2;
    const float* c_prev; int64_t ldc_prev;
    const float* drop; int64_t lddrop;
    float* h_out; int64_t ldh;       /* post-dropout h */
    float* h_out2; int64_t ldh2;     /* optional second copy (NULL to skip) */
    float* c_out; int64_t ldc_out;
    float* gates_out; int64_t ldg;   /* optional stash of the activated gates for backward, gate-interleaved: row b holds
                                        [u][4] = (i, f, g, o) of unit u (16-byte items; ldg >= 4H, 16-byte aligned rows) */
    const int32_t* len; int t;       /* optional activity predicate */
    /* x16-tiled operands (packed path only).  A (B x K) matrix in x16 layout is stored [K/16][Bp][16] with
     * Bp = round_up(B,16): one 16-row x 16-column MFMA operand tile is ONE contiguous 1 KB block, so a wave-load reads
     * 8 full cache lines instead of 16 half lines (measured 3.4 us per step at K = 1536, tools/ubench_cell.hip).
     * xt: tiled copy of the single input segment (replaces seg[0].x; rows >= B must be finite);
     * ht_out: tiled copy of h written at columns [ht_col0, ht_col0 + H) of a tiled matrix with the same Bp. */
    const float* xt;
    float* ht_out; int ht_col0;
} T2LstmStep;
/* n = 1 or 2 independent cells in one launch (the two directions of the encoder BiLSTM). */
int t2_lstm_step_fwd(const T2LstmStep* steps, int n, void* stream);
/* Weight streams re-laid once per optimisation step so that every wave-instruction of the step kernels reads one
 * contiguous 1 KB block (16 B per lane) instead of 16 rows with a power-of-two stride:
 *   fwd: out[H/4][NTpad][64][4],  NT = sum_s K_s/16;   bwd: out[ceil(ncols/16)][NCHpad][64][4] (transposed),
 *   NCH = N4/16 + N2/16; chunk counts are zero-padded (fwd: multiple of 16, bwd: multiple of 32) so the kernels' main
 *   loops are branch-free. */
int t2_lstm_pack_fwd(const T2Seg* segs, int nseg, int H, float* out, void* stream);
int t2_lstm_pack_bwd(const float* W, int64_t ldw, int N4, const float* W2, int64_t ldw2, int N2, int ncols, float* out,
                     void* stream);

/* S consecutive steps of a recurrence: step s uses base[i] with every non-NULL pointer advanced by
 * s * inc[i].<field> ELEMENTS (negative = time-descending, the reverse BiLSTM direction) and t += s*dt.
 * Replaces the packed nn.LSTM of model/encoder.py:59-65 and, with hoisted input projections, the
 * decoder-LSTMCell recurrence of model/decoder.py:94-101 over all frames (model/tacotron2.py:276-317). */
typedef struct {
    int64_t seg_x[3]; int64_t pre, c_prev, drop, h_out, h_out2, c_out, gates_out; int dt;
    int64_t xt, ht_out;
} T2LstmStride;
int t2_lstm_seq_fwd(const T2LstmStep* base, const T2LstmStride* inc, int n, int S, void* stream);
/* The same S steps of ONE cell as ONE persistent, weight-stationary launch (csrc/t2_lstm.hip): every workgroup keeps its
 * weight slice in LDS and the workgroups exchange h_s through the x16-tiled stash (write-through stores, sharded arrival
 * counters, sc1 loads), so the launch can run on its own stream next to other work without streaming weights every step.
 * Needs: packed single-segment path with K = H, `pre` (hoisted input projection), B <= 64 (rows are independent: blocks of
 * 32 rows run as consecutive launches), H/4 <= 256 workgroups, base->ht_out == base->xt + inc->xt (step s+1 reads the tiled h
 * of step s), inc->xt == inc->ht_out.  h_out2 (a second plain copy of h with its own stride) is written when given.
 * sync: >= 272 device words of scratch; words [0, 256) are the arrival counters (zeroed by every launch), word 256 is the
 * timeout flag: zeroed by the CALLER before first use and sticky - a wait that timed out (bounded spins) sets it, every launch
 * that sees it ends early (outputs unusable) until the caller has read and cleared it. */
int t2_lstm_seq_fwd_persist(const T2LstmStep* base, const T2LstmStride* inc, int S, uint32_t* sync, void* stream);
/* n = 1 or 2 INDEPENDENT cells of the same B and H in one persistent launch (grid.y = n; base[i], inc[i]): the two directions of
 * the encoder BiLSTM (model/encoder.py:47-52), each walking its own way through time (inc[i] may be negative; base[i].len and
 * base[i].t + s * inc[i].dt give the packed-sequence masking).  n * H/4 <= 256 workgroups; cell i counts arrivals in words
 * [128 i, 128 i + 128) of `sync`, the timeout flag is word 256 for all. */
int t2_lstm_seq_fwd_persist_n(const T2LstmStep* base, const T2LstmStride* inc, int n, int S, uint32_t* sync, void* stream);
/* The same with the scratch split and the clearing left to the caller: `counters` = [ceil(B/32)][256] words that the caller has
 * ZEROED on this stream since their last use (the engine clears a ring of them once per step with t2_zero_regions instead of one
 * memset per launch), `flag` = the sticky timeout word. */
int t2_lstm_seq_fwd_persist_pz(const T2LstmStep* base, const T2LstmStride* inc, int n, int S, uint32_t* counters, uint32_t* flag,
                               void* stream);
/* Residency check of the persistent launch above, without launching anything: T2_OK when H/4 workgroups with this K's weight
 * slice in LDS are all co-resident (compute units of the current device x hipOccupancyMaxActiveBlocksPerMultiprocessor),
 * T2_ERR_RESIDENCY otherwise - the caller then runs the same steps as t2_lstm_seq_fwd launches.  t2_lstm_seq_fwd_persist makes
 * the same check itself and returns the same code. */
int t2_lstm_persist_resident(int H, int K, int B);
int t2_lstm_persist_resident_n(int H, int K, int B, int n);      /* the same for n cells per launch (n * H/4 workgroups) */
/* Clears up to 64 memory regions with ONE launch (the reference's `torch.zeros` / `zero_()` of its state tensors, e.g.
 * model/tacotron2.py:126-153 init_hidden - one ATen fill each).  Region i: nrows[i] rows of row_bytes[i] bytes, stride_bytes[i] apart
 * (nrows = 1: one contiguous run); pointers and sizes are multiples of 4 bytes. */
typedef struct {
    void* p[64];
    int64_t row_bytes[64];
    int64_t nrows[64];
    int64_t stride_bytes[64];
    int n;
} T2ZeroRegions;
int t2_zero_regions(const T2ZeroRegions* r, void* stream);
/* Stream-concurrency probe (no reference counterpart: the reference is single-stream, run/train.py:235-243).  The engine needs
 * its two streams on different hardware queues (tacotron2_amd/__init__.py); Trainer.queue_check times `n` dependent one-thread
 * launches on one stream (t2_stream_probe_chain: word[0] += 1 per launch) alone and next to ONE idle wave that holds the other
 * stream for `microseconds` of wall-clock time (t2_stream_probe_spin, bounded: <= 50 ms and <= 2^24 polls).  Streams that share a
 * queue serialise: the chain then takes the spin's time longer. */
int t2_stream_probe_chain(uint32_t* word, int n, void* stream);
int t2_stream_probe_spin(uint32_t* word, int microseconds, void* stream);
/* Debug / test hook: bound of the inter-workgroup waits of t2_lstm_seq_fwd_persist in polls (default 1 << 21; < 0: every wait
 * is treated as timed out, which raises the sticky flag sync[256] deterministically).  Returns the previous value. */
int t2_debug_persist_spin_limit(int polls);
/* Makes the sticky timeout flag of the persistent launches (sync[256]) fatal without a host synchronisation: if *flag != 0,
 * x[0..n) is overwritten with NaN (the engine passes the decoder projection of the forward, so every output, the loss and
 * every gradient of that step become NaN and t2_adam_step - which skips a step whose gradient norm is not finite - leaves
 * the weights alone).  The host raises at its next synchronisation point. */
int t2_guard_poison(const uint32_t* flag, float* x, int64_t n, void* stream);
/* One step of back-propagation through time for an LSTM cell (autograd of the cells above):
 *   dx[b][u] = sum_n dg_next[b][n] * W[n*ldw + u]          (n over N4 = 4H' rows of the producing cell)
 *   epi = 0: dx_out = dx + ext1 + ext2                      (gradient w.r.t. a non-recurrent input slice)
 *   epi = 1: dh = (dx + ext1 + ext2) * drop; pointwise cell backward with the stashed activations of
 *            step t -> dg_out[b][4H] (pre-activation gate grads), dc[b][H] updated in place.
 * dg_next may be NULL (last frame: no recurrent contribution). */
typedef struct {
    int B, H, N4;
    const float* dg_next; int64_t lddg;
    const float* W; int64_t ldw;
    const float* dg2; int64_t lddg2; const float* W2; int64_t ldw2; int N2;  /* optional second K segment (+= dg2 . W2) */
    const float* wtpacked;           /* optional lane-contiguous transposed copy of W (and W2) (t2_lstm_pack_bwd) */
    int ncols; int epi;
    const float* ext1; int64_t ldx1; const float* ext2; int64_t ldx2;
    float* dx_out; int64_t lddx;
    const float* drop; int64_t lddrop;
    const float* gates; int64_t ldgs;       /* the forward stash, gate-interleaved [b][u][4] (T2LstmStep.gates_out) */
    const float* c_prev; int64_t ldcp; const float* c_cur; int64_t ldcc;
    float* dc; int64_t lddc;
    float* dg_out; int64_t ldgo;
    float* dg_out2; int64_t ldgo2;          /* optional second copy of dg_out with its own row stride */
    const int32_t* len; int t;
    /* x16-tiled operands (packed path only, layout as in T2LstmStep): dgt_next = tiled copy of dg_next ([N4/16][Bp][16]),
     * read instead of dg_next (which must still be non-NULL); dgt_out = tiled copy of dg_out ([4H/16][Bp][16]). */
    const float* dgt_next;
    float* dgt_out;
    int off_chain;                          /* 1: this step is NOT on the critical chain of its stream schedule (the decoder-LSTM BPTT,
                                               a chunk ahead on the side stream): its waves keep the default issue priority instead
                                               of the chain kernels' raised one */
} T2LstmBwdStep;
int t2_lstm_step_bwd(const T2LstmBwdStep* steps, int n, void* stream);
typedef struct { int64_t dg, dg2, ext1, ext2, drop, gates, c_prev, c_cur; int dt; int64_t dgt; } T2LstmBwdStride;
/* S steps; every pointer advances by its stride each step.  The caller lays the dgates stash out with one extra
 * zero-filled slot so that base[i].dg_next (the slot 'after' the first processed step) is valid and zero. */
int t2_lstm_seq_bwd(const T2LstmBwdStep* base, const T2LstmBwdStride* inc, int n, int S, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Location-sensitive attention, one frame (model/attention.py:52-69 + cumulative update model/decoder.py:78-90).
 *   U   [Ad][2][Kl]  = location_dense.weight . location_conv.weight, folded once per forward by
 *                      t2_attn_fold_location (pure re-association of two linear maps)
 *   pmT [B][Ad][L]   = processed memory (att_encoder output, model/tacotron2.py:229) stored transposed
 *   len [B] int32    : positions l >= len[b] are masked to -inf before the softmax (model/attention.py:63)
 *   e_part           : workspace [B][Ad/16][L];  th_out: optional stash [B][Ad][L4] of tanh(.) for backward, rows padded
 *                      to L4 = round_up(L,4) floats (16-byte aligned; must be 16-byte aligned itself)
 * Any text length whose per-sample images fit the 160 KB of LDS (energies kernel: 24 bytes per position - about 6,000 characters);
 * the kernels walk texts above 256 positions in rounds.
 * Writes w_out (new attention weights = the alignments row), cum_out = cum_prev + w, ctx_out (context). */
typedef struct {
    int B, L, A, Ad, Ef, Kl;
    const float* att_h; int64_t ldh;
    const float* Wq; const float* U; const float* v;
    const float* w_prev; int64_t ldw;        /* NULL = zeros (first frame) */
    const float* cum_prev; int64_t ldcum;    /* NULL = zeros */
    const float* pmT; const float* memory; const int32_t* len;
    float* e_part; float* th_out;
    float* w_out; int64_t ldwo; float* cum_out; int64_t ldco;
    float* ctx_out; int64_t ldctx; float* ctx_out2; int64_t ldctx2;
    float* ctxt_out; int ctxt_col0;          /* optional x16-tiled copy of the context (see T2LstmStep) */
    uint64_t* clk;                           /* diagnostic, normally NULL: 32 device words; workgroup (0,0) stamps the shader
                                                clock (s_memtime) at phase boundaries: energies [0..3], context [8..13] */
} T2AttnStep;
int t2_attn_fold_location(const float* Wd, const float* Wc, float* U, int Ad, int F, int Kl, void* stream);
int t2_attn_step_fwd(const T2AttnStep* s, void* stream);

/* Teacher-forced attention chain over all T frames (the attention half of the loop at
 * model/tacotron2.py:276-317; in teacher-forced mode it does not depend on the decoder LSTM, so the two
 * recurrences are run as separate chains with their input projections hoisted into large GEMMs).
 * Time-major stashes use slot s = t+1; the caller zero-fills slot 0 (initial states, model/tacotron2.py:126-153).
 *   pre   [T][B][4A]     prenet part of att_rnn.weight_ih + both biases (hoisted GEMM)
 *   xdec  [T+1][B][A+Ef] cols [0,A) = att_h_t (post-dropout), cols [A,A+Ef) = context_t
 *   att_c [T+1][B][A], gates [T][B][4A] (activated, optional), cum [T+1][B][L], th [T][B][Ad][L4] (optional, L4 = round_up(L,4))
 *   align [B][T][L]      the alignments output
 *   xproj_ctx            optional second copy of context_t, row (t,b) at xproj_ctx[(t*B+b)*ld_xproj] */
typedef struct {
    int B, L, T, A, Ad, Ef, Kl;
    const float* W_ih_ctx; int64_t ld_wih;
    const float* W_hh; const float* Wq; const float* U; const float* v;
    const float* wpacked;            /* optional t2_lstm_pack_fwd of segments (W_ih_ctx, Ef), (W_hh, A) */
    const float* pre; const float* pmT; const float* memory; const int32_t* len;
    const float* att_drop;
    float* xdec; float* att_c; float* gates; float* align; float* cum; float* th;
    float* xproj_ctx; int64_t ld_xproj;
    float* e_part;
    int t_begin, t_end;              /* frame range [t_begin, t_end) of this call; 0,0 = all T frames */
    float* xdec_t;                   /* optional (with wpacked): x16-tiled copy of xdec, [T+1][(A+Ef)/16][Bp][16], slot 0
                                        zero-filled by the caller; the attention-LSTM step then reads its input from it */
    uint64_t* clk;                   /* diagnostic (T2AttnStep.clk), normally NULL */
} T2AttnSeq;
int t2_attn_seq_fwd(const T2AttnSeq* a, void* stream);

/* Back-propagation through the attention chain, frames T-1 .. 0 (autograd of t2_attn_seq_fwd), 4 launches / frame:
 *   one launch for both products of dgates[t+1] (dctx_tot[t] = dctx_ext1[t] + dctx_ext2[t] + dgates[t+1].W_ih_ctx and
 *   dh_rec = dh_ext[t] + dgates[t+1].W_hh) ; attention backward (weights/energies, then per attention-dim slice) ;
 *   attention-LSTM cell backward with dh = dh_rec + dq[t].Wq.
 * Upstream gradients are time-major rows (t,b) with their own leading dimensions.  Outputs: dgates = Z [T+1][B][4A+Ad]
 * with Z[s][b] = [dgates_s (4A) | dq_{s-1} (Ad)] (the caller zero-fills slot T's first 4A columns; `dq` is unused),
 * dctx_tot [T][B][Ef] (inputs of the post-loop weight-gradient GEMMs) and the per-sample
 * accumulators dpmT [B][Ad][L], dv_part [B][Ad], dU_part [B][Ad][2][Kl] (caller zero-fills; summed over b after).
 * Workspaces: dc [B][A] (zero-filled), G [2][B][L], de [B][L], din_part [B][Ad/16][2][L].
 * Texts of up to 252 positions take the per-slice kernel in one pass; longer ones are walked in position tiles of 216 (+ 16 on
 * either side: a tile's gradient reaches 15 positions beyond it through the location filter) inside the same launch - no length
 * limit but the LDS (8 bytes per position for the location-input gradient of the whole text, 42 KB for a tile). */
typedef struct {
    int B, L, T, A, Ad, Ef, Kl;
    const float* W_ih_ctx; int64_t ld_wih;
    const float* W_hh; const float* Wq; const float* U; const float* v;
    const float* wtp_ctx;            /* optional t2_lstm_pack_bwd(W_ih_ctx, ncols = Ef) */
    const float* wtp_h;              /* t2_lstm_pack_bwd(W_hh, A, 4A, ncols = A) */
    const float* wtp_q;              /* t2_lstm_pack_bwd(Wq, A, Ad, ncols = A) */
    const float* memory; const float* xdec; const float* att_c; const float* gates; const float* align;
    const float* cum; const float* th; const float* att_drop;
    const float* dh_ext; int64_t ld_dh;
    const float* dctx_ext1; int64_t ld_dc1;
    const float* dctx_ext2; int64_t ld_dc2;
    float* dgates; float* dctx_tot; float* dq; float* dpmT; float* dv_part; float* dU_part;
    float* dc; float* G; float* de; float* din_part;
    float* dh_rec;                   /* workspace [B][A]: dh_ext[t] + dgates[t+1].W_hh */
    int t_hi, t_lo;                  /* frames t_hi-1 .. t_lo of this call (descending); 0,0 = T-1 .. 0 */
    float* dgates_t;                 /* optional x16-tiled copy of the dgates part of Z: [T+1][4A/16][Bp][16], slot T
                                        zero-filled by the caller; read by the per-frame products of dgates[t+1] */
    uint64_t* clk;                   /* diagnostic, normally NULL: 128 zeroed device words, read by the -DT2_STAMPS build only: s_memtime
                                        stamps of workgroup (0,0) at phase boundaries of the dw kernel [16..19] and the ds kernel
                                        [24..30]; [32] event counter and [40..127] a ring of the last 11 launches of the frame chain
                                        (products, dw, ds, cell backward) with wall-clock entry / exit and phase stamps (t2_common.hpp) */
    float* ws_bd;                    /* workspace of (Ad/16) * 16896 floats: the per-slice kernel runs its two correlations (dU, d_in)
                                        on the bf16 matrix pipe with exactly split operands (six products, fp32 accuracy); the
                                        workspace receives the d_in filter operand in fragment layout, rewritten by every call */
} T2AttnSeqBwd;
int t2_attn_seq_bwd(const T2AttnSeqBwd* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Conv stacks (encoder model/encoder.py:31-46,57; postnet model/postnet.py:8-49).
 * Activations use the padded channel-last layout (B, Lp = L+4, C): data rows [2, L+2), zero rows elsewhere, so a
 * k=5 'same' Conv1d is one t2_gemm with overlapping A rows (lda = C, K = 5C) against packed weights.
 *   t2_pack_conv_weight   flip=0: wp[co][k*Ci+ci] = w[co][ci][k]  (forward / wgrad layout)
 *                         flip=1: wp[ci][(K-1-k)*Co+co] = w[co][ci][k]  (dgrad: correlation with the flipped kernel)
 *   t2_unpack_conv_wgrad  g[co][ci][k] += gp[co][k*Ci+ci]
 */
int t2_embedding_fwd(const int64_t* idx, const float* table, float* out, int B, int L, int E, int pad, void* stream);
int t2_embedding_bwd(const int64_t* idx, const float* dout, float* dtable, int B, int L, int E, int Lp, int pad, void* stream);
int t2_pack_conv_weight(const float* w, float* wp, int Co, int Ci, int K, int flip, void* stream);
int t2_unpack_conv_wgrad(const float* gp, float* g, int Co, int Ci, int K, void* stream);

/* BatchNorm1d (+ activation + dropout mask [+ residual + length mask]) over the B*L valid rows of a raw conv
 * output in shifted row layout (row b*Lp_x + l).  training: batch statistics incl. padded positions, running
 * stats updated with momentum (unbiased variance), exactly nn.BatchNorm1d; eval: running statistics.
 * forward  (t2_bn_fwd): y[b][pad_y + l][c] = mask(act(bn(x))*drop + res); pad rows of y are zero-filled.
 * backward (t2_bn_bwd): dx (grad w.r.t. x, written at rows b*Lp_dx + pad_dx + l, other rows zero), dgamma/dbeta +=.
 * act: 0 none, 1 relu, 2 tanh.  sums: workspace of 2*C + 2 doubles (two sums per channel, then the row count they cover).
 * Synchronised statistics over data-parallel ranks (model/encoder.py:41, model/postnet.py:16,30,44 see the WHOLE batch in the
 * single-device reference): phase 1 = accumulate the sums only, the caller all-reduces `sums` (count included), phase 2 =
 * finalize + apply from the reduced sums; `shift` [C] must then be a rank-independent shift of the statistics sums (the
 * running mean), and in the backward `grad_share` = 1/world scales this rank's dgamma/dbeta contribution (the gradient
 * all-reduce adds the ranks' shares).  phase 0 / shift NULL / grad_share 0 = single-device behaviour. */
typedef struct {
    int B, L, C;
    const float* x; int Lp_x;
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;
    int training; float momentum, eps;
    double* sums;
    float* mean; float* invstd;
    int act;
    const float* drop;
    const float* res; int Lp_res, pad_res;
    const int32_t* len; float fill;
    float* y; int Lp_y, pad_y;
    const float* dy; int Lp_dy, pad_dy;
    float* dx; int Lp_dx, pad_dx;
    float* dgamma; float* dbeta;
    int phase; const float* shift; float grad_share;
    int sums_prezeroed;          /* 1: the caller has cleared `sums` on this stream (e.g. as one region of t2_zero_regions) */
    const float* tile_stats; int tile_M;   /* forward, training: the producing GEMM's T2Gemm.stat_out and its M (rows of the GEMM):
                                    the batch statistics come from these per-tile partials (merged in double) instead of a pass
                                    over x; NULL = the statistics kernel */
} T2Bn;
int t2_bn_fwd(const T2Bn* s, void* stream);
int t2_bn_bwd(const T2Bn* s, void* stream);

/* small data-movement / pointwise pieces of model/tacotron2.py:201-212,255,327-345 and model/tts_model.py:197-201 */
int t2_colsum(const float* x, int64_t ld, int64_t R, int C, float* out, void* stream);          /* out[c] += sum_r x[r][c] */
int t2_mel_to_tm(const float* mel, float* out, int B, int T, int M, void* stream);              /* (B,T,M) -> [T+1][B][M], slot 0 = 0 */
int t2_swap01(const float* in, float* out, int D0, int D1, int C, int accumulate, void* stream); /* (D0,D1,C) -> (D1,D0,C) */
int t2_finalize_fwd(const float* proj, int64_t ld_proj, const int32_t* len, float* mels, float* gates, float* post_in, int B,
                    int T, int M, void* stream);
int t2_finalize_bwd(const float* dpost_in, float* dproj, int B, int T, int M, void* stream);
/* upstream gradients of (mels, mels_post, gates) (any may be NULL) -> d_post_out (B,T,M) and dproj [T][B][M+1], masked */
int t2_outgrad_pack(const float* d_mels, const float* d_post, const float* d_gates, const int32_t* len, float* d_post_out,
                    float* dproj, int B, int T, int M, void* stream);
int t2_loss_fwd_bwd(const float* mels, const float* post, const float* gates, const float* mel_tgt, const float* gate_tgt,
                    const int32_t* len, int B, int T, int M, double* loss3 /* gate, mel, post */, float* d_post, float* dproj,
                    float grad_scale, void* stream);
/* The same three loss terms for the nn.Module surface (TTSModel.training_step / validation_step, model/tts_model.py:165-253): loss3
 * and - each optional - the dense gradients w.r.t. mels (B,T,M), mels_post (B,T,M) and gates (B,T,1), zero at masked positions. */
int t2_loss_terms(const float* mels, const float* post, const float* gates, const float* mel_tgt, const float* gate_tgt,
                  const int32_t* len, int B, int T, int M, double* loss3, float* d_mels, float* d_post, float* d_gates,
                  float grad_scale, void* stream);
int t2_relu_mask_bwd(const float* g, const float* y, const float* mask, float* out, int64_t n, void* stream);
int t2_condition_fwd(const float* enc, const float* spk_table, const int32_t* spk, const float* desc, float* memory, int B,
                     int L, int E, int Ef, void* stream);
/* dspk_table and ddesc are ACCUMULATED into (atomics): the caller zero-fills ddesc */
int t2_condition_bwd(const float* dmem, const float* memory, const int32_t* spk, float* denc, float* dspk_table, float* ddesc,
                     int B, int L, int E, int Ef, void* stream);
int t2_tanh_bias(float* x, const float* bias, int64_t rows, int C, void* stream);
/* HiFi-GAN V1 generator glue (model/hifi_gan.py:89-97,139-144,198-216; every convolution of the generator is t2_gemm over
 * overlapping / shifted channel-last rows): y = leaky_relu(scale * x, slope) and y += alpha * x */
int t2_leaky_relu(const float* x, float* y, int64_t n, float scale, float slope, void* stream);
int t2_axpy(const float* x, float* y, int64_t n, float alpha, void* stream);
int t2_tanh_bwd(const float* g, const float* y, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Autoregressive decoding: forward(teacher_forcing=False, max_len_override=N) of model/tacotron2.py:262-325.
 *   t2_linear_rows  out[b][n] = act(x[b][:] . w[n][:] + bias[n]) * mask[b][n]   (nn.Linear on <= batch rows; K % 16 == 0)
 *   t2_decoder_infer runs frames [t0, t1) of one group of <= 64 utterances with no host synchronisation, 6 launches per
 *   frame: state[0] = "every utterance of the group has stopped" (sticky), state[1] = frames emitted, done [B] int32.
 *   The first prenet layer is folded onto the mel projection (two linear maps with nothing in between):
 *     W_comb [(P+M+1)][D+Ef] = [W_pre1 . W_mel ; W_mel ; W_gate],  b_comb [(P+M+1)] = [W_pre1 . b_mel ; b_mel ; b_gate],
 *     row_comb (optional) [B][P+M+1] = per-utterance term of the prosody controls ([W_pre1 . cmel_b ; cmel_b ; 0])
 *   (built by the caller with t2_gemm once per call), so frame t's first launch yields p1_t AND the outputs of frame t-1
 *   from xproj_{t-1}, every sum in a fixed order (bit-reproducible stop decisions).
 *   t2_stop_scan derives the reference's break frame and `lengths` (model/tacotron2.py:319-322: counts every emitted frame
 *   whose stop logit is >= 0, Appendix C.4) from the stored logits of all groups, so groups may be decoded a few frames past
 *   the break (the host looks at state[0] only now and then; with several groups all run until ALL have stopped, exactly as
 *   the reference's single loop over the whole batch does).
 * Buffers: xs [2][(P+A+Ef+D)/16][Bp][16] = the recurrent state in the x16-tiled layout of T2LstmStep.xt, columns
 * [prenet_out | att_h | ctx | dec_h], two ping-pong slots, zero-filled by the caller (P, A, Ef, D multiples of 16);
 * att_h [B][A] and xproj [B][D+Ef] = [dec_h | ctx] row-major copies; p1 [B][P] prenet scratch;
 * att_c [2][B][A], dec_c [2][B][D], cum [2][B][L] (ping-pong, slot 0 zero-filled by the caller);
 * proj [Tcap][B][ld_proj] (cols 0..M-1 mel, col M stop logit; row t is written by the first launch of frame t+1 or by the
 * call's tail); align [B][Tcap][L]; prenet_mask [frames][2][B][P] or NULL (frame 0's masks are never read);
 * wp_att = t2_lstm_pack_fwd of the attention-LSTM weights in the column order [prenet | att_h | ctx],
 * wp_dec = t2_lstm_pack_fwd of the decoder-LSTM weights in the column order [att_h | ctx | dec_h]. */
int t2_linear_rows(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* mask,
                   int64_t ldmask, int relu, float* out, int64_t ldo, int B, int N, int K, void* stream);
typedef struct {
    int B, L, A, D, Ef, Ad, P, M, Kl, Tcap;
    const float* W_comb; const float* b_comb; const float* row_comb;
    const float* W_pre2;
    /* optional x16-tiled copies (layout of T2LstmStep.xt: [K/16][rows padded to 16][16]) of the two small linears' operands:
     * W_comb_t [(Ef+D)/16][Npad][16] with the K chunks in the order [ctx | dec_h] (the order of the tiled state), Npad =
     * round_up(P+M+1, 16); W_pre2_t [P/16][P][16]; p1_t [P/16][Bp][16] scratch.  With them every wave-load of these launches is
     * one contiguous 1 KB block (NULL: row-major operands, 16 half cache lines per wave-load). */
    const float* W_comb_t; const float* W_pre2_t; float* p1_t;
    const float* wp_att; const float* b_att_ih; const float* b_att_hh;
    const float* wp_dec; const float* b_dec_ih; const float* b_dec_hh;
    const float* Wq; const float* U; const float* v;
    const float* pmT; const float* memory; const int32_t* len;
    const float* prenet_mask;
    float* xs; float* att_h; float* att_c; float* dec_c; float* cum; float* xproj; float* p1; float* e_part;
    float* proj; int64_t ld_proj; float* align;
    int32_t* done; int32_t* state;
    const float* dec_pre;            /* optional [B][4D]: per-utterance term added to the decoder-LSTM pre-activations of
                                        every frame (controls . W_ih[:, A+Ef:]^T, model/decoder.py:94-99) */
} T2Infer;
int t2_decoder_infer(const T2Infer* a, int t0, int t1, void* stream);
/* proj[g] [nframes][Bg[g]][ld_proj] for g < ngroups (<= 64 groups of up to 64 utterances) -> lengths [sum Bg] int64,
 * out2 = {frames emitted n, 0} */
typedef struct { const float* proj[64]; int Bg[64]; int ngroups; int64_t ld_proj; int M, nframes; } T2StopScan;
int t2_stop_scan(const T2StopScan* s, int64_t* lengths, int32_t* out2, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Log-mel front-end (datasets/tts_dataset.py:166-168,204; definition restated from datasets/prosody_dataset.py:39-50,67):
 * wav [n] fp32 -> out [frames = 1 + n/hop][n_mels] natural-log mel.  basis [2*(n_fft/2+1)][n_fft] = window-folded
 * [cos ; -sin] DFT rows, fb [n_mels][ldm] mel filterbank rows zero-padded to ldm = round_up(n_fft/2+1, 4) (both built by the
 * host once, tacotron2_amd/datasets/logmel.py).  Workspaces: padded [n + n_fft], spec [frames][2*(n_fft/2+1)], mag [frames][ldm]. */
int t2_logmel_frames(int64_t n_samples, int hop);
int t2_logmel_fwd(const float* wav, int64_t n, const float* basis, const float* fb, float* padded, float* spec, float* mag,
                  float* out, int n_fft, int hop, int n_mels, void* stream);

/* The same front-end for ONE TRAINING BATCH in one pass - what the reference's 8 DataLoader workers + collate produce on the host
 * (run/train.py:150-158, datasets/tts_dataset.py:184-214, datasets/tts_dataloader.py:8-35): wavs [B][ld_wav] = the B decoded (trimmed,
 * silence-padded) utterances, zero-filled rows; n_dev [B] = their sample counts (DEVICE array, int64; n_max = the largest of
 * them, known to the host); every utterance's reflect-padded signal becomes one row of `padded` [B][Tp*hop], so that the analysis
 * windows of the WHOLE batch are the overlapping rows of ONE DFT GEMM (row b*Tp + f = frame f of utterance b, lda = hop; the
 * n_fft/hop - 1 rows that straddle two utterances are computed and dropped).  Outputs in the batch layout of the model:
 * mel (B, T_out, n_mels) zero behind each utterance's 1 + n_b/hop frames (T_out >= the longest: the caller may pad further, e.g. to a
 * data-parallel step's global shape), gate (B, T_out, 1) ones with the last valid frame 0 (datasets/tts_dataset.py:213-214; may be
 * NULL), mel_len [B] int32 (may be NULL).  Per frame the arithmetic is exactly t2_logmel_fwd's: results are bit-identical.
 * t2_logmel_batch_workspace: element counts of padded / spec / mag / tmp and Tp (out5[0..4]). */
int t2_logmel_batch_workspace(int B, int64_t n_max, int n_fft, int hop, int n_mels, int64_t* out5);
int t2_logmel_batch_fwd(const float* wavs, int64_t ld_wav, const int64_t* n_dev, int B, int64_t n_max, const float* basis,
                        const float* fb, float* padded, float* spec, float* mag, float* tmp, float* mel, int64_t T_out,
                        float* gate, int32_t* mel_len, int n_fft, int hop, int n_mels, void* stream);

/* dropout scale masks (Philox4x32-10, counter = element index) and the optimizer of model/tts_model.py:78-91 +
 * Lightning gradient_clip_val=1.0 (run/train.py:240) on one flat fp32 parameter buffer. */
int t2_philox_mask(float* out, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream);
int t2_sumsq(const float* g, int64_t n, double* out, void* stream);
/* A step whose global gradient norm (sumsq) is NaN or infinite is SKIPPED: parameters and moments stay as they are (the
 * reference would write NaN into every weight, torch clip_grad_norm_ + Adam; there is nothing to match in that state). */
int t2_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, float max_norm, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
