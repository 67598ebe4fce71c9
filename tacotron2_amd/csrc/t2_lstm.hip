// LSTM cell steps for gfx950: batch rows on the MFMA M axis (v_mfma_f32_16x16x4_f32, exact fp32).
//
// Forward step (lstm_step_fwd_kernel)
//   One workgroup = 4 hidden units x 4 gates = 16 gate columns, all batch rows (MT tiles of 16 rows).
//   grid.x = H/4 workgroups (256 for H = 1024: one per CU), grid.y = independent cells (BiLSTM directions).
//   The 4 waves split K (the concatenated input segments) in 16-deep chunks; each lane streams its
//   fragment straight from global memory as one 16-byte load per operand per chunk (weights are read once
//   per step per workgroup - the GEMV/M<=64 regime, no LDS round trip), 4 MFMA k-steps per load with the
//   k-permutation {s, 4+s, 8+s, 12+s}.  Partial tiles are summed across waves in LDS in fixed order, then
//   threads (b, unit) apply the gate non-linearities, cell update, dropout mask and write h/c/stash.
//
// Backward step (lstm_step_bwd_kernel)
//   dx[b][u] = sum_n dgates_next[b][n] * W[n][u]  (K = 4H over 4 waves, two accumulator chains), one
//   16 x 16 (batch x unit) tile per workgroup, then (epilogue) the cell's pointwise backward for step t,
//   producing dgates_t - the A operand of the next launch and the row block of the wgrad GEMMs.
#include "t2_common.hpp"
#include "t2_lstm_step.hpp"

// Diagnostic build only (-DT2_STAMPS; t2_debug_clock): when enabled, workgroup 0 of the forward fast kernel stamps the shader clock
// (s_memtime, words 0 and 6) and the 100 MHz reference (s_memrealtime, words 1 and 7) at entry and exit; nothing else
// reads these words.
__device__ unsigned long long g_t2_clk[8];
__device__ int g_t2_clk_enable;

namespace {



template <int MT>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(LstmK2 pp) {
    const LstmK& p = pp.s[blockIdx.y];
    __shared__ float red[4 * MT * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave id as a SCALAR: keeps the chunk -> segment lookup and the tail guards on the scalar unit, so the
    // vector loads below are unconditional and stay in flight across groups
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int u0 = blockIdx.x * 4;
    const int H = p.H;
    // this lane's weight row: gate block (r>>2), unit u0 + (r&3)
    const long wrow = (long)(r >> 2) * H + u0 + (r & 3);

    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Flattened 16-deep chunk list over the input segments; wave w owns chunks w, w+4, ...
    const int cb1 = p.nseg > 0 ? (p.seg[0].K >> 4) : 0;
    const int cb2 = cb1 + (p.nseg > 1 ? (p.seg[1].K >> 4) : 0);
    const int NT = cb2 + (p.nseg > 2 ? (p.seg[2].K >> 4) : 0);
    // rows >= B are clamped to row 0: their products land in accumulator rows that the epilogue never reads
    long xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) xrow[m] = (m * 16 + r) < p.B ? (m * 16 + r) : 0;

    constexpr int U = 4;
    auto load_group = [&](int c0, f32x4 (&bw)[U], f32x4 (&ax)[U][MT]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int c = c0 + 4 * j;           // scalar
            if (c < NT) {
                const int sgi = (c >= cb1 ? 1 : 0) + (c >= cb2 ? 1 : 0);
                const int lc = c - (sgi == 0 ? 0 : (sgi == 1 ? cb1 : cb2));
                const float* sx = sgi == 0 ? p.seg[0].x : (sgi == 1 ? p.seg[1].x : p.seg[2].x);
                const long ldx = sgi == 0 ? p.seg[0].ldx : (sgi == 1 ? p.seg[1].ldx : p.seg[2].ldx);
                if (p.wpacked) {
                    bw[j] = *reinterpret_cast<const f32x4*>(p.wpacked + ((long)blockIdx.x * ((NT + 15) & ~15) + c) * 256 + lane * 4);
                } else {
                    const float* sw = sgi == 0 ? p.seg[0].w : (sgi == 1 ? p.seg[1].w : p.seg[2].w);
                    const long ldw = sgi == 0 ? p.seg[0].ldw : (sgi == 1 ? p.seg[1].ldw : p.seg[2].ldw);
                    bw[j] = *reinterpret_cast<const f32x4*>(sw + wrow * ldw + 16 * lc + 4 * q);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    ax[j][m] = *reinterpret_cast<const f32x4*>(sx + xrow[m] * ldx + 16 * lc + 4 * q);
            } else {
                bw[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < MT; ++m) ax[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto mma_group = [&](const f32x4 (&bw)[U], const f32x4 (&ax)[U][MT]) {
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[j][m][s], bw[j][s], acc[m], 0, 0, 0);
    };
    {
        f32x4 bwA[U], bwB[U], axA[U][MT], axB[U][MT];
        int c = w;
        load_group(c, bwA, axA); c += 4 * U;
        while (true) {
            load_group(c, bwB, axB);
            mma_group(bwA, axA);
            if (c >= NT) break;
            c += 4 * U;
            load_group(c, bwA, axA);
            mma_group(bwB, axB);
            if (c >= NT) break;
            c += 4 * U;
        }
    }
    // C/D layout 16x16: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) red[((w * MT + m) * 16 + (q * 4 + g)) * 16 + r] = acc[m][g];
    __syncthreads();

    if (tid < MT * 64) {
        const int b = tid >> 2, uu = tid & 3;
        if (b < p.B) {
            const int u = u0 + uu;
            float gsum[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) s += red[((ww * MT + (b >> 4)) * 16 + (b & 15)) * 16 + g * 4 + uu];
                const int n = g * H + u;
                if (p.pre) s += p.pre[(long)b * p.ldpre + n];
                if (p.bias1) s += p.bias1[n];
                if (p.bias2) s += p.bias2[n];
                gsum[g] = s;
            }
            const bool active = (p.len == nullptr) || (p.t < p.len[b]);
            float gi = t2_sigmoid(gsum[0]), gf = t2_sigmoid(gsum[1]), gg = t2_tanh(gsum[2]), go = t2_sigmoid(gsum[3]);
            const float cp = p.c_prev ? p.c_prev[(long)b * p.ldc_prev + u] : 0.f;
            float cn = gf * cp + gi * gg;
            float hn = go * t2_tanh(cn);
            if (p.drop) hn *= p.drop[(long)b * p.lddrop + u];
            if (!active) { hn = 0.f; cn = 0.f; gi = gf = gg = go = 0.f; }
            p.h_out[(long)b * p.ldh + u] = hn;
            if (p.h_out2) p.h_out2[(long)b * p.ldh2 + u] = hn;
            if (p.c_out) p.c_out[(long)b * p.ldc_out + u] = cn;
            if (p.gates_out) {
                // gate-interleaved stash [b][u][4] = (i, f, g, o): one 16-byte store per thread, 64 contiguous bytes per 4 units
                *reinterpret_cast<f32x4*>(p.gates_out + (long)b * p.ldg + 4 * u) = (f32x4){gi, gf, gg, go};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fast path: packed, zero-padded weight stream + ONE contiguous activation segment.  Branch-free main loop:
// every wave runs NTpad/16 groups of 4 chunks; chunk c of group g is 16g + 4j + w.  Chunks past the real K multiply the
// all-zero padding chunks of the weight stream with (clamped, finite) activations, so no guard, select or wait sits
// between a load and the next load: two groups (2 x 12 x 1 KB per wave) are in flight while one is in the MFMAs.
// ---------------------------------------------------------------------------------------------------------
template <int MT, int U>
__global__ __launch_bounds__(256, 1) void lstm_step_fwd_fast_kernel(LstmK2 pp) {
    T2_CHAIN_PRIO();
    __shared__ float red[4 * MT * 256];
#ifdef T2_STAMPS      // diagnostic build only: every stamp is a branch the instruction scheduler does not move work across
    const bool stamp = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && g_t2_clk_enable;
    if (stamp) { g_t2_clk[0] = __builtin_amdgcn_s_memtime(); g_t2_clk[1] = __builtin_amdgcn_s_memrealtime(); }
    t2_lstm_fwd_fast_body<MT, U>(pp.s[blockIdx.y], blockIdx.x, red, stamp ? g_t2_clk : nullptr);
    if (stamp) { g_t2_clk[6] = __builtin_amdgcn_s_memtime(); g_t2_clk[7] = __builtin_amdgcn_s_memrealtime(); }
#else
    t2_lstm_fwd_fast_body<MT, U>(pp.s[blockIdx.y], blockIdx.x, red);
#endif
}

// ---------------------------------------------------------------------------------------------------------
// The same step for 33..64 batch rows with a SQUARE workgroup tile: 32 rows x 32 gate columns (8 hidden units) instead of 64 rows x
// 16 columns.  At four row tiles the step is bound by the operand bytes a CU takes in (profiles/r04_ab_cell_mfma_ablation.txt: not
// by the matrix pipe), and those are (rows + columns) x K x 4 bytes per workgroup: 64 K x 4 instead of 80 K x 4.  grid = (H/8 column
// blocks, 2 row blocks): the two row blocks of a column block are workgroups x and x + H/8 - the same XCD under round-robin
// dispatch (H/8 is a multiple of 8) - so their common weight tile is fetched into that XCD's L2 once.  Same packed weight stream
// (two consecutive 16-column blocks), same x16-tiled activations, same chunk-granular software pipeline, same epilogue.
// ---------------------------------------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(256, 1) void lstm_step_fwd_sq_kernel(LstmK2 pp) {
    T2_CHAIN_PRIO();
    __shared__ float red[4 * 4 * 256];
    const LstmK& p = pp.s[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int bx = blockIdx.x, b0 = blockIdx.y * 32, u0 = bx * 8;
    const int H = p.H;
    const int NT = p.seg[0].K >> 4, NTpad = (NT + 15) & ~15, G = NTpad / (4 * U);
    const float* wb[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) wb[ct] = p.wpacked + (long)(2 * bx + ct) * NTpad * 256 + lane * 4;
    const float* xb[2];
    const long xcs = p.xt_cs;
#pragma unroll
    for (int m = 0; m < 2; ++m) {       // (tiled rows exist up to round_up(B, 16) of the whole batch: clamp the row tiles past it)
        const long row = b0 + m * 16 + r, rows = p.xt_rows;
        xb[m] = p.xt + (row < rows ? row : rows - 1) * 16 + 4 * q;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[m][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // epilogue operands of this thread's (row, unit): hoisted, in flight during the GEMM
    const int bl = tid >> 3, uu8 = tid & 7, eb = b0 + bl, eu = u0 + uu8;
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
    auto load_chunk = [&](int g, int j, f32x4 (&bw)[2], f32x4 (&ax)[2]) {
        const int c = 4 * U * g + 4 * j + w;
        const int cx = c < NT ? c : NT - 1;     // padding chunks: any finite activations x the zero weight chunk
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) bw[ct] = *reinterpret_cast<const f32x4*>(wb[ct] + (long)c * 256);
#pragma unroll
        for (int m = 0; m < 2; ++m) ax[m] = *reinterpret_cast<const f32x4*>(xb[m] + xcs * cx);
    };
    auto mma_chunk = [&](const f32x4 (&bw)[2], const f32x4 (&ax)[2]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[m][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[m][s], bw[ct][s], acc[m][ct], 0, 0, 0);
    };
    auto pipe_group = [&](int gl, f32x4 (&bwL)[U][2], f32x4 (&axL)[U][2], const f32x4 (&bwM)[U][2], const f32x4 (&axM)[U][2]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            load_chunk(gl, j, bwL[j], axL[j]);
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk(bwM[j], axM[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        f32x4 bwA[U][2], bwB[U][2], axA[U][2], axB[U][2];
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
    // C layout 16x16: col = lane&15 (gate column of the tile), row = (lane>>4)*4 + reg (batch row of the tile)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) red[(((w * 2 + m) * 2 + ct) * 16 + (q * 4 + g)) * 16 + r] = acc[m][ct][g];
    __syncthreads();
    if (eb < p.B) {
        const int m = bl >> 4, rl = bl & 15, ct = uu8 >> 2, uu = uu8 & 3;
        float gsum[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) s += red[(((ww * 2 + m) * 2 + ct) * 16 + rl) * 16 + g * 4 + uu];
            gsum[g] = s + (p.pre ? e_pre[g] : 0.f) + (p.bias1 ? e_b1[g] : 0.f) + (p.bias2 ? e_b2[g] : 0.f);
        }
        if (!p.c_prev) e_cp = 0.f;
        if (!p.drop) e_drop = 1.f;
        if (!p.len) e_len = 0x7fffffff;
        const bool active = p.t < e_len;
        float gi = t2_sigmoid(gsum[0]), gf = t2_sigmoid(gsum[1]), gg = t2_tanh(gsum[2]), go = t2_sigmoid(gsum[3]);
        float cn = gf * e_cp + gi * gg;
        float hn = go * t2_tanh(cn) * e_drop;
        if (!active) { hn = 0.f; cn = 0.f; gi = gf = gg = go = 0.f; }
        p.h_out[(long)eb * p.ldh + eu] = hn;
        if (p.h_out2) p.h_out2[(long)eb * p.ldh2 + eu] = hn;
        if (p.ht_out) { const int col = p.ht_col0 + eu; p.ht_out[(long)(col >> 4) * p.xt_cs + eb * 16 + (col & 15)] = hn; }
        if (p.c_out) p.c_out[(long)eb * p.ldc_out + eu] = cn;
        if (p.gates_out) *reinterpret_cast<f32x4*>(p.gates_out + (long)eb * p.ldg + 4 * eu) = (f32x4){gi, gf, gg, go};
    }
}

int launch_fwd(const T2LstmStep* steps, int n, hipStream_t st) {
    T2_REQUIRE(n == 1 || n == 2, "lstm step: n must be 1 or 2");
    for (int i = 0; i < n; ++i) T2_TRY(t2_lstm_check_step(steps[i]));
    if (n == 2) T2_REQUIRE(steps[0].B == steps[1].B && steps[0].H == steps[1].H, "lstm step: cells must share B,H");
    const int B = steps[0].B;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int bn = (B - b0) < 64 ? (B - b0) : 64;
        LstmK2 kk;
        for (int i = 0; i < n; ++i) t2_lstm_to_k(steps[i], kk.s[i], b0, bn);
        if (n == 1) kk.s[1] = kk.s[0];
        dim3 grid(steps[0].H / 4, n), block(256);
        bool fast = true;
        for (int i = 0; i < n; ++i) fast = fast && steps[i].wpacked && steps[i].nseg == 1;
        if (fast) {
            // Pipeline depth: 4 chunks per group (12 x 1 KB loads per wave per group at two row tiles, two groups in flight).
            // Measured in one session on the training step, 4 chunks per group beat 8 (82.05 against 82.37 ms), and every
            // deeper variant (two groups requested ahead, all loads of the step issued at entry) was slower still: past
            // ~24 KB per wave the first operands only arrive later.
            static const int sq_tile = T2_KNOB("T2_CELL_SQ", 1);
            bool tiled_in = true;
            for (int i = 0; i < n; ++i) tiled_in = tiled_in && steps[i].xt != nullptr;
            if (bn <= 16) hipLaunchKernelGGL((lstm_step_fwd_fast_kernel<1, 4>), grid, block, 0, st, kk);
            else if (bn <= 32) hipLaunchKernelGGL((lstm_step_fwd_fast_kernel<2, 4>), grid, block, 0, st, kk);
            else if (sq_tile && tiled_in && steps[0].H % 64 == 0)      // 33..64 rows: 32 x 32 tiles, both row blocks of a column block on one XCD
                hipLaunchKernelGGL((lstm_step_fwd_sq_kernel<4>), dim3(steps[0].H / 8, t2_cdiv(bn, 32), n), block, 0, st, kk);
            else hipLaunchKernelGGL((lstm_step_fwd_fast_kernel<4, 4>), grid, block, 0, st, kk);
        } else {
            if (bn <= 16) hipLaunchKernelGGL((lstm_step_fwd_kernel<1>), grid, block, 0, st, kk);
            else if (bn <= 32) hipLaunchKernelGGL((lstm_step_fwd_kernel<2>), grid, block, 0, st, kk);
            else hipLaunchKernelGGL((lstm_step_fwd_kernel<4>), grid, block, 0, st, kk);
        }
    }
    T2_CHECK_LAUNCH();
    return T2_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Persistent, weight-stationary recurrence: S consecutive steps of ONE cell in ONE launch.
//
// Each workgroup keeps its 16 gate columns x K weights (K = 1024: 64 KB) in LDS for the whole launch, so a step streams no
// weights at all; what a step needs from the other workgroups is h_{s-1} (B x H floats = 128 KB at B = 32, H = 1024), which
// they exchange through the x16-tiled h stash itself:
//   producer: h goes out with write-through (sc1) stores, every storing wave drains them (s_waitcnt vmcnt(0)), the
//             workgroup's barrier, then ONE lane adds 1 to the workgroup's shard of an arrival counter (8 shards, 64 bytes
//             apart, agent-scope atomics);
//   consumer: the 8 lanes of one wave poll the 8 shards (relaxed sc1 loads, s_sleep in between) until shard i shows
//             n_i * s arrivals, the workgroup's barrier, then EVERY load of h is an sc1 buffer load to registers (no
//             cache-wide acquire: a buffer_inv would also throw out the L1 lines of the attention kernels that share the CU).
// (MI355X_MICROARCH.md, inter-workgroup visibility: the `sc1 stores + sc1 loads + drained counter add` form.)
// The launch runs on its own stream NEXT to the attention chain (one workgroup per CU: 4 waves, 72 KB of LDS), where the same
// steps hosted inside the attention-energies launches stretch every frame of the chain (energies 7.5 -> 11 us).
// Safety: every spin is bounded; a timeout raises a flag that ends every workgroup (checked by the host at its next
// synchronisation point); the counters are zeroed by the launch function; progress needs all H/4 workgroups resident, which
// holds for H/4 <= 256 once the kernels of the other streams drain (they never wait for this one).
// ---------------------------------------------------------------------------------------------------------
struct PersistK {
    LstmK s;                 // operand block of step 0
    long i_pre, i_drop, i_h_out, i_h_out2, i_c_out, i_gates, i_xt, i_ht;    // element increments per step
    int i_dt, steps;
    unsigned* sync;          // this cell's arrival counters: [8 shards] x 16 words
    unsigned* tmo;           // the timeout flag (shared by all persistent launches of an engine, sticky)
    int spin_limit;
};
struct PersistK2 { PersistK c[2]; };     // up to two independent cells per launch (grid.y): the two directions of the encoder BiLSTM
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (Measured alternatives, profiles/r02_ab_fwd_dec_chain.txt: the two 16-row groups of a batch as two interleaved pipelines
// inside one workgroup - no faster, a step is a chain of four dependent memory round trips (drain, counter, poll, sc1 loads)
// either way; one workgroup per group, two per CU - the 136 KB of LDS lock the attention kernels out of the CUs, the whole
// forward gets 6 ms slower.)
template <int MT>
__global__ __launch_bounds__(256, 1) void lstm_seq_persist_fwd_kernel(PersistK2 pq) {
    T2_CHAIN_PRIO();      // (default priority for this launch: +0.25 ms per step, profiles/r03_ab_bptt_priority_chunks.txt)
    extern __shared__ __attribute__((aligned(16))) float plds[];
    const PersistK& pp = pq.c[blockIdx.y];
    const LstmK& p = pp.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int bx = blockIdx.x, u0 = bx * 4, H = p.H;
    const int NT = p.seg[0].K >> 4, NTpad = (NT + 15) & ~15;
    float* wl = plds;                        // [NTpad][64 lanes][4]: this workgroup's slice of the packed stream
    float* red = plds + (long)NTpad * 256;   // [4 waves][MT][16][16] + 1 word for the wait's verdict
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.wpacked + (long)bx * NTpad * 256);
        f32x4* dst = reinterpret_cast<f32x4*>(wl);
        for (int i = tid; i < NTpad * 64; i += 256) dst[i] = src[i];
    }
    const int nwg = gridDim.x;
    unsigned* my_shard = pp.sync + (bx & 7) * 16;
    unsigned* tmo = pp.tmo;
    const unsigned want_per_step = lane < 8 ? (unsigned)((nwg + 7 - lane) >> 3) : 0u;     // workgroups that add to shard `lane`
    const int eb = tid >> 2, euu = tid & 3, eu = u0 + euu;
    const long ebc = eb < p.B ? eb : p.B - 1;
    float c_reg = (p.c_prev && tid < MT * 64) ? p.c_prev[ebc * p.ldc_prev + eu] : 0.f;
    const int e_len = p.len ? p.len[ebc] : 0x7fffffff;
    __syncthreads();
    const long xt_bytes = (long)NT * p.xt_cs * 4;      // one tiled slot of h
    bool alive = true;
    for (int s = 0; s < pp.steps && alive; ++s) {
        // epilogue operands of this step: independent of the other workgroups, requested before the wait
        float e_pre[4] = {0.f, 0.f, 0.f, 0.f}, e_drop = 1.f;
        if (tid < MT * 64) {
            const float* pre = p.pre + (long)s * pp.i_pre;
#pragma unroll
            for (int g = 0; g < 4; ++g) e_pre[g] = pre[ebc * p.ldpre + g * H + eu];
            if (p.drop) e_drop = (p.drop + (long)s * pp.i_drop)[ebc * p.lddrop + eu];
        }
        if (s > 0) {       // h_{s-1}: every workgroup has published it when shard i shows want_i * s arrivals
            if (w == 0) {
                int spins = 0;
                bool ok = false;
                while (true) {
                    unsigned got = 0xffffffffu;
                    if (lane < 8) got = __hip_atomic_load(pp.sync + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned bad = __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = __all(lane >= 8 || got >= want_per_step * (unsigned)s) && pp.spin_limit >= 0;
                    if (ok || bad) { ok = ok && !bad; break; }
                    if (++spins > pp.spin_limit || pp.spin_limit < 0) {   // (< 0: debug hook, every wait times out)
                        if (lane == 0) __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (lane == 0) red[4 * MT * 256] = ok ? 1.f : 0.f;
            }
            __syncthreads();
            alive = red[4 * MT * 256] != 0.f;
            if (!alive) break;
        }
        // gates = W . h_{s-1}: weights from LDS, activations by sc1 buffer loads (all issued before the first MFMA)
        const float* xt = p.xt + (long)s * pp.i_xt;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xt, 0, (int)xt_bytes, 0x00020000);
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int c0 = w; c0 < NT; c0 += 4 * 8) {
            f32x4 ax[8][MT], bw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = (c0 + 4 * j) < NT ? (c0 + 4 * j) : NT - 1;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int row = (m * 16 + r) < p.B ? (m * 16 + r) : p.B - 1;     // (padding rows are never written: read a real one)
                    const unsigned off = (unsigned)((((long)c * p.xt_cs) + row * 16 + 4 * q) * 4);
                    ax[j][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
                }
                bw[j] = *reinterpret_cast<const f32x4*>(wl + ((long)c * 64 + lane) * 4);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float okf = (c0 + 4 * j) < NT ? 1.f : 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[j][m][k], okf * bw[j][k], acc[m], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) red[((w * MT + m) * 16 + (q * 4 + g)) * 16 + r] = acc[m][g];
        __syncthreads();
        if (tid < MT * 64 && eb < p.B) {
            float gsum[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float sacc = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) sacc += red[((ww * MT + (eb >> 4)) * 16 + (eb & 15)) * 16 + g * 4 + euu];
                gsum[g] = sacc + e_pre[g] + (p.bias1 ? p.bias1[g * H + eu] : 0.f) + (p.bias2 ? p.bias2[g * H + eu] : 0.f);
            }
            const bool active = (p.t + s * pp.i_dt) < e_len;
            float gi = t2_sigmoid(gsum[0]), gf = t2_sigmoid(gsum[1]), gg = t2_tanh(gsum[2]), go = t2_sigmoid(gsum[3]);
            float cn = gf * c_reg + gi * gg;
            float hn = go * t2_tanh(cn) * e_drop;
            if (!active) { hn = 0.f; cn = 0.f; gi = gf = gg = go = 0.f; }
            c_reg = cn;
            (p.h_out + (long)s * pp.i_h_out)[(long)eb * p.ldh + eu] = hn;
            if (p.h_out2) (p.h_out2 + (long)s * pp.i_h_out2)[(long)eb * p.ldh2 + eu] = hn;
            {   // the exchanged copy: write-through
                float* ht = p.ht_out + (long)s * pp.i_ht;
                const int col = p.ht_col0 + eu;
                __hip_atomic_store(ht + (long)(col >> 4) * p.xt_cs + eb * 16 + (col & 15), hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (p.c_out) (p.c_out + (long)s * pp.i_c_out)[(long)eb * p.ldc_out + eu] = cn;
            if (p.gates_out) *reinterpret_cast<f32x4*>(p.gates_out + (long)s * pp.i_gates + (long)eb * p.ldg + 4 * eu) = (f32x4){gi, gf, gg, go};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores ...
        __syncthreads();                                        // ... before the workgroup signals (also frees `red`)
        if (tid == 0) __hip_atomic_fetch_add(my_shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------


__device__ __forceinline__ void bwd_epilogue(const BwdK& p, const float* red, int tid, int u0, int b0) {
    {
        const int bl = tid >> 4, ul = tid & 15;
        const int b = b0 + bl, u = u0 + ul;
        if (b < p.B && u < p.ncols) {
            float dx = red[(0 * 16 + bl) * 16 + ul] + red[(1 * 16 + bl) * 16 + ul] + red[(2 * 16 + bl) * 16 + ul] +
                       red[(3 * 16 + bl) * 16 + ul];
            if (p.ext1) dx += p.ext1[(long)b * p.ldx1 + u];
            if (p.ext2) dx += p.ext2[(long)b * p.ldx2 + u];
            if (p.epi == 0) {
                p.dx_out[(long)b * p.lddx + u] = dx;
            } else {
                const int H = p.H;
                const bool active = (p.len == nullptr) || (p.t < p.len[b]);
                float dh = dx;
                if (p.drop) dh *= p.drop[(long)b * p.lddrop + u];
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gates + (long)b * p.ldgs + 4 * u);
                const float gi = g4[0], gf = g4[1], gg = g4[2], go = g4[3];
                const float cp = p.c_prev ? p.c_prev[(long)b * p.ldcp + u] : 0.f;
                const float tc = t2_tanh(p.c_cur[(long)b * p.ldcc + u]);
                float dcv = p.dc[(long)b * p.lddc + u] + dh * go * (1.f - tc * tc);
                float d_o = dh * tc * go * (1.f - go);
                float d_i = dcv * gg * gi * (1.f - gi);
                float d_f = dcv * cp * gf * (1.f - gf);
                float d_g = dcv * gi * (1.f - gg * gg);
                float dcp = dcv * gf;
                if (!active) { d_i = d_f = d_g = d_o = 0.f; dcp = 0.f; }
                p.dc[(long)b * p.lddc + u] = dcp;
                float* dgo = p.dg_out + (long)b * p.ldgo + u;
                dgo[0] = d_i; dgo[H] = d_f; dgo[2 * H] = d_g; dgo[3 * H] = d_o;
                if (p.dg_out2) {
                    float* dg2o = p.dg_out2 + (long)b * p.ldgo2 + u;
                    dg2o[0] = d_i; dg2o[H] = d_f; dg2o[2 * H] = d_g; dg2o[3 * H] = d_o;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(BwdK2 pp) {
    const BwdK& p = pp.s[blockIdx.z];
    __shared__ float red[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar wave id (see the forward kernel)
    const int r = lane & 15, q = lane >> 4;
    const int u0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    {
        const int row = b0 + r;
        const bool cin = (u0 + r) < p.ncols;
        const long arow = row < p.B ? row : 0;
        const int ucol = u0 + (cin ? r : 0);
        const int nch1 = p.dg_next ? (p.N4 >> 4) : 0;
        const int nch = nch1 + (p.dg2 ? (p.N2 >> 4) : 0);
        constexpr int U = 4;
        auto load_group = [&](int c0, f32x4 (&a)[U], f32x4 (&b)[U]) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int c = c0 + 4 * j;       // scalar
                if (c < nch) {
                    const bool s2 = c >= nch1;
                    const int lc = s2 ? c - nch1 : c;
                    // rows >= B are clamped (arow): their results are never stored
                    const float* ap = (s2 ? p.dg2 + arow * p.lddg2 : p.dg_next + arow * p.lddg) + 16 * lc + 4 * q;
                    a[j] = *reinterpret_cast<const f32x4*>(ap);
                    if (p.wtpacked) {   // packed stream is laid out over BOTH segments even when dg_next is absent
                        const int pk1 = p.N4 >> 4, pkn = (pk1 + (p.N2 >> 4) + 31) & ~31;
                        b[j] = *reinterpret_cast<const f32x4*>(p.wtpacked + (((long)blockIdx.x * pkn + (s2 ? pk1 + lc : lc)) * 64 + lane) * 4);
                    } else {
                        const long ldw = s2 ? p.ldw2 : p.ldw;
                        const float* bq = (s2 ? p.W2 : p.W) + (long)(16 * lc + 4 * q) * ldw + ucol;
                        b[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (cin) { b[j][0] = bq[0]; b[j][1] = bq[ldw]; b[j][2] = bq[2 * ldw]; b[j][3] = bq[3 * ldw]; }
                    }
                } else {
                    a[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    b[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        };
        auto mma_group = [&](const f32x4 (&a)[U], const f32x4 (&b)[U]) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][0], b[j][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][1], b[j][1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][2], b[j][2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][3], b[j][3], acc1, 0, 0, 0);
            }
        };
        if (nch > 0) {
            f32x4 aA[U], bA[U], aB[U], bB[U];
            int c = w;
            load_group(c, aA, bA); c += 4 * U;
            while (true) {
                load_group(c, aB, bB);
                mma_group(aA, bA);
                if (c >= nch) break;
                c += 4 * U;
                load_group(c, aA, bA);
                mma_group(aB, bB);
                if (c >= nch) break;
                c += 4 * U;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) red[(w * 16 + (q * 4 + g)) * 16 + r] = acc0[g] + acc1[g];
    __syncthreads();
    bwd_epilogue(p, red, tid, u0, b0);
}

// Fast path of the backward step: ONE contiguous gradient row block dg[b][0:K) (K = N4 + N2) against the packed,
// zero-padded transposed weight stream; same branch-free double-buffered structure as the forward fast path.
__global__ __launch_bounds__(256, 1) void lstm_step_bwd_fast_kernel(BwdK2 pp) {
    if (!pp.s[blockIdx.z].off_chain) T2_CHAIN_PRIO();
    __shared__ float red[4 * 256];
    t2_lstm_bwd_fast_body<4, 4>(pp.s[blockIdx.z], blockIdx.x, blockIdx.y, red);
}

// The same with eight waves splitting K, for launches of few workgroups (the encoder BiLSTM backward: 64 workgroups with K = 1024
// on a chip of 256 compute units): half the chunks, loads and MFMAs per wave.
__global__ __launch_bounds__(512, 1) void lstm_step_bwd_fast8_kernel(BwdK2 pp) {
    if (!pp.s[blockIdx.z].off_chain) T2_CHAIN_PRIO();
    __shared__ float red[8 * 256];
    t2_lstm_bwd_fast_body<8, 2>(pp.s[blockIdx.z], blockIdx.x, blockIdx.y, red);
}
static int g_bwd8_max_wgs = T2_KNOB("T2_BWD8_MAX_WGS", 64);

int launch_bwd(const T2LstmBwdStep* steps, int n, hipStream_t st, unsigned long long* clk = nullptr) {
    T2_REQUIRE(n == 1 || n == 2, "lstm bwd step: n must be 1 or 2");
    BwdK2 kk;
    for (int i = 0; i < n; ++i) { T2_TRY(t2_lstm_check_bwd(steps[i])); t2_lstm_to_bk(steps[i], kk.s[i]); kk.s[i].clk = clk; }
    bool fast = true;
    for (int i = 0; i < n; ++i) fast = fast && steps[i].wtpacked && steps[i].dg_next && !steps[i].dg2;
    for (int i = 1; i < n; ++i)
        T2_REQUIRE(steps[0].B == steps[i].B && (fast || steps[0].ncols == steps[i].ncols), "lstm bwd step: shapes differ");
    for (int i = 0; i < n; ++i)
        T2_REQUIRE((!steps[i].dgt_next && !steps[i].dgt_out) || (fast && steps[i].H % 16 == 0 && t2_aligned16(steps[i].dgt_next)),
                   "lstm bwd step: x16-tiled operands need the packed path and H % 16 == 0");
    for (int i = n; i < 2; ++i) kk.s[i] = kk.s[0];
    int maxcols = steps[0].ncols;
    for (int i = 1; i < n; ++i) maxcols = steps[i].ncols > maxcols ? steps[i].ncols : maxcols;
    dim3 grid(t2_cdiv(maxcols, 16), t2_cdiv(steps[0].B, 16), n), block(256);
    if (fast && (int)(grid.x * grid.y * grid.z) <= g_bwd8_max_wgs) hipLaunchKernelGGL(lstm_step_bwd_fast8_kernel, grid, dim3(512), 0, st, kk);
    else if (fast) hipLaunchKernelGGL(lstm_step_bwd_fast_kernel, grid, block, 0, st, kk);
    else hipLaunchKernelGGL(lstm_step_bwd_kernel, grid, block, 0, st, kk);
    T2_CHECK_LAUNCH();
    return T2_OK;
}

struct PackSegs { int nseg; const float* w[3]; long ldw[3]; int K[3]; };

// out[((j*NT + c)*64 + lane)*4 + e] = W_seg[(g*H + 4j + uu)*ldw + 16*lc + 4q + e],  lane = q*16 + (g*4 + uu)
__global__ void lstm_pack_fwd_kernel(PackSegs s, int H, int NT, float* out) {
    const int NTpad = (NT + 15) & ~15;
    const long total = (long)(H / 4) * NTpad * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long rem = idx >> 6;
        const int c = (int)(rem % NTpad), j = (int)(rem / NTpad);
        const int r = lane & 15, q = lane >> 4;
        if (c >= NT) { *reinterpret_cast<f32x4*>(out + idx * 4) = (f32x4){0.f, 0.f, 0.f, 0.f}; continue; }
        int sgi = 0, lc = c;
        while (sgi < s.nseg - 1 && lc >= (s.K[sgi] >> 4)) { lc -= s.K[sgi] >> 4; ++sgi; }
        const long row = (long)(r >> 2) * H + 4 * j + (r & 3);
        const float* src = s.w[sgi] + row * s.ldw[sgi] + 16 * lc + 4 * q;
        f32x4 v; v[0] = src[0]; v[1] = src[1]; v[2] = src[2]; v[3] = src[3];
        *reinterpret_cast<f32x4*>(out + idx * 4) = v;
    }
}

// out[((ut*NCH + c)*64 + lane)*4 + s] = Wx[(16*lc + 4q + s)*ldwx + 16*ut + j],  lane = q*16 + j  (0 past ncols)
__global__ void lstm_pack_bwd_kernel(const float* W, long ldw, int N4, const float* W2, long ldw2, int N2, int ncols,
                                     float* out) {
    const int nch1 = N4 >> 4, nch = nch1 + (W2 ? (N2 >> 4) : 0), nchpad = (nch + 31) & ~31;
    const int tiles = (ncols + 15) / 16;
    const long total = (long)tiles * nchpad * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long rem = idx >> 6;
        const int c = (int)(rem % nchpad), ut = (int)(rem / nchpad);
        if (c >= nch) { *reinterpret_cast<f32x4*>(out + idx * 4) = (f32x4){0.f, 0.f, 0.f, 0.f}; continue; }
        const int j = lane & 15, q = lane >> 4, u = 16 * ut + j;
        const bool s2 = c >= nch1;
        const int lc = s2 ? c - nch1 : c;
        const float* Wx = s2 ? W2 : W;
        const long ld = s2 ? ldw2 : ldw;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (u < ncols) {
            const float* src = Wx + (long)(16 * lc + 4 * q) * ld + u;
            v[0] = src[0]; v[1] = src[ld]; v[2] = src[2 * ld]; v[3] = src[3 * ld];
        }
        *reinterpret_cast<f32x4*>(out + idx * 4) = v;
    }
}

template <typename T>
inline void adv(T*& p, int64_t inc) { if (p) p += inc; }

}  // namespace

void t2_lstm_fwd_advance(T2LstmStep& c, const T2LstmStride& inc) {
    for (int j = 0; j < 3; ++j) adv(c.seg[j].x, inc.seg_x[j]);
    adv(c.pre, inc.pre); adv(c.c_prev, inc.c_prev); adv(c.drop, inc.drop);
    adv(c.h_out, inc.h_out); adv(c.h_out2, inc.h_out2); adv(c.c_out, inc.c_out);
    adv(c.gates_out, inc.gates_out);
    adv(c.xt, inc.xt); adv(c.ht_out, inc.ht_out);
    c.t += inc.dt;
}
void t2_lstm_bwd_advance(T2LstmBwdStep& c, const T2LstmBwdStride& inc) {
    adv(c.dg_next, inc.dg);
    adv(c.ext1, inc.ext1); adv(c.ext2, inc.ext2); adv(c.drop, inc.drop);
    adv(c.gates, inc.gates); adv(c.c_prev, inc.c_prev); adv(c.c_cur, inc.c_cur);
    adv(c.dg_out, inc.dg);
    adv(c.dg_out2, inc.dg2);
    adv(c.dgt_next, inc.dgt); adv(c.dgt_out, inc.dgt);
    c.t += inc.dt;
}
// internal entries used by the attention sequence (t2_attention.hip)
int t2_lstm_step_fwd_launch(const T2LstmStep* steps, int n, hipStream_t st) { return launch_fwd(steps, n, st); }
int t2_lstm_step_bwd_launch(const T2LstmBwdStep* steps, int n, hipStream_t st, unsigned long long* clk) { return launch_bwd(steps, n, st, clk); }
extern "C" int t2_lstm_step_fwd(const T2LstmStep* steps, int n, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(steps != nullptr, "t2_lstm_step_fwd: null");
    return launch_fwd(steps, n, (hipStream_t)stream);
}

extern "C" int t2_lstm_step_bwd(const T2LstmBwdStep* steps, int n, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(steps != nullptr, "t2_lstm_step_bwd: null");
    return launch_bwd(steps, n, (hipStream_t)stream);
}

extern "C" int t2_lstm_seq_fwd(const T2LstmStep* base, const T2LstmStride* inc, int n, int S, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(base && inc && (n == 1 || n == 2) && S >= 0, "t2_lstm_seq_fwd: bad arguments");
    T2LstmStep cur[2];
    for (int i = 0; i < n; ++i) cur[i] = base[i];
    for (int s = 0; s < S; ++s) {
        T2_TRY(launch_fwd(cur, n, (hipStream_t)stream));
        for (int i = 0; i < n; ++i) t2_lstm_fwd_advance(cur[i], inc[i]);
    }
    return T2_OK;
}

extern "C" int t2_lstm_seq_bwd(const T2LstmBwdStep* base, const T2LstmBwdStride* inc, int n, int S, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(base && inc && (n == 1 || n == 2) && S >= 0, "t2_lstm_seq_bwd: bad arguments");
    T2LstmBwdStep cur[2];
    for (int i = 0; i < n; ++i) cur[i] = base[i];
    for (int s = 0; s < S; ++s) {
        T2_TRY(launch_bwd(cur, n, (hipStream_t)stream));
        for (int i = 0; i < n; ++i) t2_lstm_bwd_advance(cur[i], inc[i]);
    }
    return T2_OK;
}

// (T2_PERSIST_SPIN_LIMIT in the environment presets the debug value for a whole process, e.g. a CLI run under test)
static int g_persist_spin_limit = getenv("T2_PERSIST_SPIN_LIMIT") ? atoi(getenv("T2_PERSIST_SPIN_LIMIT")) : (1 << 21);
extern "C" int t2_debug_persist_spin_limit(int polls) {
    const int old = g_persist_spin_limit;
    g_persist_spin_limit = polls;
    return old;
}

// All H/4 workgroups of the persistent launch must be resident at once (they wait for each other): compute units of the
// current device x the occupancy the runtime reports for this kernel with its LDS slice.  The answer is cached per (MT, LDS).
static int persist_resident(int nwg, int MT, size_t lds) {
    static std::mutex mu;
    static std::unordered_map<long, int> cap;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { t2_set_error("hipGetDevice failed", __FILE__, __LINE__); return T2_ERR_LAUNCH; }
    const long key = ((long)dev << 40) | ((long)MT << 32) | (long)lds;
    auto it = cap.find(key);
    if (it == cap.end()) {
        int cus = 0, per_cu = 0;
        hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess)
            e = MT == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_seq_persist_fwd_kernel<1>, 256, lds)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_seq_persist_fwd_kernel<2>, 256, lds);
        if (e != hipSuccess) { t2_set_error(hipGetErrorString(e), __FILE__, __LINE__); (void)hipGetLastError(); return T2_ERR_LAUNCH; }
        it = cap.emplace(key, cus * per_cu).first;
    }
    if (nwg > it->second) {
        t2_set_error("t2_lstm_seq_fwd_persist: the launch would not be co-resident on this device (compute units x occupancy "
                     "< H/4 workgroups); use t2_lstm_seq_fwd", __FILE__, __LINE__);
        return T2_ERR_RESIDENCY;
    }
    return T2_OK;
}

static size_t persist_lds_bytes(int K) {
    const int NT = K >> 4, NTpad = (NT + 15) & ~15;
    return ((size_t)NTpad * 256 + (size_t)4 * 2 * 256 + 4) * sizeof(float);
}

extern "C" int t2_lstm_persist_resident_n(int H, int K, int B, int n) {
    (void)hipGetLastError();
    T2_REQUIRE(H >= 4 && H % 4 == 0 && K >= 16 && K % 16 == 0 && B >= 1 && n >= 1, "t2_lstm_persist_resident: bad arguments");
    const size_t lds = persist_lds_bytes(K);
    // "cannot run as ONE co-resident launch" is an answer, not an error: the caller falls back to step launches (T2_ERR_RESIDENCY)
    if (n * (H / 4) > 256 || n > 2 ||
        !(t2_allow_lds(lstm_seq_persist_fwd_kernel<1>, lds) && t2_allow_lds(lstm_seq_persist_fwd_kernel<2>, lds))) {
        t2_set_error("t2_lstm_persist_resident: cannot be ONE co-resident launch (more than 256 workgroups, more than two cells, or a "
                     "weight slice that does not fit the LDS); use t2_lstm_seq_fwd", __FILE__, __LINE__);
        return T2_ERR_RESIDENCY;
    }
    return persist_resident(n * (H / 4), B <= 16 ? 1 : 2, lds);
}
extern "C" int t2_lstm_persist_resident(int H, int K, int B) { return t2_lstm_persist_resident_n(H, K, B, 1); }

// counters: 256 words per launch (row block); flag: the sticky timeout word.  prezeroed: the caller cleared the counters of every
// row block ([ceil(B/32)][256] words) on this stream; otherwise each launch clears its 256 words itself.
static int persist_launch(const T2LstmStep* base, const T2LstmStride* inc, int n, int S, uint32_t* sync, uint32_t* flag, bool prezeroed,
                          void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(base && inc && sync && flag && S >= 0 && (n == 1 || n == 2), "t2_lstm_seq_fwd_persist: bad arguments");
    if (S == 0) return T2_OK;
    for (int i = 0; i < n; ++i) {
        T2_TRY(t2_lstm_check_step(base[i]));
        const T2LstmStep& b = base[i];
        T2_REQUIRE(b.wpacked && b.nseg == 1 && b.xt && b.ht_out && b.B <= 64 && b.pre && b.H % 4 == 0,
                   "t2_lstm_seq_fwd_persist: needs the packed single-segment path with x16-tiled h exchange and B <= 64");
        T2_REQUIRE(b.seg[0].K == b.H && b.ht_col0 == 0 && inc[i].xt == inc[i].ht_out && b.ht_out == b.xt + inc[i].xt,
                   "t2_lstm_seq_fwd_persist: the input of step s+1 must be the tiled h of step s (K = H)");
        T2_REQUIRE(b.B == base[0].B && b.H == base[0].H, "t2_lstm_seq_fwd_persist: the cells of one launch share B and H");
    }
    const T2LstmStep& b = base[0];
    T2_REQUIRE(n * (b.H / 4) <= 256, "t2_lstm_seq_fwd_persist: at most 256 workgroups (one per CU)");
    const size_t lds = persist_lds_bytes(b.seg[0].K);
    T2_REQUIRE(t2_allow_lds(lstm_seq_persist_fwd_kernel<1>, lds) && t2_allow_lds(lstm_seq_persist_fwd_kernel<2>, lds),
               "t2_lstm_seq_fwd_persist: weight slice does not fit the LDS");
    T2_TRY(persist_resident(n * (b.H / 4), (b.B < 32 ? b.B : 32) <= 16 ? 1 : 2, lds));
    hipStream_t st = (hipStream_t)stream;
    // Rows are independent: blocks of up to 32 rows (two 16-row tiles) run as consecutive launches of the same chunk.  Each
    // launch zeroes the arrival counters (128 words per cell); the timeout flag (word 256) is sticky - only the host clears it.
    for (int b0 = 0; b0 < b.B; b0 += 32) {
        const int bn = (b.B - b0) < 32 ? (b.B - b0) : 32;
        PersistK2 kk;
        for (int i = 0; i < n; ++i) {
            PersistK& k = kk.c[i];
            t2_lstm_to_k(base[i], k.s, b0, bn);
            k.i_pre = inc[i].pre; k.i_drop = inc[i].drop; k.i_h_out = inc[i].h_out; k.i_h_out2 = inc[i].h_out2;
            k.i_c_out = inc[i].c_out; k.i_gates = inc[i].gates_out;
            k.i_xt = inc[i].xt; k.i_ht = inc[i].ht_out; k.i_dt = inc[i].dt; k.steps = S;
            k.sync = sync + i * 128; k.tmo = flag; k.spin_limit = g_persist_spin_limit;
        }
        if (n == 1) kk.c[1] = kk.c[0];
        if (prezeroed) sync += 256;          // (the next row block's own counters)
        else (void)hipMemsetAsync(sync, 0, 16 * 16 * sizeof(uint32_t), st);
        dim3 grid(b.H / 4, n), block(256);
        if (bn <= 16) hipLaunchKernelGGL((lstm_seq_persist_fwd_kernel<1>), grid, block, lds, st, kk);
        else hipLaunchKernelGGL((lstm_seq_persist_fwd_kernel<2>), grid, block, lds, st, kk);
    }
    T2_CHECK_LAUNCH();
    return T2_OK;
}

extern "C" int t2_lstm_seq_fwd_persist_n(const T2LstmStep* base, const T2LstmStride* inc, int n, int S, uint32_t* sync, void* stream) {
    return persist_launch(base, inc, n, S, sync, sync ? sync + 256 : nullptr, false, stream);
}
extern "C" int t2_lstm_seq_fwd_persist(const T2LstmStep* base, const T2LstmStride* inc, int S, uint32_t* sync, void* stream) {
    return t2_lstm_seq_fwd_persist_n(base, inc, 1, S, sync, stream);
}
extern "C" int t2_lstm_seq_fwd_persist_pz(const T2LstmStep* base, const T2LstmStride* inc, int n, int S, uint32_t* counters,
                                          uint32_t* flag, void* stream) {
    return persist_launch(base, inc, n, S, counters, flag, true, stream);
}

extern "C" int t2_lstm_pack_fwd(const T2Seg* segs, int nseg, int H, float* out, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(segs && out && nseg >= 1 && nseg <= 3 && H % 4 == 0, "t2_lstm_pack_fwd: bad arguments");
    PackSegs s; s.nseg = nseg;
    int NT = 0;
    for (int i = 0; i < nseg; ++i) {
        T2_REQUIRE(segs[i].K % 16 == 0 && segs[i].w, "t2_lstm_pack_fwd: segment K must be a multiple of 16");
        s.w[i] = segs[i].w; s.ldw[i] = segs[i].ldw; s.K[i] = segs[i].K; NT += segs[i].K >> 4;
    }
    const long total = (long)(H / 4) * ((NT + 15) & ~15) * 64;
    hipLaunchKernelGGL(lstm_pack_fwd_kernel, dim3(t2_cdiv(total, 256) > 2048 ? 2048 : t2_cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, s, H, NT, out);
    T2_CHECK_LAUNCH();
    return T2_OK;
}

extern "C" int t2_lstm_pack_bwd(const float* W, int64_t ldw, int N4, const float* W2, int64_t ldw2, int N2, int ncols,
                                float* out, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(W && out && N4 % 16 == 0 && (!W2 || N2 % 16 == 0) && ncols >= 1, "t2_lstm_pack_bwd: bad arguments");
    const long total = (long)t2_cdiv(ncols, 16) * ((((N4 >> 4) + (W2 ? (N2 >> 4) : 0)) + 31) & ~31) * 64;
    hipLaunchKernelGGL(lstm_pack_bwd_kernel, dim3(t2_cdiv(total, 256) > 2048 ? 2048 : t2_cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, W, (long)ldw, N4, W2, (long)ldw2, N2, ncols, out);
    T2_CHECK_LAUNCH();
    return T2_OK;
}

// Diagnostic: enable/disable the clock stamps of the forward fast kernel and read the last 8 stamped words.
extern "C" int t2_debug_clock(int enable, uint64_t* out8) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_t2_clk_enable), &enable, sizeof(int)) != hipSuccess) return T2_ERR_LAUNCH;
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_t2_clk), 8 * sizeof(unsigned long long)) != hipSuccess) return T2_ERR_LAUNCH;
    return T2_OK;
}
