// HBM-bound elementwise / reduction kernels of the Tacotron 2 path for gfx950: embedding gather, Conv1d weight
// packing for the conv-as-GEMM formulation, BatchNorm1d (batch statistics over ALL rows incl. padding, as the
// reference does), activation/dropout epilogues, output masking, the 3-term loss, Philox dropout masks and
// the flat-buffer Adam step.  All accesses are coalesced along the channel axis (channel-last tensors).
//
// Conv stacks use one padded activation layout: (B, Lp = L+4, C), data rows [2, L+2), zero rows elsewhere.
// A 'same' k=5 convolution is then ONE GEMM whose A rows overlap: row r = b*Lp + l of A is the 5*C contiguous
// floats starting at padded row r (lda = C, K = 5C); outputs land in the 'shifted' row layout r = b*Lp + l
// (l < L valid, 4 junk rows per sample that every consumer skips or zeroes).
#include "t2_common.hpp"

namespace {

__global__ void embedding_fwd_kernel(const int64_t* idx, const float* table, float* out, int B, int L, int E, int pad) {
    const int Lp = L + 2 * pad;
    const long n = (long)B * Lp * E;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const long row = i / E;
        const int lp = (int)(row % Lp), b = (int)(row / Lp);
        const int l = lp - pad;
        float v = 0.f;
        if (l >= 0 && l < L) v = table[idx[(long)b * L + l] * E + e];
        out[i] = v;
    }
}

// dtable[idx] += dout (rows of the padded buffer); padding_idx row 0 receives nothing (model/encoder.py:25)
__global__ void embedding_bwd_kernel(const int64_t* idx, const float* dout, float* dtable, int B, int L, int E, int Lp,
                                     int pad) {
    const long n = (long)B * L * E;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const long row = i / E;
        const int l = (int)(row % L), b = (int)(row / L);
        const int64_t id = idx[row];
        if (id != 0) atomicAdd(&dtable[id * E + e], dout[((long)b * Lp + pad + l) * E + e]);
    }
}

__global__ void pack_conv_w_kernel(const float* w, float* wp, int Co, int Ci, int K, int flip) {
    const long n = (long)Co * Ci * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int ci = (int)((i / K) % Ci);
        const int co = (int)(i / ((long)K * Ci));
        if (!flip) wp[(long)co * K * Ci + (long)k * Ci + ci] = w[i];
        else wp[(long)ci * K * Co + (long)(K - 1 - k) * Co + co] = w[i];
    }
}

__global__ void unpack_conv_wgrad_kernel(const float* gp, float* g, int Co, int Ci, int K) {
    const long n = (long)Co * Ci * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int ci = (int)((i / K) % Ci);
        const int co = (int)(i / ((long)K * Ci));
        g[i] += gp[(long)co * K * Ci + (long)k * Ci + ci];
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm1d
// ---------------------------------------------------------------------------------------------
struct BnK {
    int B, L, C;
    const float* x; int Lp_x;             // raw conv output, shifted rows b*Lp_x + l
    const float* gamma; const float* beta;
    float* rmean; float* rvar;            // running stats (read in eval, updated in training)
    int training; float momentum, eps;
    double* sums;                         // [2][C] + 2: sums, then the row count they cover (summed over ranks under sync-BN)
    const float* shift;                   // [C] common shift of the statistics sums, or null (= the channel's first value)
    float* mean; float* invstd;           // [C] saved batch stats (training) / derived from running (eval)
    int act;                              // 0 none, 1 relu, 2 tanh
    const float* drop;                    // dense [B][L][C] scale mask or null
    const float* res; int Lp_res, pad_res;  // optional residual added after dropout
    const int32_t* len; float fill;       // optional: rows l >= len[b] are set to `fill`
    float* y; int Lp_y, pad_y;            // output; pad rows are zero-filled
    // backward
    const float* dy; int Lp_dy, pad_dy;
    float* dx; int Lp_dx, pad_dx;
    float* dgamma; float* dbeta;
    double grad_share;                    // 1, or 1/world under sync-BN
};

// grid (ceil(C/64), row chunks); block 256 = 64 channels x 4 row lanes
__global__ __launch_bounds__(256) void bn_stats_kernel(BnK p, int rows_per_block) {
    __shared__ float s1[4][64], s2[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long R = (long)p.B * p.L;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > R) r1 = R;
    // Sums of (x - shift) with shift = the channel's first value: E[d^2] - E[d]^2 then has no cancellation however far the
    // channel mean is from zero (a -5.5 log-mel level with 0.25 spread loses ~3 digits of the variance otherwise).
    float a = 0.f, q = 0.f;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(&p.sums[2 * p.C], (double)R);
    if (c < p.C) {
        const float sh = p.shift ? p.shift[c] : p.x[c];
        for (long r = r0 + rl; r < r1; r += 4) {
            const int b = (int)(r / p.L), l = (int)(r % p.L);
            const float v = p.x[((long)b * p.Lp_x + l) * p.C + c] - sh;
            a += v; q = fmaf(v, v, q);
        }
    }
    s1[rl][cl] = a; s2[rl][cl] = q;
    __syncthreads();
    if (rl == 0 && c < p.C) {
        const double sa = (double)s1[0][cl] + s1[1][cl] + s1[2][cl] + s1[3][cl];
        const double sq = (double)s2[0][cl] + s2[1][cl] + s2[2][cl] + s2[3][cl];
        atomicAdd(&p.sums[c], sa);
        atomicAdd(&p.sums[p.C + c], sq);
    }
}

// Batch statistics from the producing GEMM's per-tile partials (T2Gemm.stat_out): per channel, the tiles' (shift, sum d, sum d^2, n)
// are merged in double with Chan's update and written to `sums` in the representation bn_finalize_kernel (and the sync-BN
// all-reduce) expect: sums of (x - s) and (x - s)^2 about the layer's common shift s, then the row count.
__global__ __launch_bounds__(1024) void bn_merge_tiles_kernel(BnK p, const float* ts, int tile_M) {
    // block = 64 channels x 16 tile lanes: a lane merges every 16th tile (its loads are independent of each other - the whole
    // table is a few hundred KB, the cost is latency), then the 16 partial results are merged pairwise through LDS
    __shared__ double sn[16][64], sm[16][64], sq[16][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int ntile = (tile_M + 127) / 128;
    const int Lp = p.Lp_x, L = p.L;
    auto upto = [&](int x) { const int r = x % Lp; return (x / Lp) * L + (r < L ? r : L); };     // valid rows among GEMM rows [0, x)
    auto merge = [](double& n, double& mean, double& M2, double nt, double mt, double m2t) {
        if (nt <= 0) return;
        const double tot = n + nt, delta = mt - mean;
        mean += delta * nt / tot;
        M2 += m2t + delta * delta * n * nt / tot;
        n = tot;
    };
    double n = 0, mean = 0, M2 = 0;
    if (c < p.C) {
        for (int t0 = tl; t0 < ntile; t0 += 64) {
            float sh[4], s1[4], s2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 16 * u;
                const float* o = ts + (long)(t < ntile ? t : 0) * 3 * p.C + c;
                sh[u] = o[0]; s1[u] = o[p.C]; s2[u] = o[2 * (long)p.C];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 16 * u;
                if (t >= ntile) continue;
                const int r0 = t * 128, r1 = r0 + 128 < tile_M ? r0 + 128 : tile_M;
                const double nt = (double)(upto(r1) - upto(r0));
                if (nt <= 0) continue;
                merge(n, mean, M2, nt, (double)sh[u] + (double)s1[u] / nt, (double)s2[u] - (double)s1[u] * s1[u] / nt);
            }
        }
    }
    sn[tl][cl] = n; sm[tl][cl] = mean; sq[tl][cl] = M2;
    __syncthreads();
    for (int s_ = 8; s_ >= 1; s_ >>= 1) {
        if (tl < s_) {
            double a = sn[tl][cl], b = sm[tl][cl], d = sq[tl][cl];
            merge(a, b, d, sn[tl + s_][cl], sm[tl + s_][cl], sq[tl + s_][cl]);
            sn[tl][cl] = a; sm[tl][cl] = b; sq[tl][cl] = d;
        }
        __syncthreads();
    }
    if (tl == 0 && c < p.C) {
        n = sn[0][cl]; mean = sm[0][cl]; M2 = sq[0][cl];
        const double s = (double)(p.shift ? p.shift[c] : p.x[c]);
        const double dm = mean - s;
        p.sums[c] = n * dm;
        p.sums[p.C + c] = (M2 > 0 ? M2 : 0) + n * dm * dm;
        if (c == 0) p.sums[2 * p.C] = n;
    }
}

__global__ void bn_finalize_kernel(BnK p) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    if (p.training) {
        const double n = p.sums[2 * p.C];                // rows behind the sums (all ranks' rows under sync-BN)
        const double md = p.sums[c] / n;                 // mean of (x - shift) (bn_stats_kernel)
        double var = p.sums[p.C + c] / n - md * md;
        if (var < 0) var = 0;
        const double m = md + (double)(p.shift ? p.shift[c] : p.x[c]);
        p.mean[c] = (float)m;
        p.invstd[c] = (float)(1.0 / sqrt(var + (double)p.eps));
        if (p.rmean) {
            const double unb = n > 1 ? var * n / (n - 1) : var;
            p.rmean[c] = (1.f - p.momentum) * p.rmean[c] + p.momentum * (float)m;
            p.rvar[c] = (1.f - p.momentum) * p.rvar[c] + p.momentum * (float)unb;
        }
    } else {
        p.mean[c] = p.rmean[c];
        p.invstd[c] = 1.f / sqrtf(p.rvar[c] + p.eps);
    }
}

__global__ void bn_apply_kernel(BnK p) {
    const long n = (long)p.B * p.Lp_y * p.C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % p.C);
        const long row = i / p.C;
        const int lp = (int)(row % p.Lp_y), b = (int)(row / p.Lp_y);
        const int l = lp - p.pad_y;
        float v = 0.f;
        if (l >= 0 && l < p.L) {
            const float xv = p.x[((long)b * p.Lp_x + l) * p.C + c];
            v = (xv - p.mean[c]) * p.invstd[c] * p.gamma[c] + p.beta[c];
            if (p.act == 1) v = fmaxf(v, 0.f);
            else if (p.act == 2) v = t2_tanh(v);
            if (p.drop) v *= p.drop[((long)b * p.L + l) * p.C + c];
            if (p.res) v += p.res[((long)b * p.Lp_res + p.pad_res + l) * p.C + c];
            if (p.len && l >= p.len[b]) v = p.fill;
        }
        p.y[i] = v;
    }
}

// dz = dy * drop * act'(bn(x)); sums[0][c] = sum dz, sums[1][c] = sum dz * xhat
__device__ __forceinline__ float bn_dz(const BnK& p, int b, int l, int c, float& xhat) {
    const float xv = p.x[((long)b * p.Lp_x + l) * p.C + c];
    xhat = (xv - p.mean[c]) * p.invstd[c];
    float g = p.dy[((long)b * p.Lp_dy + p.pad_dy + l) * p.C + c];
    if (p.drop) g *= p.drop[((long)b * p.L + l) * p.C + c];
    const float pre = xhat * p.gamma[c] + p.beta[c];
    if (p.act == 1) g = pre > 0.f ? g : 0.f;
    else if (p.act == 2) { const float t = t2_tanh(pre); g *= (1.f - t * t); }
    return g;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnK p, int rows_per_block) {
    __shared__ float s1[4][64], s2[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long R = (long)p.B * p.L;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > R) r1 = R;
    float a = 0.f, q = 0.f;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(&p.sums[2 * p.C], (double)R);
    if (c < p.C) {
        for (long r = r0 + rl; r < r1; r += 4) {
            const int b = (int)(r / p.L), l = (int)(r % p.L);
            float xh;
            const float dz = bn_dz(p, b, l, c, xh);
            a += dz; q = fmaf(dz, xh, q);
        }
    }
    s1[rl][cl] = a; s2[rl][cl] = q;
    __syncthreads();
    if (rl == 0 && c < p.C) {
        atomicAdd(&p.sums[c], (double)s1[0][cl] + s1[1][cl] + s1[2][cl] + s1[3][cl]);
        atomicAdd(&p.sums[p.C + c], (double)s2[0][cl] + s2[1][cl] + s2[2][cl] + s2[3][cl]);
    }
}

__global__ void bn_bwd_apply_kernel(BnK p) {
    const long n = (long)p.B * p.Lp_dx * p.C;
    const float invn = (float)(1.0 / p.sums[2 * p.C]);      // rows behind the sums (all ranks' rows under sync-BN)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % p.C);
        const long row = i / p.C;
        const int lp = (int)(row % p.Lp_dx), b = (int)(row / p.Lp_dx);
        const int l = lp - p.pad_dx;
        float v = 0.f;
        if (l >= 0 && l < p.L) {
            float xh;
            const float dz = bn_dz(p, b, l, c, xh);
            const float gi = p.gamma[c] * p.invstd[c];
            if (p.training) v = gi * (dz - (float)p.sums[c] * invn - xh * (float)p.sums[p.C + c] * invn);
            else v = gi * dz;
        }
        p.dx[i] = v;
        if (lp == 0 && b == 0) {   // one thread per channel folds the parameter gradients (this rank's share under sync-BN:
                                   // the gradient all-reduce adds the ranks' shares, so the globally summed `sums` are scaled back)
            p.dbeta[c] += (float)(p.sums[c] * p.grad_share);
            p.dgamma[c] += (float)(p.sums[p.C + c] * p.grad_share);
        }
    }
}

void to_bnk(const T2Bn* s, BnK& k) {
    k.B = s->B; k.L = s->L; k.C = s->C; k.x = s->x; k.Lp_x = s->Lp_x; k.gamma = s->gamma; k.beta = s->beta;
    k.rmean = s->running_mean; k.rvar = s->running_var; k.training = s->training; k.momentum = s->momentum; k.eps = s->eps;
    k.sums = s->sums; k.mean = s->mean; k.invstd = s->invstd; k.act = s->act; k.drop = s->drop;
    k.res = s->res; k.Lp_res = s->Lp_res; k.pad_res = s->pad_res; k.len = s->len; k.fill = s->fill;
    k.y = s->y; k.Lp_y = s->Lp_y; k.pad_y = s->pad_y; k.dy = s->dy; k.Lp_dy = s->Lp_dy; k.pad_dy = s->pad_dy;
    k.dx = s->dx; k.Lp_dx = s->Lp_dx; k.pad_dx = s->pad_dx; k.dgamma = s->dgamma; k.dbeta = s->dbeta;
    k.shift = s->shift; k.grad_share = s->grad_share > 0.f ? (double)s->grad_share : 1.0;
}

inline int ew_grid(long n) { long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

// ---------------------------------------------------------------------------------------------
// misc elementwise
// ---------------------------------------------------------------------------------------------
// column sums: out[c] += sum_r x[r*ld + c]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, long ld, long R, int C, float* out, int rows_per_block) {
    __shared__ float s1[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > R) r1 = R;
    float a = 0.f;
    if (c < C) for (long r = r0 + rl; r < r1; r += 4) a += x[r * ld + c];
    s1[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && c < C) atomicAdd(&out[c], s1[0][cl] + s1[1][cl] + s1[2][cl] + s1[3][cl]);
}

// 16-byte variant (C % 4 == 0, ld % 4 == 0, aligned base): a wave reads 1 KB of one row per instruction, 8 rows in flight
__global__ __launch_bounds__(256) void colsum4_kernel(const float* x, long ld, long R, int C, float* out, int rows_per_block) {
    __shared__ f32x4 s1[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + cl * 4;
    const int cc = c < C ? c : 0;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > R) r1 = R;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    long r = r0 + rl;
    for (; r + 28 < r1; r += 32) {
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(x + (r + 4 * i) * ld + cc);
#pragma unroll
        for (int i = 0; i < 8; ++i) a += v[i];
    }
    for (; r < r1; r += 4) a += *reinterpret_cast<const f32x4*>(x + r * ld + cc);
    s1[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && c < C) {
        const f32x4 t = s1[0][cl] + s1[1][cl] + s1[2][cl] + s1[3][cl];
        atomicAdd(&out[c], t[0]); atomicAdd(&out[c + 1], t[1]); atomicAdd(&out[c + 2], t[2]); atomicAdd(&out[c + 3], t[3]);
    }
}

// (B,T,M) batch-major -> [T+1][B][M] time-major with a zero frame at slot 0 (F.pad(mel,(0,0,1,0)), model/tacotron2.py:255)
__global__ void mel_to_tm_kernel(const float* mel, float* out, int B, int T, int M) {
    const long n = (long)(T + 1) * B * M;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i % M);
        const long row = i / M;
        const int b = (int)(row % B), s = (int)(row / B);
        out[i] = s == 0 ? 0.f : mel[((long)b * T + (s - 1)) * M + m];
    }
}

// generic (D0, D1, C) -> (D1, D0, C) transpose (mask / gradient re-layout between batch- and time-major)
__global__ void swap01_kernel(const float* in, float* out, int D0, int D1, int C, int accumulate) {
    const long n = (long)D0 * D1 * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long row = i / C;
        const int d1 = (int)(row % D1), d0 = (int)(row / D1);
        const long o = ((long)d1 * D0 + d0) * C + c;
        if (accumulate) out[o] += in[i]; else out[o] = in[i];
    }
}

// proj [T][B][M+1] (time-major, col M = stop logit) -> mels (B,T,M) masked 0, gates (B,T,1) masked -1000,
// postnet input (B,T+4,M) padded layout holding the UNMASKED mels (model/tacotron2.py:327-345)
__global__ void finalize_fwd_kernel(const float* proj, long ldp, const int32_t* len, float* mels, float* gates, float* post_in,
                                    int B, int T, int M) {
    const int Tp = T + 4, M1 = M + 1;
    const long n = (long)B * Tp * M1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i % M1);
        const long row = i / M1;
        const int tp = (int)(row % Tp), b = (int)(row / Tp);
        const int t = tp - 2;
        const bool in = t >= 0 && t < T;
        const float v = in ? proj[((long)t * B + b) * ldp + m] : 0.f;
        if (m < M) {
            if (post_in) post_in[((long)b * Tp + tp) * M + m] = v;
            if (in) mels[((long)b * T + t) * M + m] = (t >= len[b]) ? 0.f : v;
        } else if (in) {
            gates[(long)b * T + t] = (t >= len[b]) ? -1000.f : v;
        }
    }
}

// Loss (model/tts_model.py:197-201) and its gradient in one pass.
//   loss = mean BCEWithLogits(gates, gate_tgt) + mean (mels - tgt)^2 + mean (post - tgt)^2   (padding included)
// Gradients are w.r.t. the UNDERLYING (pre-masking) tensors: masked positions are constants -> zero gradient.
//   d_post (B,T,M) dense;  dproj [T][B][M+1] time-major gets d_mels + d_post in cols < M and d_gate in col M.
__global__ __launch_bounds__(256) void loss_kernel(const float* mels, const float* post, const float* gates, const float* mel_tgt,
                                                   const float* gate_tgt, const int32_t* len, int B, int T, int M,
                                                   double* loss3, float* d_post, float* dproj, float gscale,
                                                   float* d_mels = nullptr, float* d_gates = nullptr) {
    __shared__ double red[3][4];
    const long nm = (long)B * T * M, ng = (long)B * T;
    const float sm = 1.f / (float)nm, sg = 1.f / (float)ng;
    double a_g = 0, a_m = 0, a_p = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nm + ng; i += (long)gridDim.x * blockDim.x) {
        if (i < nm) {
            const int m = (int)(i % M);
            const long row = i / M;
            const int t = (int)(row % T), b = (int)(row / T);
            const bool masked = t >= len[b];
            const float tg = mel_tgt[i];
            const float e1 = mels[i] - tg, e2 = post[i] - tg;
            a_m += (double)e1 * e1; a_p += (double)e2 * e2;
            const float g1 = masked ? 0.f : 2.f * e1 * sm * gscale;
            const float g2 = masked ? 0.f : 2.f * e2 * sm * gscale;
            if (d_post) d_post[i] = g2;
            if (d_mels) d_mels[i] = g1;
            if (dproj) dproj[((long)t * B + b) * (M + 1) + m] = g1 + g2;
        } else {
            const long r = i - nm;
            const int t = (int)(r % T), b = (int)(r / T);
            const bool masked = t >= len[b];
            const float x = gates[r], y = gate_tgt[r];
            a_g += (double)(fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));
            const float g = masked ? 0.f : (t2_sigmoid(x) - y) * sg * gscale;
            if (dproj) dproj[((long)t * B + b) * (M + 1) + M] = g;
            if (d_gates) d_gates[r] = g;
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    a_g = t2_wave_sum_d(a_g); a_m = t2_wave_sum_d(a_m); a_p = t2_wave_sum_d(a_p);
    if (lane == 0) { red[0][w] = a_g; red[1][w] = a_m; red[2][w] = a_p; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        atomicAdd(&loss3[threadIdx.x], s / (threadIdx.x == 0 ? (double)ng : (double)nm));
    }
}

// Arbitrary upstream gradients (autograd path) -> the two tensors backward_tf consumes.  Masked positions are constants
// (masked_fill, model/tacotron2.py:343-345) so their gradient is dropped.
__global__ void outgrad_pack_kernel(const float* d_mels, const float* d_post, const float* d_gates, const int32_t* len,
                                    float* d_post_out, float* dproj, int B, int T, int M) {
    const long n = (long)B * T * (M + 1);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i % (M + 1));
        const long row = i / (M + 1);
        const int t = (int)(row % T), b = (int)(row / T);
        const bool masked = t >= len[b];
        if (m < M) {
            const long j = row * M + m;
            const float g1 = (masked || !d_mels) ? 0.f : d_mels[j];
            const float g2 = (masked || !d_post) ? 0.f : d_post[j];
            d_post_out[j] = g2;
            dproj[((long)t * B + b) * (M + 1) + m] = g1 + g2;
        } else {
            dproj[((long)t * B + b) * (M + 1) + M] = (masked || !d_gates) ? 0.f : d_gates[row];
        }
    }
}

// Gradient of finalize + postnet residual: dproj[t][b][m] += dpost_in[b][t][m]  (dpost_in in shifted rows b*Tp + t)
__global__ void finalize_bwd_kernel(const float* dpost_in, float* dproj, int B, int T, int M) {
    const int Tp = T + 4;
    const long n = (long)B * T * M;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i % M);
        const long row = i / M;
        const int t = (int)(row % T), b = (int)(row / T);
        dproj[((long)t * B + b) * (M + 1) + m] += dpost_in[((long)b * Tp + t) * M + m];
    }
}

// out = relu'(y) * mask * g   (prenet layers: y is the stored post-dropout output; y > 0 <=> pre-activation > 0 and kept)
__global__ void relu_mask_bwd_kernel(const float* g, const float* y, const float* mask, float* out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = y[i] > 0.f ? g[i] * (mask ? mask[i] : 1.f) : 0.f;
}

// speaker conditioning: enc[b][l][:E] = tanh(enc + emb[spk[b]]) (model/tacotron2.py:202), written into the first E
// columns of memory (B,L,Ef); description vector d[b][0:Ef-E] broadcast into the remaining columns (:203-212)
__global__ void condition_fwd_kernel(const float* enc, const float* spk_table, const int32_t* spk, const float* desc,
                                     float* memory, int B, int L, int E, int Ef) {
    const long n = (long)B * L * Ef;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i % Ef);
        const long row = i / Ef;
        const int b = (int)(row / L);
        float v;
        if (e < E) {
            v = enc[row * E + e];
            if (spk_table) v = t2_tanh(v + spk_table[(long)spk[b] * E + e]);
        } else {
            v = desc[(long)b * (Ef - E) + (e - E)];
        }
        memory[i] = v;
    }
}

// backward of the above: denc = dmem[:, :, :E] * (1 - mem^2) (if speakers), dspk_table[spk[b]] += sum_l denc,
// ddesc[b][:] += sum_l dmem[b][l][E:]
__global__ void condition_bwd_kernel(const float* dmem, const float* memory, const int32_t* spk, int has_spk, float* denc,
                                     float* dspk_table, float* ddesc, int B, int L, int E, int Ef) {
    // grid (B, ceil(L/16)): 16 rows per workgroup with all their loads independent (a single workgroup per sample walking
    // L rows with a dependent accumulation took 0.67 ms); the row sums are combined with atomics
    const int b = blockIdx.x, l0 = blockIdx.y * 16;
    for (int e = threadIdx.x; e < Ef; e += blockDim.x) {
        float g[16], m[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int l = l0 + r < L ? l0 + r : L - 1;
            const long i = ((long)b * L + l) * Ef + e;
            g[r] = dmem[i];
            m[r] = (e < E && has_spk) ? memory[i] : 0.f;
        }
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (l0 + r >= L) continue;
            float v = g[r];
            if (e < E) {
                if (has_spk) v *= (1.f - m[r] * m[r]);
                denc[((long)b * L + l0 + r) * E + e] = v;
            }
            acc += v;
        }
        if (e < E) { if (has_spk) atomicAdd(&dspk_table[(long)spk[b] * E + e], acc); }
        else if (ddesc) atomicAdd(&ddesc[(long)b * (Ef - E) + (e - E)], acc);
    }
}

__global__ void tanh_bias_kernel(float* x, const float* bias, long rows, int C) {
    const long n = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] = t2_tanh(x[i] + (bias ? bias[i % C] : 0.f));
}
__global__ void tanh_bwd_kernel(const float* g, const float* y, float* out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = g[i] * (1.f - y[i] * y[i]);
}

// scale mask: 0 with probability p, 1/(1-p) otherwise; Philox4x32-10 keyed by (seed, stream), counter = element/4
__global__ void philox_mask_kernel(float* out, long n, float p, uint64_t seed, uint64_t stream_id) {
    const float keep_scale = 1.f / (1.f - p);
    const long n4 = (n + 3) / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        uint32_t r[4];
        t2_philox4((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32), (uint32_t)seed,
                   (uint32_t)(seed >> 32), r);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long e = 4 * i + j;
            if (e < n) {
                const float u = (float)(r[j] >> 8) * (1.0f / 16777216.0f);   // [0,1)
                out[e] = u < p ? 0.f : keep_scale;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// optimizer: global-norm clip + Adam with L2-in-gradient weight decay on one flat fp32 buffer
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long n, double* out) {
    __shared__ double red[4];
    double a = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = g[i];
        a += v * v;
    }
    a = t2_wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long n, const double* sumsq, float max_norm,
                            float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float gscale) {
    float clip = 1.f;
    if (sumsq && !isfinite(*sumsq)) return;    // non-finite gradient norm: skip the step (see include/tacotron2_amd.h)
    if (sumsq && max_norm > 0.f) {
        const float tot = (float)sqrt(*sumsq) * gscale;
        const float c = max_norm / (tot + 1e-6f);
        clip = c < 1.f ? c : 1.f;
    }
    clip *= gscale;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float gr = g[i] * clip + wd * pv;
        const float mv = b1 * m[i] + (1.f - b1) * gr;
        const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mv; v[i] = vv;
        p[i] = pv - lr * (mv / bc1) / (sqrtf(vv / bc2) + eps);
    }
}

__global__ void guard_poison_kernel(const uint32_t* flag, float* x, long n) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = __builtin_nanf("");
}

// t2_zero_regions: up to 64 regions (contiguous, or rows of row_bytes at a stride) cleared by ONE launch; blockIdx.y = region.
struct ZeroK { T2ZeroRegions r; };
__global__ __launch_bounds__(256) void zero_regions_kernel(ZeroK k) {
    const int y = blockIdx.y;
    char* base = (char*)k.r.p[y];
    const long rb = k.r.row_bytes[y], nr = k.r.nrows[y], sb = k.r.stride_bytes[y];
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
    if (nr == 1 || sb == rb) {                 // one contiguous run
        const long n = rb * nr;
        if ((((uintptr_t)base) & 15) == 0 && (n & 15) == 0) {
            f32x4* q = (f32x4*)base;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            for (long i = tid; i < (n >> 4); i += nth) q[i] = z;
        } else {
            uint32_t* q = (uint32_t*)base;
            for (long i = tid; i < (n >> 2); i += nth) q[i] = 0u;
        }
    } else {
        const long rw = rb >> 2;
        for (long i = tid; i < rw * nr; i += nth) {
            const long row = i / rw, c = i - row * rw;
            ((uint32_t*)(base + row * sb))[c] = 0u;
        }
    }
}

// Stream-concurrency probe (t2_stream_probe_*): a chain of dependent one-thread launches, and one wave that idles for a
// bounded wall-clock time (s_memrealtime: 100 MHz, independent of the shader clock; s_sleep between polls).
__global__ void probe_tick_kernel(uint32_t* word) { word[0] += 1u; }
__global__ void probe_spin_kernel(uint32_t* word, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned polls = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && polls < (1u << 24)) { __builtin_amdgcn_s_sleep(16); ++polls; }
    word[0] = polls;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int t2_zero_regions(const T2ZeroRegions* r, void* stream) {
    (void)hipGetLastError();
    T2_REQUIRE(r && r->n >= 0 && r->n <= 64, "t2_zero_regions: 0..64 regions");
    if (r->n == 0) return T2_OK;
    long most = 0;
    for (int i = 0; i < r->n; ++i) {
        T2_REQUIRE(r->p[i] && r->row_bytes[i] > 0 && r->nrows[i] > 0 && (r->row_bytes[i] & 3) == 0 && ((uintptr_t)r->p[i] & 3) == 0 &&
                   (r->nrows[i] == 1 || ((r->stride_bytes[i] & 3) == 0 && r->stride_bytes[i] >= r->row_bytes[i])),
                   "t2_zero_regions: regions are 4-byte aligned runs of whole words (rows: stride >= row, a multiple of 4)");
        const long n = r->row_bytes[i] * r->nrows[i];
        if (n > most) most = n;
    }
    long gx = (most + 16 * 256 * 8 - 1) / (16 * 256 * 8);          // ~8 x 16-byte stores per thread of the largest region
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    ZeroK k; k.r = *r;
    hipLaunchKernelGGL(zero_regions_kernel, dim3((unsigned)gx, r->n), dim3(256), 0, ST, k);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_stream_probe_chain(uint32_t* word, int n, void* stream) {
    (void)hipGetLastError();
    T2_REQUIRE(word && n >= 0 && n <= 100000, "t2_stream_probe_chain: bad arguments");
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(probe_tick_kernel, dim3(1), dim3(1), 0, ST, word);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_stream_probe_spin(uint32_t* word, int microseconds, void* stream) {
    (void)hipGetLastError();
    T2_REQUIRE(word && microseconds >= 0 && microseconds <= 50000, "t2_stream_probe_spin: 0..50000 us");
    hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(64), 0, ST, word, (unsigned long long)microseconds * 100ull);
    T2_CHECK_LAUNCH(); return T2_OK;
}

extern "C" int t2_guard_poison(const uint32_t* flag, float* x, int64_t n, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(flag && x && n >= 0, "t2_guard_poison: bad arguments");
    if (n == 0) return T2_OK;
    hipLaunchKernelGGL(guard_poison_kernel, dim3(ew_grid(n) > 1024 ? 1024 : ew_grid(n)), dim3(256), 0, ST, flag, x, (long)n);
    T2_CHECK_LAUNCH(); return T2_OK;
}

extern "C" int t2_embedding_fwd(const int64_t* idx, const float* table, float* out, int B, int L, int E, int pad, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(idx && table && out, "t2_embedding_fwd: null");
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(ew_grid((long)B * (L + 2 * pad) * E)), dim3(256), 0, ST, idx, table, out, B, L, E, pad);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_embedding_bwd(const int64_t* idx, const float* dout, float* dtable, int B, int L, int E, int Lp, int pad, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(idx && dout && dtable, "t2_embedding_bwd: null");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(ew_grid((long)B * L * E)), dim3(256), 0, ST, idx, dout, dtable, B, L, E, Lp, pad);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_pack_conv_weight(const float* w, float* wp, int Co, int Ci, int K, int flip, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(w && wp, "t2_pack_conv_weight: null");
    hipLaunchKernelGGL(pack_conv_w_kernel, dim3(ew_grid((long)Co * Ci * K)), dim3(256), 0, ST, w, wp, Co, Ci, K, flip);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_unpack_conv_wgrad(const float* gp, float* g, int Co, int Ci, int K, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(gp && g, "t2_unpack_conv_wgrad: null");
    hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3(ew_grid((long)Co * Ci * K)), dim3(256), 0, ST, gp, g, Co, Ci, K);
    T2_CHECK_LAUNCH(); return T2_OK;
}

extern "C" int t2_bn_fwd(const T2Bn* s, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(s && s->x && s->gamma && s->beta && s->mean && s->invstd && s->y, "t2_bn_fwd: null operand");
    T2_REQUIRE(s->training ? (s->sums != nullptr) : (s->running_mean && s->running_var), "t2_bn_fwd: stats operands");
    T2_REQUIRE(s->phase >= 0 && s->phase <= 2 && (s->phase == 0 || (s->training && s->shift)),
               "t2_bn_fwd: phases 1/2 (sync-BN) need training statistics and a rank-independent shift");
    BnK k; to_bnk(s, k);
    if (s->training && s->phase != 2) {
        if (s->tile_stats) {     // statistics from the producing GEMM's epilogue: merged per channel, `sums` is written, not added to
            T2_REQUIRE(s->tile_M > 0 && (long)s->tile_M <= (long)s->B * s->Lp_x, "t2_bn_fwd: tile_M is the producing GEMM's row count");
            hipLaunchKernelGGL(bn_merge_tiles_kernel, dim3(t2_cdiv(s->C, 64)), dim3(1024), 0, ST, k, s->tile_stats, s->tile_M);
        } else {
            if (!s->sums_prezeroed) (void)hipMemsetAsync(s->sums, 0, sizeof(double) * (2 * s->C + 2), ST);
            const long R = (long)s->B * s->L;
            const int rpb = 128;
            hipLaunchKernelGGL(bn_stats_kernel, dim3(t2_cdiv(s->C, 64), t2_cdiv(R, rpb)), dim3(256), 0, ST, k, rpb);
        }
    }
    if (s->phase != 1) {
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(t2_cdiv(s->C, 256)), dim3(256), 0, ST, k);
        hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid((long)s->B * s->Lp_y * s->C)), dim3(256), 0, ST, k);
    }
    T2_CHECK_LAUNCH(); return T2_OK;
}

extern "C" int t2_bn_bwd(const T2Bn* s, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(s && s->x && s->gamma && s->beta && s->mean && s->invstd && s->dy && s->dx && s->sums && s->dgamma && s->dbeta,
               "t2_bn_bwd: null operand");
    T2_REQUIRE(s->phase >= 0 && s->phase <= 2, "t2_bn_bwd: bad phase");
    BnK k; to_bnk(s, k);
    if (s->phase != 2) {
        if (!s->sums_prezeroed) (void)hipMemsetAsync(s->sums, 0, sizeof(double) * (2 * s->C + 2), ST);
        const long R = (long)s->B * s->L;
        const int rpb = 128;
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(t2_cdiv(s->C, 64), t2_cdiv(R, rpb)), dim3(256), 0, ST, k, rpb);
    }
    if (s->phase != 1) hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid((long)s->B * s->Lp_dx * s->C)), dim3(256), 0, ST, k);
    T2_CHECK_LAUNCH(); return T2_OK;
}

extern "C" int t2_colsum(const float* x, int64_t ld, int64_t R, int C, float* out, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(x && out && R > 0 && C > 0, "t2_colsum: bad arguments");
    if (C % 4 == 0 && ld % 4 == 0 && t2_aligned16(x) && C >= 256) {
        // enough row blocks for ~4 workgroups per CU
        const int cb = t2_cdiv(C, 256);
        long rpb = t2_cdiv(R, t2_cdiv(1024, cb)); rpb = (rpb + 31) / 32 * 32; if (rpb < 32) rpb = 32;
        hipLaunchKernelGGL(colsum4_kernel, dim3(cb, t2_cdiv(R, rpb)), dim3(256), 0, ST, x, (long)ld, (long)R, C, out, (int)rpb);
    } else {
        const int rpb = 256;
        hipLaunchKernelGGL(colsum_kernel, dim3(t2_cdiv(C, 64), t2_cdiv(R, rpb)), dim3(256), 0, ST, x, (long)ld, (long)R, C, out, rpb);
    }
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_mel_to_tm(const float* mel, float* out, int B, int T, int M, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(mel && out, "t2_mel_to_tm: null");
    hipLaunchKernelGGL(mel_to_tm_kernel, dim3(ew_grid((long)(T + 1) * B * M)), dim3(256), 0, ST, mel, out, B, T, M);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_swap01(const float* in, float* out, int D0, int D1, int C, int accumulate, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(in && out, "t2_swap01: null");
    hipLaunchKernelGGL(swap01_kernel, dim3(ew_grid((long)D0 * D1 * C)), dim3(256), 0, ST, in, out, D0, D1, C, accumulate);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_finalize_fwd(const float* proj, int64_t ld_proj, const int32_t* len, float* mels, float* gates, float* post_in,
                               int B, int T, int M, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(proj && len && mels && gates, "t2_finalize_fwd: null");
    hipLaunchKernelGGL(finalize_fwd_kernel, dim3(ew_grid((long)B * (T + 4) * (M + 1))), dim3(256), 0, ST, proj, (long)ld_proj, len, mels, gates,
                       post_in, B, T, M);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_loss_fwd_bwd(const float* mels, const float* post, const float* gates, const float* mel_tgt,
                               const float* gate_tgt, const int32_t* len, int B, int T, int M, double* loss3, float* d_post,
                               float* dproj, float grad_scale, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(mels && post && gates && mel_tgt && gate_tgt && len && loss3, "t2_loss_fwd_bwd: null");
    (void)hipMemsetAsync(loss3, 0, 3 * sizeof(double), ST);
    hipLaunchKernelGGL(loss_kernel, dim3(ew_grid((long)B * T * (M + 1))), dim3(256), 0, ST, mels, post, gates, mel_tgt, gate_tgt,
                       len, B, T, M, loss3, d_post, dproj, grad_scale);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_loss_terms(const float* mels, const float* post, const float* gates, const float* mel_tgt, const float* gate_tgt,
                             const int32_t* len, int B, int T, int M, double* loss3, float* d_mels, float* d_post, float* d_gates,
                             float grad_scale, void* stream) {
    (void)hipGetLastError();
    T2_REQUIRE(mels && post && gates && mel_tgt && gate_tgt && len && loss3, "t2_loss_terms: null");
    (void)hipMemsetAsync(loss3, 0, 3 * sizeof(double), ST);
    hipLaunchKernelGGL(loss_kernel, dim3(ew_grid((long)B * T * (M + 1))), dim3(256), 0, ST, mels, post, gates, mel_tgt, gate_tgt,
                       len, B, T, M, loss3, d_post, (float*)nullptr, grad_scale, d_mels, d_gates);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_outgrad_pack(const float* d_mels, const float* d_post, const float* d_gates, const int32_t* len,
                               float* d_post_out, float* dproj, int B, int T, int M, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(len && d_post_out && dproj, "t2_outgrad_pack: null");
    hipLaunchKernelGGL(outgrad_pack_kernel, dim3(ew_grid((long)B * T * (M + 1))), dim3(256), 0, ST, d_mels, d_post, d_gates, len,
                       d_post_out, dproj, B, T, M);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_finalize_bwd(const float* dpost_in, float* dproj, int B, int T, int M, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(dpost_in && dproj, "t2_finalize_bwd: null");
    hipLaunchKernelGGL(finalize_bwd_kernel, dim3(ew_grid((long)B * T * M)), dim3(256), 0, ST, dpost_in, dproj, B, T, M);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_relu_mask_bwd(const float* g, const float* y, const float* mask, float* out, int64_t n, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(g && y && out, "t2_relu_mask_bwd: null");
    hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, g, y, mask, out, (long)n);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_condition_fwd(const float* enc, const float* spk_table, const int32_t* spk, const float* desc, float* memory,
                                int B, int L, int E, int Ef, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(enc && memory && (Ef == E || desc), "t2_condition_fwd: bad arguments");
    hipLaunchKernelGGL(condition_fwd_kernel, dim3(ew_grid((long)B * L * Ef)), dim3(256), 0, ST, enc, spk_table, spk, desc, memory, B,
                       L, E, Ef);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_condition_bwd(const float* dmem, const float* memory, const int32_t* spk, float* denc, float* dspk_table,
                                float* ddesc, int B, int L, int E, int Ef, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(dmem && memory && denc, "t2_condition_bwd: null");
    hipLaunchKernelGGL(condition_bwd_kernel, dim3(B, t2_cdiv(L, 16)), dim3(256), 0, ST, dmem, memory, spk, dspk_table != nullptr, denc,
                       dspk_table, ddesc, B, L, E, Ef);
    T2_CHECK_LAUNCH(); return T2_OK;
}
// y = leaky_relu(scale * x, slope) and y += alpha * x (HiFi-GAN generator glue, model/hifi_gan.py:89-97,198-216)
__global__ void leaky_relu_kernel(const float* x, float* y, long n, float scale, float slope) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = scale * x[i];
        y[i] = v > 0.f ? v : slope * v;
    }
}
__global__ void axpy_kernel(const float* x, float* y, long n, float alpha) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = fmaf(alpha, x[i], y[i]);
}
extern "C" int t2_leaky_relu(const float* x, float* y, int64_t n, float scale, float slope, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(x && y && n >= 0, "t2_leaky_relu: bad arguments");
    if (n == 0) return T2_OK;
    hipLaunchKernelGGL(leaky_relu_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, x, y, (long)n, scale, slope);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_axpy(const float* x, float* y, int64_t n, float alpha, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(x && y && n >= 0, "t2_axpy: bad arguments");
    if (n == 0) return T2_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, x, y, (long)n, alpha);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_tanh_bias(float* x, const float* bias, int64_t rows, int C, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(x, "t2_tanh_bias: null");
    hipLaunchKernelGGL(tanh_bias_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, ST, x, bias, (long)rows, C);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_tanh_bwd(const float* g, const float* y, float* out, int64_t n, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(g && y && out, "t2_tanh_bwd: null");
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, g, y, out, (long)n);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_philox_mask(float* out, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(out && n >= 0 && p >= 0.f && p < 1.f, "t2_philox_mask: bad arguments");
    if (n == 0) return T2_OK;
    hipLaunchKernelGGL(philox_mask_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, ST, out, (long)n, p, seed, stream_id);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_sumsq(const float* g, int64_t n, double* out, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(g && out, "t2_sumsq: null");
    (void)hipMemsetAsync(out, 0, sizeof(double), ST);
    hipLaunchKernelGGL(sumsq_kernel, dim3(ew_grid(n) > 1024 ? 1024 : ew_grid(n)), dim3(256), 0, ST, g, (long)n, out);
    T2_CHECK_LAUNCH(); return T2_OK;
}
extern "C" int t2_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, float max_norm, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(p && g && m && v && step >= 1, "t2_adam_step: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, p, g, m, v, (long)n, sumsq, max_norm, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2, grad_scale);
    T2_CHECK_LAUNCH(); return T2_OK;
}
