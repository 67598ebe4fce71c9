// Log-mel front-end on the device (datasets/tts_dataset.py:166-168,204 -> speech_utils TacotronMelSpectrogram, restated from
// datasets/prosody_dataset.py:39-50,67 and run/say.py:161-171; parity UNPINNED, see oracle/__init__.py):
//   centred STFT (reflect pad n_fft/2), periodic Hann window, n_fft = win = 1024, hop = 256, magnitude,
//   slaney-scale slaney-normalised mel filterbank, natural log of clamp(., 1e-5); output (frames, n_mels).
// The DFT is ONE fp32-MFMA GEMM whose A rows are the overlapping analysis windows of the padded signal
// (row f = 1024 samples starting at f*hop: lda = hop < K, the same overlapping-row trick as the conv layers) against the
// window-folded [cos ; -sin] basis; a magnitude kernel, the filterbank GEMM and a log kernel follow.  HBM-trivial.
#include "t2_common.hpp"

namespace {

__global__ void reflect_pad_kernel(const float* wav, float* out, long n, int pad) {
    const long total = n + 2L * pad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long j = i - pad;
        if (j < 0) j = -j;
        if (j >= n) j = 2 * (n - 1) - j;
        if (j < 0) j = 0;
        out[i] = wav[j];
    }
}

// spec [frames][2*nb] (re | im) -> mag [frames][ldm] (zero-padded columns)
__global__ void magnitude_kernel(const float* spec, float* mag, long frames, int nb, int ldm) {
    const long total = frames * ldm;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % ldm);
        const long f = i / ldm;
        float v = 0.f;
        if (k < nb) {
            const float re = spec[f * 2 * nb + k], im = spec[f * 2 * nb + nb + k];
            v = sqrtf(re * re + im * im);
        }
        mag[i] = v;
    }
}

__global__ void log_clamp_kernel(float* x, long n, float floor_) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] = logf(fmaxf(x[i], floor_));
}

// ---- batched variant: B utterances of one training batch in one pass (t2_logmel_batch_fwd) -------------------------------------
// padded [B][Np] (Np = Tp * hop): row b = the reflect-padded utterance b (n_b + 2 * pad samples), zeros behind it
__global__ void reflect_pad_batch_kernel(const float* wavs, long ld_wav, const int64_t* n, float* out, long Np, int pad) {
    const int b = blockIdx.y;
    const long nb = n[b];
    const float* w = wavs + (long)b * ld_wav;
    float* o = out + (long)b * Np;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < Np; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < nb + 2L * pad) {
            long j = i - pad;
            if (j < 0) j = -j;
            if (j >= nb) j = 2 * (nb - 1) - j;
            if (j < 0) j = 0;
            v = w[j];
        }
        o[i] = v;
    }
}

// tmp [B*Tp][n_mels] (linear mel energies; row b*Tp + f = frame f of utterance b) -> mel (B, T_out, n_mels) natural log, zero
// behind each utterance's frames; gate (B, T_out, 1) = 1 for every valid frame but the last (datasets/tts_dataset.py:213-214), 0
// elsewhere; mel_len[b] = frames of utterance b
__global__ void logmel_finalize_batch_kernel(const float* tmp, const int64_t* n, float* mel, float* gate, int32_t* mel_len, int Tp,
                                             long T_out, int n_mels, int hop, float floor_) {
    const int b = blockIdx.y;
    const long frames = 1 + n[b] / hop;
    const long total = T_out * n_mels;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long f = i / n_mels;
        const int m = (int)(i - f * n_mels);
        float v = 0.f;
        if (f < frames) v = logf(fmaxf(tmp[((long)b * Tp + f) * n_mels + m], floor_));
        mel[(long)b * total + i] = v;
        if (m == 0 && gate) gate[(long)b * T_out + f] = (f < frames - 1) ? 1.f : 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && mel_len) mel_len[b] = (int32_t)frames;
}

inline int grid_for(long n) { long g = (n + 255) / 256; return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" int t2_logmel_frames(int64_t n_samples, int hop) { return (int)(1 + n_samples / hop); }

extern "C" int t2_logmel_fwd(const float* wav, int64_t n, const float* basis, const float* fb, float* padded, float* spec,
                             float* mag, float* out, int n_fft, int hop, int n_mels, void* stream) {
    (void)hipGetLastError();   // drop stale sticky errors of other HIP users in this thread: only OUR launches are checked
    T2_REQUIRE(wav && basis && fb && padded && spec && mag && out, "t2_logmel_fwd: null operand");
    T2_REQUIRE(n > n_fft / 2 && n_fft % 4 == 0 && hop % 4 == 0, "t2_logmel_fwd: need n > n_fft/2 (reflect padding), n_fft,hop % 4 == 0");
    hipStream_t st = (hipStream_t)stream;
    const int pad = n_fft / 2, nb = n_fft / 2 + 1, ldm = (nb + 3) & ~3;
    const long frames = 1 + n / hop;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(grid_for(n + 2 * pad)), dim3(256), 0, st, wav, padded, (long)n, pad);
    T2Gemm g;
    memset(&g, 0, sizeof(g));
    g.A = padded; g.B = basis; g.C = spec; g.M = (int)frames; g.N = 2 * nb; g.K = n_fft;
    g.lda = hop; g.ldb = n_fft; g.ldc = 2 * nb; g.a_kmajor = 1; g.b_kmajor = 1; g.alpha = 1.f; g.splitk = 1; g.batch = 1;
    T2_TRY(t2_gemm(&g, stream));
    hipLaunchKernelGGL(magnitude_kernel, dim3(grid_for(frames * ldm)), dim3(256), 0, st, spec, mag, frames, nb, ldm);
    T2Gemm m;
    memset(&m, 0, sizeof(m));
    m.A = mag; m.B = fb; m.C = out; m.M = (int)frames; m.N = n_mels; m.K = ldm;
    m.lda = ldm; m.ldb = ldm; m.ldc = n_mels; m.a_kmajor = 1; m.b_kmajor = 1; m.alpha = 1.f; m.splitk = 1; m.batch = 1;
    T2_TRY(t2_gemm(&m, stream));
    hipLaunchKernelGGL(log_clamp_kernel, dim3(grid_for(frames * n_mels)), dim3(256), 0, st, out, frames * n_mels, 1e-5f);
    T2_CHECK_LAUNCH();
    return T2_OK;
}

// ---- one training batch ---------------------------------------------------------------------------------------------------------
extern "C" int t2_logmel_batch_workspace(int B, int64_t n_max, int n_fft, int hop, int n_mels, int64_t* out5) {
    T2_REQUIRE(out5 && B >= 1 && n_max > n_fft / 2 && hop > 0 && n_fft % hop == 0, "t2_logmel_batch_workspace: bad arguments");
    const int nb = n_fft / 2 + 1, ldm = (nb + 3) & ~3;
    const int64_t Tp = 1 + n_max / hop + n_fft / hop;           // frames of the longest utterance + the rows that run into the next one
    out5[0] = (int64_t)B * Tp * hop;                             // padded
    out5[1] = (int64_t)B * Tp * 2 * nb;                          // spec
    out5[2] = (int64_t)B * Tp * ldm;                             // mag
    out5[3] = (int64_t)B * Tp * n_mels;                          // tmp
    out5[4] = Tp;
    return T2_OK;
}

extern "C" int t2_logmel_batch_fwd(const float* wavs, int64_t ld_wav, const int64_t* n_dev, int B, int64_t n_max, const float* basis,
                                   const float* fb, float* padded, float* spec, float* mag, float* tmp, float* mel, int64_t T_out,
                                   float* gate, int32_t* mel_len, int n_fft, int hop, int n_mels, void* stream) {
    (void)hipGetLastError();
    T2_REQUIRE(wavs && n_dev && basis && fb && padded && spec && mag && tmp && mel, "t2_logmel_batch_fwd: null operand");
    T2_REQUIRE(B >= 1 && B <= 65535 && n_max > n_fft / 2 && n_max <= ld_wav && n_fft % 4 == 0 && hop % 4 == 0 && n_fft % hop == 0,
               "t2_logmel_batch_fwd: need 1 <= B, n_fft/2 < n_max <= ld_wav, n_fft % hop == 0, n_fft,hop % 4 == 0");
    T2_REQUIRE(T_out >= 1 + n_max / hop, "t2_logmel_batch_fwd: T_out is shorter than the longest utterance's frame count");
    hipStream_t st = (hipStream_t)stream;
    const int pad = n_fft / 2, nb = n_fft / 2 + 1, ldm = (nb + 3) & ~3;
    const long Tp = 1 + n_max / hop + n_fft / hop, Np = Tp * hop;
    const long rows = (long)B * Tp - (n_fft / hop - 1);          // the last window that still lies inside padded[B][Np]
    hipLaunchKernelGGL(reflect_pad_batch_kernel, dim3(grid_for(Np), B), dim3(256), 0, st, wavs, (long)ld_wav, n_dev, padded, Np, pad);
    T2Gemm g;
    memset(&g, 0, sizeof(g));
    g.A = padded; g.B = basis; g.C = spec; g.M = (int)rows; g.N = 2 * nb; g.K = n_fft;
    g.lda = hop; g.ldb = n_fft; g.ldc = 2 * nb; g.a_kmajor = 1; g.b_kmajor = 1; g.alpha = 1.f; g.splitk = 1; g.batch = 1;
    T2_TRY(t2_gemm(&g, stream));
    hipLaunchKernelGGL(magnitude_kernel, dim3(grid_for(rows * ldm)), dim3(256), 0, st, spec, mag, rows, nb, ldm);
    T2Gemm m;
    memset(&m, 0, sizeof(m));
    m.A = mag; m.B = fb; m.C = tmp; m.M = (int)rows; m.N = n_mels; m.K = ldm;
    m.lda = ldm; m.ldb = ldm; m.ldc = n_mels; m.a_kmajor = 1; m.b_kmajor = 1; m.alpha = 1.f; m.splitk = 1; m.batch = 1;
    T2_TRY(t2_gemm(&m, stream));
    hipLaunchKernelGGL(logmel_finalize_batch_kernel, dim3(grid_for(T_out * n_mels), B), dim3(256), 0, st, tmp, n_dev, mel, gate, mel_len,
                       (int)Tp, (long)T_out, n_mels, hop, 1e-5f);
    T2_CHECK_LAUNCH();
    return T2_OK;
}
