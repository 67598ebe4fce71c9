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
