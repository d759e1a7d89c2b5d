// Paraformer mel/STFT frontend on the GPU.
//   reference: funasr-mlx/src/paraformer.rs:195-412 (MelFrontend::{new, forward, compute_stft,
//   create_mel_filterbank}) -- single-threaded CPU Rust + rustfft, with a GPU->CPU->GPU round trip
//   of the audio and the features (:279-282, :363-366).  Here the samples stay in HBM.
//
// Kernel 1 (one block per frame): x*32768 -> pre-emphasis -> Hamming -> 400-point DFT -> |X|^2 ->
//   80 HTK mel filters -> ln(max(., 1e-10)).  n_fft = 400 is not a power of two; at 201 x 400 MACs
//   per frame (0.5 GFLOP for 30 s) an exact DFT is cheaper than staging a mixed-radix FFT: the twiddle
//   of (k, n) is entry (k*n mod n_fft) of a cos/sin table built in fp64, so every factor is correctly
//   rounded (the reference's own O(N^2) check, examples/validate_correctness.rs:19-59, evaluates the
//   angle in f32).  The frame, the table and the power spectrum live in LDS.
// Kernel 2: LFR stacking (m = 7, n = 6, left pad 3 x frame 0, tail clamp) + CMVN, one thread per
//   output element.
#include <math.h>

#include <algorithm>

#include <vector>

#include "common.hpp"

namespace omx {
namespace {

constexpr int kMaxFft = 512;
constexpr int kMaxMels = 128;

__global__ __launch_bounds__(256) void mel_power_kernel(const float* __restrict__ audio, int64_t n_samples, int n_frames,
                                                        int n_fft, int hop, const float* __restrict__ window,
                                                        const float* __restrict__ tw_cos, const float* __restrict__ tw_sin,
                                                        const float* __restrict__ fbank, int n_mels,
                                                        float* __restrict__ power_out, float* __restrict__ logmel,
                                                        float in_scale, float preemph, int log10_mode, int pad_partial = 0,
                                                        int mel_major = 0) {
    __shared__ float s_frame[kMaxFft], s_cos[kMaxFft], s_sin[kMaxFft], s_pow[kMaxFft / 2 + 1];
    const int frame = blockIdx.x;
    const int n_freqs = n_fft / 2 + 1;
    // Paraformer: an input shorter than one window gives ONE all-zero power frame (paraformer.rs:386-388); the SenseVoice frontend
    // (pad_partial) zero-pads every frame that runs past the end instead (funasr-nano-mlx/src/audio.rs:127-134)
    const bool too_short = !pad_partial && n_samples < n_fft;
    for (int i = threadIdx.x; i < n_fft; i += blockDim.x) {
        float w = 0.f;
        if (!too_short) {
            const int64_t g = (int64_t)frame * hop + i;
            // Paraformer: x*32768 then y[n] = x[n] - 0.97 x[n-1]; Whisper-style: the raw sample (scale 1, no pre-emphasis)
            const bool in_range = g < n_samples;
            const float cur = in_range ? audio[g] * in_scale : 0.f;
            const float y = (g == 0 || preemph == 0.f || !in_range) ? cur : cur - preemph * (audio[g - 1] * in_scale);
            w = y * window[i];
        }
        s_frame[i] = w;
        s_cos[i] = tw_cos[i];
        s_sin[i] = tw_sin[i];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n_freqs; k += blockDim.x) {
        float re = 0.f, im = 0.f;
        int idx = 0;   // (k * n) mod n_fft, advanced incrementally
        for (int n = 0; n < n_fft; ++n) {
            const float x = s_frame[n];
            re = fmaf(x, s_cos[idx], re);
            im = fmaf(-x, s_sin[idx], im);
            idx += k;
            if (idx >= n_fft) idx -= n_fft;
        }
        const float p = re * re + im * im;
        s_pow[k] = p;
        if (power_out) power_out[(size_t)frame * n_freqs + k] = p;
    }
    __syncthreads();
    for (int m = threadIdx.x; m < n_mels; m += blockDim.x) {
        const float* f = fbank + (size_t)m * n_freqs;
        float sum = 0.f;
        for (int k = 0; k < n_freqs; ++k) sum = fmaf(s_pow[k], f[k], sum);
        const float v = log10_mode ? log10f(fmaxf(sum, 1e-10f)) : logf(fmaxf(sum, 1e-10f));
        if (mel_major) logmel[(size_t)m * n_frames + frame] = v;      // [n_mels, n_frames]
        else logmel[(size_t)frame * n_mels + m] = v;
    }
}

__global__ __launch_bounds__(256) void lfr_cmvn_kernel(const float* __restrict__ logmel, int n_frames, int n_mels, int lfr_m,
                                                       int lfr_n, const float* __restrict__ addshift,
                                                       const float* __restrict__ rescale, float* __restrict__ out,
                                                       int t_out) {
    const int dim = lfr_m * n_mels;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)t_out * dim) return;
    const int t = (int)(i / dim), d = (int)(i % dim);
    const int m = d / n_mels, f = d % n_mels;
    const int left = (lfr_m - 1) / 2;
    int src = t * lfr_n + m - left;
    src = src < 0 ? 0 : (src >= n_frames ? n_frames - 1 : src);
    float v = logmel[(size_t)src * n_mels + f];
    if (addshift && rescale) v = (v + addshift[d]) * rescale[d];
    out[i] = v;
}

// funasr-nano-mlx/src/audio.rs:345-412: centre-based stacking over a MEL-MAJOR spectrogram [n_mels, n_frames]
__global__ __launch_bounds__(256) void lfr_center_kernel(const float* __restrict__ mel, int n_frames, int n_mels, int lfr_m, int lfr_n,
                                                         float* __restrict__ out, int t_out) {
    const int dim = lfr_m * n_mels;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)t_out * dim) return;
    const int t = (int)(i / dim), d = (int)(i % dim);
    const int j = d / n_mels, f = d % n_mels;
    int src = t * lfr_n + j - lfr_m / 2;
    src = src < 0 ? 0 : (src >= n_frames ? n_frames - 1 : src);
    out[i] = mel[(size_t)f * n_frames + src];
}

__global__ __launch_bounds__(256) void nonfinite_count_kernel(const float* __restrict__ x, int64_t n, unsigned* count) {
    unsigned bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad += !isfinite(x[i]);
    if (bad) atomicAdd(count, bad);
}

// Whisper normalisation (qwen3-asr-mlx/src/audio.rs:113-123): max over the whole spectrogram, clip at max - 8,
// (x + 4) / 4, written [n_mels, n_frames]
__device__ __forceinline__ unsigned orderable(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(256) void max_kernel(const float* __restrict__ x, int64_t n, unsigned* out) {
    unsigned best = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned k = orderable(x[i]);
        best = k > best ? k : best;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, best);
}
__global__ __launch_bounds__(256) void whisper_norm_kernel(const float* __restrict__ logmel, int n_frames, int n_mels,
                                                           const unsigned* __restrict__ max_key, float* __restrict__ out) {
    const unsigned k = *max_key;
    const float mx = __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
    const float floor_v = mx - 8.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)n_frames * n_mels; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n_frames), t = (int)(i % n_frames);
        out[i] = (fmaxf(logmel[(size_t)t * n_mels + m], floor_v) + 4.0f) / 4.0f;
    }
}

float hz_to_slaney_mel(float freq) {   // qwen3-asr-mlx/src/audio.rs:229-240
    const float f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f, min_log_mel = min_log_hz / f_sp, logstep = logf(6.4f) / 27.0f;
    return freq < min_log_hz ? freq / f_sp : min_log_mel + logf(freq / min_log_hz) / logstep;
}
float slaney_mel_to_hz(float mel) {    // :242-253
    const float f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f, min_log_mel = min_log_hz / f_sp, logstep = logf(6.4f) / 27.0f;
    return mel < min_log_mel ? f_sp * mel : min_log_hz * expf(logstep * (mel - min_log_mel));
}

float hz_to_mel(float hz) { return 2595.0f * log10f(1.0f + hz / 700.0f); }
float mel_to_hz(float mel) { return 700.0f * (powf(10.0f, mel / 2595.0f) - 1.0f); }

}  // namespace
}  // namespace omx

struct omx_mel_frontend_ {
    omx_mel_config cfg;
    float *window = nullptr, *tw_cos = nullptr, *tw_sin = nullptr, *fbank = nullptr, *addshift = nullptr, *rescale = nullptr;
    float* logmel = nullptr;
    int logmel_cap = 0;
    unsigned* bad = nullptr;
    bool whisper = false;     // Slaney filters, periodic Hann, log10 + Whisper normalisation (qwen3-asr-mlx/src/audio.rs)
    bool sensevoice = false;  // symmetric Hann, FFT-bin triangles, ln, frames from sample 0 (funasr-nano-mlx/src/audio.rs)
    int64_t max_samples = 0;  // sensevoice: max_length * sample_rate
};

extern "C" {

int omx_mel_frontend_create(omx_mel_frontend* out, const omx_mel_config* cfg) {
    OMX_REQUIRE(out && cfg, "omx_mel_frontend_create: null argument");
    OMX_REQUIRE(cfg->n_fft >= 2 && cfg->n_fft <= omx::kMaxFft && cfg->n_mels >= 1 && cfg->n_mels <= omx::kMaxMels &&
                    cfg->hop_length >= 1 && cfg->lfr_m >= 1 && cfg->lfr_n >= 1 && cfg->sample_rate > 0,
                "InvalidConfig: mel frontend n_fft=%d n_mels=%d hop=%d lfr=(%d,%d)", cfg->n_fft, cfg->n_mels, cfg->hop_length, cfg->lfr_m, cfg->lfr_n);
    omx_mel_frontend f = new omx_mel_frontend_();
    f->cfg = *cfg;
    const int n_fft = cfg->n_fft, n_mels = cfg->n_mels, n_freqs = n_fft / 2 + 1;
    std::vector<float> window(n_fft), c(n_fft), s(n_fft), fb((size_t)n_mels * n_freqs, 0.f);
    for (int i = 0; i < n_fft; ++i) {
        const float t = (float)i / (float)(n_fft - 1);                      // paraformer.rs:202-207
        window[i] = 0.54f - 0.46f * cosf(2.0f * (float)M_PI * t);
        c[i] = (float)cos(2.0 * M_PI * (double)i / (double)n_fft);
        s[i] = (float)sin(2.0 * M_PI * (double)i / (double)n_fft);
    }
    {   // create_mel_filterbank, paraformer.rs:239-275
        const float sr = (float)cfg->sample_rate, mel_min = omx::hz_to_mel(0.0f), mel_max = omx::hz_to_mel(sr / 2.0f);
        std::vector<float> pts(n_mels + 2);
        for (int i = 0; i < n_mels + 2; ++i) pts[i] = omx::mel_to_hz(mel_min + (mel_max - mel_min) * (float)i / (float)(n_mels + 1));
        for (int m = 0; m < n_mels; ++m) {
            const float fl = pts[m], fc = pts[m + 1], fr = pts[m + 2];
            for (int k = 0; k < n_freqs; ++k) {
                const float freq = (float)k * sr / (float)n_fft;
                if (freq >= fl && freq <= fc) fb[(size_t)m * n_freqs + k] = (freq - fl) / (fc - fl);
                else if (freq > fc && freq <= fr) fb[(size_t)m * n_freqs + k] = (fr - freq) / (fr - fc);
            }
        }
    }
    auto up = [&](float** dst, const std::vector<float>& src) -> int {
        OMX_HIP_CHECK(hipMalloc((void**)dst, src.size() * 4));
        OMX_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * 4, hipMemcpyHostToDevice));
        return 0;
    };
    if (up(&f->window, window) || up(&f->tw_cos, c) || up(&f->tw_sin, s) || up(&f->fbank, fb)) return 1;
    *out = f;
    return 0;
}

int omx_mel_frontend_destroy(omx_mel_frontend f) {
    if (!f) return 0;
    for (float* p : {f->window, f->tw_cos, f->tw_sin, f->fbank, f->addshift, f->rescale, f->logmel})
        if (p) (void)hipFree(p);
    if (f->bad) (void)hipFree(f->bad);
    delete f;
    return 0;
}

int omx_mel_frontend_set_cmvn(omx_mel_frontend f, const float* addshift_host, const float* rescale_host, int dim) {
    OMX_REQUIRE(f && addshift_host && rescale_host, "omx_mel_frontend_set_cmvn: null argument");
    OMX_REQUIRE(dim == f->cfg.lfr_m * f->cfg.n_mels, "CMVN must be %d-dimensional (got %d)", f->cfg.lfr_m * f->cfg.n_mels, dim);   // paraformer.rs:1550
    if (!f->addshift) OMX_HIP_CHECK(hipMalloc((void**)&f->addshift, (size_t)dim * 4));
    if (!f->rescale) OMX_HIP_CHECK(hipMalloc((void**)&f->rescale, (size_t)dim * 4));
    OMX_HIP_CHECK(hipMemcpy(f->addshift, addshift_host, (size_t)dim * 4, hipMemcpyHostToDevice));
    OMX_HIP_CHECK(hipMemcpy(f->rescale, rescale_host, (size_t)dim * 4, hipMemcpyHostToDevice));
    return 0;
}

int omx_mel_frontend_frames(omx_mel_frontend f, int64_t n_samples, int* n_frames, int* n_lfr) {
    OMX_REQUIRE(f && n_frames && n_lfr, "omx_mel_frontend_frames: null argument");
    const int nf = n_samples >= f->cfg.n_fft ? (int)((n_samples - f->cfg.n_fft) / f->cfg.hop_length + 1) : 1;
    *n_frames = nf;
    *n_lfr = (nf + (f->cfg.lfr_m - 1) / 2 + f->cfg.lfr_n - 1) / f->cfg.lfr_n;
    return 0;
}

int omx_mel_frontend_forward(omx_mel_frontend f, const float* audio, int64_t n_samples, float* feats, float* logmel_out,
                             float* power_out, omx_stream stream) {
    OMX_REQUIRE(f && audio && feats, "omx_mel_frontend_forward: null argument");
    OMX_REQUIRE(n_samples >= 1, "Audio too short for mel spectrogram");
    int nf = 0, nl = 0;
    omx_mel_frontend_frames(f, n_samples, &nf, &nl);
    hipStream_t s = (hipStream_t)stream;
    {   // paraformer.rs:284-286: reject NaN / Inf before doing any work
        if (!f->bad) OMX_HIP_CHECK(hipMalloc((void**)&f->bad, 4));
        OMX_HIP_CHECK(hipMemsetAsync(f->bad, 0, 4, s));
        int64_t blocks = (n_samples + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        omx::nonfinite_count_kernel<<<(unsigned)blocks, 256, 0, s>>>(audio, n_samples, f->bad);
        OMX_LAUNCH_CHECK();
        unsigned bad = 0;
        OMX_HIP_CHECK(hipMemcpyAsync(&bad, f->bad, 4, hipMemcpyDeviceToHost, s));
        OMX_HIP_CHECK(hipStreamSynchronize(s));
        OMX_REQUIRE(bad == 0, "Audio contains NaN or Inf values");
    }
    if (nf > f->logmel_cap) {
        if (f->logmel) OMX_HIP_CHECK(hipFree(f->logmel));
        OMX_HIP_CHECK(hipMalloc((void**)&f->logmel, (size_t)nf * f->cfg.n_mels * 4));
        f->logmel_cap = nf;
    }
    float* lm = logmel_out ? logmel_out : f->logmel;
    omx::mel_power_kernel<<<nf, 256, 0, s>>>(audio, n_samples, nf, f->cfg.n_fft, f->cfg.hop_length, f->window, f->tw_cos,
                                             f->tw_sin, f->fbank, f->cfg.n_mels, power_out, lm, 32768.0f, 0.97f, 0);
    OMX_LAUNCH_CHECK();
    const int64_t total = (int64_t)nl * f->cfg.lfr_m * f->cfg.n_mels;
    omx::lfr_cmvn_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(lm, nf, f->cfg.n_mels, f->cfg.lfr_m, f->cfg.lfr_n,
                                                                          f->addshift, f->rescale, feats, nl);
    OMX_LAUNCH_CHECK();
    return 0;
}

// ---- sibling frontend: the WhisperFeatureExtractor-compatible log-mel of
//      qwen3-asr-mlx/src/audio.rs:24-128 -- periodic Hann, 400-pt DFT power, 128 Slaney filters with Slaney
//      normalisation, log10(max(., 1e-10)), clip at (global max - 8), (x + 4) / 4; out [n_mels, n_frames] ----
int omx_whisper_mel_create(omx_mel_frontend* out, int sample_rate, int n_mels, int n_fft, int hop_length) {
    OMX_REQUIRE(out, "omx_whisper_mel_create: null argument");
    OMX_REQUIRE(n_fft >= 2 && n_fft <= omx::kMaxFft && n_mels >= 1 && n_mels <= omx::kMaxMels && hop_length >= 1 && sample_rate > 0,
                "InvalidConfig: whisper mel n_fft=%d n_mels=%d hop=%d", n_fft, n_mels, hop_length);
    omx_mel_frontend f = new omx_mel_frontend_();
    f->cfg = omx_mel_config{sample_rate, n_mels, n_fft, hop_length, 1, 1};
    f->whisper = true;
    const int n_freqs = n_fft / 2 + 1;
    std::vector<float> window(n_fft), c(n_fft), s(n_fft), fb((size_t)n_mels * n_freqs, 0.f);
    for (int i = 0; i < n_fft; ++i) {
        window[i] = 0.5f * (1.0f - cosf(2.0f * (float)M_PI * (float)i / (float)n_fft));      // audio.rs:50-54
        c[i] = (float)cos(2.0 * M_PI * (double)i / (double)n_fft);
        s[i] = (float)sin(2.0 * M_PI * (double)i / (double)n_fft);
    }
    {   // create_whisper_mel_filterbank, audio.rs:260-319
        const float fmax = (float)sample_rate / 2.0f;
        const float mel_min = omx::hz_to_slaney_mel(0.0f), mel_max = omx::hz_to_slaney_mel(fmax);
        std::vector<float> ff(n_mels + 2);
        for (int i = 0; i < n_mels + 2; ++i) ff[i] = omx::slaney_mel_to_hz(mel_min + (mel_max - mel_min) * (float)i / (float)(n_mels + 1));
        for (int m = 0; m < n_mels; ++m) {
            const float lower = ff[m], center = ff[m + 1], upper = ff[m + 2];
            const float bw = upper - lower;
            const float norm = bw > 0.f ? 2.0f / bw : 1.0f;
            for (int k = 0; k < n_freqs; ++k) {
                const float freq = (float)k * fmax / (float)(n_freqs - 1);
                float v = 0.f;
                if (freq >= lower && freq <= center && center > lower) v = (freq - lower) / (center - lower);
                else if (freq > center && freq <= upper && upper > center) v = (upper - freq) / (upper - center);
                fb[(size_t)m * n_freqs + k] = v * norm;
            }
        }
    }
    auto up = [&](float** dst, const std::vector<float>& src) -> int {
        OMX_HIP_CHECK(hipMalloc((void**)dst, src.size() * 4));
        OMX_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * 4, hipMemcpyHostToDevice));
        return 0;
    };
    if (up(&f->window, window) || up(&f->tw_cos, c) || up(&f->tw_sin, s) || up(&f->fbank, fb)) return 1;
    *out = f;
    return 0;
}

int omx_whisper_mel_frames(omx_mel_frontend f, int64_t n_samples, int* n_frames) {
    OMX_REQUIRE(f && n_frames && f->whisper, "omx_whisper_mel_frames: not a whisper frontend");
    OMX_REQUIRE(n_samples > 0, "Audio samples are empty");                                                    // audio.rs:71-75
    OMX_REQUIRE(n_samples >= f->cfg.n_fft, "Audio too short: %lld ms, need at least %lld ms",                 // audio.rs:81-86
                (long long)(n_samples * 1000 / f->cfg.sample_rate), (long long)((int64_t)f->cfg.n_fft * 1000 / f->cfg.sample_rate));
    *n_frames = 1 + (int)((n_samples - f->cfg.n_fft) / f->cfg.hop_length);
    return 0;
}

int omx_whisper_mel_forward(omx_mel_frontend f, const float* audio, int64_t n_samples, float* out, omx_stream stream) {
    OMX_REQUIRE(f && audio && out && f->whisper, "omx_whisper_mel_forward: null argument or not a whisper frontend");
    int nf = 0;
    if (omx_whisper_mel_frames(f, n_samples, &nf)) return 1;
    hipStream_t s = (hipStream_t)stream;
    if (nf > f->logmel_cap) {
        if (f->logmel) OMX_HIP_CHECK(hipFree(f->logmel));
        OMX_HIP_CHECK(hipMalloc((void**)&f->logmel, (size_t)nf * f->cfg.n_mels * 4));
        f->logmel_cap = nf;
    }
    if (!f->bad) OMX_HIP_CHECK(hipMalloc((void**)&f->bad, 4));
    OMX_HIP_CHECK(hipMemsetAsync(f->bad, 0, 4, s));
    omx::mel_power_kernel<<<nf, 256, 0, s>>>(audio, n_samples, nf, f->cfg.n_fft, f->cfg.hop_length, f->window, f->tw_cos,
                                             f->tw_sin, f->fbank, f->cfg.n_mels, nullptr, f->logmel, 1.0f, 0.0f, 1);
    OMX_LAUNCH_CHECK();
    const int64_t total = (int64_t)nf * f->cfg.n_mels;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 1024);
    omx::max_kernel<<<blocks, 256, 0, s>>>(f->logmel, total, f->bad);
    omx::whisper_norm_kernel<<<blocks, 256, 0, s>>>(f->logmel, nf, f->cfg.n_mels, f->bad, out);
    OMX_LAUNCH_CHECK();
    return 0;
}

// ---- sibling frontend: the Fun-ASR-Nano / SenseVoice log-mel of funasr-nano-mlx/src/audio.rs:44-157
//      (`MelFrontend::{new, compute_mel_spectrogram}`, defaults 16 kHz / 80 mels / n_fft 400 / hop 160 / 30 s): symmetric Hann
//      (i / (n_fft - 1)), frames start at sample frame * hop with zero padding past the end, n_frames = max(len / hop, 1) after
//      truncation to max_length, 400-pt DFT power, triangles between floor((n_fft + 1) hz / sr) FFT bins (:287-339),
//      ln(max(., 1e-10)); out [n_mels, n_frames].  `apply_lfr` (:345-412) follows as its own entry point. ----
int omx_sensevoice_mel_create(omx_mel_frontend* out, int sample_rate, int n_mels, int n_fft, int hop_length, float max_length_s) {
    OMX_REQUIRE(out, "omx_sensevoice_mel_create: null argument");
    OMX_REQUIRE(n_fft >= 2 && n_fft <= omx::kMaxFft && n_mels >= 1 && n_mels <= omx::kMaxMels && hop_length >= 1 && sample_rate > 0 &&
                    max_length_s > 0.f,
                "InvalidConfig: sensevoice mel n_fft=%d n_mels=%d hop=%d max_length=%g", n_fft, n_mels, hop_length, (double)max_length_s);
    omx_mel_frontend f = new omx_mel_frontend_();
    f->cfg = omx_mel_config{sample_rate, n_mels, n_fft, hop_length, 1, 1};
    f->sensevoice = true;
    f->max_samples = (int64_t)(max_length_s * (float)sample_rate);                                     // audio.rs:112
    const int n_freqs = n_fft / 2 + 1;
    std::vector<float> window(n_fft), c(n_fft), s(n_fft), fb((size_t)n_mels * n_freqs, 0.f);
    for (int i = 0; i < n_fft; ++i) {
        window[i] = 0.5f * (1.0f - cosf(2.0f * (float)M_PI * (float)i / (float)(n_fft - 1)));          // audio.rs:68-72
        c[i] = (float)cos(2.0 * M_PI * (double)i / (double)n_fft);
        s[i] = (float)sin(2.0 * M_PI * (double)i / (double)n_fft);
    }
    {   // create_mel_filterbank, audio.rs:287-339
        const float sr = (float)sample_rate, mel_low = omx::hz_to_mel(0.0f), mel_high = omx::hz_to_mel(sr / 2.0f);
        std::vector<int64_t> bin(n_mels + 2);
        for (int i = 0; i < n_mels + 2; ++i) {
            const float hz = omx::mel_to_hz(mel_low + (mel_high - mel_low) * (float)i / (float)(n_mels + 1));
            bin[i] = (int64_t)floorf((float)(n_fft + 1) * hz / sr);
        }
        for (int m = 0; m < n_mels; ++m) {
            const int64_t left = bin[m], center = bin[m + 1], right = bin[m + 2];
            for (int64_t k = left; k < center; ++k)
                if (k < n_freqs && center > left) fb[(size_t)m * n_freqs + k] = (float)(k - left) / (float)(center - left);
            for (int64_t k = center; k < right; ++k)
                if (k < n_freqs && right > center) fb[(size_t)m * n_freqs + k] = (float)(right - k) / (float)(right - center);
        }
    }
    auto up = [&](float** dst, const std::vector<float>& src) -> int {
        OMX_HIP_CHECK(hipMalloc((void**)dst, src.size() * 4));
        OMX_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * 4, hipMemcpyHostToDevice));
        return 0;
    };
    if (up(&f->window, window) || up(&f->tw_cos, c) || up(&f->tw_sin, s) || up(&f->fbank, fb)) return 1;
    *out = f;
    return 0;
}

int omx_sensevoice_mel_frames(omx_mel_frontend f, int64_t n_samples, int* n_frames) {
    OMX_REQUIRE(f && n_frames && f->sensevoice, "omx_sensevoice_mel_frames: not a sensevoice frontend");
    OMX_REQUIRE(n_samples > 0, "Cannot compute mel spectrogram: audio samples are empty");                     // audio.rs:95-99
    OMX_REQUIRE(n_samples >= f->cfg.hop_length, "Audio too short: %lld ms, need at least %lld ms",              // audio.rs:105-110
                (long long)(n_samples * 1000 / f->cfg.sample_rate), (long long)((int64_t)f->cfg.hop_length * 1000 / f->cfg.sample_rate));
    const int64_t n = std::min<int64_t>(n_samples, f->max_samples);
    *n_frames = (int)std::max<int64_t>(n / f->cfg.hop_length, 1);                                               // audio.rs:121
    return 0;
}

int omx_sensevoice_mel_forward(omx_mel_frontend f, const float* audio, int64_t n_samples, float* out, omx_stream stream) {
    OMX_REQUIRE(f && audio && out && f->sensevoice, "omx_sensevoice_mel_forward: null argument or not a sensevoice frontend");
    int nf = 0;
    if (omx_sensevoice_mel_frames(f, n_samples, &nf)) return 1;
    const int64_t n = std::min<int64_t>(n_samples, f->max_samples);
    omx::mel_power_kernel<<<nf, 256, 0, (hipStream_t)stream>>>(audio, n, nf, f->cfg.n_fft, f->cfg.hop_length, f->window, f->tw_cos, f->tw_sin,
                                                               f->fbank, f->cfg.n_mels, nullptr, out, 1.0f, 0.0f, 0, /*pad_partial=*/1,
                                                               /*mel_major=*/1);
    OMX_LAUNCH_CHECK();
    return 0;
}

/* apply_lfr (funasr-nano-mlx/src/audio.rs:345-412): mel [n_mels, n_frames] -> [ceil(n_frames / lfr_n), lfr_m * n_mels]; output frame t
 * stacks frames t * lfr_n + (j - lfr_m / 2), j = 0 .. lfr_m - 1, clamped to [0, n_frames - 1]. */
int omx_apply_lfr(float* out, const float* mel, int n_mels, int n_frames, int lfr_m, int lfr_n, omx_stream stream) {
    OMX_REQUIRE(out && mel && n_mels >= 1 && n_frames >= 1 && lfr_m >= 1 && lfr_n >= 1, "omx_apply_lfr: bad arguments");
    const int t_out = (n_frames + lfr_n - 1) / lfr_n;
    const int64_t total = (int64_t)t_out * lfr_m * n_mels;
    omx::lfr_center_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(mel, n_frames, n_mels, lfr_m, lfr_n, out, t_out);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
