// Windowed-sinc resampler on the GPU -- `audio::resample` (mlx-rs-core/src/audio.rs:178-277), the one audio function besides WAV IO
// that every ASR crate of the reference calls (funasr-mlx/src/audio.rs:6).  The reference runs rubato 0.14's SincFixedIn on one CPU
// thread: sinc_len 256, oversampling 256, cubic interpolation between the 4 nearest phases, squared Blackman-Harris window, cutoff
// 0.95 (x ratio when downsampling), chunks of min(4096, len), a zero-padded remainder of which only ceil(remaining * ratio) outputs
// are kept, one all-zero flush chunk, truncation to round(len * ratio).  oracle/ref_resample.py restates all of it.
//
// Split of work: WHICH read positions are visited is a sequential float64 recurrence (idx += 1 / ratio, re-based every chunk) whose
// floor / fraction decide sample and phase -- computed on the host exactly as rubato does (one add per output, ~0.5 M for 30 s), so the
// positions are bit-identical to the oracle's.  The arithmetic -- 4 x 256 multiply-adds per output against a 256 KiB phase table that
// stays in L2 -- runs on the device: one wave per output, each lane 4 consecutive taps of each of the 4 points (16-byte loads of the
// table rows; the samples of the 4 points overlap and come from 2 consecutive windows), DPP wave sums, lane 0 evaluates the cubic.
#include <math.h>

#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace omx {
namespace {

constexpr int kSincLen = 256, kPhases = 256, kMaxChunk = 4096;

struct ResamplePoint { int64_t g0; int sub0; float frac; };   // stream index / phase of the first of the 4 points; cubic fraction

// rubato make_sincs + windows.rs, evaluated in f32 like the reference's SincFixedIn::<f32>
std::vector<float> make_sinc_table(float cutoff) {
    const int tot = kSincLen * kPhases;
    std::vector<float> y(tot), table((size_t)tot);
    const float n = (float)tot, pi2 = (float)(2.0 * M_PI), pi4 = (float)(4.0 * M_PI), pi6 = (float)(6.0 * M_PI), pi = (float)M_PI;
    float sum = 0.f;
    for (int x = 0; x < tot; ++x) {
        const float xf = (float)x;
        float w = 0.35875f - 0.48829f * cosf(pi2 * xf / n) + 0.14128f * cosf(pi4 * xf / n) - 0.01168f * cosf(pi6 * xf / n);
        w = w * w;
        const float arg = (xf - (float)(tot / 2)) * cutoff / (float)kPhases;
        const float s = arg == 0.f ? 1.f : sinf(arg * pi) / (arg * pi);
        y[x] = w * s;
        sum += y[x];
    }
    sum /= (float)kPhases;
    for (int p = 0; p < kSincLen; ++p)
        for (int k = 0; k < kPhases; ++k) table[(size_t)(kPhases - k - 1) * kSincLen + p] = y[kPhases * p + k] / sum;
    return table;
}

struct TableCache {
    std::mutex mu;
    std::map<uint32_t, float*> dev;   // keyed by the cutoff's bit pattern
} g_tables;

int sinc_table_device(float cutoff, const float** out) {
    std::lock_guard<std::mutex> lock(g_tables.mu);
    uint32_t key;
    memcpy(&key, &cutoff, 4);
    auto it = g_tables.dev.find(key);
    if (it == g_tables.dev.end()) {
        const std::vector<float> t = make_sinc_table(cutoff);
        float* d = nullptr;
        OMX_HIP_CHECK(hipMalloc((void**)&d, t.size() * 4));
        OMX_HIP_CHECK(hipMemcpy(d, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        it = g_tables.dev.emplace(key, d).first;
    }
    *out = it->second;
    return 0;
}

// the read positions the reference's driver makes rubato visit (oracle/ref_resample.py `plan`)
std::vector<ResamplePoint> resample_plan(int64_t n_in, double ratio) {
    const int64_t chunk = std::min<int64_t>(kMaxChunk, n_in);
    const double t_ratio = 1.0 / ratio;
    const double end_idx = (double)(chunk - (kSincLen + 1) - (int64_t)ceil(t_ratio));
    const int64_t n_full = n_in / chunk, remaining = n_in % chunk;
    std::vector<int64_t> keep((size_t)n_full, -1);                       // -1: every output of the chunk
    if (remaining) keep.push_back((int64_t)ceil((double)remaining * ratio));   // audio.rs:250-252
    keep.push_back(-1);                                                  // the flush chunk, :258-267
    const int64_t expected = (int64_t)llround((double)n_in * ratio);     // :270
    std::vector<ResamplePoint> pts;
    pts.reserve((size_t)expected + 16);
    double idx = -(double)(kSincLen / 2);
    for (size_t c = 0; c < keep.size(); ++c) {
        int64_t n = 0;
        while (idx < end_idx) {
            idx += t_ratio;
            if (keep[c] < 0 || n < keep[c]) {
                const double fl = floor(idx);
                int64_t index = (int64_t)fl;
                int sub = (int)floor((idx - fl) * (double)kPhases) - 1;  // get_nearest_times_4: the first point is one phase earlier
                if (sub < 0) { sub += kPhases; index -= 1; }
                const double v = idx * (double)kPhases;
                pts.push_back({(int64_t)c * chunk + index, sub, (float)(v - floor(v))});
            }
            ++n;
        }
        idx -= (double)chunk;
    }
    if ((int64_t)pts.size() > expected) pts.resize((size_t)expected);    // :271-273
    return pts;
}

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int64_t n_in, const float* __restrict__ table,
                                                       const ResamplePoint* __restrict__ pts, int64_t n_out, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= n_out) return;
    const ResamplePoint p = pts[o];
    // the 4 points are consecutive phases: point k = (sample g0 + (sub0 + k) / 256, phase (sub0 + k) % 256); their sample windows start
    // at g0 or g0 + 1.  Each lane holds taps 4 lane .. 4 lane + 4 of the window at g0 (5 samples cover both starts).
    float s[5];
    const int64_t base = p.g0 + 4 * lane;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int64_t g = base + i;
        s[i] = (g >= 0 && g < n_in) ? x[g] : 0.f;                        // the stream: samples, zeros before and after
    }
    float y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sub = p.sub0 + k, shift = sub >> 8, phase = sub & (kPhases - 1);
        const f32x4 h = *reinterpret_cast<const f32x4*>(table + (size_t)phase * kSincLen + 4 * lane);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = fmaf(shift ? s[i + 1] : s[i], h[i], acc);
        y[k] = wave_sum(acc);
    }
    if (lane == 0) {   // rubato interp_cubic: y[1] at 0, y[2] at 1
        const float a0 = y[1];
        const float a1 = -(1.0f / 3.0f) * y[0] - 0.5f * y[1] + y[2] - (1.0f / 6.0f) * y[3];
        const float a2 = 0.5f * (y[0] + y[2]) - y[1];
        const float a3 = 0.5f * (y[1] - y[2]) + (1.0f / 6.0f) * (y[3] - y[0]);
        const float f = p.frac, f2 = f * f;
        out[o] = a0 + a1 * f + a2 * f2 + a3 * f2 * f;
    }
}

}  // namespace
}  // namespace omx

extern "C" {

int64_t omx_resample_len(int64_t n_in, uint32_t src_rate, uint32_t dst_rate) {
    if (n_in <= 0 || src_rate == 0 || dst_rate == 0) return 0;
    if (src_rate == dst_rate) return n_in;
    return (int64_t)llround((double)n_in * ((double)dst_rate / (double)src_rate));
}

int omx_resample_sinc(const float* in, int64_t n_in, uint32_t src_rate, uint32_t dst_rate, float* out, int64_t out_cap, int64_t* n_out,
                      omx_stream stream) {
    OMX_REQUIRE(n_out && src_rate > 0 && dst_rate > 0 && n_in >= 0, "omx_resample_sinc: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (n_in == 0) { *n_out = 0; return 0; }                             // audio.rs:179-181
    OMX_REQUIRE(in && out, "omx_resample_sinc: null buffer");
    if (src_rate == dst_rate) {
        OMX_REQUIRE(out_cap >= n_in, "omx_resample_sinc: output holds %lld samples, need %lld", (long long)out_cap, (long long)n_in);
        OMX_HIP_CHECK(hipMemcpyAsync(out, in, (size_t)n_in * 4, hipMemcpyDeviceToDevice, s));
        *n_out = n_in;
        return 0;
    }
    const double ratio = (double)dst_rate / (double)src_rate;
    // rubato rejects ratios outside what its buffers were sized for only when CHANGING the ratio; construction needs ratio > 0
    const float cutoff = ratio >= 1.0 ? 0.95f : 0.95f * (float)ratio;
    const float* table = nullptr;
    if (omx::sinc_table_device(cutoff, &table)) return 1;
    const std::vector<omx::ResamplePoint> pts = omx::resample_plan(n_in, ratio);
    const int64_t n = (int64_t)pts.size();
    OMX_REQUIRE(out_cap >= n, "omx_resample_sinc: output holds %lld samples, need %lld", (long long)out_cap, (long long)n);
    *n_out = n;
    if (n == 0) return 0;
    omx::ResamplePoint* d_pts = nullptr;
    OMX_HIP_CHECK(hipMalloc((void**)&d_pts, (size_t)n * sizeof(omx::ResamplePoint)));
    hipError_t e = hipMemcpyAsync(d_pts, pts.data(), (size_t)n * sizeof(omx::ResamplePoint), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        omx::resample_kernel<<<(unsigned)((n + 3) / 4), 256, 0, s>>>(in, n_in, table, d_pts, n, out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);                    // the plan buffers are released below
    (void)hipFree(d_pts);
    OMX_REQUIRE(e == hipSuccess, "omx_resample_sinc: %s", hipGetErrorString(e));
    return 0;
}

}  // extern "C"
