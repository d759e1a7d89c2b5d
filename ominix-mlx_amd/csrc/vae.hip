// FLUX VAE decoder (latents -> image) on gfx950: SURVEY.md 8f rank 3, the last step of generate_klein.rs.
//   reference: flux-klein-mlx/src/autoencoder.rs -- Decoder::forward :375-412, ResnetBlock::forward :139-157,
//   AttnBlock::forward :195-232, config :22-81; GroupNorm (pytorch compatible, 32 groups, eps 1e-5)
//   mlx-rs/src/nn/normalization.rs:363-392; weights by the names weights.rs:164-217 produces (NHWC convolutions,
//   MLX Conv2d weight [out, kH, kW, in]).
// Mapping: activations NHWC bf16 (fp32 accumulate).  A 3x3 / pad-1 convolution is the bf16 MFMA GEMM
//   out[H*W, Cout] = im2col(x)[H*W, 9*Cin] . W[Cout, 9*Cin]^T  -- MLX's weight layout IS that B matrix --
//   with bias and the ResNet shortcut fused into the GEMM epilogue; nearest-neighbour 2x upsampling is folded into
//   the im2col gather of the convolution that follows it (the upsampled tensor never exists).  GroupNorm + SiLU is
//   one HBM-bound pass after a deterministic two-level statistics reduction.  The single-head mid-block attention
//   (head dim = 512 channels) runs as two GEMMs around a row softmax.
#include <math.h>

#include <map>
#include <string>
#include <vector>

#include "gemm.hpp"

namespace omx {
namespace {

constexpr int kGnChunks = 512;  // pixel chunks (= blocks) of the statistics pass

// col[(y*W + x), (kh*3 + kw)*C + c] = src[sy, sx, c] or 0; (sy, sx) = (y + kh - 1, x + kw - 1) on the OUTPUT grid,
// read from the half-resolution source when `up` (Upsample(2, nearest) folded in)
__global__ __launch_bounds__(576) void im2col3x3_kernel(bf16_t* __restrict__ col, const bf16_t* __restrict__ src, int H, int W, int C,
                                                        int up) {
    // One output pixel = one row of 9 * C/8 16-byte vectors.  A block walks pixels blockIdx.x, + gridDim.x, ... and a thread
    // keeps its (tap, channel vector) for the whole walk: the only division left is pixel -> (y, x), 32-bit, once per pixel.
    // (Flat 64-bit index arithmetic -- five divisions per vector -- made this pass instruction-bound at 1.5 TB/s.)
    // The block is 576 threads = 4, 2 or 1 whole pixel rows at 128, 256 or 512 channels (other widths: a thread loops over the row).
    const int c8 = C / 8, nv = 9 * c8;
    const int Ws = up ? W / 2 : W;
    const unsigned npix = (unsigned)H * (unsigned)W;
    const int ppi = (int)blockDim.x >= nv ? (int)blockDim.x / nv : 1;      // pixels per block iteration
    const int pl = (int)threadIdx.x / nv;                                  // which of them this thread serves
    if (pl >= ppi && (int)blockDim.x >= nv) return;
    for (int t = (int)threadIdx.x - pl * nv; t < nv; t += blockDim.x) {
        const int tap = t / c8, cv = t - tap * c8;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        for (unsigned pix = blockIdx.x * ppi + pl; pix < npix; pix += gridDim.x * ppi) {
            const int y = (int)(pix / (unsigned)W), x = (int)(pix - (unsigned)y * (unsigned)W);
            const int sy = y + dy, sx = x + dx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
                const int py = up ? sy >> 1 : sy, px = up ? sx >> 1 : sx;
                v = *reinterpret_cast<const u32x4*>(src + ((int64_t)py * Ws + px) * C + cv * 8);
            }
            *reinterpret_cast<u32x4*>(col + ((int64_t)pix * nv + t) * 8) = v;
        }
    }
}

// dst [(H+2), (W+2), C]: src (or its 2x nearest upsampling when `up`) inside a one-pixel zero border -- the A operand of the
// implicit 3x3 convolution (gemm.hpp: launch_conv3x3_implicit); 1x (4x when upsampling) the activation instead of im2col's 9x
__global__ __launch_bounds__(256) void pad_nhwc_kernel(bf16_t* __restrict__ dst, const bf16_t* __restrict__ src, int H, int W, int C, int up) {
    const int c8 = C / 8, Wp = W + 2, Ws = up ? W / 2 : W;
    const unsigned total = (unsigned)(H + 2) * (unsigned)Wp * (unsigned)c8;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned pp = i / (unsigned)c8;
        const int cv = (int)(i - pp * (unsigned)c8);
        const int yp = (int)(pp / (unsigned)Wp), xp = (int)(pp - (unsigned)yp * (unsigned)Wp);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (yp >= 1 && yp <= H && xp >= 1 && xp <= W) {
            const int sy = up ? (yp - 1) >> 1 : yp - 1, sx = up ? (xp - 1) >> 1 : xp - 1;
            v = *reinterpret_cast<const u32x4*>(src + ((int64_t)sy * Ws + sx) * C + cv * 8);
        }
        *reinterpret_cast<u32x4*>(dst + (int64_t)i * 8) = v;
    }
}

// statistics: block j sums its pixel chunk for ALL groups with full-row coalesced reads -- thread t owns the 8-channel
// vector (t mod C/8) of every (256 / (C/8))-th pixel -- then folds the per-thread sums group by group in a fixed
// order (no atomics: the result is reproducible)
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const bf16_t* __restrict__ x, int64_t HW, int C, int G,
                                                              double* __restrict__ partials) {
    __shared__ float sm_s[256][8], sm_q[256][8];
    const int j = blockIdx.x, c8 = C / 8, cg = C / G;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    // threads beyond a multiple of c8 idle, so that a thread's vector index never changes
    const int lanes = (256 / c8) * c8;
    // block j takes every kGnChunks-th slab of `lanes` vectors (4 KiB at 128 channels): at any moment the blocks of the grid
    // read one contiguous stretch of the tensor -- with a contiguous range per block the 512 streams sit 1/512 of the tensor
    // apart (poor DRAM-page / translation locality: the 1024^2 x 128 pass took 640 us that way, 97 us interleaved)
    if ((int)threadIdx.x < lanes || c8 > 256) {
        const int64_t slab = c8 > 256 ? 256 : lanes, total = HW * c8;
#pragma unroll 4   // four loads in flight per thread; the accumulation order stays the sequential one
        for (int64_t i = (int64_t)j * slab + threadIdx.x; i < total; i += slab * kGnChunks) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float lo = bf16lo(v[w]), hi = bf16hi(v[w]);
                s[2 * w] += lo; q[2 * w] = fmaf(lo, lo, q[2 * w]);
                s[2 * w + 1] += hi; q[2 * w + 1] = fmaf(hi, hi, q[2 * w + 1]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm_s[threadIdx.x][e] = s[e]; sm_q[threadIdx.x][e] = q[e]; }
    __syncthreads();
    // group g = channels [g*cg, (g+1)*cg): owned by the threads whose vector covers them
    for (int g = threadIdx.x; g < G; g += 256) {
        double ts = 0.0, tq = 0.0;
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            const int vec = c / 8, e = c % 8;
            if (c8 > 256) continue;
            for (int t = vec; t < lanes; t += c8) { ts += sm_s[t][e]; tq += sm_q[t][e]; }
        }
        partials[((size_t)g * kGnChunks + j) * 2] = ts;
        partials[((size_t)g * kGnChunks + j) * 2 + 1] = tq;
    }
}

// fold the chunk partials of every group (fixed order, double) into mean and 1/sqrt(var + eps)
__global__ void groupnorm_finalize_kernel(const double* __restrict__ partials, float* __restrict__ mean_rstd, int64_t HW, int C, int G,
                                          float eps) {
    // one wave per group: lane l sums chunks l, l + 64, ... in order, then a fixed xor-tree over the lanes (deterministic)
    const int g = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, ss = 0.0;
    for (int j = lane; j < kGnChunks; j += 64) {
        s += partials[((size_t)g * kGnChunks + j) * 2];
        ss += partials[((size_t)g * kGnChunks + j) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        ss += __shfl_xor(ss, o, 64);
    }
    if (lane != 0) return;
    const double n = (double)HW * (C / G), mean = s / n;
    const double var = fmax(ss / n - mean * mean, 0.0);
    mean_rstd[g] = (float)mean;
    mean_rstd[G + g] = (float)(1.0 / sqrt(var + (double)eps));
}

// y = (x - mean_g) * rstd_g * weight[c] + bias[c], optionally SiLU, one rounding to bf16
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ x, int64_t HW, int C,
                                                              int G, const float* __restrict__ mean_rstd,
                                                              const bf16_t* __restrict__ weight, const bf16_t* __restrict__ bias,
                                                              int silu) {
    extern __shared__ float gn_sm[];   // [G] mean, [G] rstd
    const int cg = C / G;
    for (int g = threadIdx.x; g < 2 * G; g += blockDim.x) gn_sm[g] = mean_rstd[g];
    __syncthreads();
    const int c8 = C / 8;
#pragma unroll 2
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < HW * c8; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % c8) * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
        const u32x4 wv = *reinterpret_cast<const u32x4*>(weight + c0);
        const u32x4 bv = *reinterpret_cast<const u32x4*>(bias + c0);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float r[2];
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int c = c0 + 2 * q + hlf, g = c / cg;
                const float xv = hlf ? bf16hi(v[q]) : bf16lo(v[q]);
                const float w = hlf ? bf16hi(wv[q]) : bf16lo(wv[q]);
                const float b = hlf ? bf16hi(bv[q]) : bf16lo(bv[q]);
                float y = (xv - gn_sm[g]) * gn_sm[G + g] * w + b;
                if (silu) y = y / (1.0f + expf(-y));
                r[hlf] = y;
            }
            o[q] = pack_bf16(r[0], r[1]);
        }
        *reinterpret_cast<u32x4*>(out + i * 8) = o;
    }
}

__global__ __launch_bounds__(256) void scale_shift_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ x, int64_t n, float scale,
                                                          float shift) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f32_to_bf16(round_bf16(bf16_to_f32(x[i]) / scale) + shift);   // divide, then add: two primitives
}

// [T, C] -> [C, T] through an LDS tile
__global__ __launch_bounds__(256) void transpose_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ in, int T, int C) {
    __shared__ bf16_t tile[32][33];
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int r = threadIdx.x / 32; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + (threadIdx.x & 31);
        if (t < T && c < C) tile[r][threadIdx.x & 31] = in[(size_t)t * C + c];
    }
    __syncthreads();
    for (int r = threadIdx.x / 32; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + (threadIdx.x & 31);
        if (t < T && c < C) out[(size_t)c * T + t] = tile[threadIdx.x & 31][r];
    }
}

// in place: row <- softmax(row / scale) in fp32, one rounding (autoencoder.rs:215-219)
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* __restrict__ s, int T, float scale) {
    __shared__ float red[4];
    bf16_t* row = s + (size_t)blockIdx.x * T;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < T; i += 256) mx = fmaxf(mx, round_bf16(bf16_to_f32(row[i]) / scale));
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < T; i += 256) sum += expf(round_bf16(bf16_to_f32(row[i]) / scale) - mx);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    for (int i = threadIdx.x; i < T; i += 256) row[i] = f32_to_bf16(expf(round_bf16(bf16_to_f32(row[i]) / scale) - mx) / sum);
}

}  // namespace
}  // namespace omx

using namespace omx;

struct omx_vae_decoder_ {
    omx_vae_config cfg;
    std::map<std::string, const bf16_t*> w;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    bf16_t* buf[4] = {nullptr, nullptr, nullptr, nullptr};   // activations, each max(H*W*C) elements
    size_t buf_cap = 0;
    bf16_t* col = nullptr;          // im2col scratch
    size_t col_cap = 0;
    bf16_t* attn = nullptr;         // q, k, v, vT, o [T, C] x 5 + scores [T, T]
    size_t attn_cap = 0;
    double* stats = nullptr;
};

namespace {

int vget(omx_vae_decoder m, const std::string& name, const bf16_t** out) {
    auto it = m->w.find(name);
    if (it == m->w.end()) return set_error("WeightNotFound: %s", name.c_str());
    *out = it->second;
    return 0;
}

int grow(bf16_t** p, size_t* cap, size_t need) {
    if (need <= *cap) return 0;
    if (*p) OMX_HIP_CHECK(hipFree(*p));
    OMX_HIP_CHECK(hipMalloc((void**)p, need * 2));
    *cap = need;
    return 0;
}

// out [H*W, Cout] = conv3x3(src [Hs, Ws, Cin], pad 1) + bias (+ resid); up = the source is half resolution
int conv3x3(omx_vae_decoder m, bf16_t* out, const bf16_t* src, const std::string& name, int H, int W, int Cin, int Cout, int up,
            const bf16_t* resid) {
    const bf16_t *wt = nullptr, *bs = nullptr;
    if (vget(m, name + ".weight", &wt) || vget(m, name + ".bias", &bs)) return 1;
    OMX_REQUIRE(Cin % 8 == 0, "vae: %s input channels %d must be a multiple of 8", name.c_str(), Cin);
    const int64_t M = (int64_t)H * W;
    const char* im_env = getenv("OMX_VAE_IMPLICIT");   // 0: im2col + GEMM everywhere (A/B, tests)
    const bool implicit_on = !(im_env && im_env[0] == '0');
    if (implicit_on && conv3x3_implicit_supported(H, W, Cin, Cout)) {
        // zero-bordered copy of the input, then the convolution as one GEMM whose A tiles are read straight from it
        const size_t padded = (size_t)(H + 2) * (W + 2) * Cin;
        if (grow(&m->col, &m->col_cap, padded)) return 1;
        pad_nhwc_kernel<<<4096, 256, 0, m->stream>>>(m->col, src, H, W, Cin, up);
        OMX_LAUNCH_CHECK();
        return launch_conv3x3_implicit(out, m->col, wt, bs, resid, H, W, Cin, Cout, m->stream);
    }
    if (grow(&m->col, &m->col_cap, (size_t)M * 9 * Cin)) return 1;
    im2col3x3_kernel<<<(unsigned)std::min<int64_t>(M, 4096), 576, 0, m->stream>>>(m->col, src, H, W, Cin, up);
    OMX_LAUNCH_CHECK();
    return launch_gemm_bf16_ex(out, m->col, wt, bs, resid, (int)M, Cout, 9 * Cin, m->stream);
}

int conv1x1(omx_vae_decoder m, bf16_t* out, const bf16_t* src, const std::string& name, int64_t M, int Cin, int Cout, const bf16_t* resid) {
    const bf16_t *wt = nullptr, *bs = nullptr;
    if (vget(m, name + ".weight", &wt) || vget(m, name + ".bias", &bs)) return 1;
    return launch_gemm_bf16_ex(out, src, wt, bs, resid, (int)M, Cout, Cin, m->stream);
}

int group_norm(omx_vae_decoder m, bf16_t* out, const bf16_t* x, const std::string& name, int64_t HW, int C, int silu) {
    const bf16_t *wt = nullptr, *bs = nullptr;
    if (vget(m, name + ".weight", &wt) || vget(m, name + ".bias", &bs)) return 1;
    constexpr int G = 32;
    OMX_REQUIRE(C % G == 0 && C % 8 == 0, "vae: GroupNorm over %d channels needs a multiple of 32", C);
    OMX_REQUIRE(C / 8 <= 256, "vae: GroupNorm over %d channels (max 2048)", C);
    groupnorm_stats_kernel<<<kGnChunks, 256, 0, m->stream>>>(x, HW, C, G, m->stats);
    float* mr = reinterpret_cast<float*>(m->stats + (size_t)32 * kGnChunks * 2);
    groupnorm_finalize_kernel<<<G, 64, 0, m->stream>>>(m->stats, mr, HW, C, G, 1e-5f);
    groupnorm_apply_kernel<<<2048, 256, 2 * G * sizeof(float), m->stream>>>(out, x, HW, C, G, mr, wt, bs, silu);
    OMX_LAUNCH_CHECK();
    return 0;
}

// ResnetBlock::forward (:139-157): x in buf[xi] -> result in buf[(xi + 1) & 3]; uses the two other buffers
int resnet(omx_vae_decoder m, int* xi, const std::string& p, int H, int W, int Cin, int Cout) {
    bf16_t *x = m->buf[*xi], *y = m->buf[(*xi + 1) & 3], *t = m->buf[(*xi + 2) & 3], *sc = m->buf[(*xi + 3) & 3];
    const int64_t HW = (int64_t)H * W;
    if (group_norm(m, t, x, p + "norm1", HW, Cin, 1)) return 1;
    if (conv3x3(m, y, t, p + "conv1", H, W, Cin, Cout, 0, nullptr)) return 1;
    if (group_norm(m, t, y, p + "norm2", HW, Cout, 1)) return 1;
    const bf16_t* shortcut = x;
    if (Cin != Cout) {
        if (conv1x1(m, sc, x, p + "conv_shortcut", HW, Cin, Cout, nullptr)) return 1;
        shortcut = sc;
    }
    if (conv3x3(m, y, t, p + "conv2", H, W, Cout, Cout, 0, shortcut)) return 1;
    *xi = (*xi + 1) & 3;
    return 0;
}

// AttnBlock::forward (:195-232): one head over T = H*W tokens of C channels
int attn_block(omx_vae_decoder m, int* xi, const std::string& p, int H, int W, int C) {
    bf16_t *x = m->buf[*xi], *y = m->buf[(*xi + 1) & 3], *t = m->buf[(*xi + 2) & 3];
    const int64_t T = (int64_t)H * W;
    if (grow(&m->attn, &m->attn_cap, (size_t)T * C * 5 + (size_t)T * T)) return 1;
    bf16_t *q = m->attn, *k = q + T * C, *v = k + T * C, *vT = v + T * C, *o = vT + T * C, *S = o + T * C;
    if (group_norm(m, t, x, p + "group_norm", T, C, 0)) return 1;
    if (conv1x1(m, q, t, p + "to_q", T, C, C, nullptr) || conv1x1(m, k, t, p + "to_k", T, C, C, nullptr) ||
        conv1x1(m, v, t, p + "to_v", T, C, C, nullptr))
        return 1;
    if (launch_gemm_bf16(S, q, k, nullptr, (int)T, (int)T, C, m->stream)) return 1;          // q k^T
    softmax_rows_kernel<<<(unsigned)T, 256, 0, m->stream>>>(S, (int)T, sqrtf((float)C));
    transpose_kernel<<<dim3((unsigned)((T + 31) / 32), (unsigned)((C + 31) / 32)), 256, 0, m->stream>>>(vT, v, (int)T, C);
    OMX_LAUNCH_CHECK();
    if (launch_gemm_bf16(o, S, vT, nullptr, (int)T, C, (int)T, m->stream)) return 1;          // P v
    if (conv1x1(m, y, o, p + "to_out", T, C, C, x)) return 1;                                  // + x
    *xi = (*xi + 1) & 3;
    return 0;
}

}  // namespace

extern "C" {

int omx_vae_decoder_create(omx_vae_decoder* out, const omx_vae_config* cfg) {
    OMX_REQUIRE(out && cfg, "omx_vae_decoder_create: null argument");
    OMX_REQUIRE(cfg->n_mult >= 1 && cfg->n_mult <= 8 && cfg->ch % 32 == 0 && cfg->z_channels % 8 == 0 && cfg->num_res_blocks >= 0 &&
                    cfg->out_ch >= 1 && cfg->scale_factor != 0.f,
                "InvalidConfig: vae ch=%d n_mult=%d z=%d", cfg->ch, cfg->n_mult, cfg->z_channels);
    omx_vae_decoder m = new omx_vae_decoder_();
    m->cfg = *cfg;
    OMX_HIP_CHECK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    OMX_HIP_CHECK(hipEventCreate(&m->ev0));
    OMX_HIP_CHECK(hipEventCreate(&m->ev1));
    OMX_HIP_CHECK(hipMalloc((void**)&m->stats, (size_t)32 * kGnChunks * 2 * sizeof(double) + 64 * sizeof(float)));
    *out = m;
    return 0;
}

int omx_vae_decoder_destroy(omx_vae_decoder m) {
    if (!m) return 0;
    for (bf16_t* p : {m->buf[0], m->buf[1], m->buf[2], m->buf[3], m->col, m->attn})
        if (p) (void)hipFree(p);
    if (m->stats) (void)hipFree(m->stats);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return 0;
}

int omx_vae_decoder_set_weight(omx_vae_decoder m, const char* name, const void* ptr) {
    OMX_REQUIRE(m && name && ptr, "omx_vae_decoder_set_weight: null argument");
    m->w[name] = (const bf16_t*)ptr;
    return 0;
}

int omx_vae_decoder_out_shape(omx_vae_decoder m, int h, int w, int* out_h, int* out_w, int* out_c) {
    OMX_REQUIRE(m && out_h && out_w && out_c, "omx_vae_decoder_out_shape: null argument");
    const int f = 1 << (m->cfg.n_mult - 1);
    *out_h = h * f; *out_w = w * f; *out_c = m->cfg.out_ch;
    return 0;
}

int omx_vae_decode(omx_vae_decoder m, void* image, const void* latent, int h, int w) {
    OMX_REQUIRE(m && image && latent && h >= 1 && w >= 1, "omx_vae_decode: bad argument");
    const omx_vae_config& c = m->cfg;
    hipStream_t s = m->stream;
    const int nres = c.n_mult, f = 1 << (nres - 1);
    const int block_in = c.ch * c.ch_mult[nres - 1];
    // largest activation: full resolution x ch*ch_mult[0] or any level's H*W*C
    size_t need = (size_t)h * w * std::max(block_in, c.z_channels);
    {
        int H = h, W = w;
        for (int i = nres - 1; i >= 0; --i) {
            const int Cout = c.ch * c.ch_mult[i];
            need = std::max(need, (size_t)H * W * std::max(Cout, i < nres - 1 ? c.ch * c.ch_mult[i + 1] : block_in));
            if (i > 0) { H *= 2; W *= 2; need = std::max(need, (size_t)H * W * Cout); }
        }
    }
    {   // im2col and attention scratch for the whole pass, allocated before the timed region
        size_t col_need = (size_t)h * w * 9 * std::max(c.z_channels, block_in);
        int H = h, W = w, cur = block_in;
        for (int i = nres - 1; i >= 0; --i) {
            const int Cout = c.ch * c.ch_mult[i];
            col_need = std::max(col_need, (size_t)H * W * 9 * std::max(cur, Cout));
            if (i > 0) { H *= 2; W *= 2; col_need = std::max(col_need, (size_t)H * W * 9 * Cout); }
            cur = Cout;
        }
        const size_t T = (size_t)h * w;
        if (grow(&m->col, &m->col_cap, col_need) || grow(&m->attn, &m->attn_cap, T * block_in * 5 + T * T)) return 1;
    }
    if (need > m->buf_cap) {
        OMX_HIP_CHECK(hipStreamSynchronize(s));
        for (int i = 0; i < 4; ++i) {
            if (m->buf[i]) OMX_HIP_CHECK(hipFree(m->buf[i]));
            OMX_HIP_CHECK(hipMalloc((void**)&m->buf[i], need * 2));
        }
        m->buf_cap = need;
    }
    OMX_HIP_CHECK(hipEventRecord(m->ev0, s));
    int H = h, W = w, xi = 0;
    const int64_t hw = (int64_t)h * w;
    // z / scale_factor + shift_factor, post_quant_conv (1x1), conv_in      (:377-387)
    scale_shift_kernel<<<1024, 256, 0, s>>>(m->buf[1], (const bf16_t*)latent, hw * c.z_channels, c.scale_factor, c.shift_factor);
    OMX_LAUNCH_CHECK();
    if (conv1x1(m, m->buf[2], m->buf[1], "post_quant_conv", hw, c.z_channels, c.z_channels, nullptr)) return 1;
    if (conv3x3(m, m->buf[0], m->buf[2], "conv_in", H, W, c.z_channels, block_in, 0, nullptr)) return 1;
    // middle (:390-392)
    if (resnet(m, &xi, "mid_block_resnets_0.", H, W, block_in, block_in)) return 1;
    if (attn_block(m, &xi, "mid_block_attentions_0.", H, W, block_in)) return 1;
    if (resnet(m, &xi, "mid_block_resnets_1.", H, W, block_in, block_in)) return 1;
    // up blocks, lowest resolution first (:395-407); block index b = position in the reference's up_blocks vector
    int cur = block_in;
    for (int i = nres - 1, b = 0; i >= 0; --i, ++b) {
        const int Cout = c.ch * c.ch_mult[i];
        const std::string ub = "up_blocks." + std::to_string(b) + ".";
        for (int j = 0; j <= c.num_res_blocks; ++j) {
            if (resnet(m, &xi, ub + "resnets." + std::to_string(j) + ".", H, W, j == 0 ? cur : Cout, Cout)) return 1;
        }
        if (i > 0) {   // Upsample(2, nearest) + 3x3 convolution, fused
            H *= 2; W *= 2;
            if (conv3x3(m, m->buf[(xi + 1) & 3], m->buf[xi], ub + "upsamplers_0_conv", H, W, Cout, Cout, 1, nullptr)) return 1;
            xi = (xi + 1) & 3;
        }
        cur = Cout;
    }
    // conv_norm_out + SiLU + conv_out (:410-412)
    if (group_norm(m, m->buf[(xi + 1) & 3], m->buf[xi], "conv_norm_out", (int64_t)H * W, c.ch, 1)) return 1;
    if (conv3x3(m, (bf16_t*)image, m->buf[(xi + 1) & 3], "conv_out", H, W, c.ch, c.out_ch, 0, nullptr)) return 1;
    OMX_HIP_CHECK(hipEventRecord(m->ev1, s));
    OMX_HIP_CHECK(hipStreamSynchronize(s));
    OMX_HIP_CHECK(hipEventElapsedTime(&m->last_ms, m->ev0, m->ev1));
    (void)f;
    return 0;
}

int omx_vae_decoder_last_ms(omx_vae_decoder m, float* ms) {
    OMX_REQUIRE(m && ms, "omx_vae_decoder_last_ms: null argument");
    *ms = m->last_ms;
    return 0;
}

}  // extern "C"
