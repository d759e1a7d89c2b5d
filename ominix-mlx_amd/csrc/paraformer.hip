// Paraformer body pieces: SAN-M encoder layer and the CIF integrate-and-fire.
//   reference: funasr-mlx/src/paraformer.rs -- SanmAttention::forward :496-532, FeedForward :560-570,
//   SanmEncoderLayer::forward :618-634, CIFPredictor::cif_fire :779-879.
// The reference runs this model in float32 with explicit QK^T / softmax / PV matmuls and a CPU loop for CIF
// (with a device->host->device round trip).  Here: bf16 activations with fp32 accumulation, the fused
// projection consumed in place through strides by the flash-attention kernel, the FSMN depthwise
// convolution + both residual adds in one pass, and CIF as one block per utterance (the scalar recurrence
// is replayed by every thread, each thread owns one hidden column) -- no host round trip.
#include <math.h>

#include "gemm.hpp"
#include "workspace.hpp"

namespace omx {
namespace {

// out[t, c] = attn_proj[t, c] + v[t, c] + sum_j w[c, j] * v[t + j - pad, c]     (depthwise conv, zero padded)
__global__ __launch_bounds__(256) void fsmn_add_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ attn_proj,
                                                       const bf16_t* __restrict__ v, int64_t ldv,
                                                       const bf16_t* __restrict__ w, int T, int C, int ksize) {
    const int pad = ksize / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)T * C; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i / C), c = (int)(i % C);
        float acc = 0.f;
        for (int j = 0; j < ksize; ++j) {
            const int tt = t + j - pad;
            if (tt >= 0 && tt < T) acc = fmaf(bf16_to_f32(w[(size_t)c * ksize + j]), bf16_to_f32(v[(size_t)tt * ldv + c]), acc);
        }
        // fsmn_out = conv(v) + v (bf16 arrays in the reference's op chain), then attn_proj + fsmn_out
        const float fsmn = round_bf16(round_bf16(acc) + bf16_to_f32(v[(size_t)t * ldv + c]));
        out[i] = f32_to_bf16(bf16_to_f32(attn_proj[i]) + fsmn);
    }
}

// CIF integrate-and-fire, one block per batch item; thread d owns hidden column d
__global__ __launch_bounds__(256) void cif_fire_kernel(const float* __restrict__ hidden, const float* __restrict__ alphas,
                                                       int T, int H, float threshold, float tail_threshold,
                                                       float* __restrict__ frames, int max_frames, int* __restrict__ counts) {
    const int b = blockIdx.x;
    const float* hb = hidden + (size_t)b * T * H;
    const float* ab = alphas + (size_t)b * T;
    float* fb = frames + (size_t)b * max_frames * H;
    for (int d = threadIdx.x; d < H; d += blockDim.x) {
        float integrate = 0.f, frame = 0.f;
        int n = 0;
        for (int t = 0; t < T; ++t) {
            const float alpha = ab[t];
            const float completion = 1.0f - integrate;
            integrate += alpha;
            const bool fire = integrate >= threshold;
            if (fire) integrate -= 1.0f;
            const float cur = fire ? completion : alpha;
            const float remainds = alpha - cur;
            const float hv = hb[(size_t)t * H + d];
            frame += cur * hv;
            if (fire) {
                if (n < max_frames) fb[(size_t)n * H + d] = frame;
                ++n;
                frame = remainds * hv;
            }
        }
        if (integrate > tail_threshold) {
            if (n < max_frames) fb[(size_t)n * H + d] = frame;
            ++n;
        }
        if (d == 0) counts[b] = n;
    }
}

}  // namespace
}  // namespace omx

extern "C" {

int omx_sanm_encoder_layer(void* out, const void* x, const omx_sanm_layer_weights* w, int T, int in_dim, int dim,
                           int heads, int ffn_dim, int kernel_size, omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(out && x && w, "omx_sanm_encoder_layer: null argument");
    OMX_REQUIRE(T > 0 && dim % heads == 0 && dim / heads == 128, "omx_sanm_encoder_layer: head_dim must be 128 (dim %d, heads %d)", dim, heads);
    OMX_REQUIRE(kernel_size % 2 == 1 && kernel_size <= 31, "omx_sanm_encoder_layer: odd kernel_size <= 31 expected");
    hipStream_t s = (hipStream_t)stream;
    const size_t need = ((size_t)T * (in_dim + 3 * dim + 3 * dim + ffn_dim + dim) + 1024) * 2;
    void* ws = nullptr;
    if (get_workspace(&ws, need)) return 1;
    bf16_t* h1 = (bf16_t*)ws;                     // [T, in_dim]  LN1(x)
    bf16_t* qkv = h1 + (size_t)T * in_dim;        // [T, 3*dim]
    bf16_t* att = qkv + (size_t)T * 3 * dim;      // [T, dim]
    bf16_t* prj = att + (size_t)T * dim;          // [T, dim]
    bf16_t* xr = prj + (size_t)T * dim;           // [T, dim]  x after the attention residual
    bf16_t* h2 = xr + (size_t)T * dim;            // [T, dim]  LN2
    bf16_t* ff = h2 + (size_t)T * dim;            // [T, ffn_dim]
    const bf16_t* xin = (const bf16_t*)x;
    // h = norm1(x) ; qkv = linear_q_k_v(h)                                           (paraformer.rs:619, 500)
    if (omx_layer_norm(h1, xin, w->norm1_w, w->norm1_b, T, in_dim, 1e-5f, OMX_BFLOAT16, stream)) return 1;
    if (launch_gemm_bf16(qkv, h1, (const bf16_t*)w->qkv_w, (const bf16_t*)w->qkv_b, T, 3 * dim, in_dim, s)) return 1;
    // softmax(q k^T * d^-1/2) v, 4 heads x 128, operands read in place from the fused projection    (:503-522)
    AttnLayout L = {0, 128, 3 * (int64_t)dim, 3 * (int64_t)dim, 0, 128, dim};
    if (launch_attn_prefill(att, qkv, qkv + dim, qkv + 2 * dim, 1, heads, heads, T, T, 128, 0, 128, 1.0f / sqrtf(128.0f),
                            OMX_MASK_NONE, nullptr, s, false, &L))
        return 1;
    if (launch_gemm_bf16(prj, att, (const bf16_t*)w->out_w, (const bf16_t*)w->out_b, T, dim, dim, s)) return 1;
    // out_proj(attn) + (fsmn_block(v) + v)                                             (:524-529)
    fsmn_add_kernel<<<1024, 256, 0, s>>>(att, prj, qkv + 2 * dim, 3 * (int64_t)dim, (const bf16_t*)w->fsmn_w, T, dim, kernel_size);
    OMX_LAUNCH_CHECK();
    // residual only when the layer keeps its width (first layer maps 560 -> 512 without it, :625-629)
    const bf16_t* xa = att;
    if (in_dim == dim) {
        if (omx_add(xr, xin, att, (int64_t)T * dim, OMX_BFLOAT16, stream)) return 1;
        xa = xr;
    }
    if (omx_layer_norm(h2, xa, w->norm2_w, w->norm2_b, T, dim, 1e-5f, OMX_BFLOAT16, stream)) return 1;
    if (launch_gemm_bf16_bias_relu(ff, h2, (const bf16_t*)w->ffn_up_w, (const bf16_t*)w->ffn_up_b, T, ffn_dim, dim, s)) return 1;
    if (launch_gemm_bf16(prj, ff, (const bf16_t*)w->ffn_down_w, (const bf16_t*)w->ffn_down_b, T, dim, ffn_dim, s)) return 1;
    return omx_add(out, xa, prj, (int64_t)T * dim, OMX_BFLOAT16, stream);
}

int omx_cif_fire(float* frames, int* counts, const float* hidden, const float* alphas, int batch, int T, int H,
                 float threshold, float tail_threshold, int max_frames, omx_stream stream) {
    OMX_REQUIRE(frames && counts && hidden && alphas, "omx_cif_fire: null argument");
    OMX_REQUIRE(batch > 0 && T > 0 && H > 0 && max_frames > 0, "omx_cif_fire: bad shape");
    OMX_HIP_CHECK(hipMemsetAsync(frames, 0, (size_t)batch * max_frames * H * 4, (hipStream_t)stream));
    omx::cif_fire_kernel<<<batch, 256, 0, (hipStream_t)stream>>>(hidden, alphas, T, H, threshold, tail_threshold, frames,
                                                                 max_frames, counts);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
