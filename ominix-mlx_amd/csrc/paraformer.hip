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
// resid != null: out = resid + bf16(that)   (the layer's attention residual, paraformer.rs:625-629, in the same launch)
__global__ __launch_bounds__(256) void fsmn_add_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ attn_proj,
                                                       const bf16_t* __restrict__ v, int64_t ldv,
                                                       const bf16_t* __restrict__ w, int T, int C, int ksize,
                                                       const bf16_t* __restrict__ resid) {
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
        float o = bf16_to_f32(attn_proj[i]) + fsmn;
        if (resid) o = bf16_to_f32(resid[i]) + round_bf16(o);
        out[i] = f32_to_bf16(o);
    }
}

// CIF integrate-and-fire (paraformer.rs:779-879).  The fire decisions form a scalar recurrence over time that does not depend
// on the hidden column, and a fired frame only sums the few time steps between two fires.  So: every block of the (batch, NB)
// grid replays the scalar recurrence from LDS (one thread, ~10 ns per step; T = 501 -> 5 us) and records, per fired frame, the
// time step and the weight of that step; then the blocks share the frames, thread d owning hidden column d, each frame summed
// in time order with the operations of the serial definition (frame = remainder * h; frame += alpha * h ...; frame +=
// completion * h) -- the same bits as a one-thread-per-column walk over all T steps, which took 280 us for 30 s of audio.
__global__ __launch_bounds__(256) void cif_fire_kernel(const float* __restrict__ hidden, const float* __restrict__ alphas,
                                                       int T, int H, float threshold, float tail_threshold,
                                                       float* __restrict__ frames, int max_frames, int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cif_smem[];
    float* al = reinterpret_cast<float*>(cif_smem);                 // [T]
    int* fire_t = reinterpret_cast<int*>(al + T);                   // [max_frames] time step of the n-th fire
    float* fire_w = reinterpret_cast<float*>(fire_t + max_frames);  // [max_frames] weight of that step in the fired frame
    __shared__ int s_fired, s_tail;
    const int b = blockIdx.x;
    const float* hb = hidden + (size_t)b * T * H;
    const float* ab = alphas + (size_t)b * T;
    float* fb = frames + (size_t)b * max_frames * H;
    for (int t = threadIdx.x; t < T; t += blockDim.x) al[t] = ab[t];
    __syncthreads();
    if (threadIdx.x == 0) {
        float integrate = 0.f;
        int n = 0;
#pragma unroll 4
        for (int t = 0; t < T; ++t) {
            const float alpha = al[t];
            const float completion = 1.0f - integrate;
            integrate += alpha;
            if (integrate >= threshold) {
                integrate -= 1.0f;
                if (n < max_frames) { fire_t[n] = t; fire_w[n] = completion; }
                ++n;
            }
        }
        s_fired = n;
        s_tail = integrate > tail_threshold ? 1 : 0;
        if (blockIdx.y == 0) counts[b] = n + s_tail;
    }
    __syncthreads();
    const int fired = s_fired;
    const int n_out = min(fired + s_tail, max_frames);
    for (int n = blockIdx.y; n < n_out; n += gridDim.y) {
        const int t_prev = n > 0 ? fire_t[n - 1] : -1;              // the fire that opened this frame
        const bool closes = n < fired;                              // ends with a fire (else: the tail frame)
        const int t_end = closes ? fire_t[n] : T;                   // plain-alpha steps are (t_prev, t_end)
        for (int d = threadIdx.x; d < H; d += blockDim.x) {
            float frame = 0.f;
            if (t_prev >= 0) frame = (al[t_prev] - fire_w[n - 1]) * hb[(size_t)t_prev * H + d];
            for (int t = t_prev + 1; t < t_end; ++t) frame += al[t] * hb[(size_t)t * H + d];
            if (closes) frame += fire_w[n] * hb[(size_t)t_end * H + d];
            fb[(size_t)n * H + d] = frame;
        }
    }
}

// out = bf16(mel * sqrt(512) + PE), PE[pos, i] = sin((pos+1) * ts_i), PE[pos, half+i] = cos(...), ts_i = exp(-i ln(1e4)/(half-1))
__global__ __launch_bounds__(256) void paraformer_embed_kernel(bf16_t* __restrict__ out, const float* __restrict__ mel, int T, int dim) {
    const int half = dim / 2;
    const float inc = logf(10000.0f) / ((float)half - 1.0f);
    const float scale = sqrtf(512.0f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)T * dim; i += (int64_t)gridDim.x * 256) {
        const int pos = (int)(i / dim), c = (int)(i % dim);
        const int k = c < half ? c : c - half;
        const float st = (float)(pos + 1) * expf(-(float)k * inc);
        const float pe = c < half ? sinf(st) : cosf(st);
        out[i] = f32_to_bf16(mel[i] * scale + pe);
    }
}

// im2col for a dense Conv1d over time: col[t, j*C + c] = x[t + j - pad, c] (zero padded); also x as f32
__global__ __launch_bounds__(256) void im2col_time_kernel(bf16_t* __restrict__ col, float* __restrict__ xf, const bf16_t* __restrict__ x,
                                                          int T, int C, int ksize) {
    const int pad = ksize / 2;
    const int64_t n = (int64_t)T * ksize * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C), j = (int)((i / C) % ksize), t = (int)(i / ((int64_t)C * ksize));
        const int tt = t + j - pad;
        const bf16_t v = (tt >= 0 && tt < T) ? x[(size_t)tt * C + c] : (bf16_t)0;
        col[i] = v;
        if (xf && j == pad) xf[(size_t)t * C + c] = bf16_to_f32(v);
    }
}

// alphas[t] = sigmoid(bf16(h[t] . w + b)), one wave per row
__global__ __launch_bounds__(256) void alpha_head_kernel(float* __restrict__ alphas, const bf16_t* __restrict__ h,
                                                         const bf16_t* __restrict__ w, const bf16_t* __restrict__ b, int T, int C) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc = fmaf(bf16_to_f32(h[(size_t)t * C + c]), bf16_to_f32(w[c]), acc);
    acc = wave_sum(acc);
    if (lane == 0) {
        const float z = round_bf16(acc + (b ? bf16_to_f32(b[0]) : 0.f));
        alphas[t] = round_bf16(1.0f / (1.0f + expf(-z)));
    }
}

template <int DD, int DS>
__global__ __launch_bounds__(256) void cast_dt_kernel(typename Elem<DD>::T* __restrict__ dst, const typename Elem<DS>::T* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        Elem<DD>::st(dst + i, Elem<DS>::ld(src + i));
}

}  // namespace
}  // namespace omx

extern "C" {

int omx_cast(void* dst, omx_dtype dd, const void* src, omx_dtype ds, int64_t n, omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(dst && src && n >= 0, "omx_cast: bad arguments");
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipStream_t s = (hipStream_t)stream;
#define OMX_CAST_CASE(A, B)                                                                                                 \
    if (dd == A && ds == B) {                                                                                               \
        cast_dt_kernel<A, B><<<blocks, 256, 0, s>>>((Elem<A>::T*)dst, (const Elem<B>::T*)src, n);                           \
        OMX_LAUNCH_CHECK();                                                                                                 \
        return 0;                                                                                                           \
    }
    OMX_CAST_CASE(OMX_BFLOAT16, OMX_FLOAT32) OMX_CAST_CASE(OMX_FLOAT32, OMX_BFLOAT16) OMX_CAST_CASE(OMX_FLOAT16, OMX_FLOAT32)
    OMX_CAST_CASE(OMX_FLOAT32, OMX_FLOAT16) OMX_CAST_CASE(OMX_BFLOAT16, OMX_FLOAT16) OMX_CAST_CASE(OMX_FLOAT16, OMX_BFLOAT16)
    OMX_CAST_CASE(OMX_FLOAT32, OMX_FLOAT32) OMX_CAST_CASE(OMX_BFLOAT16, OMX_BFLOAT16) OMX_CAST_CASE(OMX_FLOAT16, OMX_FLOAT16)
#undef OMX_CAST_CASE
    return set_error("omx_cast: unsupported conversion %d -> %d", (int)ds, (int)dd);
}

int omx_paraformer_embed(void* out, const float* mel, int T, int dim, omx_stream stream) {
    OMX_REQUIRE(out && mel && T > 0 && dim >= 4 && dim % 2 == 0, "omx_paraformer_embed: bad arguments");
    omx::paraformer_embed_kernel<<<1024, 256, 0, (hipStream_t)stream>>>((omx::bf16_t*)out, mel, T, dim);
    OMX_LAUNCH_CHECK();
    return 0;
}

int omx_cif_alphas(float* alphas, float* hidden_f32, const void* enc, const void* conv_w, const void* conv_b, const void* proj_w,
                   const void* proj_b, int T, int dim, int kernel_size, omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(alphas && enc && conv_w && proj_w, "omx_cif_alphas: null argument");
    OMX_REQUIRE(T > 0 && dim % 64 == 0 && kernel_size % 2 == 1 && kernel_size <= 15, "omx_cif_alphas: bad shape");
    hipStream_t s = (hipStream_t)stream;
    void* ws = nullptr;
    if (get_workspace(&ws, ((size_t)T * kernel_size * dim + (size_t)T * dim) * 2 + 1024)) return 1;
    bf16_t* col = (bf16_t*)ws;
    bf16_t* h = col + (size_t)T * kernel_size * dim;
    im2col_time_kernel<<<1024, 256, 0, s>>>(col, hidden_f32, (const bf16_t*)enc, T, dim, kernel_size);
    OMX_LAUNCH_CHECK();
    if (launch_gemm_bf16_bias_relu(h, col, (const bf16_t*)conv_w, (const bf16_t*)conv_b, T, dim, kernel_size * dim, s)) return 1;
    alpha_head_kernel<<<(T + 3) / 4, 256, 0, s>>>(alphas, h, (const bf16_t*)proj_w, (const bf16_t*)proj_b, T, dim);
    OMX_LAUNCH_CHECK();
    return 0;
}

int omx_paraformer_decoder_layer(void* out, const void* x, const void* enc, const omx_paraformer_decoder_weights* w, int N, int Ts,
                                 int dim, int enc_dim, int heads, int ffn_dim, int kernel_size, omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(out && x && enc && w, "omx_paraformer_decoder_layer: null argument");
    OMX_REQUIRE(N > 0 && Ts > 0 && dim % heads == 0 && dim / heads == 128, "omx_paraformer_decoder_layer: head_dim must be 128 (dim %d, heads %d)", dim, heads);
    OMX_REQUIRE(kernel_size % 2 == 1 && kernel_size <= 31, "omx_paraformer_decoder_layer: odd kernel_size <= 31 expected");
    hipStream_t s = (hipStream_t)stream;
    const size_t need = ((size_t)N * (5 * (size_t)dim + 2 * (size_t)ffn_dim) + (size_t)Ts * 2 * dim + 1024) * 2;
    void* ws = nullptr;
    if (get_workspace(&ws, need)) return 1;
    bf16_t* h = (bf16_t*)ws;                       // [N, dim]   scratch LN outputs
    bf16_t* ff = h + (size_t)N * dim;              // [N, ffn]
    bf16_t* ffn = ff + (size_t)N * ffn_dim;        // [N, ffn]   LN(ffn)
    bf16_t* tgt = ffn + (size_t)N * ffn_dim;       // [N, dim]
    bf16_t* x1 = tgt + (size_t)N * dim;            // [N, dim]   after the FSMN residual
    bf16_t* q = x1 + (size_t)N * dim;              // [N, dim]
    bf16_t* att = q + (size_t)N * dim;             // [N, dim]
    bf16_t* kv = att + (size_t)N * dim;            // [Ts, 2*dim]
    const bf16_t* xin = (const bf16_t*)x;
    // tgt = down(LN_ffn(relu(up(norm1(x)))))                                                     (:1036-1042)
    if (omx_layer_norm(h, xin, w->norm1_w, w->norm1_b, N, dim, 1e-5f, OMX_BFLOAT16, stream)) return 1;
    if (launch_gemm_bf16_bias_relu(ff, h, (const bf16_t*)w->ffn_up_w, (const bf16_t*)w->ffn_up_b, N, ffn_dim, dim, s)) return 1;
    if (omx_layer_norm(ffn, ff, w->ffn_norm_w, w->ffn_norm_b, N, ffn_dim, 1e-5f, OMX_BFLOAT16, stream)) return 1;
    if (launch_gemm_bf16(tgt, ffn, (const bf16_t*)w->ffn_down_w, nullptr, N, dim, ffn_dim, s)) return 1;
    // x1 = x + (fsmn(norm2(tgt)) + norm2(tgt))                                                   (:1044-1047)
    if (omx_layer_norm(h, tgt, w->norm2_w, w->norm2_b, N, dim, 1e-5f, OMX_BFLOAT16, stream)) return 1;
    fsmn_add_kernel<<<1024, 256, 0, s>>>(x1, xin, h, dim, (const bf16_t*)w->fsmn_w, N, dim, kernel_size, nullptr);
    OMX_LAUNCH_CHECK();
    // out = x1 + src_attn_out(softmax(q k^T * d^-1/2) v), q from norm3(x1), k/v from the encoder output      (:1049-1052, 981-1017)
    if (omx_layer_norm(h, x1, w->norm3_w, w->norm3_b, N, dim, 1e-5f, OMX_BFLOAT16, stream)) return 1;
    if (launch_gemm_bf16(q, h, (const bf16_t*)w->q_w, (const bf16_t*)w->q_b, N, dim, dim, s)) return 1;
    if (launch_gemm_bf16(kv, (const bf16_t*)enc, (const bf16_t*)w->kv_w, (const bf16_t*)w->kv_b, Ts, 2 * dim, enc_dim, s)) return 1;
    AttnLayout L = {0, 128, dim, 2 * (int64_t)dim, 0, 128, dim};
    if (launch_attn_prefill(att, q, kv, kv + dim, 1, heads, heads, N, Ts, 128, 0, 128, 1.0f / sqrtf(128.0f), OMX_MASK_NONE,
                            nullptr, s, false, &L))
        return 1;
    return launch_gemm_bf16_ex((bf16_t*)out, att, (const bf16_t*)w->out_w, (const bf16_t*)w->out_b, x1, N, dim, dim, s);
}

int omx_paraformer_decoder_tail(void* logits, const void* x, const omx_paraformer_tail_weights* w, int N, int dim, int ffn_dim,
                                int vocab, omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(logits && x && w && N > 0, "omx_paraformer_decoder_tail: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    void* ws = nullptr;
    if (get_workspace(&ws, ((size_t)N * (2 * (size_t)dim + 2 * (size_t)ffn_dim) + 1024) * 2)) return 1;
    bf16_t* h = (bf16_t*)ws;
    bf16_t* ff = h + (size_t)N * dim;
    bf16_t* ffn = ff + (size_t)N * ffn_dim;
    bf16_t* t = ffn + (size_t)N * ffn_dim;
    if (omx_layer_norm(h, x, w->norm1_w, w->norm1_b, N, dim, 1e-5f, OMX_BFLOAT16, stream)) return 1;
    if (launch_gemm_bf16_bias_relu(ff, h, (const bf16_t*)w->up_w, (const bf16_t*)w->up_b, N, ffn_dim, dim, s)) return 1;
    if (omx_layer_norm(ffn, ff, w->ffn_norm_w, w->ffn_norm_b, N, ffn_dim, 1e-5f, OMX_BFLOAT16, stream)) return 1;
    if (launch_gemm_bf16(t, ffn, (const bf16_t*)w->down_w, nullptr, N, dim, ffn_dim, s)) return 1;
    if (omx_layer_norm(h, t, w->after_norm_w, w->after_norm_b, N, dim, 1e-5f, OMX_BFLOAT16, stream)) return 1;
    return launch_gemm_bf16((bf16_t*)logits, h, (const bf16_t*)w->out_w, (const bf16_t*)w->out_b, N, vocab, dim, s);
}

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
    // + the layer residual in the same launch; only when the layer keeps its width (the first maps 560 -> 512 without it, :625-629)
    const bf16_t* xa = xr;
    fsmn_add_kernel<<<1024, 256, 0, s>>>(xr, prj, qkv + 2 * dim, 3 * (int64_t)dim, (const bf16_t*)w->fsmn_w, T, dim, kernel_size,
                                         in_dim == dim ? xin : nullptr);
    OMX_LAUNCH_CHECK();
    if (omx_layer_norm(h2, xa, w->norm2_w, w->norm2_b, T, dim, 1e-5f, OMX_BFLOAT16, stream)) return 1;
    if (launch_gemm_bf16_bias_relu(ff, h2, (const bf16_t*)w->ffn_up_w, (const bf16_t*)w->ffn_up_b, T, ffn_dim, dim, s)) return 1;
    // out = xa + bf16(ffn_down(ff) + bias): the FFN residual in the GEMM epilogue
    return launch_gemm_bf16_ex((bf16_t*)out, ff, (const bf16_t*)w->ffn_down_w, (const bf16_t*)w->ffn_down_b, xa, T, dim, ffn_dim, s);
}

int omx_cif_fire(float* frames, int* counts, const float* hidden, const float* alphas, int batch, int T, int H,
                 float threshold, float tail_threshold, int max_frames, omx_stream stream) {
    OMX_REQUIRE(frames && counts && hidden && alphas, "omx_cif_fire: null argument");
    OMX_REQUIRE(batch > 0 && T > 0 && H > 0 && max_frames > 0, "omx_cif_fire: bad shape");
    OMX_HIP_CHECK(hipMemsetAsync(frames, 0, (size_t)batch * max_frames * H * 4, (hipStream_t)stream));
    const size_t lds = (size_t)T * 4 + (size_t)max_frames * 8;
    OMX_REQUIRE(lds <= 160 * 1024 - 64, "omx_cif_fire: %d steps / %d frames exceed the LDS tables (%zu bytes)", T, max_frames, lds);
    static bool attr_set = false;
    if (!attr_set) {
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)omx::cif_fire_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        attr_set = true;
    }
    omx::cif_fire_kernel<<<dim3(batch, 32), 256, lds, (hipStream_t)stream>>>(hidden, alphas, T, H, threshold, tail_threshold, frames,
                                                                            max_frames, counts);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
