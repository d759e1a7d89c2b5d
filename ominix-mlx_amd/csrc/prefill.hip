// Glue kernels of the batched (T > 1) prefill step of the decode engine; each fuses what the
// reference issues as several lazy ops.
#include "prefill.hpp"
#include "act16.hpp"

namespace omx {
namespace {

// One D/8-lane group per (token, head) row.  q rows: per-head RMSNorm -> RoPE -> q_out[h][t][:]
// (the [B,H,T,D] operand of SDPA); k rows: same, written straight into the KV slab at offset+t
// (KVCache::update_and_fetch, cache.rs:183-188); v rows: copied into the slab.
//   reference: qwen3-mlx/src/model.rs:172-196 (reshape/transpose, q_norm/k_norm, rope, cache update).
// F16: a float16 model (float16 rows, norm weights, cache slabs and rounding points: act16.hpp)
template <int D, bool F16 = false>
__global__ __launch_bounds__(256) void qk_norm_rope_scatter_kernel(
    const bf16_t* __restrict__ q_lin, const bf16_t* __restrict__ k_lin, const bf16_t* __restrict__ v_lin,
    const bf16_t* __restrict__ q_norm_w, const bf16_t* __restrict__ k_norm_w, const float* __restrict__ rope_cos,
    const float* __restrict__ rope_sin, bf16_t* __restrict__ q_out, bf16_t* __restrict__ kcache,
    bf16_t* __restrict__ vcache, int T, int H, int Hkv, int cap, int offset, float eps) {
    typedef Act16<F16> A16;
    constexpr int LPR = D / 8;
    const int lane = threadIdx.x & 63;
    const int c = lane % LPR;
    const int rows_per_block = 256 / LPR;
    const int64_t row = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / LPR;
    const int per_tok = H + 2 * Hkv;
    if (row >= (int64_t)T * per_tok) return;
    const int t = (int)(row / per_tok), hh = (int)(row % per_tok);
    const int pos = offset + t;
    if (hh >= H + Hkv) {   // v: plain copy into the slab
        const int kvh = hh - H - Hkv;
        *reinterpret_cast<u32x4*>(vcache + ((size_t)kvh * cap + pos) * D + c * 8) =
            *reinterpret_cast<const u32x4*>(v_lin + ((size_t)t * Hkv + kvh) * D + c * 8);
        return;
    }
    const bool is_q = hh < H;
    const bf16_t* src = is_q ? q_lin + ((size_t)t * H + hh) * D : k_lin + ((size_t)t * Hkv + (hh - H)) * D;
    const bf16_t* w = is_q ? q_norm_w : k_norm_w;
    const u32x4 r = *reinterpret_cast<const u32x4*>(src + c * 8);
    const u32x4 wr = w ? *reinterpret_cast<const u32x4*>(w + c * 8) : (F16 ? u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u} : u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});   // no q/k norm (Mixtral): weight 1
    float x[8], wv[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x[2 * e] = A16::lo(r[e]); x[2 * e + 1] = A16::hi(r[e]);
        wv[2 * e] = A16::lo(wr[e]); wv[2 * e + 1] = A16::hi(wr[e]);
    }
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(x[e], x[e], ss);
    ss = group_sum<LPR>(ss);
    const float rstd = w ? 1.0f / sqrtf(ss / (float)D + eps) : 1.0f;   // without a norm the projection goes to RoPE as it is
    const int i0 = (c % (LPR / 2)) * 8;
    const bool first_half = c < LPR / 2;
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xn = A16::rnd(x[e] * rstd * wv[e]);
        // partner element i +- D/2 lives in lane c ^ (LPR/2)
        const float other = (LPR == 16) ? dpp_f<0x128>(xn) : dpp_f<0x1B>(dpp_f<kDppHalfMirror>(xn));
        const float cs = rope_cos[(size_t)pos * (D / 2) + i0 + e], sn = rope_sin[(size_t)pos * (D / 2) + i0 + e];
        y[e] = first_half ? xn * cs - other * sn : other * sn + xn * cs;
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = A16::pack(y[2 * e], y[2 * e + 1]);
    bf16_t* dst = is_q ? q_out + ((size_t)hh * T + t) * D : kcache + ((size_t)(hh - H) * cap + pos) * D;
    *reinterpret_cast<u32x4*>(dst + c * 8) = o;
}

// nn::silu(gate) * up with every primitive's result held in bf16 (qwen3-mlx/src/model.rs:264-265)
__global__ __launch_bounds__(256) void silu_mul_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ gate,
                                                       const bf16_t* __restrict__ up, int64_t n_vec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * 256) {
        const u32x4 g = reinterpret_cast<const u32x4*>(gate)[i];
        const u32x4 u = reinterpret_cast<const u32x4*>(up)[i];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r[2];
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const float gv = hlf ? bf16hi(g[e]) : bf16lo(g[e]);
                const float uv = hlf ? bf16hi(u[e]) : bf16lo(u[e]);
                const float sg = round_bf16(1.0f / (1.0f + expf(-gv)));
                r[hlf] = round_bf16(gv * sg) * uv;
            }
            o[e] = pack_bf16(r[0], r[1]);
        }
        reinterpret_cast<u32x4*>(out)[i] = o;
    }
}

}  // namespace

int launch_qk_norm_rope_scatter(const bf16_t* q_lin, const bf16_t* k_lin, const bf16_t* v_lin, const bf16_t* q_norm_w,
                                const bf16_t* k_norm_w, const float* rope_cos, const float* rope_sin, bf16_t* q_out,
                                bf16_t* kcache, bf16_t* vcache, int T, int H, int Hkv, int D, int cap, int offset,
                                float eps, hipStream_t s, bool f16) {
    OMX_REQUIRE(D == 64 || D == 128, "qk_norm_rope: head_dim %d unsupported", D);
    const int64_t rows = (int64_t)T * (H + 2 * Hkv);
    const int rpb = 256 / (D / 8);
    const unsigned blocks = (unsigned)((rows + rpb - 1) / rpb);
    if (f16) {
        OMX_REQUIRE(D == 128, "qk_norm_rope: float16 models have head_dim 128");
        qk_norm_rope_scatter_kernel<128, true><<<blocks, 256, 0, s>>>(q_lin, k_lin, v_lin, q_norm_w, k_norm_w, rope_cos, rope_sin,
                                                                      q_out, kcache, vcache, T, H, Hkv, cap, offset, eps);
    } else
    if (D == 128)
        qk_norm_rope_scatter_kernel<128><<<blocks, 256, 0, s>>>(q_lin, k_lin, v_lin, q_norm_w, k_norm_w, rope_cos, rope_sin,
                                                                q_out, kcache, vcache, T, H, Hkv, cap, offset, eps);
    else
        qk_norm_rope_scatter_kernel<64><<<blocks, 256, 0, s>>>(q_lin, k_lin, v_lin, q_norm_w, k_norm_w, rope_cos, rope_sin,
                                                               q_out, kcache, vcache, T, H, Hkv, cap, offset, eps);
    OMX_LAUNCH_CHECK();
    return 0;
}

int launch_silu_mul(bf16_t* out, const bf16_t* gate, const bf16_t* up, int64_t n, hipStream_t s) {
    OMX_REQUIRE(n % 8 == 0, "silu_mul: element count %lld must be a multiple of 8", (long long)n);
    const int64_t nv = n / 8;
    int64_t blocks = (nv + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    silu_mul_kernel<<<(unsigned)blocks, 256, 0, s>>>(out, gate, up, nv);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // namespace omx
