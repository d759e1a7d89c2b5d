// KV-cached decode attention (Tq == 1) for gfx950: split-KV flash-decode.
//   reference: mlx_rs_core::scaled_dot_product_attention (mlx-rs-core/src/utils.rs:191-209) ->
//   mlx_fast_scaled_dot_product_attention (mlx-c fast.h:189-198), whose Tq==1 case MLX serves with
//   a dedicated vector kernel (mlx-rs/src/fast.rs:114).  HBM-bound: 2*Hkv*T*D*2 bytes per layer.
//
// Layout / mapping (wave64):
//   * K/V rows are D bf16 = D/8 lanes x 16 B; a wave-instruction covers 64/(D/8) consecutive
//     tokens (4 for D=128) as ONE contiguous 1 KiB burst;
//   * the G = H/Hkv query heads that share a KV head are processed together in registers, so
//     each K/V byte is read once per KV head, not once per query head (GQA without tiling,
//     fast.rs:118);
//   * scores: per-lane 8-element partial dot, xor-shuffle reduce inside the D/8-lane group;
//     softmax state (m, l) in fp32 (fast.rs:116); a wave keeps ONE running max per head so the
//     final merge across its token sub-groups is a plain sum;
//   * grid = (B*Hkv) x nsplit; each block writes an un-normalised partial (m, l, o[D]) per head;
//     attn_combine_kernel merges the splits and rounds once to the output dtype.
//   * FUSED variant (decode engine): the block also applies the per-head q/k RMSNorm and RoPE of
//     qwen3-mlx/src/model.rs:172-194 and writes the new K/V row into the cache (cache.rs:183-188),
//     removing 6 launches per layer.
#include "attn.hpp"

namespace omx {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = 4;
constexpr int kUnroll = 4;   // token rows per lane-group per step -> 4 K + 4 V loads in flight

template <int D, int GT, bool FUSED>
__global__ __launch_bounds__(kBlock) void attn_decode_kernel(const AttnDecodeArgs a) {
    constexpr int LPR = D / 8;          // lanes per K/V row
    constexpr int TPW = 64 / LPR;       // tokens per wave-instruction
    constexpr int STEP = TPW * kUnroll; // tokens per wave per step
    __shared__ float sm_m[kWaves][GT];
    __shared__ float sm_l[kWaves][GT];
    __shared__ float sm_o[kWaves][GT][D];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane % LPR;           // 8-element chunk of the head dim owned by this lane
    const int sg = lane / LPR;          // token sub-group inside the wave
    const int bk = blockIdx.x;          // b * Hkv + kvh
    const int b = bk / a.Hkv, kvh = bk % a.Hkv;
    const int split = blockIdx.y;
    const int G = a.H / a.Hkv;

    int Tk = a.Tk;
    int pos = 0;
    if (FUSED) {
        pos = *a.pos_ptr;   // tokens already in the cache == RoPE offset (model.rs:186-194)
        Tk = pos + 1;
    }
    // token range of this split: multiples of the block step
    const int per = (Tk + a.nsplit - 1) / a.nsplit;
    const int chunk = ((per + STEP * kWaves - 1) / (STEP * kWaves)) * (STEP * kWaves);
    const int t_begin = split * chunk;
    const int t_end = min(Tk, t_begin + chunk);

    const bf16_t* Kb = a.k + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;
    const bf16_t* Vb = a.v + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;

    // ---- query (G heads) -> registers, pre-multiplied by scale in fp32 ----
    float q[GT][8];
    if (FUSED) {
        // raw projections: [H*D | Hkv*D | Hkv*D] bf16 (gemv epilogue output)
        const bf16_t* qraw = a.qkv;
        const bf16_t* kraw = a.qkv + (size_t)a.H * D;
        const bf16_t* vraw = kraw + (size_t)a.Hkv * D;
        const float* cosr = a.rope_cos + (size_t)pos * (D / 2);
        const float* sinr = a.rope_sin + (size_t)pos * (D / 2);
        float cs[8], sn[8], wq[8], wk[8];
        {
            const int i0 = (c % (LPR / 2)) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                cs[e] = cosr[i0 + e];
                sn[e] = sinr[i0 + e];
            }
            const u32x4 wqv = *reinterpret_cast<const u32x4*>(a.q_norm_w + c * 8);
            const u32x4 wkv = *reinterpret_cast<const u32x4*>(a.k_norm_w + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wq[2 * e] = bf16lo(wqv[e]);
                wq[2 * e + 1] = bf16hi(wqv[e]);
                wk[2 * e] = bf16lo(wkv[e]);
                wk[2 * e + 1] = bf16hi(wkv[e]);
            }
        }
        const bool first_half = c < LPR / 2;
        auto norm_rope = [&](const bf16_t* src, const float (&w)[8], float (&out)[8]) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(src + c * 8);
            float x[8];
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[2 * e] = bf16lo(r[e]);
                x[2 * e + 1] = bf16hi(r[e]);
                ss = fmaf(x[2 * e], x[2 * e], ss);
                ss = fmaf(x[2 * e + 1], x[2 * e + 1], ss);
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            const float rstd = 1.0f / sqrtf(ss / (float)D + a.eps);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xn = round_bf16(x[e] * rstd * w[e]);          // RMSNorm output is bf16
                const float other = __shfl_xor(xn, LPR / 2, 64);           // partner element i +- D/2
                const float y = first_half ? xn * cs[e] - other * sn[e] : other * sn[e] + xn * cs[e];
                out[e] = round_bf16(y);                                    // RoPE output is bf16
            }
        };
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            const int h = kvh * G + min(g, G - 1);
            norm_rope(qraw + (size_t)h * D, wq, q[g]);
#pragma unroll
            for (int e = 0; e < 8; ++e) q[g][e] *= a.scale;
        }
        // the split that owns token `pos` appends the new K/V row before anyone reads it
        if (pos >= t_begin && pos < t_begin + chunk) {
            if (wave == 0 && sg == 0) {
                float kn[8];
                norm_rope(kraw + (size_t)kvh * D, wk, kn);
                u32x4 kp;
#pragma unroll
                for (int e = 0; e < 4; ++e) kp[e] = pack_bf16(kn[2 * e], kn[2 * e + 1]);
                bf16_t* Kw = const_cast<bf16_t*>(Kb);
                bf16_t* Vw = const_cast<bf16_t*>(Vb);
                *reinterpret_cast<u32x4*>(Kw + (size_t)pos * D + c * 8) = kp;
                *reinterpret_cast<u32x4*>(Vw + (size_t)pos * D + c * 8) =
                    *reinterpret_cast<const u32x4*>(vraw + (size_t)kvh * D + c * 8);
            }   // (norm_rope's shuffles stay inside the LPR-lane group, all of whose lanes are active)
            __threadfence_block();
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            const int h = kvh * G + min(g, G - 1);
            const u32x4 r = *reinterpret_cast<const u32x4*>(a.q + ((size_t)b * a.H + h) * D + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                q[g][2 * e] = bf16lo(r[e]) * a.scale;
                q[g][2 * e + 1] = bf16hi(r[e]) * a.scale;
            }
        }
    }

    float m[GT], l[GT], o[GT][8];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[g][e] = 0.f;
    }

    for (int t0 = t_begin + wave * STEP; t0 < t_end; t0 += STEP * kWaves) {
        u32x4 kr[kUnroll], vr[kUnroll];
        int tok[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            tok[u] = t0 + u * TPW + sg;
            const int tc = min(tok[u], t_end - 1);
            kr[u] = *reinterpret_cast<const u32x4*>(Kb + (size_t)tc * D + c * 8);
            vr[u] = *reinterpret_cast<const u32x4*>(Vb + (size_t)tc * D + c * 8);
        }
        float s[kUnroll][GT];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            float kf[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kf[2 * e] = bf16lo(kr[u][e]);
                kf[2 * e + 1] = bf16hi(kr[u][e]);
            }
#pragma unroll
            for (int g = 0; g < GT; ++g) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf(q[g][e], kf[e], d);
#pragma unroll
                for (int ofs = LPR / 2; ofs > 0; ofs >>= 1) d += __shfl_xor(d, ofs, 64);
                if (a.mask_mode == OMX_MASK_BOOL) {
                    if (tok[u] < t_end && !reinterpret_cast<const uint8_t*>(a.mask)[tok[u]]) d = -INFINITY;
                } else if (a.mask_mode == OMX_MASK_ADDITIVE) {
                    if (tok[u] < t_end) d += bf16_to_f32(reinterpret_cast<const bf16_t*>(a.mask)[tok[u]]);
                }
                s[u][g] = tok[u] < t_end ? d : -INFINITY;
            }
        }
        // one running max per head for the whole wave
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            float mx = s[0][g];
#pragma unroll
            for (int u = 1; u < kUnroll; ++u) mx = fmaxf(mx, s[u][g]);
#pragma unroll
            for (int ofs = LPR; ofs < 64; ofs <<= 1) mx = fmaxf(mx, __shfl_xor(mx, ofs, 64));
            const float mn = fmaxf(m[g], mx);
            const float alpha = (mn == -INFINITY) ? 1.f : __expf(m[g] - mn);
            m[g] = mn;
            l[g] *= alpha;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[g][e] *= alpha;
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const float p = (mn == -INFINITY) ? 0.f : __expf(s[u][g] - mn);
                l[g] += p;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[g][2 * e] = fmaf(p, bf16lo(vr[u][e]), o[g][2 * e]);
                    o[g][2 * e + 1] = fmaf(p, bf16hi(vr[u][e]), o[g][2 * e + 1]);
                }
            }
        }
    }

    // ---- merge token sub-groups of the wave (same m): plain sums ----
#pragma unroll
    for (int g = 0; g < GT; ++g) {
#pragma unroll
        for (int ofs = LPR; ofs < 64; ofs <<= 1) {
            l[g] += __shfl_xor(l[g], ofs, 64);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[g][e] += __shfl_xor(o[g][e], ofs, 64);
        }
    }
    // l was accumulated per lane over ITS tokens; every lane of a sub-group saw the same p's, so
    // the LPR lanes of a group hold identical l -- the xor over sub-groups above is the total.
    if (sg == 0) {
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            if (c == 0) {
                sm_m[wave][g] = m[g];
                sm_l[wave][g] = l[g];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) sm_o[wave][g][c * 8 + e] = o[g][e];
        }
    }
    __syncthreads();
    // ---- merge the 4 waves, write the split's partial ----
    for (int idx = threadIdx.x; idx < G * D; idx += kBlock) {
        const int g = idx / D, d = idx % D;
        float M = sm_m[0][g];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) M = fmaxf(M, sm_m[w][g]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const float f = (sm_m[w][g] == -INFINITY) ? 0.f : __expf(sm_m[w][g] - M);
            L = fmaf(f, sm_l[w][g], L);
            O = fmaf(f, sm_o[w][g][d], O);
        }
        const size_t head = (size_t)b * a.H + kvh * G + g;
        a.ws_o[(head * a.nsplit + split) * D + d] = O;
        if (d == 0) {
            a.ws_ml[(head * a.nsplit + split) * 2] = M;
            a.ws_ml[(head * a.nsplit + split) * 2 + 1] = L;
        }
    }
}

// merge splits: out[head, d] = sum_i e^{m_i-M} o_i[d] / sum_i e^{m_i-M} l_i, rounded once to bf16
template <int D>
__global__ __launch_bounds__(D) void attn_combine_kernel(bf16_t* __restrict__ out, const float* __restrict__ ws_o,
                                                         const float* __restrict__ ws_ml, int nsplit) {
    const size_t head = blockIdx.x;
    const int d = threadIdx.x;
    const float* ml = ws_ml + head * nsplit * 2;
    float M = -INFINITY;
    for (int i = 0; i < nsplit; ++i) M = fmaxf(M, ml[2 * i]);
    float L = 0.f, O = 0.f;
    for (int i = 0; i < nsplit; ++i) {
        const float mi = ml[2 * i];
        const float f = (mi == -INFINITY) ? 0.f : __expf(mi - M);
        L = fmaf(f, ml[2 * i + 1], L);
        O = fmaf(f, ws_o[(head * nsplit + i) * D + d], O);
    }
    out[head * D + d] = f32_to_bf16(O / L);
}

}  // namespace

size_t attn_decode_ws_bytes(int BH, int nsplit, int D) { return (size_t)BH * nsplit * (D + 2) * sizeof(float); }

int launch_attn_decode(const AttnDecodeArgs& a, int D, bool fused, hipStream_t s) {
    const int G = a.H / a.Hkv;
    OMX_REQUIRE(a.H % a.Hkv == 0, "sdpa: H=%d not a multiple of Hkv=%d", a.H, a.Hkv);
    OMX_REQUIRE(G >= 1 && G <= 8, "sdpa decode: %d query heads per KV head unsupported (max 8)", G);
    OMX_REQUIRE(a.nsplit >= 1, "sdpa decode: nsplit must be >= 1");
    const dim3 grid(a.B * a.Hkv, a.nsplit), block(kBlock);
    const int gt = G <= 1 ? 1 : G <= 2 ? 2 : G <= 4 ? 4 : 8;
#define OMX_ATTN_CASE(DD, GG)                                                                   \
    if (D == DD && gt == GG) {                                                                  \
        if (fused) attn_decode_kernel<DD, GG, true><<<grid, block, 0, s>>>(a);                  \
        else attn_decode_kernel<DD, GG, false><<<grid, block, 0, s>>>(a);                       \
        OMX_LAUNCH_CHECK();                                                                     \
        attn_combine_kernel<DD><<<a.B * a.H, DD, 0, s>>>(a.out, a.ws_o, a.ws_ml, a.nsplit);     \
        OMX_LAUNCH_CHECK();                                                                     \
        return 0;                                                                               \
    }
    OMX_ATTN_CASE(128, 1) OMX_ATTN_CASE(128, 2) OMX_ATTN_CASE(128, 4) OMX_ATTN_CASE(128, 8)
    OMX_ATTN_CASE(64, 1) OMX_ATTN_CASE(64, 2) OMX_ATTN_CASE(64, 4) OMX_ATTN_CASE(64, 8)
#undef OMX_ATTN_CASE
    return set_error("sdpa decode: head_dim %d unsupported (64 or 128)", D);
}

}  // namespace omx
