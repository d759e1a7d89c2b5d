// KV-cached decode attention (Tq == 1) for gfx950: split-KV flash-decode.
//   reference: mlx_rs_core::scaled_dot_product_attention (mlx-rs-core/src/utils.rs:191-209) ->
//   mlx_fast_scaled_dot_product_attention (mlx-c fast.h:189-198), whose Tq==1 case MLX serves with
//   a dedicated vector kernel (mlx-rs/src/fast.rs:114).  HBM-bound: 2*Hkv*T*D*2 bytes per layer,
//   but at batch 1 / ctx 2k it is LATENCY that matters (9 MB per layer): the kernel is written to
//   have a short dependent chain, not just coalesced loads.
//
// Layout / mapping (wave64):
//   * K/V rows are D bf16 = D/8 lanes x 16 B; a wave-instruction covers 64/(D/8) consecutive
//     tokens (4 for D=128) as ONE contiguous 1 KiB burst;
//   * the G = H/Hkv query heads that share a KV head are processed together in registers, so
//     each K/V byte is read once per KV head, not once per query head (GQA without tiling,
//     fast.rs:118);
//   * scores: per-lane 8-element partial dot, reduced over the D/8-lane group with DPP row ops
//     (no LDS crossbar); softmax state (m, l) in fp32 (fast.rs:116); a wave keeps ONE running max
//     per head (v_readlane across its token sub-groups) so sub-group partials merge by plain sums;
//   * grid = (B*Hkv) x nsplit; each block writes an un-normalised partial (m, l, o[D]) per head;
//     attn_combine_kernel merges the splits and rounds once to the output dtype;
// The decode ENGINE does not use this kernel: its attention launch (q/k norm + RoPE + cache append + SDPA + split merge with a
// position-independent first load round) is attn_step.hip.
#include <algorithm>

#include "attn.hpp"

namespace omx {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = 4;
constexpr int kUnroll = 4;   // token rows per lane-group per step -> 4 K + 4 V loads in flight

__device__ __forceinline__ void unpack8(const u32x4 r, float (&x)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x[2 * e] = bf16lo(r[e]);
        x[2 * e + 1] = bf16hi(r[e]);
    }
}

template <int D, int GT>
__global__ __launch_bounds__(kBlock) void attn_decode_kernel(const AttnDecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bk = blockIdx.x, split = blockIdx.y;
    constexpr int LPR = D / 8;          // lanes per K/V row
    constexpr int TPW = 64 / LPR;       // tokens per wave-instruction == token sub-groups per wave
    constexpr int STEP = TPW * kUnroll; // tokens per wave per step
    float* sm_o = reinterpret_cast<float*>(smem);                 // [kWaves][TPW][GT][D]
    float* sm_m = sm_o + kWaves * TPW * GT * D;                   // [kWaves][GT]
    float* sm_l = sm_m + kWaves * GT;                             // [kWaves][GT]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane % LPR;           // 8-element chunk of the head dim owned by this lane
    const int sg = lane / LPR;          // token sub-group inside the wave
    const int b = bk / a.Hkv, kvh = bk % a.Hkv;
    const int G = a.H / a.Hkv;
    const int Tk = a.Tk - (a.causal_tail ? a.B - 1 - b : 0);
    // token range of this split: multiples of the block step
    const int per = (Tk + a.nsplit - 1) / a.nsplit;
    const int chunk = ((per + STEP * kWaves - 1) / (STEP * kWaves)) * (STEP * kWaves);
    const int t_begin = split * chunk;
    const int t_end = min(Tk, t_begin + chunk);

    const bf16_t* Kb = a.k + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;
    const bf16_t* Vb = a.v + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;

    // ---- first K/V step goes out before anything else ----
    u32x4 kr[kUnroll], vr[kUnroll];
    int t0 = t_begin + wave * STEP;
    auto issue_kv = [&](int tbase) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int tc = max(min(tbase + u * TPW + sg, t_end - 1), 0);
            kr[u] = *reinterpret_cast<const u32x4*>(Kb + (size_t)tc * D + c * 8);
            vr[u] = *reinterpret_cast<const u32x4*>(Vb + (size_t)tc * D + c * 8);
        }
    };
    if (t0 < t_end) issue_kv(t0);

    // ---- query (G heads) -> registers, pre-multiplied by scale in fp32 ----
    float q[GT][8];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        const int h = kvh * G + min(g, G - 1);
        float x[8];
        unpack8(*reinterpret_cast<const u32x4*>(a.q + (size_t)b * (a.q_bs ? a.q_bs : (int64_t)a.H * D) + (size_t)h * (a.q_hs ? a.q_hs : D) + c * 8), x);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[g][e] = x[e] * a.scale;
    }

    float m[GT], l[GT], o[GT][8];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[g][e] = 0.f;
    }

    for (; t0 < t_end; t0 += STEP * kWaves) {
        float s[kUnroll][GT];
        float vf[kUnroll][8];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int tok = t0 + u * TPW + sg;
            float kf[8];
            unpack8(kr[u], kf);
            unpack8(vr[u], vf[u]);
            if (tok >= t_end) {   // clamped duplicate row: its p is 0, but 0 * garbage must stay 0
#pragma unroll
                for (int e = 0; e < 8; ++e) vf[u][e] = 0.f;
            }
#pragma unroll
            for (int g = 0; g < GT; ++g) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf(q[g][e], kf[e], d);
                d = group_sum<LPR>(d);
                if (a.mask_mode == OMX_MASK_BOOL) {
                    if (tok < t_end && !reinterpret_cast<const uint8_t*>(a.mask)[tok]) d = -INFINITY;
                } else if (a.mask_mode == OMX_MASK_ADDITIVE) {
                    if (tok < t_end) d += bf16_to_f32(reinterpret_cast<const bf16_t*>(a.mask)[tok]);
                }
                s[u][g] = tok < t_end ? d : -INFINITY;
            }
        }
        // next step's loads are independent of the softmax below
        if (t0 + STEP * kWaves < t_end) issue_kv(t0 + STEP * kWaves);
        // one running max per head for the whole wave
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            float mx = s[0][g];
#pragma unroll
            for (int u = 1; u < kUnroll; ++u) mx = fmaxf(mx, s[u][g]);
            float wmx = readlane_f(mx, 0);
#pragma unroll
            for (int r = 1; r < TPW; ++r) wmx = fmaxf(wmx, readlane_f(mx, r * LPR));
            const float mn = fmaxf(m[g], wmx);
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
                for (int e = 0; e < 8; ++e) o[g][e] = fmaf(p, vf[u][e], o[g][e]);
            }
        }
    }

    // ---- every token sub-group parks its partial in LDS (same m inside a wave: plain sums) ----
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        float* dst = sm_o + (((size_t)(wave * TPW + sg) * GT + g) * D + c * 8);
        *reinterpret_cast<f32x4*>(dst) = f32x4{o[g][0], o[g][1], o[g][2], o[g][3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[g][4], o[g][5], o[g][6], o[g][7]};
        // the LPR lanes of a sub-group hold identical l; sum the sub-groups' l by readlane
        float lw = readlane_f(l[g], 0);
#pragma unroll
        for (int r = 1; r < TPW; ++r) lw += readlane_f(l[g], r * LPR);
        if (lane == 0) {
            sm_m[wave * GT + g] = m[g];
            sm_l[wave * GT + g] = lw;
        }
    }
    __syncthreads();
    // ---- merge the 4 waves x TPW sub-groups, write the split's partial ----
    for (int idx = threadIdx.x; idx < G * D; idx += kBlock) {
        const int g = idx / D, d = idx % D;
        float M = sm_m[g];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) M = fmaxf(M, sm_m[w * GT + g]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const float mw = sm_m[w * GT + g];
            const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            float ow = 0.f;
#pragma unroll
            for (int r = 0; r < TPW; ++r) ow += sm_o[((size_t)(w * TPW + r) * GT + g) * D + d];
            L = fmaf(f, sm_l[w * GT + g], L);
            O = fmaf(f, ow, O);
        }
        const size_t head = (size_t)b * a.H + kvh * G + g;
        a.ws_o[(head * a.nsplit + split) * D + d] = O;
        if (d == 0) {
            a.ws_ml[(head * a.nsplit + split) * 2] = M;
            a.ws_ml[(head * a.nsplit + split) * 2 + 1] = L;
        }
    }
}

// merge splits: out[head, d] = sum_i e^{m_i-M} o_i[d] / sum_i e^{m_i-M} l_i, rounded once to bf16.
// Phase 1 is split-parallel (one lane per split, wave reductions), phase 2 is d-parallel with the
// split loop unrolled so its loads pipeline instead of forming a dependent chain.
template <int D>
__global__ __launch_bounds__(D) void attn_combine_kernel(bf16_t* __restrict__ out, const float* __restrict__ ws_o,
                                                         const float* __restrict__ ws_ml, int nsplit) {
    __shared__ float sm_f[512];
    __shared__ float sm_L;
    const size_t head = blockIdx.x;
    const int d = threadIdx.x, lane = threadIdx.x & 63;
    const float* ml = ws_ml + head * nsplit * 2;
    if (threadIdx.x < 64) {
        float mloc = -INFINITY;
        for (int i = lane; i < nsplit; i += 64) mloc = fmaxf(mloc, ml[2 * i]);
        const float M = wave_max(mloc);
        float lloc = 0.f;
        for (int i = lane; i < nsplit; i += 64) {
            const float mi = ml[2 * i];
            const float f = (mi == -INFINITY) ? 0.f : __expf(mi - M);
            sm_f[i] = f;
            lloc = fmaf(f, ml[2 * i + 1], lloc);
        }
        const float L = wave_sum(lloc);
        if (lane == 0) sm_L = L;
    }
    __syncthreads();
    const float* src = ws_o + head * nsplit * D + d;
    float acc0 = 0.f, acc1 = 0.f;
    int i = 0;
    for (; i + 16 <= nsplit; i += 16) {   // 16 independent loads in flight, then the FMAs
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = src[(size_t)(i + j) * D];
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            acc0 = fmaf(sm_f[i + j], v[j], acc0);
            acc1 = fmaf(sm_f[i + j + 1], v[j + 1], acc1);
        }
    }
    for (; i + 4 <= nsplit; i += 4) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = src[(size_t)(i + j) * D];
        acc0 = fmaf(sm_f[i], v[0], acc0);
        acc1 = fmaf(sm_f[i + 1], v[1], acc1);
        acc0 = fmaf(sm_f[i + 2], v[2], acc0);
        acc1 = fmaf(sm_f[i + 3], v[3], acc1);
    }
    for (; i < nsplit; ++i) acc0 = fmaf(sm_f[i], src[(size_t)i * D], acc0);
    out[head * D + d] = f32_to_bf16((acc0 + acc1) / sm_L);
}

}  // namespace

size_t attn_decode_ws_bytes(int BH, int nsplit, int D) { return (size_t)BH * nsplit * (D + 2) * sizeof(float); }

int launch_attn_decode(const AttnDecodeArgs& a, int D, hipStream_t s) {
    const int G = a.H / a.Hkv;
    OMX_REQUIRE(a.H % a.Hkv == 0, "sdpa: H=%d not a multiple of Hkv=%d", a.H, a.Hkv);
    OMX_REQUIRE(G >= 1 && G <= 8, "sdpa decode: %d query heads per KV head unsupported (max 8)", G);
    OMX_REQUIRE(a.nsplit >= 1 && a.nsplit <= 512, "sdpa decode: nsplit %d out of range (1..512)", a.nsplit);
    const dim3 grid(a.B * a.Hkv, a.nsplit), block(kBlock);
    const int gt = G <= 1 ? 1 : G <= 2 ? 2 : G <= 4 ? 4 : 8;
#define OMX_ATTN_CASE(DD, GG)                                                                           \
    if (D == DD && gt == GG) {                                                                          \
        const size_t shmem = ((size_t)kWaves * (64 / (DD / 8)) * GG * DD + 2 * kWaves * GG + 4) * sizeof(float); \
        if (shmem > 48 * 1024)                                                                          \
            OMX_HIP_CHECK(hipFuncSetAttribute((const void*)attn_decode_kernel<DD, GG>,                  \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
        attn_decode_kernel<DD, GG><<<grid, block, shmem, s>>>(a);                                       \
        OMX_LAUNCH_CHECK();                                                                             \
        attn_combine_kernel<DD><<<a.B * a.H, DD, 0, s>>>(a.out, a.ws_o, a.ws_ml, a.nsplit);             \
        OMX_LAUNCH_CHECK();                                                                             \
        return 0;                                                                                       \
    }
    OMX_ATTN_CASE(128, 1) OMX_ATTN_CASE(128, 2) OMX_ATTN_CASE(128, 4) OMX_ATTN_CASE(128, 8)
    OMX_ATTN_CASE(64, 1) OMX_ATTN_CASE(64, 2) OMX_ATTN_CASE(64, 4) OMX_ATTN_CASE(64, 8)
#undef OMX_ATTN_CASE
    return set_error("sdpa decode: head_dim %d unsupported (64 or 128)", D);
}

}  // namespace omx
