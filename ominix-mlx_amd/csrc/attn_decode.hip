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
//   * FUSED variant (decode engine): the block also applies the per-head q/k RMSNorm and RoPE of
//     qwen3-mlx/src/model.rs:172-194; the lane group whose token is the NEW position builds that
//     K/V row in registers, uses it, and appends it to the cache (cache.rs:183-188) -- no barrier,
//     no fence, and 6 launches per layer less than the per-op sequence.
#include <algorithm>

#include "attn.hpp"
#include "gridsync.hpp"

namespace omx {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = 4;
constexpr int kUnroll = 4;   // token rows per lane-group per step -> 4 K + 4 V loads in flight

// value held by lane (l ^ N/2) of the aligned N-lane group, N = 8 or 16 (RoPE partner i <-> i + D/2)
template <int N>
__device__ __forceinline__ float swap_halves(float v) {
    if (N == 16) return dpp_f<0x128>(v);                       // row_ror:8
    return dpp_f<0x1B>(dpp_f<kDppHalfMirror>(v));              // (7 - i) then quad reverse == i ^ 4
}

__device__ __forceinline__ void unpack8(const u32x4 r, float (&x)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x[2 * e] = bf16lo(r[e]);
        x[2 * e + 1] = bf16hi(r[e]);
    }
}

template <int D, int GT, bool FUSED>
__device__ __forceinline__ void attn_decode_body(const AttnDecodeArgs& a, const int bk, const int split, unsigned char* smem) {
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

    int Tk = a.Tk;
    int pos = -1;
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

    // ---- first K/V step goes out before anything else: it only depends on `pos` ----
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
    float cs[8], sn[8], wk[8];   // FUSED: RoPE row of this position + k_norm weight chunk
    const bool first_half = c < LPR / 2;
    auto norm_rope = [&](const bf16_t* src, const float (&w)[8], float (&out)[8]) {
        float x[8];
        unpack8(*reinterpret_cast<const u32x4*>(src + c * 8), x);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(x[e], x[e], ss);
        ss = group_sum<LPR>(ss);
        const float rstd = a.q_norm_w ? 1.0f / sqrtf(ss / (float)D + a.eps) : 1.0f;   // no q/k norm (Mixtral): x goes to RoPE as it is
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xn = round_bf16(x[e] * rstd * w[e]);            // RMSNorm output is bf16
            const float other = swap_halves<LPR>(xn);                    // element i +- D/2
            const float y = first_half ? xn * cs[e] - other * sn[e] : other * sn[e] + xn * cs[e];
            out[e] = round_bf16(y);                                      // RoPE output is bf16
        }
    };
    if (FUSED) {
        // raw projections: [H*D | Hkv*D | Hkv*D] bf16 (QKV GEMV output)
        const bf16_t* qraw = a.qkv;
        const int i0 = (c % (LPR / 2)) * 8;
        const f32x4* cp = reinterpret_cast<const f32x4*>(a.rope_cos + (size_t)pos * (D / 2) + i0);
        const f32x4* sp = reinterpret_cast<const f32x4*>(a.rope_sin + (size_t)pos * (D / 2) + i0);
        const f32x4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
        float wq[8];
        if (a.q_norm_w) {
            unpack8(*reinterpret_cast<const u32x4*>(a.q_norm_w + c * 8), wq);
            unpack8(*reinterpret_cast<const u32x4*>(a.k_norm_w + c * 8), wk);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) wq[e] = wk[e] = 1.0f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cs[e] = c0[e]; cs[4 + e] = c1[e];
            sn[e] = s0[e]; sn[4 + e] = s1[e];
        }
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            const int h = kvh * G + min(g, G - 1);
            norm_rope(qraw + (size_t)h * D, wq, q[g]);
#pragma unroll
            for (int e = 0; e < 8; ++e) q[g][e] *= a.scale;
        }
    } else {
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            const int h = kvh * G + min(g, G - 1);
            float x[8];
            unpack8(*reinterpret_cast<const u32x4*>(a.q + ((size_t)b * a.H + h) * D + c * 8), x);
#pragma unroll
            for (int e = 0; e < 8; ++e) q[g][e] = x[e] * a.scale;
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

    for (; t0 < t_end; t0 += STEP * kWaves) {
        float s[kUnroll][GT];
        float vf[kUnroll][8];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int tok = t0 + u * TPW + sg;
            float kf[8];
            if (FUSED && tok == pos) {
                // this lane group owns the NEW token: build its K/V row, use it, append it to the cache
                const bf16_t* kraw = a.qkv + (size_t)a.H * D + (size_t)kvh * D;
                const bf16_t* vraw = kraw + (size_t)a.Hkv * D;
                norm_rope(kraw, wk, kf);
                u32x4 kp;
#pragma unroll
                for (int e = 0; e < 4; ++e) kp[e] = pack_bf16(kf[2 * e], kf[2 * e + 1]);
                const u32x4 vp = *reinterpret_cast<const u32x4*>(vraw + c * 8);
                *reinterpret_cast<u32x4*>(const_cast<bf16_t*>(Kb) + (size_t)pos * D + c * 8) = kp;
                *reinterpret_cast<u32x4*>(const_cast<bf16_t*>(Vb) + (size_t)pos * D + c * 8) = vp;
                unpack8(vp, vf[u]);
            } else {
                unpack8(kr[u], kf);
                unpack8(vr[u], vf[u]);
            }
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
        if (a.arrive) {   // read by another block of this launch: write through (sc1), see below
            st_coh_f32(a.ws_o + (head * a.nsplit + split) * D + d, O);
            if (d == 0) {
                st_coh_f32(a.ws_ml + (head * a.nsplit + split) * 2, M);
                st_coh_f32(a.ws_ml + (head * a.nsplit + split) * 2 + 1, L);
            }
        } else {
            a.ws_o[(head * a.nsplit + split) * D + d] = O;
            if (d == 0) {
                a.ws_ml[(head * a.nsplit + split) * 2] = M;
                a.ws_ml[(head * a.nsplit + split) * 2 + 1] = L;
            }
        }
    }
    if (!a.arrive) return;
    // ---- in-launch combine: the block that arrives LAST at its KV head's counter merges all splits of its G heads.
    //      Publication: write-through partial stores, vmcnt(0), block barrier, then ONE relaxed device-scope atomic;
    //      the reader uses sc1 loads (no fences: cdna_hip_programming.md, split-K seam).  Arithmetic and order are
    //      attn_combine_kernel's, so both routes give the same bits. ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* sm_flag = reinterpret_cast<int*>(sm_l + kWaves * GT);
    if (threadIdx.x == 0) {
        unsigned* cnt = a.arrive + (size_t)bk * 16;
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old + 1u == (unsigned)a.nsplit);
        if (last) st_coh32(cnt, 0u);   // ready for the next launch (stream order separates them)
        *sm_flag = last;
    }
    __syncthreads();
    if (!*sm_flag) return;
    float* sm_f = sm_o;                 // [G][nsplit] rescale factors (the merge scratch is free again)
    float* sm_L = sm_m;                 // [G]
    __syncthreads();
    for (int g = wave; g < G; g += kWaves) {   // phase 1: one lane per split
        const float* ml = a.ws_ml + ((size_t)b * a.H + kvh * G + g) * a.nsplit * 2;
        float mloc = -INFINITY;
        for (int i = lane; i < a.nsplit; i += 64) mloc = fmaxf(mloc, ld_coh_f32(ml + 2 * i));
        const float M = wave_max(mloc);
        float lloc = 0.f;
        for (int i = lane; i < a.nsplit; i += 64) {
            const float mi = ld_coh_f32(ml + 2 * i);
            const float f = (mi == -INFINITY) ? 0.f : __expf(mi - M);
            sm_f[g * a.nsplit + i] = f;
            lloc = fmaf(f, ld_coh_f32(ml + 2 * i + 1), lloc);
        }
        const float L = wave_sum(lloc);
        if (lane == 0) sm_L[g] = L;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < G * D; idx += kBlock) {   // phase 2: one thread per (head, d)
        const int g = idx / D, d = idx % D;
        const size_t head = (size_t)b * a.H + kvh * G + g;
        const float* src = a.ws_o + head * a.nsplit * D + d;
        const float* f = sm_f + g * a.nsplit;
        float acc0 = 0.f, acc1 = 0.f;
        int i = 0;
        for (; i + 16 <= a.nsplit; i += 16) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld_coh_f32(src + (size_t)(i + j) * D);
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                acc0 = fmaf(f[i + j], v[j], acc0);
                acc1 = fmaf(f[i + j + 1], v[j + 1], acc1);
            }
        }
        for (; i + 4 <= a.nsplit; i += 4) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ld_coh_f32(src + (size_t)(i + j) * D);
            acc0 = fmaf(f[i], v[0], acc0);
            acc1 = fmaf(f[i + 1], v[1], acc1);
            acc0 = fmaf(f[i + 2], v[2], acc0);
            acc1 = fmaf(f[i + 3], v[3], acc1);
        }
        for (; i < a.nsplit; ++i) acc0 = fmaf(f[i], ld_coh_f32(src + (size_t)i * D), acc0);
        if (a.done) st_coh_bf16(a.out + head * D + d, f32_to_bf16((acc0 + acc1) / sm_L[g]));
        else a.out[head * D + d] = f32_to_bf16((acc0 + acc1) / sm_L[g]);
    }
    if (a.done) {   // this KV head's slice of the attention output is final: publish it to the O-projection blocks
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) st_coh32(a.done + (size_t)bk * 16, *a.seq_ptr);
    }
}

template <int D, int GT, bool FUSED>
__global__ __launch_bounds__(kBlock) void attn_decode_kernel(const AttnDecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    attn_decode_body<D, GT, FUSED>(a, blockIdx.x, blockIdx.y, smem);
}

// ---- fused launch: blocks [0, n_attn) are the attention above; the others own O-projection rows.  They issue their
//      weight rows first (registers: two batches of two rows), wait until every KV head has published its output,
//      read the attention vector with coherent loads and finish with the residual epilogue of gemv_kernel
//      (same per-row arithmetic: identical bits to the separate launch). ----
constexpr unsigned kWaitLimit = 1u << 22;

template <int NV>
__device__ __forceinline__ void oproj_body(const AttnDecodeArgs& a, const OProjArgs& o, const int ob, unsigned char* smem) {
    u32x4* xs = reinterpret_cast<u32x4*>(smem);   // [NV * 64] attention output as packed bf16
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = o.rows_per_wave;
    const int row_begin = (ob * kWaves + wave) * rpw;
    const int row_end = min(row_begin + rpw, o.N);
    u32x4 wA[2][NV], wB[2][NV];
    auto issue = [&](u32x4 (&W)[2][NV], int r0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const u32x4* p = reinterpret_cast<const u32x4*>(o.w + (size_t)min(r0 + r, o.N - 1) * o.K);
#pragma unroll
            for (int j = 0; j < NV; ++j) W[r][j] = __builtin_nontemporal_load(p + j * 64 + lane);
        }
    };
    auto compute = [&](const u32x4 (&W)[2][NV], int r0) {
        float acc[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const u32x4 xp = xs[j * 64 + lane];
            float xf[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { xf[2 * q] = bf16lo(xp[q]); xf[2 * q + 1] = bf16hi(xp[q]); }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[r] = fmaf(bf16lo(W[r][j][i]), xf[2 * i], acc[r]);
                    acc[r] = fmaf(bf16hi(W[r][j][i]), xf[2 * i + 1], acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = r0 + r;
                if (row < row_end) o.out[row] = f32_to_bf16(bf16_to_f32(o.resid[row]) + round_bf16(acc[r]));
            }
        }
    };
    if (row_begin < row_end) issue(wA, row_begin);
    if (row_begin + 2 < row_end) issue(wB, row_begin + 2);
    // wait for every KV head of this launch (bounded: a lost block must not hang the device)
    const unsigned want = *a.seq_ptr;
    const int n_heads = a.B * a.Hkv;
    if ((int)threadIdx.x < n_heads) {
        unsigned it = 0;
        while (ld_coh32(a.done + (size_t)threadIdx.x * 16) != want) {
            __builtin_amdgcn_s_sleep(2);
            if (++it >= kWaitLimit) { st_coh32(o.abort_flag, 1u); break; }
        }
    }
    __syncthreads();
    for (int v = threadIdx.x; v < NV * 64; v += kBlock) xs[v] = ld_coh128(reinterpret_cast<const u32x4*>(a.out) + v);
    __syncthreads();
    for (int r0 = row_begin; r0 < row_end; r0 += 4) {
        compute(wA, r0);
        if (r0 + 2 >= row_end) break;
        if (r0 + 4 < row_end) issue(wA, r0 + 4);
        compute(wB, r0 + 2);
        if (r0 + 6 < row_end) issue(wB, r0 + 6);
    }
}

template <int D, int GT, int NV>
__global__ __launch_bounds__(kBlock) void attn_oproj_kernel(const AttnDecodeArgs a, const OProjArgs o) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int heads = a.B * a.Hkv, n_attn = heads * a.nsplit;
    if ((int)blockIdx.x < n_attn) attn_decode_body<D, GT, true>(a, blockIdx.x % heads, blockIdx.x / heads, smem);
    else oproj_body<NV>(a, o, blockIdx.x - n_attn, smem);
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

namespace {
template <class F>
int with_fused_kernel(int D, int gt, int nv, F&& f) {
#define OMX_AO_CASE(DD, GG, VV) if (D == DD && gt == GG && nv == VV) return f((const void*)attn_oproj_kernel<DD, GG, VV>, \
        ((size_t)kWaves * (64 / (DD / 8)) * GG * DD + 2 * kWaves * GG + 4) * sizeof(float));
    OMX_AO_CASE(128, 4, 8) OMX_AO_CASE(128, 4, 2) OMX_AO_CASE(128, 2, 8) OMX_AO_CASE(128, 2, 4) OMX_AO_CASE(128, 1, 1)
    OMX_AO_CASE(128, 1, 2) OMX_AO_CASE(64, 2, 1) OMX_AO_CASE(64, 4, 1) OMX_AO_CASE(128, 8, 7) OMX_AO_CASE(128, 8, 8)
#undef OMX_AO_CASE
    return -1;
}
}  // namespace

int attn_oproj_capacity(int D, int G, int K) {
    if (K % 512 != 0) return 0;
    const int gt = G <= 1 ? 1 : G <= 2 ? 2 : G <= 4 ? 4 : 8;
    int per_cu = 0;
    const int rc = with_fused_kernel(D, gt, K / 512, [&](const void* fn, size_t shmem) -> int {
        if (shmem > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) return -1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, shmem) != hipSuccess) return -1;
        return 0;
    });
    if (rc != 0) { (void)hipGetLastError(); return 0; }
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return per_cu * cus;
}

int launch_attn_oproj(const AttnDecodeArgs& a, const OProjArgs& o, int D, hipStream_t s) {
    const int G = a.H / a.Hkv;
    OMX_REQUIRE(a.arrive && a.done && a.seq_ptr && o.abort_flag, "fused attention + O projection: missing synchronisation words");
    OMX_REQUIRE(o.K == a.H * D && o.K % 512 == 0, "fused attention + O projection: K=%d", o.K);
    const int gt = G <= 1 ? 1 : G <= 2 ? 2 : G <= 4 ? 4 : 8;
    const int n_attn = a.B * a.Hkv * a.nsplit;
    const int rc = with_fused_kernel(D, gt, o.K / 512, [&](const void* fn, size_t shmem_attn) -> int {
        const size_t shmem = std::max(shmem_attn, (size_t)(o.K / 512) * 64 * 16);
        void* args[] = {(void*)&a, (void*)&o};
        if (hipLaunchKernel(fn, dim3(n_attn + o.n_blocks), dim3(kBlock), args, shmem, s) != hipSuccess) return 1;
        return 0;
    });
    if (rc < 0) return set_error("fused attention + O projection: no kernel for D=%d G=%d K=%d", D, G, o.K);
    if (rc > 0) return set_error("fused attention + O projection: launch failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

int launch_attn_decode(const AttnDecodeArgs& a, int D, bool fused, hipStream_t s) {
    const int G = a.H / a.Hkv;
    OMX_REQUIRE(a.H % a.Hkv == 0, "sdpa: H=%d not a multiple of Hkv=%d", a.H, a.Hkv);
    OMX_REQUIRE(G >= 1 && G <= 8, "sdpa decode: %d query heads per KV head unsupported (max 8)", G);
    OMX_REQUIRE(a.nsplit >= 1 && a.nsplit <= 512, "sdpa decode: nsplit %d out of range (1..512)", a.nsplit);
    const dim3 grid(a.B * a.Hkv, a.nsplit), block(kBlock);
    const int gt = G <= 1 ? 1 : G <= 2 ? 2 : G <= 4 ? 4 : 8;
#define OMX_ATTN_CASE(DD, GG)                                                                           \
    if (D == DD && gt == GG) {                                                                          \
        const size_t shmem = ((size_t)kWaves * (64 / (DD / 8)) * GG * DD + 2 * kWaves * GG + 4) * sizeof(float); \
        if (shmem > 48 * 1024) {                                                                        \
            OMX_HIP_CHECK(hipFuncSetAttribute((const void*)attn_decode_kernel<DD, GG, true>,            \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
            OMX_HIP_CHECK(hipFuncSetAttribute((const void*)attn_decode_kernel<DD, GG, false>,           \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
        }                                                                                               \
        if (fused) attn_decode_kernel<DD, GG, true><<<grid, block, shmem, s>>>(a);                      \
        else attn_decode_kernel<DD, GG, false><<<grid, block, shmem, s>>>(a);                           \
        OMX_LAUNCH_CHECK();                                                                             \
        if (!a.arrive) {                                                                                \
            attn_combine_kernel<DD><<<a.B * a.H, DD, 0, s>>>(a.out, a.ws_o, a.ws_ml, a.nsplit);         \
            OMX_LAUNCH_CHECK();                                                                         \
        }                                                                                               \
        return 0;                                                                                       \
    }
    OMX_ATTN_CASE(128, 1) OMX_ATTN_CASE(128, 2) OMX_ATTN_CASE(128, 4) OMX_ATTN_CASE(128, 8)
    OMX_ATTN_CASE(64, 1) OMX_ATTN_CASE(64, 2) OMX_ATTN_CASE(64, 4) OMX_ATTN_CASE(64, 8)
#undef OMX_ATTN_CASE
    return set_error("sdpa decode: head_dim %d unsupported (64 or 128)", D);
}

}  // namespace omx
