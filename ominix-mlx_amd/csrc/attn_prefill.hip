// SDPA with Tq > 1 on the gfx950 matrix cores: flash-attention forward (prefill, DiT joint
// attention, Paraformer-sized encoders).
//   reference: mlx_fast_scaled_dot_product_attention (mlx-c fast.h:189-198; mlx-rs/src/fast.rs:121-151)
//   as called by mlx_rs_core::scaled_dot_product_attention (utils.rs:191-209) with mask = explicit
//   bool array at prefill (utils.rs:134-153, qwen3-mlx/src/model.rs:401), "causal", or none
//   (FLUX joint attention, flux-klein-mlx/src/klein_model.rs:474-483, which the reference computes by
//   materialising the [H,S,S] scores).  Scores never leave registers here.
//
// Mapping (wave64, MFMA 16x16x32 bf16, fp32 accumulate):
//   * block = 4 waves = 64 query rows (16 per wave) of one (batch, head); KV tiles of 64 keys staged
//     in LDS: K row-major with a 16-B-chunk XOR swizzle, V TRANSPOSED ([d][key]) so that both MFMA
//     operands are read as contiguous k-runs;
//   * "swapped" products: S^T = K Q^T and O^T = V^T P^T.  In both C layouts a lane's column is its
//     query row (lane & 15), so the online-softmax state (m, l) and the O rescale are lane-local;
//     row max / sum across the 4 lanes that share a query use v_permlane16/32_swap (no LDS);
//   * P goes from the S^T accumulator straight into the B operand of the second product: the MFMA
//     contraction index is permuted identically on the V^T side, so no transpose of P is needed;
//   * softmax in fp32 (fast.rs:116); P is rounded to bf16 for the second product.
#include "gemm.hpp"

namespace omx {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4v = __attribute__((ext_vector_type(4))) float;
using u32x2v = __attribute__((ext_vector_type(2))) uint32_t;

constexpr int QB = 64;    // query rows per block
constexpr int KB = 64;    // keys per LDS tile
constexpr int VT_STRIDE = KB + 8;   // bf16 elements per V^T row (144 B): conflict-free 8-B fragment reads

struct PrefillArgs {
    const bf16_t *q, *k, *v;
    bf16_t* out;
    int B, H, Hkv, Tq, Tk;
    int64_t kv_batch_stride, kv_head_stride;
    float scale;
    int mask_mode;
    const void* mask;
    // element strides of q and out: [B,H,Tq,D] by default; the engine writes out as [Tq, H*D]
    int64_t q_bs, q_hs, q_ts, o_bs, o_hs, o_ts;
    int64_t kv_ts;   // elements between consecutive key rows (D when K/V are [.., T, D] contiguous)
};

// max / sum over the 4 lanes {l, l^16, l^32, l^48} that hold the same query column
__device__ __forceinline__ float quad_rows_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_rows_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int D>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const PrefillArgs a) {
    constexpr int DC = D / 8;       // 16-B chunks per K row
    constexpr int NI = D / 32;      // MFMA k-steps over the head dim
    constexpr int NDT = D / 16;     // 16-wide output tiles over the head dim
    __shared__ __attribute__((aligned(16))) bf16_t sK[KB * D];
    __shared__ __attribute__((aligned(16))) bf16_t sVt[D * VT_STRIDE];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qcol = lane & 15, rg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kvh = h / (a.H / a.Hkv);
    const int q0 = blockIdx.x * QB;
    const int qrow = q0 + wave * 16 + qcol;              // this lane's query row
    const int qrow_c = min(qrow, a.Tq - 1);
    const int shift = a.Tk - a.Tq;                       // causal: query i sees keys <= i + shift

    const bf16_t* Qp = a.q + (size_t)b * a.q_bs + (size_t)h * a.q_hs + (size_t)qrow_c * a.q_ts;
    const bf16_t* Kb = a.k + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;
    const bf16_t* Vb = a.v + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;

    bf16x8 qf[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) qf[i] = *reinterpret_cast<const bf16x8*>(Qp + i * 32 + rg * 8);

    f32x4v o[NDT];
#pragma unroll
    for (int t = 0; t < NDT; ++t) o[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;   // l_run is this lane's PARTIAL sum (reduced at the end)

    int kv_end = a.Tk;
    if (a.mask_mode == OMX_MASK_CAUSAL) kv_end = max(0, min(a.Tk, q0 + QB + shift));

    for (int k0 = 0; k0 < kv_end; k0 += KB) {
        __syncthreads();   // previous tile fully consumed
        // ---- stage K (swizzled rows) and V^T ----
        for (int ci = threadIdx.x; ci < KB * DC; ci += 256) {
            const int row = ci / DC, ch = ci % DC;
            const int key = min(k0 + row, a.Tk - 1);
            const u32x4 kv = *reinterpret_cast<const u32x4*>(Kb + (size_t)key * a.kv_ts + ch * 8);
            *reinterpret_cast<u32x4*>(&sK[(row * DC + (ch ^ (row & (DC - 1)))) * 8]) = kv;
            const u32x4 vv = *reinterpret_cast<const u32x4*>(Vb + (size_t)key * a.kv_ts + ch * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sVt[(ch * 8 + 2 * e) * VT_STRIDE + row] = (bf16_t)(vv[e] & 0xFFFFu);
                sVt[(ch * 8 + 2 * e + 1) * VT_STRIDE + row] = (bf16_t)(vv[e] >> 16);
            }
        }
        __syncthreads();

        // ---- S^T = K Q^T for 4 key tiles of 16 ----
        f32x4v s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4v{0.f, 0.f, 0.f, 0.f};
            const int row = kt * 16 + qcol;   // A operand: lane & 15 indexes the key row
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int ch = i * 4 + rg;
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&sK[(row * DC + (ch ^ (row & (DC - 1)))) * 8]);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[i], s[kt], 0, 0, 0);
            }
        }
        // ---- scale, mask, online softmax (lane: query qcol, keys kt*16 + rg*4 + r) ----
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + kt * 16 + rg * 4 + r;
                float v = s[kt][r] * a.scale;
                bool keep = key < a.Tk;
                if (a.mask_mode == OMX_MASK_CAUSAL) keep = keep && (key <= qrow + shift);
                else if (a.mask_mode == OMX_MASK_BOOL)
                    keep = keep && reinterpret_cast<const uint8_t*>(a.mask)[(size_t)qrow_c * a.Tk + min(key, a.Tk - 1)];
                else if (a.mask_mode == OMX_MASK_ADDITIVE)
                    v += bf16_to_f32(reinterpret_cast<const bf16_t*>(a.mask)[(size_t)qrow_c * a.Tk + min(key, a.Tk - 1)]);
                v = keep ? v : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = quad_rows_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int t = 0; t < NDT; ++t) o[t] *= alpha;
        bf16x8 pf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sv = s[2 * j + (e >> 2)][e & 3];
                const float p = (m_new == -INFINITY) ? 0.f : __expf(sv - m_new);
                const bf16_t pb = f32_to_bf16(p);
                l_run += bf16_to_f32(pb);   // normaliser of the bf16 probabilities actually multiplied
                pf[j][e] = __builtin_bit_cast(__bf16, pb);
            }
        // ---- O^T += V^T P^T : k-slot (rg*8 + e) <-> key (2j + (e>>2))*16 + rg*4 + (e&3) on BOTH operands ----
#pragma unroll
        for (int t = 0; t < NDT; ++t) {
            const bf16_t* vrow = &sVt[(t * 16 + qcol) * VT_STRIDE];   // A operand: lane & 15 indexes d
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x2v lo = *reinterpret_cast<const u32x2v*>(vrow + (2 * j) * 16 + rg * 4);
                const u32x2v hi = *reinterpret_cast<const u32x2v*>(vrow + (2 * j + 1) * 16 + rg * 4);
                const u32x4 packed = {lo[0], lo[1], hi[0], hi[1]};
                o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, packed), pf[j], o[t], 0, 0, 0);
            }
        }
    }

    // ---- normalise and store: lane holds out[qrow][t*16 + rg*4 + 0..3] ----
    const float l_tot = quad_rows_sum(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qrow < a.Tq) {
        bf16_t* op = a.out + (size_t)b * a.o_bs + (size_t)h * a.o_hs + (size_t)qrow * a.o_ts;
#pragma unroll
        for (int t = 0; t < NDT; ++t) {
            u32x2v w = {pack_bf16(o[t][0] * inv, o[t][1] * inv), pack_bf16(o[t][2] * inv, o[t][3] * inv)};
            *reinterpret_cast<u32x2v*>(op + t * 16 + rg * 4) = w;
        }
    }
}

}  // namespace

int launch_attn_prefill(bf16_t* out, const bf16_t* q, const bf16_t* k, const bf16_t* v, int B, int H, int Hkv, int Tq,
                        int Tk, int D, int64_t kv_batch_stride, int64_t kv_head_stride, float scale, int mask_mode,
                        const void* mask, hipStream_t s, bool out_token_major, const AttnLayout* layout) {
    OMX_REQUIRE(D == 64 || D == 128, "sdpa prefill: head_dim %d unsupported (64 or 128)", D);
    PrefillArgs a = {q, k, v, out, B, H, Hkv, Tq, Tk, kv_batch_stride, kv_head_stride, scale, mask_mode, mask,
                     (int64_t)H * Tq * D, (int64_t)Tq * D, D, (int64_t)H * Tq * D, (int64_t)Tq * D, D, D};
    if (out_token_major) {   // out[b][t][h][d]: what o_proj consumes after the reference's transpose+reshape (model.rs:211-213)
        a.o_hs = D;
        a.o_ts = (int64_t)H * D;
    }
    if (layout) {            // fully strided operands (DiT blocks read q/k/v straight out of a fused projection)
        a.q_bs = layout->q_bs; a.q_hs = layout->q_hs; a.q_ts = layout->q_ts;
        a.o_bs = layout->o_bs; a.o_hs = layout->o_hs; a.o_ts = layout->o_ts;
        a.kv_ts = layout->kv_ts;
    }
    const dim3 grid((Tq + QB - 1) / QB, H, B), block(256);
    if (D == 128) attn_prefill_kernel<128><<<grid, block, 0, s>>>(a);
    else attn_prefill_kernel<64><<<grid, block, 0, s>>>(a);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // namespace omx
