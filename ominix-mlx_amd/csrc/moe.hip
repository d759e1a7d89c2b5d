// Sparse mixture-of-experts block: router + top-k + SwitchGLU + weighted sum.
//   reference: MixtralSparseMoeBlock::forward  mixtral-mlx/src/model.rs:296-308
//              MoeBlock::forward (Qwen3-MoE)   qwen3-mlx/src/qwen3_moe.rs:475-503
//              SwitchGLU::forward_experts      mixtral-mlx/src/model.rs:243-274  (gather_qmm x3 + fused_swiglu,
//                                              rows sorted by expert when B*L*k >= 64)
// The reference issues ~12 lazy ops per block (argpartition, take_along_axis, softmax, argsort x2,
// floor_divide, take x3, gather_qmm x3, fused_swiglu, multiply, sum).  Here:
//   moe_router_kernel    gate GEMV + top-k + softmax per token in one block (logits in bf16, softmax fp32)
//   decode  (N*k <= 32)  two expert-selected batched GEMV launches (gate/up + fused_swiglu epilogue, down)
//   prefill              counting sort by expert on the device (moe_plan_kernel) -> grouped MFMA GEMM with row
//                        gather (no materialised x_sorted) -> fused_swiglu -> grouped GEMM
//   moe_combine_kernel   y * score summed over k, un-sorting through the inverse permutation
// Expert weights are dense bf16 [E, out, in] (the bf16 build of BASELINE config 3; the reference's 4-bit
// gather_qmm is a SURVEY 8f "next" row).
#include "gemm.hpp"
#include "gemv.hpp"
#include "quant.hpp"
#include "workspace.hpp"
#include "moe_route.hpp"

namespace omx {
namespace {


// One block (16 waves) per token.  Phase 1: the gate Linear -- wave w owns experts w, w+16, ..., four at a time so that
// their row loads are in flight together.  Phase 2 (wave 0): softmax / top-k / renormalisation with the experts spread
// over the lanes (E <= 256 -> four per lane): wave reductions instead of a serial scan by one thread (113 us -> a few us
// at 128 experts).  Ties go to the lower expert index, selections come out in descending score order.
constexpr int kRouterThreads = 1024;
constexpr int kRouterXnMax = 8192;   // widest hidden size whose normalised row the few-experts path keeps in LDS
static int router_threads(int n_experts, int hidden) { return (n_experts <= 8 && hidden <= 4096 && (hidden & 511) == 0) ? 512 : kRouterThreads; }
__global__ __launch_bounds__(kRouterThreads) void moe_router_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gate_w,
                                                                    int h, int E, int k, int mode, int renorm,
                                                                    uint32_t* __restrict__ inds, bf16_t* __restrict__ scores,
                                                                    const bf16_t* __restrict__ norm_w = nullptr, float eps = 0.f,
                                                                    bf16_t* __restrict__ xn_out = nullptr) {
    __shared__ float s_logit[kMaxExperts];
    __shared__ float s_red[kRouterThreads / 64];
    __shared__ __attribute__((aligned(16))) bf16_t s_xn[kRouterXnMax];
    // (launched with kRouterThreads, or with 512 threads for a few experts over a row of <= 4096: half the waves to start and to meet at
    //  the barriers, the same sums -- the missing waves' partials are zeros)
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = (int)blockDim.x / 64;
    const bf16_t* xr = x + (size_t)t * h;
    // few experts (Mixtral: 8): wave e owns expert e and holds its WHOLE gate row in registers, loaded before anything depends on the
    // activation; the normalised row then goes through LDS instead of a global round trip.  Same per-lane fma chains and wave sums as
    // the loop below (element it * 512 + lane * 8 in iteration it), so the logits are bit-identical.  (Decode: this launch is a
    // chain of dependent round trips, 9.9 us per layer at Mixtral-8x7B shapes before, 32 layers per token.)
    const bool few = norm_w != nullptr && E <= n_waves && h <= kRouterXnMax && h <= (int)blockDim.x * 8 && (h & 511) == 0;
    u32x4 gw[kRouterXnMax / 512];
    if (few && wave < E) {
#pragma unroll
        for (int it = 0; it < kRouterXnMax / 512; ++it)
            if (it * 512 < h) gw[it] = *reinterpret_cast<const u32x4*>(gate_w + (size_t)wave * h + it * 512 + lane * 8);
    }
    if (norm_w && few) {   // (h <= 8192: one vector per thread -- the row and the norm weights are read ONCE, before the reduction)
        const int i = threadIdx.x * 8;
        u32x4 a = {0u, 0u, 0u, 0u}, w = {0u, 0u, 0u, 0u};
        if (i < h) {
            a = *reinterpret_cast<const u32x4*>(xr + i);
            w = *reinterpret_cast<const u32x4*>(norm_w + i);
        }
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { ss = fmaf(bf16lo(a[q]), bf16lo(a[q]), ss); ss = fmaf(bf16hi(a[q]), bf16hi(a[q]), ss); }
        ss = wave_sum(ss);
        if (lane == 0) s_red[wave] = ss;
        __syncthreads();
        float tot = 0.f;
        for (int wv = 0; wv < n_waves; ++wv) tot += s_red[wv];
        const float rstd = 1.0f / sqrtf(tot / (float)h + eps);
        if (i < h) {
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pack_bf16(bf16lo(a[q]) * rstd * bf16lo(w[q]), bf16hi(a[q]) * rstd * bf16hi(w[q]));
            *reinterpret_cast<u32x4*>(xn_out + (size_t)t * h + i) = o;
            *reinterpret_cast<u32x4*>(s_xn + i) = o;
        }
        __syncthreads();
    } else if (norm_w) {   // post-attention RMSNorm of the decoder block folded in: xn = bf16(x * rstd * w), written for the experts
        float ss = 0.f;
        for (int i = threadIdx.x * 8; i < h; i += (int)blockDim.x * 8) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(xr + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) { ss = fmaf(bf16lo(a[q]), bf16lo(a[q]), ss); ss = fmaf(bf16hi(a[q]), bf16hi(a[q]), ss); }
        }
        ss = wave_sum(ss);
        if (lane == 0) s_red[wave] = ss;
        __syncthreads();
        float tot = 0.f;
        for (int w = 0; w < n_waves; ++w) tot += s_red[w];
        const float rstd = 1.0f / sqrtf(tot / (float)h + eps);
        bf16_t* xo = xn_out + (size_t)t * h;
        for (int i = threadIdx.x * 8; i < h; i += (int)blockDim.x * 8) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(xr + i);
            const u32x4 w = *reinterpret_cast<const u32x4*>(norm_w + i);
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pack_bf16(bf16lo(a[q]) * rstd * bf16lo(w[q]), bf16hi(a[q]) * rstd * bf16hi(w[q]));
            *reinterpret_cast<u32x4*>(xo + i) = o;
            if (few) *reinterpret_cast<u32x4*>(s_xn + i) = o;
        }
        __syncthreads();   // this block reads its own xn row back below (same CU: L1/L2 coherent within the block after the barrier)
        if (!few) __threadfence_block();
        xr = xo;
    }
    if (few) {
        if (wave < E) {
            float acc = 0.f;
#pragma unroll
            for (int it = 0; it < kRouterXnMax / 512; ++it) {
                if (it * 512 >= h) break;
                const u32x4 a = *reinterpret_cast<const u32x4*>(s_xn + it * 512 + lane * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc = fmaf(bf16lo(a[q]), bf16lo(gw[it][q]), acc);
                    acc = fmaf(bf16hi(a[q]), bf16hi(gw[it][q]), acc);
                }
            }
            const float v = wave_sum(acc);
            if (lane == 0) s_logit[wave] = round_bf16(v);
        }
        __syncthreads();
        if (wave != 0) return;
        route_from_logits(s_logit, t, lane, E, k, mode, renorm, inds, scores);
        return;
    }
    for (int e0 = wave; e0 < E; e0 += 4 * n_waves) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = lane * 8; i < h; i += 64 * 8) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(xr + i);
            u32x4 b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const u32x4*>(gate_w + (size_t)min(e0 + u * n_waves, E - 1) * h + i);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[u] = fmaf(bf16lo(a[q]), bf16lo(b[u][q]), acc[u]);
                    acc[u] = fmaf(bf16hi(a[q]), bf16hi(b[u][q]), acc[u]);
                }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float v = wave_sum(acc[u]);
            const int e = e0 + u * n_waves;
            if (lane == 0 && e < E) s_logit[e] = round_bf16(v);   // the gate Linear's output is a bf16 array
        }
    }
    __syncthreads();
    if (wave != 0) return;
    route_from_logits(s_logit, t, lane, E, k, mode, renorm, inds, scores);
}

// logits [n_tokens, E] already computed (quantised router: qgemv on the packed gate): selection only, one wave per token
template <bool F16 = false>
__global__ __launch_bounds__(64) void moe_route_logits_kernel(const bf16_t* __restrict__ logits, int E, int k, int mode, int renorm,
                                                              uint32_t* __restrict__ inds, bf16_t* __restrict__ scores) {
    __shared__ float s_logit[kMaxExperts];
    const int t = blockIdx.x, lane = threadIdx.x;
    for (int e = lane; e < E; e += 64) s_logit[e] = Act16<F16>::val(logits[(size_t)t * E + e]);
    __syncthreads();
    route_from_logits<F16>(s_logit, t, lane, E, k, mode, renorm, inds, scores);
}

// counting sort of the N*k (token, slot) pairs by expert + tile table of the grouped GEMM (single block)
__global__ __launch_bounds__(1024) void moe_plan_kernel(const uint32_t* __restrict__ inds, int n_slots, int E, int k,
                                                        int* __restrict__ seg_start, uint32_t* __restrict__ row_src,
                                                        uint32_t* __restrict__ pos_of_slot, int* __restrict__ tile_expert,
                                                        int* __restrict__ tile_m0, int* __restrict__ n_tiles, int tile_rows,
                                                        int n_tile_experts = -1) {
    // n_tile_experts >= 0: only experts [0, n_tile_experts) get GEMM tiles -- the expert-parallel batched form sorts the slots routed
    // to OTHER ranks into one trailing pseudo-expert that nobody multiplies
    __shared__ int s_cnt[kMaxExperts + 1], s_start[kMaxExperts + 2], s_fill[kMaxExperts + 1];   // (+1: the trailing pseudo-expert of an expert-parallel shard)
    for (int e = threadIdx.x; e < E; e += blockDim.x) { s_cnt[e] = 0; s_fill[e] = 0; }
    __syncthreads();
    for (int i = threadIdx.x; i < n_slots; i += blockDim.x) atomicAdd(&s_cnt[inds[i]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0, tiles = 0;
        for (int e = 0; e < E; ++e) {
            s_start[e] = acc;
            if (n_tile_experts < 0 || e < n_tile_experts)
                for (int m0 = 0; m0 < s_cnt[e]; m0 += tile_rows) { tile_expert[tiles] = e; tile_m0[tiles] = m0; ++tiles; }
            acc += s_cnt[e];
        }
        s_start[E] = acc;
        *n_tiles = tiles;
    }
    __syncthreads();
    for (int e = threadIdx.x; e <= E; e += blockDim.x) seg_start[e] = s_start[e];
    // order inside an expert segment is arbitrary (atomics): every row is computed independently and
    // un-sorted again through pos_of_slot, so the result does not depend on it
    for (int i = threadIdx.x; i < n_slots; i += blockDim.x) {
        const int e = (int)inds[i];
        const int p = s_start[e] + atomicAdd(&s_fill[e], 1);
        row_src[p] = (uint32_t)(i / k);      // x_sorted = x_flat[order // k]  (model.rs:214-216)
        pos_of_slot[i] = (uint32_t)p;
    }
}

// Row-tile height of the grouped GEMMs: 256-row tiles (the deep-pipelined kernel, gate/up/fused_swiglu in ONE launch) once an
// expert sees 256 rows on average -- the 128 expected padding rows per expert then cost less than the kernel gains;
// 128-row tiles otherwise (many small experts).  OMX_MOE_TILE=128|256 forces one.
int moe_tile_rows(int rows, int n_experts, int hidden, int inter) {
    const char* env = getenv("OMX_MOE_TILE");
    const int forced = env ? atoi(env) : 0;
    const bool ok = hidden % 64 == 0 && inter % 64 == 0;
    if (forced == 256 && ok) return 256;
    if (forced == 128) return 128;
    return ok && rows / n_experts >= 256 ? 256 : 128;
}

// SwitchGLU over expert-sorted rows with 256-row tiles: [gate | up] GEMM with fused_swiglu(up, gate) in the epilogue, then down
int grouped_glu_256(bf16_t* ybuf, bf16_t* gbuf, const bf16_t* x, const bf16_t* w_gate, const bf16_t* w_up, const bf16_t* w_down,
                    int rows, int hidden, int inter, int n_experts, GroupedDesc g, hipStream_t s) {
    const int mt = rows / 256 + n_experts + 1;
    GemmSegs gu = {};
    gu.w_gate = w_gate; gu.w_up = w_up; gu.out_act = gbuf; gu.half = inter; gu.ld_act = inter; gu.act_mode = 0;
    g.w_estride = (size_t)inter * hidden;
    if (launch_gemm_bf16_segmented_grouped(x, rows, hidden, gu, g, mt, s)) return 1;
    g.row_src = nullptr;   // activations are already in expert-sorted order
    g.w_estride = (size_t)hidden * inter;
    GemmSegs dn = {};
    dn.n_plain = 1;
    dn.plain[0] = {w_down, nullptr, ybuf, hidden, hidden, 0};
    return launch_gemm_bf16_segmented_grouped(gbuf, rows, inter, dn, g, mt, s);
}

// out[t] = bf16( sum_j bf16( y[pos(t,j)] * score[t,j] ) )     (model.rs:304-307)
template <bool F16 = false>
__global__ __launch_bounds__(256) void moe_combine_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ y,
                                                          const bf16_t* __restrict__ scores,
                                                          const uint32_t* __restrict__ pos_of_slot, int h, int k,
                                                          const bf16_t* __restrict__ resid = nullptr) {
    typedef Act16<F16> A16;
    const int t = blockIdx.x;
    for (int i = threadIdx.x * 8; i < h; i += 256 * 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < k; ++j) {
            const size_t slot = (size_t)t * k + j;
            const size_t p = pos_of_slot ? pos_of_slot[slot] : slot;
            const float sc = A16::val(scores[slot]);
            const u32x4 v = *reinterpret_cast<const u32x4*>(y + p * h + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[2 * q] += A16::rnd(A16::lo(v[q]) * sc);
                acc[2 * q + 1] += A16::rnd(A16::hi(v[q]) * sc);
            }
        }
        u32x4 o;
        if (resid) {   // decoder block: h + moe(h_normed), the block output rounded first (model.rs:343-344)
            const u32x4 r = *reinterpret_cast<const u32x4*>(resid + (size_t)t * h + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = A16::pack(A16::lo(r[q]) + A16::rnd(acc[2 * q]), A16::hi(r[q]) + A16::rnd(acc[2 * q + 1]));
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = A16::pack(acc[2 * q], acc[2 * q + 1]);
        }
        *reinterpret_cast<u32x4*>(out + (size_t)t * h + i) = o;
    }
}

// y_rows[i] = y_sorted[pos_of_slot[i]]  (scatter_unsort, model.rs:226-233)
__global__ __launch_bounds__(256) void moe_unsort_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ y,
                                                         const uint32_t* __restrict__ pos_of_slot, int h) {
    const size_t src = pos_of_slot[blockIdx.x];
    for (int i = threadIdx.x * 8; i < h; i += 256 * 8)
        *reinterpret_cast<u32x4*>(out + (size_t)blockIdx.x * h + i) = *reinterpret_cast<const u32x4*>(y + src * h + i);
}


// expert-parallel partial of the weighted sum: out[t] (f32) = sum over the slots whose expert lives on this rank of
// bf16(y * score) -- the ranks' partials add up (all-reduce) to the single-device sum before its rounding
template <bool F16 = false>
__global__ __launch_bounds__(256) void moe_combine_partial_kernel(float* __restrict__ out, const bf16_t* __restrict__ y,
                                                                  const bf16_t* __restrict__ scores, const uint32_t* __restrict__ inds,
                                                                  int h, int k, int e_lo, int e_n,
                                                                  const uint32_t* __restrict__ pos_of_slot = nullptr) {
    typedef Act16<F16> A16;      // (F16: a float16 checkpoint's slot outputs and scores, products rounded to float16)
    const int t = blockIdx.x;
    for (int i = threadIdx.x * 8; i < h; i += 256 * 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < k; ++j) {
            const size_t slot = (size_t)t * k + j;
            const int e = (int)inds[slot];
            if (e < e_lo || e >= e_lo + e_n) continue;
            const float sc = A16::val(scores[slot]);
            const size_t p = pos_of_slot ? pos_of_slot[slot] : slot;     // batched form: the row's place in the expert-sorted order
            const u32x4 v = *reinterpret_cast<const u32x4*>(y + p * h + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[2 * q] += A16::rnd(A16::lo(v[q]) * sc);
                acc[2 * q + 1] += A16::rnd(A16::hi(v[q]) * sc);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) out[(size_t)t * h + i + q] = acc[q];
    }
}

// expert-parallel batched form: expert ids relative to this rank's shard, slots routed elsewhere -> the pseudo-expert e_n
__global__ void moe_localize_kernel(uint32_t* __restrict__ local, const uint32_t* __restrict__ inds, int n, int e_lo, int e_n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)inds[i];
    local[i] = (e >= e_lo && e < e_lo + e_n) ? (uint32_t)(e - e_lo) : (uint32_t)e_n;
}

}  // namespace
}  // namespace omx

extern "C" int omx_moe_workspace_bytes(int n_tokens, int hidden, int inter, int n_experts, int top_k, size_t* bytes) {
    OMX_REQUIRE(bytes, "omx_moe_workspace_bytes: null argument");
    const size_t slots = (size_t)n_tokens * top_k;
    const size_t tiles = slots / 128 + n_experts + 1;
    *bytes = slots * 4 * 3 + slots * 2 + (size_t)(n_experts + 2) * 4 + tiles * 8 + 4096 +   // plan
             slots * (size_t)inter * 2 * 2 + slots * (size_t)hidden * 2;                     // gate/up (act in place), y
    return 0;
}

namespace {
// quantized expert stacks (mixtral-mlx/src/model.rs:182-201 QuantizedSwitchLinear): packed [E, out, in*bits/32] u32,
// scales / biases [E, out, in/group]
struct QExperts {
    omx::QMat gate, up, down;
    int group, bits;
};
omx::bf16_t* g_dq = nullptr;      // dequantised expert stacks for the grouped-GEMM (many tokens) route
size_t g_dq_cap = 0;
}  // namespace

struct BlockFusion {            // decoder-block glue folded into the MoE launches (engine): out = resid + moe(rmsnorm(x))
    const void* norm_w = nullptr;
    float eps = 0.f;
    void* xn = nullptr;         // [n_tokens, hidden] scratch receiving the normalised rows
    const void* resid = nullptr;
    float* partials = nullptr;  // decode (bf16 experts): [top_k, hidden] f32 receiving bf16(bf16(y_j) * score_j) from the down GEMVs'
                                // epilogue INSTEAD of the weighted-sum launch; the caller's next GEMV folds them into the residual
};

static int moe_forward_impl(void* out, const void* x, const void* gate_w, const void* w_gate, const void* w_up,
                            const void* w_down, const QExperts* q, int n_tokens, int hidden, int inter, int n_experts, int top_k,
                            int mode, int norm_topk_prob, uint32_t* inds_out, void* scores_out, omx_stream stream,
                            const BlockFusion* bf = nullptr) {
    using namespace omx;
    OMX_REQUIRE(out && x && gate_w && (q || (w_gate && w_up && w_down)), "omx_moe_forward: null tensor");
    OMX_REQUIRE(n_tokens >= 0 && hidden > 0 && inter > 0, "omx_moe_forward: bad shape");
    OMX_REQUIRE(n_experts >= 1 && n_experts <= kMaxExperts && top_k >= 1 && top_k <= kMaxTopK && top_k <= n_experts,
                "omx_moe_forward: n_experts=%d (max %d), top_k=%d (max %d)", n_experts, kMaxExperts, top_k, kMaxTopK);
    OMX_REQUIRE(hidden % 64 == 0 && inter % 64 == 0, "omx_moe_forward: hidden=%d and intermediate=%d must be multiples of 64", hidden, inter);
    OMX_REQUIRE(mode == 0 || mode == 1, "omx_moe_forward: mode must be 0 (Mixtral) or 1 (Qwen3-MoE)");
    if (n_tokens == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int slots = n_tokens * top_k;
    const int max_tiles = slots / 128 + n_experts + 1;
    size_t need = 0;
    omx_moe_workspace_bytes(n_tokens, hidden, inter, n_experts, top_k, &need);
    void* ws = nullptr;
    if (get_workspace(&ws, need)) return 1;
    char* p = (char*)ws;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    uint32_t* inds = (uint32_t*)take((size_t)slots * 4);
    uint32_t* row_src = (uint32_t*)take((size_t)slots * 4);
    uint32_t* pos_of_slot = (uint32_t*)take((size_t)slots * 4);
    bf16_t* scores = (bf16_t*)take((size_t)slots * 2);
    int* seg_start = (int*)take((size_t)(n_experts + 2) * 4);
    int* tile_expert = (int*)take((size_t)max_tiles * 4);
    int* tile_m0 = (int*)take((size_t)max_tiles * 4);
    int* n_tiles = (int*)take(256);
    bf16_t* gbuf = (bf16_t*)take((size_t)slots * inter * 2);
    bf16_t* ubuf = (bf16_t*)take((size_t)slots * inter * 2);
    bf16_t* ybuf = (bf16_t*)take((size_t)slots * hidden * 2);

    const bool decode = slots <= 32 && gemv_k_supported(hidden, false) && gemv_k_supported(inter, false);
    // one token of a decoder block, bf16 experts, few experts, OMX_MOE_ROUTE_FUSED=1 (opt-in): NO router launch -- every block of the
    // experts' gate/up GEMV normalises and routes the row itself (gemv.hip PRO_ROUTE: the router kernel's arithmetic, bit-identical),
    // block (0, 0) leaves inds / scores for the down projection.  Measured SLOWER at Mixtral-8x7B shapes: 198.8 tok/s with the usual
    // row groups (3.5 rounds of blocks, each paying the routing prologue before its first weight byte), 207.9 with one round of long
    // row groups, against 209.7 with the 8.5 us router launch -- the prologue cannot hide behind a first weight batch any more.
    const char* rf_env = getenv("OMX_MOE_ROUTE_FUSED");
    const bool route_fused = decode && !q && bf && bf->norm_w && n_tokens == 1 && gemv_route_supported(hidden, n_experts, top_k) &&
                             router_threads(n_experts, hidden) == 512 && rf_env && rf_env[0] == '1';
    const void* x_raw = x;
    if (!route_fused) {
        moe_router_kernel<<<n_tokens, router_threads(n_experts, hidden), 0, s>>>((const bf16_t*)x, (const bf16_t*)gate_w, hidden, n_experts, top_k, mode,
                                                              norm_topk_prob, inds, scores, bf ? (const bf16_t*)bf->norm_w : nullptr,
                                                              bf ? bf->eps : 0.f, bf ? (bf16_t*)bf->xn : nullptr);
        OMX_LAUNCH_CHECK();
        if (bf && bf->norm_w) x = bf->xn;               // the experts read the normalised rows
    }
    const bf16_t* resid = bf ? (const bf16_t*)bf->resid : nullptr;
    if (decode && q) {
        // gather_qmm x3 on the PACKED weights (model.rs:262-272 unsorted branch): expert-selected batched quantised GEMVs
        QGemvArgs a = {};
        a.m[0] = q->gate; a.m[1] = q->up; a.N = inter; a.K = hidden; a.group = q->group;
        a.x = (const bf16_t*)x; a.out = gbuf;
        a.n_batch = slots; a.x_div = top_k; a.w_sel = inds;
        a.w_estride = (size_t)inter * hidden * q->bits / 32; a.s_estride = (size_t)inter * (hidden / q->group);
        a.swiglu_single_round = 1;
        if (launch_qgemv(a, q->bits, PRO_NONE, EPI_SWIGLU, s)) return 1;
        QGemvArgs d = {};
        d.m[0] = q->down; d.N = hidden; d.K = inter; d.group = q->group;
        d.x = gbuf; d.out = ybuf;
        d.n_batch = slots; d.x_div = 1; d.w_sel = inds;
        d.w_estride = (size_t)hidden * inter * q->bits / 32; d.s_estride = (size_t)hidden * (inter / q->group);
        if (launch_qgemv(d, q->bits, PRO_NONE, EPI_STORE, s)) return 1;
        moe_combine_kernel<false><<<n_tokens, 256, 0, s>>>((bf16_t*)out, ybuf, scores, nullptr, hidden, top_k, resid);
        OMX_LAUNCH_CHECK();
        if (inds_out) OMX_HIP_CHECK(hipMemcpyAsync(inds_out, inds, (size_t)slots * 4, hipMemcpyDeviceToDevice, s));
        if (scores_out) OMX_HIP_CHECK(hipMemcpyAsync(scores_out, scores, (size_t)slots * 2, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    if (q) {
        // many tokens: dequantise the three stacks once into a scratch (what MLX's qmm does per tile) and take the
        // grouped MFMA route on bf16 weights
        const size_t per = (size_t)n_experts * inter * hidden;
        if (3 * per > g_dq_cap) {
            OMX_HIP_CHECK(hipStreamSynchronize(s));
            if (g_dq) OMX_HIP_CHECK(hipFree(g_dq));
            OMX_HIP_CHECK(hipMalloc((void**)&g_dq, 3 * per * 2));
            g_dq_cap = 3 * per;
        }
        if (omx_dequantize(g_dq, q->gate.w, q->gate.scales, q->gate.biases, (int64_t)n_experts * inter, hidden, q->group, q->bits, OMX_BFLOAT16, stream) ||
            omx_dequantize(g_dq + per, q->up.w, q->up.scales, q->up.biases, (int64_t)n_experts * inter, hidden, q->group, q->bits, OMX_BFLOAT16, stream) ||
            omx_dequantize(g_dq + 2 * per, q->down.w, q->down.scales, q->down.biases, (int64_t)n_experts * hidden, inter, q->group, q->bits, OMX_BFLOAT16, stream))
            return 1;
        w_gate = g_dq; w_up = g_dq + per; w_down = g_dq + 2 * per;
    }
    if (decode) {
        // SwitchGLU without the sort (model.rs:262-272): expert-selected batched GEMVs
        GemvArgs a = {};
        a.w0 = (const bf16_t*)w_gate; a.w1 = (const bf16_t*)w_up; a.n0 = inter; a.N = inter; a.K = hidden;
        a.x = (const bf16_t*)x; a.out = gbuf;
        a.n_batch = slots; a.x_div = top_k; a.x_bstride = hidden; a.out_bstride_bytes = (size_t)inter * 2;
        a.w_sel = inds; a.w_estride = (size_t)inter * hidden; a.swiglu_single_round = 1;
        if (route_fused) {
            a.x = (const bf16_t*)x_raw; a.w_sel = nullptr;
            a.norm_w = (const bf16_t*)bf->norm_w; a.eps = bf->eps;
            a.route_gate = (const bf16_t*)gate_w; a.route_E = n_experts; a.route_k = top_k; a.route_mode = mode; a.route_renorm = norm_topk_prob;
            a.route_inds = inds; a.route_scores = scores;
            // ONE round of blocks (every block pays the routing prologue before its first weight byte moves, and nothing hides it):
            // row groups sized so that slots x blocks fit the chip's 2 x 256 resident blocks
            a.rows_per_wave = std::max(4, (inter * slots + 4 * 512 - 1) / (4 * 512));
            if (const char* e = getenv("OMX_MOE_ROUTE_RPW")) a.rows_per_wave = atoi(e);
            if (launch_gemv(a, PRO_ROUTE, EPI_SWIGLU, s)) return 1;
        } else if (launch_gemv(a, PRO_NONE, EPI_SWIGLU, s)) return 1;
        GemvArgs d = {};
        d.w0 = (const bf16_t*)w_down; d.n0 = hidden; d.N = hidden; d.K = inter;
        d.x = gbuf; d.out = ybuf;
        d.n_batch = slots; d.x_div = 1; d.x_bstride = inter; d.out_bstride_bytes = (size_t)hidden * 2;
        d.w_sel = inds; d.w_estride = (size_t)hidden * inter;
        if (bf && bf->partials && n_tokens == 1) {
            d.out = bf->partials; d.out_bstride_bytes = (size_t)hidden * 4; d.out_scale = scores;
            if (launch_gemv(d, PRO_NONE, EPI_F32, s)) return 1;
            return 0;
        }
        if (launch_gemv(d, PRO_NONE, EPI_STORE, s)) return 1;
        moe_combine_kernel<false><<<n_tokens, 256, 0, s>>>((bf16_t*)out, ybuf, scores, nullptr, hidden, top_k, resid);
        OMX_LAUNCH_CHECK();
    } else {
        const int tile_rows = moe_tile_rows(slots, n_experts, hidden, inter);
        moe_plan_kernel<<<1, 1024, 0, s>>>(inds, slots, n_experts, top_k, seg_start, row_src, pos_of_slot, tile_expert,
                                           tile_m0, n_tiles, tile_rows);
        OMX_LAUNCH_CHECK();
        GroupedDesc g;
        g.tile_expert = tile_expert; g.tile_m0 = tile_m0; g.seg_start = seg_start; g.n_tiles = n_tiles;
        g.row_src = row_src; g.w_estride = (size_t)inter * hidden;
        if (tile_rows == 256) {
            if (grouped_glu_256(ybuf, gbuf, (const bf16_t*)x, (const bf16_t*)w_gate, (const bf16_t*)w_up, (const bf16_t*)w_down, slots,
                                hidden, inter, n_experts, g, s))
                return 1;
        } else {
        if (launch_gemm_bf16_grouped(gbuf, (const bf16_t*)x, (const bf16_t*)w_gate, slots, inter, hidden, g, max_tiles, s)) return 1;
        if (launch_gemm_bf16_grouped(ubuf, (const bf16_t*)x, (const bf16_t*)w_up, slots, inter, hidden, g, max_tiles, s)) return 1;
        if (omx_fused_swiglu(gbuf, ubuf, gbuf, (int64_t)slots * inter, OMX_BFLOAT16, stream)) return 1;   // fused_swiglu(up, gate)
        g.row_src = nullptr;   // activations are already in expert-sorted order
        g.w_estride = (size_t)hidden * inter;
        if (launch_gemm_bf16_grouped(ybuf, gbuf, (const bf16_t*)w_down, slots, hidden, inter, g, max_tiles, s)) return 1;
        }
        moe_combine_kernel<false><<<n_tokens, 256, 0, s>>>((bf16_t*)out, ybuf, scores, pos_of_slot, hidden, top_k, resid);
        OMX_LAUNCH_CHECK();
    }
    if (inds_out) OMX_HIP_CHECK(hipMemcpyAsync(inds_out, inds, (size_t)slots * 4, hipMemcpyDeviceToDevice, s));
    if (scores_out) OMX_HIP_CHECK(hipMemcpyAsync(scores_out, scores, (size_t)slots * 2, hipMemcpyDeviceToDevice, s));
    return 0;
}

extern "C" int omx_moe_forward(void* out, const void* x, const void* gate_w, const void* w_gate, const void* w_up,
                               const void* w_down, int n_tokens, int hidden, int inter, int n_experts, int top_k,
                               int mode, int norm_topk_prob, uint32_t* inds_out, void* scores_out, omx_stream stream) {
    return moe_forward_impl(out, x, gate_w, w_gate, w_up, w_down, nullptr, n_tokens, hidden, inter, n_experts, top_k, mode,
                            norm_topk_prob, inds_out, scores_out, stream);
}

/* decoder-block form for the engines: out = resid + moe(rmsnorm(x) * norm_w) -- the norm runs in the router launch, the
 * residual in the combine launch; xn is a [n_tokens, hidden] scratch */
extern "C" int omx_moe_block_forward(void* out, const void* resid, const void* x, const void* norm_w, float eps, void* xn,
                                     const void* gate_w, const void* w_gate, const void* w_up, const void* w_down, int n_tokens,
                                     int hidden, int inter, int n_experts, int top_k, int mode, int norm_topk_prob,
                                     omx_stream stream) {
    OMX_REQUIRE(resid && norm_w && xn, "omx_moe_block_forward: null tensor");
    BlockFusion bf;
    bf.norm_w = norm_w; bf.eps = eps; bf.xn = xn; bf.resid = resid;
    return moe_forward_impl(out, x, gate_w, w_gate, w_up, w_down, nullptr, n_tokens, hidden, inter, n_experts, top_k, mode,
                            norm_topk_prob, nullptr, nullptr, stream, &bf);
}

/* one token, bf16 experts: the block WITHOUT its weighted sum -- partials[j, :] (f32) = bf16(bf16(y_j) * score_j) for the top_k routed
 * experts in slot order; out = bf16(resid + bf16(sum_j partials[j])) is left to the caller's next GEMV prologue (GemvArgs::x_partial_n).
 * Returns 2 when the shape does not take the batched-GEMV route (the caller then uses omx_moe_block_forward).                       */
extern "C" int omx_moe_block_partials(float* partials, const void* x, const void* norm_w, float eps, void* xn, const void* gate_w,
                                      const void* w_gate, const void* w_up, const void* w_down, int hidden, int inter, int n_experts,
                                      int top_k, int mode, int norm_topk_prob, omx_stream stream) {
    OMX_REQUIRE(partials && norm_w && xn, "omx_moe_block_partials: null tensor");
    if (!(top_k <= 32 && omx::gemv_k_supported(hidden, false) && omx::gemv_k_supported(inter, false))) return 2;
    BlockFusion bf;
    bf.norm_w = norm_w; bf.eps = eps; bf.xn = xn; bf.resid = x; bf.partials = partials;
    return moe_forward_impl(partials, x, gate_w, w_gate, w_up, w_down, nullptr, 1, hidden, inter, n_experts, top_k, mode, norm_topk_prob,
                            nullptr, nullptr, stream, &bf);
}

/* decoder-block form on a quantised checkpoint (mixtral-mlx/src/model.rs:560-600: gate = QuantizedLinear, switch_mlp =
 * QuantizedSwitchLinear x3).  Few tokens (<= 32 routed slots): router logits = quantised GEMV on the packed gate with the
 * RMSNorm prologue, selection, expert GEMVs on the packed stacks (RMSNorm prologue again: x is raw), weighted sum + residual.
 * More tokens: RMSNorm rows, router and experts dequantised once (what MLX's qmm does per tile), grouped GEMM route. */
extern "C" int omx_moe_block_forward_q_ex(void* out, const void* resid, const void* x, const void* norm_w, float eps, void* xn,
                                          const void* q_router, const void* s_router, const void* b_router, const void* q_gate,
                                          const void* s_gate, const void* b_gate, const void* q_up, const void* s_up, const void* b_up,
                                          const void* q_down, const void* s_down, const void* b_down, int n_tokens, int hidden, int inter,
                                          int n_experts, int top_k, int mode, int norm_topk_prob, int group_size, int bits, int f16,
                                          omx_stream stream);
extern "C" int omx_moe_block_forward_q(void* out, const void* resid, const void* x, const void* norm_w, float eps, void* xn,
                                       const void* q_router, const void* s_router, const void* b_router, const void* q_gate,
                                       const void* s_gate, const void* b_gate, const void* q_up, const void* s_up, const void* b_up,
                                       const void* q_down, const void* s_down, const void* b_down, int n_tokens, int hidden, int inter,
                                       int n_experts, int top_k, int mode, int norm_topk_prob, int group_size, int bits,
                                       omx_stream stream) {
    return omx_moe_block_forward_q_ex(out, resid, x, norm_w, eps, xn, q_router, s_router, b_router, q_gate, s_gate, b_gate, q_up, s_up, b_up,
                                      q_down, s_down, b_down, n_tokens, hidden, inter, n_experts, top_k, mode, norm_topk_prob, group_size, bits, 0,
                                      stream);
}
/* ... f16 != 0: a float16 checkpoint (the reference's own Mixtral format: 4-bit triplets, float16 in the MLX community builds) -- x, norm
 * weights, scales / biases and the result are float16 and every rounding point is float16, like MLX runs it.  More than 32 routed
 * slots (a prompt): the router exactly as the decode form computes it (packed GEMV per token, so a prompt routes like its tokens
 * would one by one), the stacks dequantised to float16, the grouped 256-row GEMMs' float16 instantiation, the float16 combine. */
extern "C" int omx_moe_block_forward_q_ex(void* out, const void* resid, const void* x, const void* norm_w, float eps, void* xn,
                                          const void* q_router, const void* s_router, const void* b_router, const void* q_gate,
                                          const void* s_gate, const void* b_gate, const void* q_up, const void* s_up, const void* b_up,
                                          const void* q_down, const void* s_down, const void* b_down, int n_tokens, int hidden, int inter,
                                          int n_experts, int top_k, int mode, int norm_topk_prob, int group_size, int bits, int f16,
                                          omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(out && resid && x && norm_w && xn && q_router && s_router && b_router && q_gate && s_gate && b_gate && q_up && s_up &&
                    b_up && q_down && s_down && b_down, "omx_moe_block_forward_q: null tensor");
    OMX_REQUIRE(bits == 4 || bits == 8, "omx_moe_block_forward_q: bits=%d (4 or 8)", bits);
    OMX_REQUIRE(hidden % 512 == 0 && inter % 512 == 0, "omx_moe_block_forward_q: hidden=%d and intermediate=%d must be multiples of 512", hidden, inter);
    OMX_REQUIRE(n_experts >= 1 && n_experts <= kMaxExperts && top_k >= 1 && top_k <= kMaxTopK && top_k <= n_experts, "omx_moe_block_forward_q: experts %d top-%d", n_experts, top_k);
    if (n_tokens == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    QExperts q;
    q.gate = QMat{(const uint32_t*)q_gate, (const bf16_t*)s_gate, (const bf16_t*)b_gate, inter};
    q.up = QMat{(const uint32_t*)q_up, (const bf16_t*)s_up, (const bf16_t*)b_up, inter};
    q.down = QMat{(const uint32_t*)q_down, (const bf16_t*)s_down, (const bf16_t*)b_down, hidden};
    q.gate.sb = quant_find_sb(q.gate.scales); q.up.sb = quant_find_sb(q.up.scales); q.down.sb = quant_find_sb(q.down.scales);
    q.group = group_size; q.bits = bits;
    const int slots = n_tokens * top_k;
    if (slots > 32 && f16) {
        size_t need = 0;
        omx_moe_workspace_bytes(n_tokens, hidden, inter, n_experts, top_k, &need);
        void* ws = nullptr;
        if (get_workspace(&ws, need)) return 1;
        char* p = (char*)ws;
        auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
        uint32_t* inds = (uint32_t*)take((size_t)slots * 4);
        uint32_t* row_src = (uint32_t*)take((size_t)slots * 4);
        uint32_t* pos_of_slot = (uint32_t*)take((size_t)slots * 4);
        bf16_t* scores = (bf16_t*)take((size_t)slots * 2);
        int* seg_start = (int*)take((size_t)(n_experts + 2) * 4);
        const int max_tiles = slots / 128 + n_experts + 1;
        int* tile_expert = (int*)take((size_t)max_tiles * 4);
        int* tile_m0 = (int*)take((size_t)max_tiles * 4);
        int* n_tiles = (int*)take(256);
        bf16_t* gbuf = (bf16_t*)take((size_t)slots * inter * 2);
        bf16_t* logits = (bf16_t*)take((size_t)slots * inter * 2);      // the `ubuf` slot (the GLU epilogue leaves it unused)
        bf16_t* ybuf = (bf16_t*)take((size_t)slots * hidden * 2);
        QGemvArgs a = {};
        a.m[0] = QMat{(const uint32_t*)q_router, (const bf16_t*)s_router, (const bf16_t*)b_router, n_experts};
        a.N = n_experts; a.K = hidden; a.group = group_size;
        a.x = (const bf16_t*)x; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.out = logits;
        a.n_batch = n_tokens; a.x_div = 1; a.scales_f16 = 1;
        if (launch_qgemv(a, bits, PRO_RMSNORM, EPI_STORE, s)) return 1;
        moe_route_logits_kernel<true><<<n_tokens, 64, 0, s>>>(logits, n_experts, top_k, mode, norm_topk_prob, inds, scores);
        OMX_LAUNCH_CHECK();
        if (omx_rms_norm(xn, x, norm_w, n_tokens, hidden, eps, OMX_FLOAT16, stream)) return 1;
        const size_t per = (size_t)n_experts * inter * hidden;
        if (3 * per > g_dq_cap) {
            OMX_HIP_CHECK(hipStreamSynchronize(s));
            if (g_dq) OMX_HIP_CHECK(hipFree(g_dq));
            OMX_HIP_CHECK(hipMalloc((void**)&g_dq, 3 * per * 2));
            g_dq_cap = 3 * per;
        }
        if (omx_dequantize(g_dq, q_gate, s_gate, b_gate, (int64_t)n_experts * inter, hidden, group_size, bits, OMX_FLOAT16, stream) ||
            omx_dequantize(g_dq + per, q_up, s_up, b_up, (int64_t)n_experts * inter, hidden, group_size, bits, OMX_FLOAT16, stream) ||
            omx_dequantize(g_dq + 2 * per, q_down, s_down, b_down, (int64_t)n_experts * hidden, inter, group_size, bits, OMX_FLOAT16, stream))
            return 1;
        moe_plan_kernel<<<1, 1024, 0, s>>>(inds, slots, n_experts, top_k, seg_start, row_src, pos_of_slot, tile_expert, tile_m0, n_tiles, 256);
        OMX_LAUNCH_CHECK();
        GroupedDesc g;
        g.tile_expert = tile_expert; g.tile_m0 = tile_m0; g.seg_start = seg_start; g.n_tiles = n_tiles;
        g.row_src = row_src; g.w_estride = (size_t)inter * hidden;
        const bool was_f16 = gemm_set_f16(true);    // (the engine's prompt pass has it on already: restore, do not clear)
        const int rc = grouped_glu_256(ybuf, gbuf, (const bf16_t*)xn, g_dq, g_dq + per, g_dq + 2 * per, slots, hidden, inter, n_experts, g, s);
        gemm_set_f16(was_f16);
        if (rc) return 1;
        moe_combine_kernel<true><<<n_tokens, 256, 0, s>>>((bf16_t*)out, ybuf, scores, pos_of_slot, hidden, top_k, (const bf16_t*)resid);
        OMX_LAUNCH_CHECK();
        return 0;
    }
    if (slots > 32) {
        // dequantised router (tiny) lives behind the expert scratch; rows normalised once
        static bf16_t* router_dq = nullptr;
        static size_t router_cap = 0;
        if ((size_t)n_experts * hidden > router_cap) {
            OMX_HIP_CHECK(hipStreamSynchronize(s));
            if (router_dq) OMX_HIP_CHECK(hipFree(router_dq));
            OMX_HIP_CHECK(hipMalloc((void**)&router_dq, (size_t)n_experts * hidden * 2));
            router_cap = (size_t)n_experts * hidden;
        }
        if (omx_dequantize(router_dq, q_router, s_router, b_router, n_experts, hidden, group_size, bits, OMX_BFLOAT16, stream)) return 1;
        if (omx_rms_norm(xn, x, norm_w, n_tokens, hidden, eps, OMX_BFLOAT16, stream)) return 1;
        BlockFusion bf;
        bf.resid = resid;
        return moe_forward_impl(out, xn, router_dq, nullptr, nullptr, nullptr, &q, n_tokens, hidden, inter, n_experts, top_k, mode,
                                norm_topk_prob, nullptr, nullptr, stream, &bf);
    }
    size_t need = 0;
    omx_moe_workspace_bytes(n_tokens, hidden, inter, n_experts, top_k, &need);
    void* ws = nullptr;
    if (get_workspace(&ws, need)) return 1;
    char* p = (char*)ws;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    uint32_t* inds = (uint32_t*)take((size_t)slots * 4);
    take((size_t)slots * 4); take((size_t)slots * 4);
    bf16_t* scores = (bf16_t*)take((size_t)slots * 2);
    take((size_t)(n_experts + 2) * 4);
    const int max_tiles = slots / 128 + n_experts + 1;
    take((size_t)max_tiles * 4); take((size_t)max_tiles * 4); take(256);
    bf16_t* gbuf = (bf16_t*)take((size_t)slots * inter * 2);
    bf16_t* logits = (bf16_t*)take((size_t)slots * inter * 2);      // the `ubuf` slot: n_tokens * E <= slots * inter
    bf16_t* ybuf = (bf16_t*)take((size_t)slots * hidden * 2);
    {   // router: logits[t, e] = rmsnorm(x_t) . dequant(gate[e])  (quantized_matmul, nn/quantized.rs:366-375)
        QGemvArgs a = {};
        a.m[0] = QMat{(const uint32_t*)q_router, (const bf16_t*)s_router, (const bf16_t*)b_router, n_experts};
        a.N = n_experts; a.K = hidden; a.group = group_size;
        a.x = (const bf16_t*)x; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.out = logits;
        a.n_batch = n_tokens; a.x_div = 1; a.scales_f16 = f16;
        if (launch_qgemv(a, bits, PRO_RMSNORM, EPI_STORE, s)) return 1;
        if (f16) moe_route_logits_kernel<true><<<n_tokens, 64, 0, s>>>(logits, n_experts, top_k, mode, norm_topk_prob, inds, scores);
        else moe_route_logits_kernel<false><<<n_tokens, 64, 0, s>>>(logits, n_experts, top_k, mode, norm_topk_prob, inds, scores);
        OMX_LAUNCH_CHECK();
    }
    QGemvArgs a = {};
    a.m[0] = q.gate; a.m[1] = q.up; a.N = inter; a.K = hidden; a.group = group_size;
    a.x = (const bf16_t*)x; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.out = gbuf;
    a.n_batch = slots; a.x_div = top_k; a.w_sel = inds;
    a.w_estride = (size_t)inter * hidden * bits / 32; a.s_estride = (size_t)inter * (hidden / group_size);
    a.swiglu_single_round = 1; a.scales_f16 = f16;
    if (launch_qgemv(a, bits, PRO_RMSNORM, EPI_SWIGLU, s)) return 1;
    QGemvArgs d = {};
    d.m[0] = q.down; d.N = hidden; d.K = inter; d.group = group_size;
    d.x = gbuf; d.out = ybuf;
    d.n_batch = slots; d.x_div = 1; d.w_sel = inds; d.scales_f16 = f16;
    d.w_estride = (size_t)hidden * inter * bits / 32; d.s_estride = (size_t)hidden * (inter / group_size);
    if (launch_qgemv(d, bits, PRO_NONE, EPI_STORE, s)) return 1;
    if (f16) moe_combine_kernel<true><<<n_tokens, 256, 0, s>>>((bf16_t*)out, ybuf, scores, nullptr, hidden, top_k, (const bf16_t*)resid);
    else moe_combine_kernel<false><<<n_tokens, 256, 0, s>>>((bf16_t*)out, ybuf, scores, nullptr, hidden, top_k, (const bf16_t*)resid);
    OMX_LAUNCH_CHECK();
    return 0;
}

// omx_moe_block_slots_ep runs omx_moe_block_partial_ep's batched branch up to the expert outputs (same launches, same scratch)
static thread_local omx_moe_ep_slots* g_ep_slots_out = nullptr;

/* expert-parallel decode form (simple variant: activations replicated, experts sharded): this rank
 * holds experts [e_lo, e_lo + e_n) (w_* = ITS stacks [e_n, ...]); partial [n_tokens, hidden] f32 receives its share of
 * sum_j bf16(y_j * score_j).  The ranks' partials are summed by one all-reduce; the caller then forms
 * h + bf16(sum) (the residual).  n_tokens * top_k <= 32. */
extern "C" int omx_moe_block_partial_ep(float* partial, const void* x, const void* norm_w, float eps, void* xn, const void* gate_w,
                                        const void* w_gate, const void* w_up, const void* w_down, int n_tokens, int hidden, int inter,
                                        int n_experts, int top_k, int mode, int norm_topk_prob, int e_lo, int e_n,
                                        omx_stream stream) {
    using namespace omx;
    // norm_w == null: x holds the normalised rows already (the batched prefill runs its own RMSNorm launch, like the single-rank pass)
    OMX_REQUIRE(partial && x && (xn || !norm_w) && gate_w && w_gate && w_up && w_down, "omx_moe_block_partial_ep: null tensor");
    if (!norm_w) xn = const_cast<void*>(x);
    OMX_REQUIRE(n_experts >= 1 && n_experts <= kMaxExperts && top_k >= 1 && top_k <= kMaxTopK && top_k <= n_experts &&
                    e_lo >= 0 && e_n >= 1 && e_lo + e_n <= n_experts, "omx_moe_block_partial_ep: experts %d top-%d shard [%d, +%d)", n_experts, top_k, e_lo, e_n);
    const int slots = n_tokens * top_k;
    OMX_REQUIRE(slots >= 1 && hidden % 64 == 0 && inter % 64 == 0, "omx_moe_block_partial_ep: %d routed slots, hidden %d, intermediate %d", slots, hidden, inter);
    const bool decode = slots <= 32 && gemv_k_supported(hidden, false) && gemv_k_supported(inter, false);
    hipStream_t s = (hipStream_t)stream;
    size_t need = 0;
    omx_moe_workspace_bytes(n_tokens, hidden, inter, n_experts, top_k, &need);
    // the STREAM's scratch, not the library-wide one: several expert-parallel ranks may live in one process (the loopback tests, a
    // multi-GPU host process), each on its own stream.  The engine sizes it once for its largest batch (engine.hip), so the pointers a
    // captured decode step holds never move.
    void* ws = nullptr;
    if (get_workspace_aux(&ws, need, s)) return 1;
    char* p = (char*)ws;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    uint32_t* inds = (uint32_t*)take((size_t)slots * 4);
    uint32_t* row_src = (uint32_t*)take((size_t)slots * 4);
    uint32_t* pos_of_slot = (uint32_t*)take((size_t)slots * 4);
    bf16_t* scores = (bf16_t*)take((size_t)slots * 2);
    int* seg_start = (int*)take((size_t)(n_experts + 2) * 4);
    const int max_tiles = slots / 128 + n_experts + 1;
    int* tile_expert = (int*)take((size_t)max_tiles * 4);
    int* tile_m0 = (int*)take((size_t)max_tiles * 4);
    int* n_tiles = (int*)take(256);
    bf16_t* gbuf = (bf16_t*)take((size_t)slots * inter * 2);
    bf16_t* ubuf = (bf16_t*)take((size_t)slots * inter * 2);
    bf16_t* ybuf = (bf16_t*)take((size_t)slots * hidden * 2);
    moe_router_kernel<<<n_tokens, router_threads(n_experts, hidden), 0, s>>>((const bf16_t*)x, (const bf16_t*)gate_w, hidden, n_experts, top_k, mode,
                                                          norm_topk_prob, inds, scores, (const bf16_t*)norm_w, eps, norm_w ? (bf16_t*)xn : nullptr);
    OMX_LAUNCH_CHECK();
    if (!decode) {
        // batched form (a prompt under expert parallelism): the slots routed to THIS rank's experts are counting-sorted by local
        // expert and run through the grouped matrix-core GEMMs; everything routed elsewhere lands in one trailing pseudo-expert that
        // gets no tiles.  The partial weighted sum reads its rows back through the sort permutation.
        // (partial == the sentinel of omx_moe_block_slots_ep: stop after the experts and hand the slot tables to the caller's combine)
        uint32_t* local_inds = nullptr;
        OMX_HIP_CHECK(hipMallocAsync((void**)&local_inds, (size_t)slots * 4, s));
        moe_localize_kernel<<<(slots + 255) / 256, 256, 0, s>>>(local_inds, inds, slots, e_lo, e_n);
        OMX_LAUNCH_CHECK();
        const int tile_rows = moe_tile_rows(slots / (n_experts / e_n > 0 ? n_experts / e_n : 1), e_n, hidden, inter);
        moe_plan_kernel<<<1, 1024, 0, s>>>(local_inds, slots, e_n + 1, top_k, seg_start, row_src, pos_of_slot, tile_expert, tile_m0, n_tiles,
                                           tile_rows, e_n);
        OMX_LAUNCH_CHECK();
        (void)hipFreeAsync(local_inds, s);
        GroupedDesc g;
        g.tile_expert = tile_expert; g.tile_m0 = tile_m0; g.seg_start = seg_start; g.n_tiles = n_tiles;
        g.row_src = row_src; g.w_estride = (size_t)inter * hidden;
        if (tile_rows == 256) {
            if (grouped_glu_256(ybuf, gbuf, (const bf16_t*)xn, (const bf16_t*)w_gate, (const bf16_t*)w_up, (const bf16_t*)w_down, slots, hidden,
                                inter, n_experts, g, s))
                return 1;
        } else {
            if (launch_gemm_bf16_grouped(gbuf, (const bf16_t*)xn, (const bf16_t*)w_gate, slots, inter, hidden, g, max_tiles, s)) return 1;
            if (launch_gemm_bf16_grouped(ubuf, (const bf16_t*)xn, (const bf16_t*)w_up, slots, inter, hidden, g, max_tiles, s)) return 1;
            if (omx_fused_swiglu(gbuf, ubuf, gbuf, (int64_t)slots * inter, OMX_BFLOAT16, stream)) return 1;
            g.row_src = nullptr;
            g.w_estride = (size_t)hidden * inter;
            if (launch_gemm_bf16_grouped(ybuf, gbuf, (const bf16_t*)w_down, slots, hidden, inter, g, max_tiles, s)) return 1;
        }
        if (g_ep_slots_out) {
            *g_ep_slots_out = omx_moe_ep_slots{ybuf, pos_of_slot, inds, scores};
            return 0;
        }
        moe_combine_partial_kernel<false><<<n_tokens, 256, 0, s>>>(partial, ybuf, scores, inds, hidden, top_k, e_lo, e_n, pos_of_slot);
        OMX_LAUNCH_CHECK();
        return 0;
    }
    OMX_REQUIRE(!g_ep_slots_out, "omx_moe_block_slots_ep: the batched form only (more than 32 routed slots)");
    GemvArgs a = {};
    a.w0 = (const bf16_t*)w_gate; a.w1 = (const bf16_t*)w_up; a.n0 = inter; a.N = inter; a.K = hidden;
    a.x = (const bf16_t*)xn; a.out = gbuf;
    a.n_batch = slots; a.x_div = top_k; a.x_bstride = hidden; a.out_bstride_bytes = (size_t)inter * 2;
    a.w_sel = inds; a.w_estride = (size_t)inter * hidden; a.swiglu_single_round = 1;
    a.w_sel_lo = e_lo; a.w_sel_n = e_n;
    if (launch_gemv(a, PRO_NONE, EPI_SWIGLU, s)) return 1;
    GemvArgs d = {};
    d.w0 = (const bf16_t*)w_down; d.n0 = hidden; d.N = hidden; d.K = inter;
    d.x = gbuf; d.out = ybuf;
    d.n_batch = slots; d.x_div = 1; d.x_bstride = inter; d.out_bstride_bytes = (size_t)hidden * 2;
    d.w_sel = inds; d.w_estride = (size_t)hidden * inter;
    d.w_sel_lo = e_lo; d.w_sel_n = e_n;
    if (launch_gemv(d, PRO_NONE, EPI_STORE, s)) return 1;
    moe_combine_partial_kernel<false><<<n_tokens, 256, 0, s>>>(partial, ybuf, scores, inds, hidden, top_k, e_lo, e_n);
    OMX_LAUNCH_CHECK();
    return 0;
}

/* The batched expert-parallel block WITHOUT its weighted sum: router, this rank's slots sorted by local expert, grouped matrix-core
 * GEMMs -- then the tables an exchange needs instead of a [T, hidden] partial: y (the sorted expert outputs, bf16), pos_of_slot
 * [T * k] (slot -> row of y), inds [T * k] (global expert ids) and scores [T * k] (bf16), all in the stream's scratch (valid until the
 * next MoE call on this stream).  x: the NORMALISED rows.  Consumer: omx_peer_moe_combine (peer_allreduce.hip). */
extern "C" int omx_moe_block_slots_ep(omx_moe_ep_slots* out, const void* x, const void* gate_w, const void* w_gate, const void* w_up,
                                      const void* w_down, int n_tokens, int hidden, int inter, int n_experts, int top_k, int mode,
                                      int norm_topk_prob, int e_lo, int e_n, omx_stream stream) {
    OMX_REQUIRE(out, "omx_moe_block_slots_ep: null output");
    float sentinel = 0.f;
    g_ep_slots_out = out;
    const int rc = omx_moe_block_partial_ep(&sentinel, x, nullptr, 0.f, nullptr, gate_w, w_gate, w_up, w_down, n_tokens, hidden, inter, n_experts,
                                            top_k, mode, norm_topk_prob, e_lo, e_n, stream);
    g_ep_slots_out = nullptr;
    return rc;
}

/* expert TENSOR parallel, decode form (n_tokens * top_k <= 32): every rank holds ALL experts but only `inter` = I / tp of each expert's
 * intermediate columns (gate / up rows [r I/tp, +I/tp) of every expert, the same columns of its down projection), so a token's two
 * experts stream from all ranks at once -- what expert parallelism cannot give batch-1 decode, where top-2 routing touches at most
 * two ranks' experts per layer.  The router runs replicated (same row, same weights: the same selection on every rank) and leaves
 * its choice in route_inds [slots] / route_scores [slots]; y_partial [slots, hidden] f32 receives the UNROUNDED partial of each routed
 * slot's down projection.  After the all-reduce over the ranks omx_moe_combine_slots forms the block output with the single-device
 * roundings: bf16(resid + bf16(sum_j bf16(bf16(y_j) * score_j))).  Scratch: the STREAM's (several ranks may share a process). */
extern "C" int omx_moe_block_partial_tp(float* y_partial, uint32_t* route_inds, void* route_scores, const void* x, const void* norm_w,
                                        float eps, void* xn, const void* gate_w, const void* w_gate, const void* w_up, const void* w_down,
                                        int n_tokens, int hidden, int inter, int n_experts, int top_k, int mode, int norm_topk_prob,
                                        omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(y_partial && route_inds && route_scores && x && norm_w && xn && gate_w && w_gate && w_up && w_down, "omx_moe_block_partial_tp: null tensor");
    OMX_REQUIRE(n_experts >= 1 && n_experts <= kMaxExperts && top_k >= 1 && top_k <= kMaxTopK && top_k <= n_experts,
                "omx_moe_block_partial_tp: experts %d top-%d", n_experts, top_k);
    const int slots = n_tokens * top_k;
    OMX_REQUIRE(slots >= 1 && slots <= 32 && hidden % 64 == 0 && inter % 64 == 0 && gemv_k_supported(hidden, false) && gemv_k_supported(inter, false),
                "omx_moe_block_partial_tp: %d routed slots (at most 32), hidden %d, per-rank intermediate %d", slots, hidden, inter);
    hipStream_t s = (hipStream_t)stream;
    void* ws = nullptr;
    if (get_workspace_aux(&ws, ((size_t)slots * inter * 2 + 256), s)) return 1;
    bf16_t* gbuf = (bf16_t*)ws;
    moe_router_kernel<<<n_tokens, router_threads(n_experts, hidden), 0, s>>>((const bf16_t*)x, (const bf16_t*)gate_w, hidden, n_experts, top_k, mode,
                                                          norm_topk_prob, route_inds, (bf16_t*)route_scores, (const bf16_t*)norm_w, eps, (bf16_t*)xn);
    OMX_LAUNCH_CHECK();
    GemvArgs a = {};
    a.w0 = (const bf16_t*)w_gate; a.w1 = (const bf16_t*)w_up; a.n0 = inter; a.N = inter; a.K = hidden;
    a.x = (const bf16_t*)xn; a.out = gbuf;
    a.n_batch = slots; a.x_div = top_k; a.x_bstride = hidden; a.out_bstride_bytes = (size_t)inter * 2;
    a.w_sel = route_inds; a.w_estride = (size_t)inter * hidden; a.swiglu_single_round = 1;
    if (launch_gemv(a, PRO_NONE, EPI_SWIGLU, s)) return 1;
    GemvArgs d = {};
    d.w0 = (const bf16_t*)w_down; d.n0 = hidden; d.N = hidden; d.K = inter;
    d.x = gbuf; d.out = y_partial;
    d.n_batch = slots; d.x_div = 1; d.x_bstride = inter; d.out_bstride_bytes = (size_t)hidden * 4;
    d.w_sel = route_inds; d.w_estride = (size_t)hidden * inter;
    return launch_gemv(d, PRO_NONE, EPI_F32, s);
}

/* ---- the same two sharded forms on the reference's REAL Mixtral format: MLX-packed expert stacks (round 5; mixtral-mlx refuses anything
 *      else: /root/reference/mixtral-mlx/src/model.rs:554-556, stacking at :466-548).  The triplet of a slice is the slice of the triplet
 *      (quantisation is per group of one row), so a rank holds the packed rows / whole-group K slices of ITS experts or columns.
 *      Few tokens (<= 32 routed slots): the packed GEMVs with the expert filter; more: the rank's stacks dequantised once per call into
 *      stream-ordered scratch, then the bf16 functions above (what the single-rank packed block does too).  bf16 scales / activations. ---- */
namespace {
struct DqScratch {      // dequantised router + expert stacks of one call, freed in stream order
    omx::bf16_t *router = nullptr, *g = nullptr, *u = nullptr, *d = nullptr;
    hipStream_t s;
    explicit DqScratch(hipStream_t s_) : s(s_) {}
    ~DqScratch() { for (void* p : {(void*)router, (void*)g, (void*)u, (void*)d}) if (p) (void)hipFreeAsync(p, s); }
};
int dequant_stacks(DqScratch& q, const void* q_router, const void* s_router, const void* b_router, const void* q_gate, const void* s_gate,
                   const void* b_gate, const void* q_up, const void* s_up, const void* b_up, const void* q_down, const void* s_down,
                   const void* b_down, int hidden, int inter, int n_experts, int e_n, int group_size, int bits, omx_dtype dt = OMX_BFLOAT16) {
    using namespace omx;
    const size_t per = (size_t)e_n * inter * hidden;
    // the stream-ordered pool hands freed blocks back to the system at the next synchronisation unless told to keep them (release threshold 0 by
    // default): a Mixtral-size shard would re-allocate ~1.4 GB per layer and prompt.  Keep them: the next layer's scratch is this layer's
    static const bool pool_keeps = [] {
        int dev = 0;
        hipMemPool_t pool = nullptr;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) {
            uint64_t keep = UINT64_MAX;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        return true;
    }();
    (void)pool_keeps;
    OMX_HIP_CHECK(hipMallocAsync((void**)&q.router, (size_t)n_experts * hidden * 2, q.s));
    OMX_HIP_CHECK(hipMallocAsync((void**)&q.g, per * 2, q.s));
    OMX_HIP_CHECK(hipMallocAsync((void**)&q.u, per * 2, q.s));
    OMX_HIP_CHECK(hipMallocAsync((void**)&q.d, per * 2, q.s));
    return omx_dequantize(q.router, q_router, s_router, b_router, n_experts, hidden, group_size, bits, dt, q.s) ||
           omx_dequantize(q.g, q_gate, s_gate, b_gate, (int64_t)e_n * inter, hidden, group_size, bits, dt, q.s) ||
           omx_dequantize(q.u, q_up, s_up, b_up, (int64_t)e_n * inter, hidden, group_size, bits, dt, q.s) ||
           omx_dequantize(q.d, q_down, s_down, b_down, (int64_t)e_n * hidden, inter, group_size, bits, dt, q.s);
}
// router of the packed block: logits by the packed GEMV (RMSNorm prologue when norm_w), then the selection
int route_packed(const void* x, const void* norm_w, float eps, const void* q_router, const void* s_router, const void* b_router, int n_tokens,
                 int hidden, int n_experts, int top_k, int mode, int norm_topk_prob, int group_size, int bits, omx::bf16_t* logits, uint32_t* inds,
                 omx::bf16_t* scores, hipStream_t s, bool f16 = false) {
    using namespace omx;
    QGemvArgs a = {};
    a.m[0] = QMat{(const uint32_t*)q_router, (const bf16_t*)s_router, (const bf16_t*)b_router, n_experts};
    a.N = n_experts; a.K = hidden; a.group = group_size;
    a.x = (const bf16_t*)x; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.out = logits;
    a.n_batch = n_tokens; a.x_div = 1; a.scales_f16 = f16;
    if (launch_qgemv(a, bits, norm_w ? PRO_RMSNORM : PRO_NONE, EPI_STORE, s)) return 1;
    if (f16) moe_route_logits_kernel<true><<<n_tokens, 64, 0, s>>>(logits, n_experts, top_k, mode, norm_topk_prob, inds, scores);
    else moe_route_logits_kernel<false><<<n_tokens, 64, 0, s>>>(logits, n_experts, top_k, mode, norm_topk_prob, inds, scores);
    OMX_LAUNCH_CHECK();
    return 0;
}
}  // namespace

/* expert parallel on packed stacks: q_gate / q_up / q_down (+ scales, biases) are THIS RANK's experts [e_lo, e_lo + e_n) -- [e_n, inter,
 * hidden * bits / 32] etc.; the router triplet is the whole (replicated) gate.  Same contract as omx_moe_block_partial_ep. */
extern "C" int omx_moe_block_partial_ep_q(float* partial, const void* x, const void* norm_w, float eps, void* xn, const void* q_router,
                                          const void* s_router, const void* b_router, const void* q_gate, const void* s_gate,
                                          const void* b_gate, const void* q_up, const void* s_up, const void* b_up, const void* q_down,
                                          const void* s_down, const void* b_down, int n_tokens, int hidden, int inter, int n_experts, int top_k,
                                          int mode, int norm_topk_prob, int e_lo, int e_n, int group_size, int bits, int f16, omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(partial && x && q_router && s_router && b_router && q_gate && s_gate && b_gate && q_up && s_up && b_up && q_down && s_down && b_down,
                "omx_moe_block_partial_ep_q: null tensor");
    OMX_REQUIRE(bits == 4 || bits == 8, "omx_moe_block_partial_ep_q: bits=%d (4 or 8)", bits);
    OMX_REQUIRE(hidden % 512 == 0 && inter % 512 == 0, "omx_moe_block_partial_ep_q: hidden=%d and intermediate=%d must be multiples of 512", hidden, inter);
    OMX_REQUIRE(n_experts >= 1 && n_experts <= kMaxExperts && top_k >= 1 && top_k <= kMaxTopK && top_k <= n_experts && e_lo >= 0 && e_n >= 1 &&
                    e_lo + e_n <= n_experts, "omx_moe_block_partial_ep_q: experts %d top-%d shard [%d, +%d)", n_experts, top_k, e_lo, e_n);
    const int slots = n_tokens * top_k;
    OMX_REQUIRE(slots >= 1, "omx_moe_block_partial_ep_q: no tokens");
    hipStream_t s = (hipStream_t)stream;
    if (slots > 32 && f16) {
        // a float16 checkpoint's prompt on a shard (round 6; token-serial before): the router exactly as the decode form computes it (packed
        // GEMV per token with the RMSNorm prologue, so a prompt routes like its tokens would one by one), this rank's stacks dequantised to
        // float16, the slots of its experts through the grouped 256-row GEMMs' float16 instantiation (everything routed elsewhere lands in a
        // trailing pseudo-expert without tiles), the f32 partial of every token's weighted sum with the float16 roundings of the decode form
        OMX_REQUIRE(norm_w && xn, "omx_moe_block_partial_ep_q: the float16 prompt form normalises its rows here (norm_w, xn)");
        size_t need = 0;
        omx_moe_workspace_bytes(n_tokens, hidden, inter, n_experts, top_k, &need);
        need += (size_t)n_tokens * n_experts * 2 + 1024;
        void* ws = nullptr;
        if (get_workspace_aux(&ws, need, s)) return 1;
        char* p = (char*)ws;
        auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
        uint32_t* inds = (uint32_t*)take((size_t)slots * 4);
        uint32_t* row_src = (uint32_t*)take((size_t)slots * 4);
        uint32_t* pos_of_slot = (uint32_t*)take((size_t)slots * 4);
        bf16_t* scores = (bf16_t*)take((size_t)slots * 2);
        int* seg_start = (int*)take((size_t)(n_experts + 2) * 4);
        const int max_tiles = slots / 128 + n_experts + 1;
        int* tile_expert = (int*)take((size_t)max_tiles * 4);
        int* tile_m0 = (int*)take((size_t)max_tiles * 4);
        int* n_tiles = (int*)take(256);
        bf16_t* gbuf = (bf16_t*)take((size_t)slots * inter * 2);
        (void)take((size_t)slots * inter * 2);                          // (the `ubuf` slot of the layout: the GLU epilogue leaves it unused)
        bf16_t* ybuf = (bf16_t*)take((size_t)slots * hidden * 2);
        bf16_t* logits = (bf16_t*)take((size_t)n_tokens * n_experts * 2);
        if (route_packed(x, norm_w, eps, q_router, s_router, b_router, n_tokens, hidden, n_experts, top_k, mode, norm_topk_prob, group_size, bits,
                         logits, inds, scores, s, true))
            return 1;
        if (omx_rms_norm(xn, x, norm_w, n_tokens, hidden, eps, OMX_FLOAT16, stream)) return 1;
        DqScratch dq(s);
        if (dequant_stacks(dq, q_router, s_router, b_router, q_gate, s_gate, b_gate, q_up, s_up, b_up, q_down, s_down, b_down, hidden, inter, n_experts,
                           e_n, group_size, bits, OMX_FLOAT16))
            return 1;
        uint32_t* local_inds = nullptr;
        OMX_HIP_CHECK(hipMallocAsync((void**)&local_inds, (size_t)slots * 4, s));
        moe_localize_kernel<<<(slots + 255) / 256, 256, 0, s>>>(local_inds, inds, slots, e_lo, e_n);
        OMX_LAUNCH_CHECK();
        moe_plan_kernel<<<1, 1024, 0, s>>>(local_inds, slots, e_n + 1, top_k, seg_start, row_src, pos_of_slot, tile_expert, tile_m0, n_tiles, 256, e_n);
        OMX_LAUNCH_CHECK();
        (void)hipFreeAsync(local_inds, s);
        GroupedDesc g;
        g.tile_expert = tile_expert; g.tile_m0 = tile_m0; g.seg_start = seg_start; g.n_tiles = n_tiles;
        g.row_src = row_src; g.w_estride = (size_t)inter * hidden;
        const bool was_f16 = gemm_set_f16(true);    // (the engine's prompt pass has it on already: restore, do not clear)
        const int rc = grouped_glu_256(ybuf, gbuf, (const bf16_t*)xn, dq.g, dq.u, dq.d, slots, hidden, inter, n_experts, g, s);
        gemm_set_f16(was_f16);
        if (rc) return 1;
        moe_combine_partial_kernel<true><<<n_tokens, 256, 0, s>>>(partial, ybuf, scores, inds, hidden, top_k, e_lo, e_n, pos_of_slot);
        OMX_LAUNCH_CHECK();
        return 0;
    }
    if (slots > 32) {
        // a prompt: normalised rows (the caller's, or made here), the rank's stacks and the router dequantised, the bf16 batched form
        const void* rows = x;
        if (norm_w) {
            OMX_REQUIRE(xn, "omx_moe_block_partial_ep_q: xn scratch needed with norm_w");
            if (omx_rms_norm(xn, x, norm_w, n_tokens, hidden, eps, OMX_BFLOAT16, stream)) return 1;
            rows = xn;
        }
        DqScratch dq(s);
        if (dequant_stacks(dq, q_router, s_router, b_router, q_gate, s_gate, b_gate, q_up, s_up, b_up, q_down, s_down, b_down, hidden, inter, n_experts,
                           e_n, group_size, bits))
            return 1;
        return omx_moe_block_partial_ep(partial, rows, nullptr, eps, nullptr, dq.router, dq.g, dq.u, dq.d, n_tokens, hidden, inter, n_experts, top_k,
                                        mode, norm_topk_prob, e_lo, e_n, stream);
    }
    OMX_REQUIRE(norm_w, "omx_moe_block_partial_ep_q: the decode form normalises in its GEMV prologues (norm_w)");
    void* ws = nullptr;
    const size_t need = (size_t)slots * 8 + 1024 + (size_t)n_tokens * n_experts * 2 + (size_t)slots * inter * 2 + (size_t)slots * hidden * 2 + 1024;
    if (get_workspace_aux(&ws, need, s)) return 1;     // the STREAM's scratch (several ranks may share a process; a captured step keeps these pointers)
    char* p = (char*)ws;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    uint32_t* inds = (uint32_t*)take((size_t)slots * 4);
    bf16_t* scores = (bf16_t*)take((size_t)slots * 2);
    bf16_t* logits = (bf16_t*)take((size_t)n_tokens * n_experts * 2);
    bf16_t* gbuf = (bf16_t*)take((size_t)slots * inter * 2);
    bf16_t* ybuf = (bf16_t*)take((size_t)slots * hidden * 2);
    if (route_packed(x, norm_w, eps, q_router, s_router, b_router, n_tokens, hidden, n_experts, top_k, mode, norm_topk_prob, group_size, bits, logits,
                     inds, scores, s, f16 != 0))
        return 1;
    QGemvArgs a = {};
    a.m[0] = QMat{(const uint32_t*)q_gate, (const bf16_t*)s_gate, (const bf16_t*)b_gate, inter};
    a.m[1] = QMat{(const uint32_t*)q_up, (const bf16_t*)s_up, (const bf16_t*)b_up, inter};
    a.m[0].sb = quant_find_sb(a.m[0].scales); a.m[1].sb = quant_find_sb(a.m[1].scales);
    a.N = inter; a.K = hidden; a.group = group_size;
    a.x = (const bf16_t*)x; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.out = gbuf;
    a.n_batch = slots; a.x_div = top_k; a.w_sel = inds; a.w_sel_lo = e_lo; a.w_sel_n = e_n;
    a.w_estride = (size_t)inter * hidden * bits / 32; a.s_estride = (size_t)inter * (hidden / group_size);
    a.swiglu_single_round = 1; a.scales_f16 = f16 != 0;
    if (launch_qgemv(a, bits, PRO_RMSNORM, EPI_SWIGLU, s)) return 1;
    QGemvArgs d = {};
    d.m[0] = QMat{(const uint32_t*)q_down, (const bf16_t*)s_down, (const bf16_t*)b_down, hidden};
    d.m[0].sb = quant_find_sb(d.m[0].scales);
    d.N = hidden; d.K = inter; d.group = group_size;
    d.x = gbuf; d.out = ybuf;
    d.n_batch = slots; d.x_div = 1; d.w_sel = inds; d.w_sel_lo = e_lo; d.w_sel_n = e_n;
    d.w_estride = (size_t)hidden * inter * bits / 32; d.s_estride = (size_t)hidden * (inter / group_size);
    d.scales_f16 = f16 != 0;
    if (launch_qgemv(d, bits, PRO_NONE, EPI_STORE, s)) return 1;
    if (f16) moe_combine_partial_kernel<true><<<n_tokens, 256, 0, s>>>(partial, ybuf, scores, inds, hidden, top_k, e_lo, e_n);
    else moe_combine_partial_kernel<false><<<n_tokens, 256, 0, s>>>(partial, ybuf, scores, inds, hidden, top_k, e_lo, e_n);
    OMX_LAUNCH_CHECK();
    return 0;
}

/* expert TENSOR parallel on packed stacks, decode form (<= 32 routed slots): every expert's packed gate / up rows [r inter, + inter) and the
 * whole-group K slice [r inter, + inter) of its down projection (inter = I / tp, a multiple of 512).  Same contract as
 * omx_moe_block_partial_tp: the router's choice in route_inds / route_scores, the UNROUNDED f32 partial of every routed slot's down
 * projection in y_partial [slots, hidden]; omx_moe_combine_slots follows the all-reduce. */
extern "C" int omx_moe_block_partial_tp_q(float* y_partial, uint32_t* route_inds, void* route_scores, const void* x, const void* norm_w, float eps,
                                          const void* q_router, const void* s_router, const void* b_router, const void* q_gate, const void* s_gate,
                                          const void* b_gate, const void* q_up, const void* s_up, const void* b_up, const void* q_down,
                                          const void* s_down, const void* b_down, int n_tokens, int hidden, int inter, int n_experts, int top_k,
                                          int mode, int norm_topk_prob, int group_size, int bits, int f16, omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(y_partial && route_inds && route_scores && x && norm_w && q_router && s_router && b_router && q_gate && s_gate && b_gate && q_up &&
                    s_up && b_up && q_down && s_down && b_down, "omx_moe_block_partial_tp_q: null tensor");
    OMX_REQUIRE(bits == 4 || bits == 8, "omx_moe_block_partial_tp_q: bits=%d (4 or 8)", bits);
    const int slots = n_tokens * top_k;
    OMX_REQUIRE(slots >= 1 && slots <= 32 && hidden % 512 == 0 && inter % 512 == 0,
                "omx_moe_block_partial_tp_q: %d routed slots (at most 32), hidden %d and per-rank intermediate %d multiples of 512", slots, hidden, inter);
    OMX_REQUIRE(n_experts >= 1 && n_experts <= kMaxExperts && top_k >= 1 && top_k <= kMaxTopK && top_k <= n_experts, "omx_moe_block_partial_tp_q: experts %d top-%d", n_experts, top_k);
    hipStream_t s = (hipStream_t)stream;
    void* ws = nullptr;
    if (get_workspace_aux(&ws, (size_t)n_tokens * n_experts * 2 + 512 + (size_t)slots * inter * 2 + 256, s)) return 1;
    bf16_t* logits = (bf16_t*)ws;
    bf16_t* gbuf = (bf16_t*)((char*)ws + (((size_t)n_tokens * n_experts * 2 + 255) & ~(size_t)255));
    if (route_packed(x, norm_w, eps, q_router, s_router, b_router, n_tokens, hidden, n_experts, top_k, mode, norm_topk_prob, group_size, bits, logits,
                     route_inds, (bf16_t*)route_scores, s, f16 != 0))
        return 1;
    QGemvArgs a = {};
    a.m[0] = QMat{(const uint32_t*)q_gate, (const bf16_t*)s_gate, (const bf16_t*)b_gate, inter};
    a.m[1] = QMat{(const uint32_t*)q_up, (const bf16_t*)s_up, (const bf16_t*)b_up, inter};
    a.m[0].sb = quant_find_sb(a.m[0].scales); a.m[1].sb = quant_find_sb(a.m[1].scales);
    a.N = inter; a.K = hidden; a.group = group_size;
    a.x = (const bf16_t*)x; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.out = gbuf;
    a.n_batch = slots; a.x_div = top_k; a.w_sel = route_inds;
    a.w_estride = (size_t)inter * hidden * bits / 32; a.s_estride = (size_t)inter * (hidden / group_size);
    a.swiglu_single_round = 1; a.scales_f16 = f16 != 0;
    if (launch_qgemv(a, bits, PRO_RMSNORM, EPI_SWIGLU, s)) return 1;
    QGemvArgs d = {};
    d.m[0] = QMat{(const uint32_t*)q_down, (const bf16_t*)s_down, (const bf16_t*)b_down, hidden};
    d.m[0].sb = quant_find_sb(d.m[0].scales);
    d.N = hidden; d.K = inter; d.group = group_size;
    d.x = gbuf; d.out_f32 = y_partial;
    d.n_batch = slots; d.x_div = 1; d.w_sel = route_inds;
    d.w_estride = (size_t)hidden * inter * bits / 32; d.s_estride = (size_t)hidden * (inter / group_size);
    d.scales_f16 = f16 != 0;
    return launch_qgemv(d, bits, PRO_NONE, EPI_F32, s);
}

namespace omx {
namespace {
// out[t] = bf16(resid[t] + bf16(sum_j bf16(bf16(y[t k + j]) * score[t k + j])))  -- moe_combine_kernel on f32 (all-reduced) slot outputs
template <bool F16 = false>
__global__ __launch_bounds__(256) void moe_combine_slots_kernel(bf16_t* __restrict__ out, const float* __restrict__ y, const bf16_t* __restrict__ scores,
                                                                const bf16_t* __restrict__ resid, int h, int k) {
    typedef Act16<F16> A16;
    const int t = blockIdx.x;
    for (int i = threadIdx.x; i < h; i += 256) {
        float acc = 0.f;
        for (int j = 0; j < k; ++j) acc += A16::rnd(A16::rnd(y[((size_t)t * k + j) * h + i]) * A16::val(scores[(size_t)t * k + j]));
        out[(size_t)t * h + i] = A16::bits(resid ? A16::val(resid[(size_t)t * h + i]) + A16::rnd(acc) : acc);
    }
}
}  // namespace
}  // namespace omx

extern "C" int omx_moe_combine_slots(void* out, const float* y_slots, const void* scores, const void* resid, int n_tokens, int hidden,
                                     int top_k, omx_stream stream) {
    OMX_REQUIRE(out && y_slots && scores && n_tokens >= 0 && hidden > 0 && top_k >= 1, "omx_moe_combine_slots: bad arguments");
    if (n_tokens == 0) return 0;
    omx::moe_combine_slots_kernel<false><<<n_tokens, 256, 0, (hipStream_t)stream>>>((omx::bf16_t*)out, y_slots, (const omx::bf16_t*)scores,
                                                                                    (const omx::bf16_t*)resid, hidden, top_k);
    OMX_LAUNCH_CHECK();
    return 0;
}

/* the same for a float16 checkpoint (f16 != 0): scores, residual and result float16, every rounding point float16 */
extern "C" int omx_moe_combine_slots_ex(void* out, const float* y_slots, const void* scores, const void* resid, int n_tokens, int hidden,
                                        int top_k, int f16, omx_stream stream) {
    if (!f16) return omx_moe_combine_slots(out, y_slots, scores, resid, n_tokens, hidden, top_k, stream);
    OMX_REQUIRE(out && y_slots && scores && n_tokens >= 0 && hidden > 0 && top_k >= 1, "omx_moe_combine_slots_ex: bad arguments");
    if (n_tokens == 0) return 0;
    omx::moe_combine_slots_kernel<true><<<n_tokens, 256, 0, (hipStream_t)stream>>>((omx::bf16_t*)out, y_slots, (const omx::bf16_t*)scores,
                                                                                   (const omx::bf16_t*)resid, hidden, top_k);
    OMX_LAUNCH_CHECK();
    return 0;
}

/* The reference's own Mixtral format: 4/8-bit expert stacks through gather_qmm (mixtral-mlx/src/model.rs:182-274).
 * q_* = packed u32 [E, out, in*bits/32]; s_* / b_* = scales / biases bf16 [E, out, in/group_size].  The router gate is bf16. */
extern "C" int omx_moe_forward_q(void* out, const void* x, const void* gate_w, const void* q_gate, const void* s_gate,
                                 const void* b_gate, const void* q_up, const void* s_up, const void* b_up, const void* q_down,
                                 const void* s_down, const void* b_down, int n_tokens, int hidden, int inter, int n_experts,
                                 int top_k, int mode, int norm_topk_prob, int group_size, int bits, uint32_t* inds_out,
                                 void* scores_out, omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(q_gate && s_gate && b_gate && q_up && s_up && b_up && q_down && s_down && b_down, "omx_moe_forward_q: null tensor");
    OMX_REQUIRE(bits == 4 || bits == 8, "omx_moe_forward_q: bits=%d (4 or 8)", bits);
    OMX_REQUIRE(group_size == 32 || group_size == 64 || group_size == 128, "omx_moe_forward_q: group_size=%d (32, 64, 128)", group_size);
    OMX_REQUIRE(hidden % 512 == 0 && inter % 512 == 0, "omx_moe_forward_q: hidden=%d and intermediate=%d must be multiples of 512", hidden, inter);
    QExperts q;
    q.gate = QMat{(const uint32_t*)q_gate, (const bf16_t*)s_gate, (const bf16_t*)b_gate, inter};
    q.up = QMat{(const uint32_t*)q_up, (const bf16_t*)s_up, (const bf16_t*)b_up, inter};
    q.down = QMat{(const uint32_t*)q_down, (const bf16_t*)s_down, (const bf16_t*)b_down, hidden};
    q.gate.sb = quant_find_sb(q.gate.scales); q.up.sb = quant_find_sb(q.up.scales); q.down.sb = quant_find_sb(q.down.scales);
    q.group = group_size; q.bits = bits;
    return moe_forward_impl(out, x, gate_w, nullptr, nullptr, nullptr, &q, n_tokens, hidden, inter, n_experts, top_k, mode,
                            norm_topk_prob, inds_out, scores_out, stream);
}

/* ---- the three stages of the block as separate entry points: what an expert-parallel host needs between its
 * all-to-all exchanges (ominix-mlx_amd/ep.py) ---- */
extern "C" int omx_moe_route(uint32_t* inds_out, void* scores_out, const void* x, const void* gate_w, int n_tokens,
                             int hidden, int n_experts, int top_k, int mode, int norm_topk_prob, omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(inds_out && scores_out && x && gate_w, "omx_moe_route: null tensor");
    OMX_REQUIRE(n_experts >= 1 && n_experts <= kMaxExperts && top_k >= 1 && top_k <= kMaxTopK && top_k <= n_experts,
                "omx_moe_route: n_experts=%d (max %d), top_k=%d (max %d)", n_experts, kMaxExperts, top_k, kMaxTopK);
    OMX_REQUIRE(hidden > 0 && hidden % 64 == 0, "omx_moe_route: hidden=%d must be a multiple of 64", hidden);
    OMX_REQUIRE(mode == 0 || mode == 1, "omx_moe_route: mode must be 0 (Mixtral) or 1 (Qwen3-MoE)");
    if (n_tokens <= 0) return 0;
    moe_router_kernel<<<n_tokens, router_threads(n_experts, hidden), 0, (hipStream_t)stream>>>((const bf16_t*)x, (const bf16_t*)gate_w, hidden, n_experts,
                                                                 top_k, mode, norm_topk_prob, inds_out, (bf16_t*)scores_out);
    OMX_LAUNCH_CHECK();
    return 0;
}

/* y[i] = down_e( fused_swiglu(up_e x[i], gate_e x[i]) ), e = expert_ids[i]: SwitchGLU::forward_experts on rows that
 * already carry their expert (mixtral-mlx/src/model.rs:243-274); expert_ids index the n_experts matrices passed in */
extern "C" int omx_moe_experts(void* y, const void* x_rows, const uint32_t* expert_ids, int n_rows, const void* w_gate,
                               const void* w_up, const void* w_down, int hidden, int inter, int n_experts,
                               omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(y && x_rows && expert_ids && w_gate && w_up && w_down, "omx_moe_experts: null tensor");
    OMX_REQUIRE(n_experts >= 1 && n_experts <= kMaxExperts, "omx_moe_experts: n_experts=%d (max %d)", n_experts, kMaxExperts);
    OMX_REQUIRE(hidden % 64 == 0 && inter % 64 == 0, "omx_moe_experts: hidden=%d and intermediate=%d must be multiples of 64", hidden, inter);
    if (n_rows <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int max_tiles = n_rows / 128 + n_experts + 1;
    size_t need = 0;
    omx_moe_workspace_bytes(n_rows, hidden, inter, n_experts, 1, &need);
    void* ws = nullptr;
    if (get_workspace(&ws, need)) return 1;
    char* p = (char*)ws;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    (void)take((size_t)n_rows * 4);
    uint32_t* row_src = (uint32_t*)take((size_t)n_rows * 4);
    uint32_t* pos_of_slot = (uint32_t*)take((size_t)n_rows * 4);
    (void)take((size_t)n_rows * 2);
    int* seg_start = (int*)take((size_t)(n_experts + 2) * 4);
    int* tile_expert = (int*)take((size_t)max_tiles * 4);
    int* tile_m0 = (int*)take((size_t)max_tiles * 4);
    int* n_tiles = (int*)take(256);
    bf16_t* gbuf = (bf16_t*)take((size_t)n_rows * inter * 2);
    bf16_t* ubuf = (bf16_t*)take((size_t)n_rows * inter * 2);
    bf16_t* ybuf = (bf16_t*)take((size_t)n_rows * hidden * 2);
    if (n_rows <= 32 && gemv_k_supported(hidden, false) && gemv_k_supported(inter, false)) {
        GemvArgs a = {};
        a.w0 = (const bf16_t*)w_gate; a.w1 = (const bf16_t*)w_up; a.n0 = inter; a.N = inter; a.K = hidden;
        a.x = (const bf16_t*)x_rows; a.out = gbuf;
        a.n_batch = n_rows; a.x_div = 1; a.x_bstride = hidden; a.out_bstride_bytes = (size_t)inter * 2;
        a.w_sel = expert_ids; a.w_estride = (size_t)inter * hidden; a.swiglu_single_round = 1;
        if (launch_gemv(a, PRO_NONE, EPI_SWIGLU, s)) return 1;
        GemvArgs d = {};
        d.w0 = (const bf16_t*)w_down; d.n0 = hidden; d.N = hidden; d.K = inter;
        d.x = gbuf; d.out = y;
        d.n_batch = n_rows; d.x_div = 1; d.x_bstride = inter; d.out_bstride_bytes = (size_t)hidden * 2;
        d.w_sel = expert_ids; d.w_estride = (size_t)hidden * inter;
        return launch_gemv(d, PRO_NONE, EPI_STORE, s);
    }
    const int tile_rows = moe_tile_rows(n_rows, n_experts, hidden, inter);
    moe_plan_kernel<<<1, 1024, 0, s>>>(expert_ids, n_rows, n_experts, 1, seg_start, row_src, pos_of_slot, tile_expert, tile_m0,
                                       n_tiles, tile_rows);
    OMX_LAUNCH_CHECK();
    GroupedDesc g;
    g.tile_expert = tile_expert; g.tile_m0 = tile_m0; g.seg_start = seg_start; g.n_tiles = n_tiles;
    g.row_src = row_src; g.w_estride = (size_t)inter * hidden;
    if (tile_rows == 256) {
        if (grouped_glu_256(ybuf, gbuf, (const bf16_t*)x_rows, (const bf16_t*)w_gate, (const bf16_t*)w_up, (const bf16_t*)w_down, n_rows,
                            hidden, inter, n_experts, g, s))
            return 1;
    } else {
        if (launch_gemm_bf16_grouped(gbuf, (const bf16_t*)x_rows, (const bf16_t*)w_gate, n_rows, inter, hidden, g, max_tiles, s)) return 1;
        if (launch_gemm_bf16_grouped(ubuf, (const bf16_t*)x_rows, (const bf16_t*)w_up, n_rows, inter, hidden, g, max_tiles, s)) return 1;
        if (omx_fused_swiglu(gbuf, ubuf, gbuf, (int64_t)n_rows * inter, OMX_BFLOAT16, stream)) return 1;
        g.row_src = nullptr;
        g.w_estride = (size_t)hidden * inter;
        if (launch_gemm_bf16_grouped(ybuf, gbuf, (const bf16_t*)w_down, n_rows, hidden, inter, g, max_tiles, s)) return 1;
    }
    moe_unsort_kernel<<<n_rows, 256, 0, s>>>((bf16_t*)y, ybuf, pos_of_slot, hidden);
    OMX_LAUNCH_CHECK();
    return 0;
}

/* out[t] = bf16( sum_j bf16( y[t, j] * scores[t, j] ) ): the weighted sum of model.rs:304-307 on slot-ordered rows */
extern "C" int omx_moe_combine(void* out, const void* y_slots, const void* scores, int n_tokens, int hidden, int top_k,
                               omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(out && y_slots && scores, "omx_moe_combine: null tensor");
    OMX_REQUIRE(hidden > 0 && hidden % 8 == 0 && top_k >= 1, "omx_moe_combine: bad shape");
    if (n_tokens <= 0) return 0;
    moe_combine_kernel<false><<<n_tokens, 256, 0, (hipStream_t)stream>>>((bf16_t*)out, (const bf16_t*)y_slots, (const bf16_t*)scores,
                                                                 nullptr, hidden, top_k);
    OMX_LAUNCH_CHECK();
    return 0;
}

extern "C" int omx_gather_mm(void* out, const void* x, const void* w, const uint32_t* rhs_indices, int n_rows, int x_div, int N, int K,
                             int n_experts, omx_dtype dtype, omx_stream stream) {
    using namespace omx;
    OMX_REQUIRE(out && x && w && rhs_indices, "omx_gather_mm: null tensor");
    OMX_REQUIRE(dtype == OMX_BFLOAT16, "omx_gather_mm: only bfloat16 is implemented (got dtype %d)", (int)dtype);
    OMX_REQUIRE(n_rows >= 0 && x_div >= 1 && N > 0 && n_experts >= 1 && gemv_k_supported(K, false), "omx_gather_mm: bad shape (N=%d K=%d)", N, K);
    if (n_rows == 0) return 0;
    GemvArgs a = {};
    a.w0 = (const bf16_t*)w; a.n0 = N; a.N = N; a.K = K;
    a.x = (const bf16_t*)x; a.out = out;
    a.n_batch = n_rows; a.x_div = x_div; a.x_bstride = (size_t)K; a.out_bstride_bytes = (size_t)N * 2;
    a.w_sel = rhs_indices; a.w_estride = (size_t)N * K;
    return launch_gemv(a, PRO_NONE, EPI_STORE, (hipStream_t)stream);
}
