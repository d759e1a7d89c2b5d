// Fused Qwen3 decode engine (see include/omx.h "Fused decode engine").
//
// Host-side mirror, in C++, of the caller of the hot path: qwen3-mlx's Model/Generate
// (qwen3-mlx/src/model.rs:387-433, 473-498, 743-844) driving mlx-rs-core's KVCache
// (mlx-rs-core/src/cache.rs:91-194).  The reference records ~1000 lazy graph nodes per token and
// lets MLX schedule them; here one decode step is 6 launches per layer captured once in a
// hipGraph and replayed per token, with the step state (position, current token, token ring)
// kept in device memory so that replay needs no host patching.
//
// HBM layout (288 GB part: everything resident, nothing paged):
//   weights      borrowed pointers, bf16 [out,in] row-major (checkpoint layout, nn/linear.rs)
//   KV cache     per layer K and V slabs [Hkv_local, cap, D] bf16, cap = max_context rounded up
//                to the 256-token step of cache.rs:110-117 (same API-visible growth semantics,
//                no reallocation + concatenate on growth)
//   rope tables  cos/sin [cap, D/2] f32, built once in fp64
//   step state   pos, cur_token, out ring, argmax partials
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "attn.hpp"
#include "workspace.hpp"
#include "step_state.hpp"
#include "gemm.hpp"
#include "gemv.hpp"
#include "random.hpp"
#include "prefill.hpp"
#include "quant.hpp"
#include "launch_timing.hpp"
#include "aql_step.hpp"
#include "act16.hpp"
#include "peer.hpp"
#include "step_engine.hpp"
#include <hip/hip_fp16.h>

namespace omx {
namespace {

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int kNcclFloat32 = 7, kNcclUint64 = 5, kNcclBfloat16 = 9, kNcclSum = 0, kNcclMax = 2;

struct LayerW {
    const bf16_t *q, *k, *v, *o, *gate, *up, *down, *q_norm, *k_norm, *in_ln, *post_ln;
    const bf16_t *moe_gate, *moe_wg, *moe_wu, *moe_wd;   // sparse-MoE feed-forward (router + stacked experts)
    const bf16_t *q_bias, *k_bias, *v_bias, *qkv_bias;   // Qwen2: projection biases; qkv_bias = the three concatenated (owned)
};
// quantized checkpoint (config.json "quantization", qwen3-mlx/src/model.rs:621-727): every Linear and the embedding are
// (weight u32, scales, biases) triplets; the norm weights stay bf16 in LayerW
struct LayerQ {
    QMat q, k, v, o, gate, up, down;
    QMat moe_router, moe_g, moe_u, moe_d;   // sparse-MoE feed-forward: quantised router and expert stacks
};

// dequantise ONE embedding row (QuantizedEmbedding::forward, mlx-rs/src/nn/quantized.rs:192-203) chosen by the step state
// the step's first kernel also refreshes what the rest of the step reads instead of chasing the position: the step sequence
// number (granule tags of attn_step.hip) and the RoPE row of the current position, rope_cur = cos[pos, :] | sin[pos, :]
__device__ __forceinline__ void step_begin(const StepState* st, unsigned* seq, float* rope_cur, const float* rope_cos,
                                           const float* rope_sin, int half) {
    if (blockIdx.x != 0) return;
    if (seq && threadIdx.x == 0) *seq += 1u;
    if (rope_cur && (int)threadIdx.x < 2 * half) {
        const int t = threadIdx.x, pos = st->pos;
        rope_cur[t] = t < half ? rope_cos[(size_t)pos * half + t] : rope_sin[(size_t)pos * half + t - half];
    }
}

template <int BITS>
__global__ __launch_bounds__(256) void qembed_kernel(bf16_t* __restrict__ h, const uint32_t* __restrict__ w,
                                                     const bf16_t* __restrict__ scales, const bf16_t* __restrict__ biases,
                                                     const StepState* st, int hidden, int group, unsigned* seq, float* rope_cur,
                                                     const float* rope_cos, const float* rope_sin, int half, bool scales_f16) {
    constexpr int EPW = 32 / BITS;
    step_begin(st, seq, rope_cur, rope_cos, rope_sin, half);
    const size_t row = st->cur_token;
    const int words = hidden / EPW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) {
        const uint32_t wd = w[row * words + i];
        const int g = i * EPW / group;
        const bf16_t sb16 = scales[row * (hidden / group) + g], bb16 = biases ? biases[row * (hidden / group) + g] : (bf16_t)0;
        const float sc = scales_f16 ? __half2float(__ushort_as_half(sb16)) : bf16_to_f32(sb16);
        const float bi = scales_f16 ? __half2float(__ushort_as_half(bb16)) : bf16_to_f32(bb16);
#pragma unroll
        for (int e = 0; e < EPW; ++e) {   // (dequantise: the result has the scales' dtype, quantized.rs:192-203)
            const float v = (float)((wd >> (e * BITS)) & ((1u << BITS) - 1u)) * sc + bi;
            h[i * EPW + e] = scales_f16 ? Act16<true>::bits(v) : f32_to_bf16(v);
        }
    }
}

// h = bf16(resid + bf16(all-reduced partial)): the residual of a block whose output arrives as an f32 sum over the ranks
// (f16: a float16 model -- the same two roundings in float16)
__global__ void ep_fold_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ resid, const float* __restrict__ partial, int64_t n, bool f16 = false) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (f16) out[i] = Act16<true>::bits(Act16<true>::val(resid[i]) + Act16<true>::rnd(partial[i]));
        else out[i] = f32_to_bf16(bf16_to_f32(resid[i]) + round_bf16(partial[i]));
    }
}

// a float16 partial product widened for the f32 all-reduce of a tensor-parallel float16 prompt pass
__global__ void f16_widen_kernel(float* __restrict__ out, const bf16_t* __restrict__ in, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = Act16<true>::val(in[i]);
}

uint32_t crc32_str(const char* s) {
    uint32_t crc = 0xFFFFFFFFu;
    for (; *s; ++s) {
        crc ^= (uint8_t)*s;
        for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    }
    return ~crc;
}

__global__ void rope_table_kernel(float* cos_t, float* sin_t, int cap, int half, double neg_log_base_over_half,
                                  double scale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cap * half) return;
    const int t = idx / half, i = idx % half;
    const double ang = ((double)t * scale) * exp((double)i * neg_log_base_over_half);
    double s, c;
    sincos(ang, &s, &c);
    cos_t[idx] = (float)c;
    sin_t[idx] = (float)s;
}

// step state updates (single thread; a few dozen ns of work, they only order the graph); StepState: step_state.hpp

__global__ void feed_prompt_kernel(StepState* st, const uint32_t* prompt) {
    // after a no-head prefill step: advance and feed the next prompt token
    st->pos += 1;
    st->prompt_idx += 1;
    st->cur_token = prompt[st->prompt_idx];
}

__global__ __launch_bounds__(256) void sample_finalize_kernel(StepState* st, const unsigned long long* partials,
                                                               int n_partials, uint32_t* out_ring, int ring_cap,
                                                               unsigned long long* key_out) {
    __shared__ unsigned long long red[4];
    unsigned long long best = 0;
    for (int i = threadIdx.x; i < n_partials; i += 256) best = partials[i] > best ? partials[i] : best;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) best = red[w] > best ? red[w] : best;
        if (key_out) {
            *key_out = best;   // TP: all-reduced (max) across ranks before apply_token_kernel
        } else {
            const uint32_t tok = ~(uint32_t)(best & 0xFFFFFFFFull);
            out_ring[st->out_count % ring_cap] = tok;
            st->out_count += 1;
            st->cur_token = tok;
            st->pos += 1;
        }
    }
}

__global__ void apply_token_kernel(StepState* st, const unsigned long long* key, uint32_t* out_ring, int ring_cap) {
    const uint32_t tok = ~(uint32_t)(*key & 0xFFFFFFFFull);
    out_ring[st->out_count % ring_cap] = tok;
    st->out_count += 1;
    st->cur_token = tok;
    st->pos += 1;
}

__global__ __launch_bounds__(256) void embed_kernel(bf16_t* __restrict__ h, const bf16_t* __restrict__ table,
                                                    const StepState* st, int hidden, unsigned* seq, float* rope_cur,
                                                    const float* rope_cos, const float* rope_sin, int half) {
    step_begin(st, seq, rope_cur, rope_cos, rope_sin, half);
    const u32x4* src = reinterpret_cast<const u32x4*>(table + (size_t)st->cur_token * hidden);
    u32x4* dst = reinterpret_cast<u32x4*>(h);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hidden / 8; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

}  // namespace
}  // namespace omx

using namespace omx;

struct omx_qwen3_ {
    omx_qwen3_config cfg;
    int H, Hkv, I, V;            // local (per-rank) heads / intermediate / vocab
    int cap;                     // KV slab capacity in tokens
    std::map<std::string, const void*> named;
    std::vector<void*> owned;    // allocations made by synth_weights
    std::vector<LayerW> layers;
    std::vector<LayerQ> qlayers;             // quantized mode (cfg.quant_bits != 0)
    QMat q_embed = {}, q_head = {};
    std::vector<const bf16_t*> sb_keys;   // scales pointers registered with quant_register_sb
    bf16_t* dq_buf = nullptr;                // dequantised weight of the GEMM in flight (batched prefill)
    size_t dq_cap = 0;
    // dequantised copies of the layers' packed matrices kept BETWEEN prompts (round 4): 288 GB of HBM hold a dense 8B model's 14 GB of
    // them next to the packed weights, and every prompt after the first skips the dequantise launches (key: the packed words)
    std::map<const uint32_t*, bf16_t*> dq_cache;
    char* dq_slab = nullptr;                 // ONE allocation for all of them (252 hipMallocs between the launches cost a first prompt up to 240 ms)
    size_t dq_slab_bytes = 0, dq_cache_bytes = 0;
    int dq_cache_mode = -1;                  // -1 undecided, 0 off, 1 on
    const bf16_t *embed = nullptr, *final_norm = nullptr, *lm_head = nullptr;
    bool weights_resolved = false;

    hipStream_t stream = nullptr;
    std::vector<bf16_t*> kcache, vcache;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    StepState* st = nullptr;
    uint32_t *out_ring = nullptr, *prompt_dev = nullptr;
    int ring_cap = 4096, prompt_cap = 0;
    bf16_t *h = nullptr, *h2 = nullptr, *qkv = nullptr, *attn_out = nullptr, *act = nullptr, *logits = nullptr;
    bf16_t *moe_xn = nullptr, *moe_out = nullptr;   // MoE feed-forward: normalised input row, block output
    float *partial_a = nullptr, *partial_b = nullptr;   // TP: f32 partial sums awaiting all-reduce
    float* moe_partials = nullptr;                      // MoE decode: [top_k, hidden] weighted expert outputs awaiting the next GEMV's fold
    // expert TENSOR parallel (tp_size > 1 with experts): every expert's intermediate columns sharded over the ranks
    int moe_I = 0;                                      // per-rank expert intermediate width
    float* moe_y = nullptr;                             // [top_k, hidden] f32 partial down projections of the routed slots (all-reduced)
    uint32_t* moe_inds = nullptr;                       // the replicated router's choice, kept for the combine after the all-reduce
    bf16_t* moe_scores = nullptr;
    unsigned long long *argmax_partials = nullptr, *argmax_key = nullptr;
    int n_argmax_partials = 0;
    unsigned *step_seq = nullptr, *wait_abort = nullptr;   // step sequence number (granule tags), word a gather that gave up raises
    // attention of the decode step (attn_step.hip): the split plan is fixed per captured graph and covers positions < graph_tk_max;
    // the graphs are rebuilt when the context outgrows that bucket
    float* rope_cur = nullptr;            // [D] cos | sin of the current position
    uint64_t* attn_gran = nullptr;        // split partials as tagged granules
    uint64_t* attn_xg = nullptr;          // the merged attention vector as granules (O projection in the attention launch)
    int attn_chunk = 0, attn_nsplit = 0, graph_tk_max = 0;
    unsigned long long* attn_trace = nullptr;   // set for one eager step by omx_qwen3_debug_trace_step
    bf16_t* verify_logits = nullptr;            // [verify_cap, V]: every row's logits of the last omx_qwen3_verify
    uint32_t* verify_tokens = nullptr;
    int verify_cap = 0, verify_rows = 0;
    std::vector<hipEvent_t>* kernel_events = nullptr;   // set for eager steps by omx_qwen3_time_step_kernels: [layer][class][begin, end]

    void* comm = nullptr;
    nccl_allreduce_fn allreduce = nullptr;
    const PeerDev* peer_dev = nullptr;   // the communicator is a peer-store one (peer_allreduce.hip): O / down reduce inside their GEMV

    // sampler (sampler.rs:9-18): 0 = greedy; otherwise categorical(logits / temperature) with the key sequence
    // of mlx-rs RandomState kept on the device: rng[0..1] = state, rng[2..3] = the key of the current draw
    float temperature = 0.f;
    uint32_t* rng = nullptr;

    // batched-prefill activations (allocated on first use, sized for pf_cap tokens)
    int pf_cap = 0;
    bf16_t *pf_h = nullptr, *pf_h2 = nullptr, *pf_xn = nullptr, *pf_q = nullptr, *pf_k = nullptr, *pf_v = nullptr,
           *pf_qt = nullptr, *pf_attn = nullptr, *pf_g = nullptr, *pf_u = nullptr;
    float* pf_ep_partial = nullptr;      // expert-parallel batched prefill: [pf_ep_cap, hidden] f32 partial of the MoE block
    int pf_ep_cap = 0;
    float last_prefill_ms = 0.f;

    // persistent decode step (step_engine.hip): the layers of a token in one launch
    int cus = 0;                               // compute units of the device: one resident workgroup each
    std::vector<StepEngineLayer> se_layers_host;
    StepEngineLayer* se_layers = nullptr;
    uint64_t* se_gran = nullptr;               // granule buffers of the five vector edges
    unsigned long long* se_trace = nullptr;    // set for one eager step by omx_qwen3_debug_trace_engine
    bool se_disabled = false;                  // a step gave up waiting (a workgroup was not resident): back to one launch per op
    bool oproj_disabled = false;               // the same for the O projection inside the attention launch

    hipGraphExec_t g_full = nullptr, g_nohead = nullptr;
    AqlProgram* aql_full = nullptr;            // the with-head step as AQL packets on the engine's own HSA queue (aql_step.hpp)
    bool aql_disabled = false;                 // building or replaying it failed once: hipGraph from then on
    bool eager = false;          // fallback when stream capture is unavailable (e.g. a collective refuses capture)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_decode_ms = 0.f;
};

namespace {

// the captured forms of the step (hipGraph executables, the AQL program) hold the split plan and every pointer: dropped together
void drop_graphs(omx_qwen3 m) {
    if (m->g_full) { (void)hipGraphExecDestroy(m->g_full); m->g_full = nullptr; }
    if (m->g_nohead) { (void)hipGraphExecDestroy(m->g_nohead); m->g_nohead = nullptr; }
    if (m->aql_full) { aql_destroy(m->aql_full); m->aql_full = nullptr; }
    if (m->cfg.num_experts > 0 && (m->cfg.ep_size > 1 || m->cfg.tp_size > 1) && m->stream) workspace_aux_pin(m->stream, false);
}

template <class T>
int dev_alloc(omx_qwen3 m, T** p, size_t n) {
    void* q = nullptr;
    OMX_HIP_CHECK(hipMalloc(&q, n * sizeof(T) + 64));
    // same stream as every later writer: a null-stream hipMemset is not ordered against the
    // engine's non-blocking stream and could zero a buffer after it was filled
    OMX_HIP_CHECK(hipMemsetAsync(q, 0, n * sizeof(T) + 64, m->stream));
    *p = (T*)q;
    m->owned.push_back(q);
    return 0;
}

// temperature sampling: replace the greedy per-block partials by those of logits / T + Gumbel noise, the noise
// of vocabulary row v being word v of a V-word draw from the step's key (random.hip).  The greedy partials the
// lm_head epilogue wrote are simply overwritten; sample_finalize_kernel / the TP max all-reduce are unchanged.
int add_sampling_noise(omx_qwen3 m, hipStream_t s) {
    if (m->temperature == 0.f) return 0;
    const omx_qwen3_config& c = m->cfg;
    const int tp = c.tp_size > 1 ? c.tp_size : 1;
    if (launch_rng_next(m->rng, s)) return 1;
    return launch_sample_noise(m->argmax_partials, m->n_argmax_partials, m->logits, m->rng + 2, m->V, c.tp_rank * m->V,
                               m->V * tp, 1.0f / m->temperature, c.quant_scales_f16 != 0, s);
}

int resolve_weights(omx_qwen3 m) {
    if (m->weights_resolved) return 0;
    if (!m->dq_cache.empty()) {   // the weights changed under the dequantised copies of the prompt pass
        (void)hipStreamSynchronize(m->stream);
        m->dq_cache.clear();          // (the slab stays: the same shapes come back)
        m->dq_cache_bytes = 0;
    }
    auto get = [&](const std::string& n, const bf16_t** out) -> int {
        auto it = m->named.find(n);
        if (it == m->named.end()) return set_error("WeightNotFound: %s", n.c_str());   // error.rs:6-32
        *out = (const bf16_t*)it->second;
        return 0;
    };
    m->layers.resize(m->cfg.num_hidden_layers);
    if (m->cfg.quant_bits) {
        const int D = m->cfg.head_dim, hd = m->cfg.hidden_size;
        const bool interleave = !(getenv("OMX_QUANT_INTERLEAVE") && getenv("OMX_QUANT_INTERLEAVE")[0] == '0');
        // K: contraction width; stack: matrices stacked in the tensor (experts)
        auto getq = [&](const std::string& prefix, int n, QMat* out, int K = 0, int stack = 1) -> int {
            const bf16_t *w = nullptr, *sc = nullptr, *bi = nullptr;
            if (get(prefix + ".weight", &w) || get(prefix + ".scales", &sc) || get(prefix + ".biases", &bi)) return 1;
            *out = QMat{(const uint32_t*)w, sc, bi, n};
            if (interleave && K > 0 && K % 2048 == 0) {   // (scale, bias) words for the packed-weight GEMV (quant.hpp)
                const size_t ng = (size_t)stack * n * (K / m->cfg.quant_group);
                uint32_t* sb = nullptr;
                if (dev_alloc(m, &sb, ng) || launch_quant_interleave(sb, sc, bi, ng, m->stream)) return 1;
                out->sb = sb;
                quant_register_sb(sc, sb);
                m->sb_keys.push_back(sc);
            }
            // the dense decode step's matrices once more as matrix-core tiles (qgemv_mfma.hip; OMX_QGEMV_MFMA=0: the VALU kernel only)
            const char* mfma_env = getenv("OMX_QGEMV_MFMA");        // (read per model: tests compare the two kernels in one process)
            const bool tiles_off = mfma_env && mfma_env[0] == '0';
            if (!tiles_off && stack == 1 && K > 0 && !m->cfg.quant_scales_f16 && qgemv4m_shape_ok(K, m->cfg.quant_group, m->cfg.quant_bits)) {
                uint32_t* tiles = nullptr;
                if (dev_alloc(m, &tiles, qgemv4m_tile_words(n, K)) || launch_qgemv4m_repack(tiles, (const uint32_t*)w, sc, bi, n, K, m->stream)) return 1;
                out->tiles = tiles;
            }
            return 0;
        };
        m->qlayers.resize(m->cfg.num_hidden_layers);
        for (int i = 0; i < m->cfg.num_hidden_layers; ++i) {
            const std::string p = "model.layers." + std::to_string(i) + ".";
            LayerW& L = m->layers[i];
            LayerQ& Q = m->qlayers[i];
            L = LayerW{};
            if (getq(p + "self_attn.q_proj", m->H * D, &Q.q, hd) || getq(p + "self_attn.k_proj", m->Hkv * D, &Q.k, hd) ||
                getq(p + "self_attn.v_proj", m->Hkv * D, &Q.v, hd) || getq(p + "self_attn.o_proj", hd, &Q.o, m->H * D) ||
                get(p + "input_layernorm.weight", &L.in_ln) || get(p + "post_attention_layernorm.weight", &L.post_ln))
                return 1;
            if (!m->cfg.no_qk_norm && (get(p + "self_attn.q_norm.weight", &L.q_norm) || get(p + "self_attn.k_norm.weight", &L.k_norm))) return 1;
            if (m->cfg.num_experts > 0) {
                const std::string mp = p + (m->cfg.moe_mode == 0 ? "block_sparse_moe." : "mlp.");
                // (expert tensor parallel: this rank's columns of every expert; expert parallel: this rank's experts)
                const int Im = m->cfg.tp_size > 1 ? m->moe_I : m->cfg.moe_intermediate_size;
                const int E = m->cfg.num_experts, El = m->cfg.ep_size > 1 ? E / m->cfg.ep_size : E;
                if (getq(mp + "gate", E, &Q.moe_router, hd) || getq(mp + "switch_mlp.gate_proj", Im, &Q.moe_g, hd, El) ||
                    getq(mp + "switch_mlp.up_proj", Im, &Q.moe_u, hd, El) || getq(mp + "switch_mlp.down_proj", hd, &Q.moe_d, Im, El))
                    return 1;
            } else if (getq(p + "mlp.gate_proj", m->I, &Q.gate, hd) || getq(p + "mlp.up_proj", m->I, &Q.up, hd) || getq(p + "mlp.down_proj", hd, &Q.down, m->I)) {
                return 1;
            }
        }
        if (getq("model.embed_tokens", m->cfg.vocab_size, &m->q_embed) || get("model.norm.weight", &m->final_norm)) return 1;
        if (m->cfg.tie_word_embeddings && m->cfg.tp_size <= 1) m->q_head = m->q_embed;   // QuantizedEmbedding::as_linear (quantized.rs:166-180)
        else if (getq("lm_head", m->V, &m->q_head, hd)) return 1;                        // (tied under TP: the caller registers the table's vocabulary shard as lm_head.*)
        m->weights_resolved = true;
        return 0;
    }
    for (int i = 0; i < m->cfg.num_hidden_layers; ++i) {
        const std::string p = "model.layers." + std::to_string(i) + ".";
        LayerW& L = m->layers[i];
        L = LayerW{};
        if (get(p + "self_attn.q_proj.weight", &L.q) || get(p + "self_attn.k_proj.weight", &L.k) ||
            get(p + "self_attn.v_proj.weight", &L.v) || get(p + "self_attn.o_proj.weight", &L.o) ||
            get(p + "input_layernorm.weight", &L.in_ln) || get(p + "post_attention_layernorm.weight", &L.post_ln))
            return 1;
        if (!m->cfg.no_qk_norm && (get(p + "self_attn.q_norm.weight", &L.q_norm) || get(p + "self_attn.k_norm.weight", &L.k_norm))) return 1;
        if (m->cfg.attention_bias) {   // qwen2.rs:112-124: Linear with bias for q/k/v only
            if (get(p + "self_attn.q_proj.bias", &L.q_bias) || get(p + "self_attn.k_proj.bias", &L.k_bias) || get(p + "self_attn.v_proj.bias", &L.v_bias)) return 1;
            const size_t nq = (size_t)m->H * m->cfg.head_dim, nk = (size_t)m->Hkv * m->cfg.head_dim;
            bf16_t* cat = nullptr;
            if (dev_alloc(m, &cat, nq + 2 * nk)) return 1;
            OMX_HIP_CHECK(hipMemcpyAsync(cat, L.q_bias, nq * 2, hipMemcpyDeviceToDevice, m->stream));
            OMX_HIP_CHECK(hipMemcpyAsync(cat + nq, L.k_bias, nk * 2, hipMemcpyDeviceToDevice, m->stream));
            OMX_HIP_CHECK(hipMemcpyAsync(cat + nq + nk, L.v_bias, nk * 2, hipMemcpyDeviceToDevice, m->stream));
            L.qkv_bias = cat;
        }
        if (m->cfg.num_experts > 0) {
            const std::string mp = p + (m->cfg.moe_mode == 0 ? "block_sparse_moe." : "mlp.");
            if (get(mp + "gate.weight", &L.moe_gate) || get(mp + "switch_mlp.gate_proj.weight", &L.moe_wg) ||
                get(mp + "switch_mlp.up_proj.weight", &L.moe_wu) || get(mp + "switch_mlp.down_proj.weight", &L.moe_wd))
                return 1;
        } else if (get(p + "mlp.gate_proj.weight", &L.gate) || get(p + "mlp.up_proj.weight", &L.up) || get(p + "mlp.down_proj.weight", &L.down)) {
            return 1;
        }
    }
    if (get("model.embed_tokens.weight", &m->embed) || get("model.norm.weight", &m->final_norm)) return 1;
    if (m->cfg.tie_word_embeddings) {
        // tied head = Embedding::as_linear (model.rs:485-488); under TP the caller registers the vocab shard
        auto it = m->named.find("lm_head.weight");
        m->lm_head = it != m->named.end() ? (const bf16_t*)it->second : m->embed;
        OMX_REQUIRE(m->cfg.tp_size == 1 || it != m->named.end(), "tied lm_head under TP needs a vocab shard registered as lm_head.weight");
    } else if (get("lm_head.weight", &m->lm_head)) {
        return 1;
    }
    if (m->cfg.num_experts == 0) {   // layer table of the persistent step (step_engine.hip)
        m->se_layers_host.resize(m->cfg.num_hidden_layers);
        for (int i = 0; i < m->cfg.num_hidden_layers; ++i) {
            const LayerW& L = m->layers[i];
            m->se_layers_host[i] = StepEngineLayer{L.q, L.k, L.v, L.o, L.gate, L.up, L.down, L.in_ln, L.post_ln, L.q_norm, L.k_norm,
                                                   m->kcache[i], m->vcache[i]};
        }
        if (!m->se_layers && dev_alloc(m, &m->se_layers, m->se_layers_host.size())) return 1;
        OMX_HIP_CHECK(hipMemcpyAsync(m->se_layers, m->se_layers_host.data(), m->se_layers_host.size() * sizeof(StepEngineLayer),
                                     hipMemcpyHostToDevice, m->stream));
    }
    m->weights_resolved = true;
    return 0;
}

// O projection inside the attention launch (csrc/attn_step.hip): bf16 weights, single rank, a shape the kernel has a register layout
// for; OMX_ATTN_OPROJ=0 keeps the two launches
bool attention_takes_oproj(omx_qwen3 m) {
    const char* e = getenv("OMX_ATTN_OPROJ");          // (read per call: tests flip it between engines of one process)
    const bool off = e && e[0] == '0';
    const omx_qwen3_config& c = m->cfg;
    // (tensor parallel: the rank's heads and columns -- the launch then leaves the f32 partial for the all-reduce)
    // every block of that launch waits on others: all Hkv * nsplit of them must be resident, one per CU
    if (off || m->oproj_disabled || m->Hkv * m->attn_nsplit > m->cus) return false;
    if (c.quant_bits == 4)   // 4-bit checkpoint: the packed O matrix with its interleaved scale | bias words (built at load for K % 2048 == 0)
        return c.ep_size <= 1 && c.tp_size <= 1 && !c.quant_scales_f16 && !m->qlayers.empty() && m->qlayers[0].o.sb != nullptr &&
               attn_step_oproj_q4_ok(m->H, m->Hkv, c.head_dim, m->attn_nsplit, c.hidden_size, c.quant_group);
    return c.quant_bits == 0 && attn_step_oproj_ok(m->H, m->Hkv, c.head_dim, m->attn_nsplit, c.hidden_size);
}

// OMX_PEER_FUSED: a GEMV whose blocks poll their peers' stores must be resident as a whole -- a block waiting for a peer's row while
// that peer's matching block waits for a free CU behind OUR unscheduled blocks never finishes (it gives up after 2^23 polls and voids
// the step).  The streaming kernels hold at least two 4-wave blocks per CU (<= 256 VGPRs, a few KB of LDS); beyond that the standalone
// all-reduce launch follows the GEMV as usual.
bool peer_fused_fits(omx_qwen3 m, int N, int K) {
    return m->peer_dev != nullptr && gemv_grid(N, K, EPI_F32, 0) <= 2 * m->cus;
}

// the attention launch of layer l of a decode step (both the bf16 and the packed-weight step use the bf16 KV kernels);
// resid / out != null: the layer's O projection + residual rides in the same launch (attention_takes_oproj)
int enqueue_attention(omx_qwen3 m, int l, hipStream_t s, const bf16_t* resid = nullptr, bf16_t* out = nullptr, float* out_f32 = nullptr) {
    const omx_qwen3_config& c = m->cfg;
    const int D = c.head_dim;
    const LayerW& L = m->layers[l];
    {
        AttnStepArgs a = {};
        a.qkv = m->qkv;
        a.k = m->kcache[l]; a.v = m->vcache[l];
        a.kv_head_stride = (int64_t)m->cap * D;
        a.H = m->H; a.Hkv = m->Hkv; a.cap = m->cap;
        a.scale = 1.0f / sqrtf((float)D);
        a.eps = c.rms_norm_eps;
        a.q_norm_w = L.q_norm; a.k_norm_w = L.k_norm;
        a.rope_cur = m->rope_cur;
        a.pos_ptr = &m->st->pos;
        a.seq_ptr = m->step_seq;
        a.tag_mul = (unsigned)c.num_hidden_layers; a.tag_add = (unsigned)l + 1u;
        a.chunk = m->attn_chunk; a.nsplit = m->attn_nsplit;
        a.ws = m->attn_gran;
        a.out = m->attn_out;
        a.abort_flag = m->wait_abort;
        a.trace = m->attn_trace ? m->attn_trace + (size_t)l * m->attn_nsplit * m->Hkv * 8 : nullptr;
        a.f16 = c.quant_scales_f16;   // a float16 checkpoint runs in float16 end to end
        if (resid && (out || out_f32)) {
            a.o_resid = resid; a.o_out = out; a.o_out_f32 = out_f32; a.o_rows = c.hidden_size; a.xg = m->attn_xg;
            if (c.quant_bits) {
                const QMat& o = m->qlayers[l].o;
                a.o_wq = o.w; a.o_sb = o.sb; a.o_group = c.quant_group;
            } else {
                a.o_w = L.o;
            }
        }
        return launch_attn_step(a, D, s);
    }
}

// the same step on a quantized checkpoint: packed-weight GEMVs (quant.hip) with the prologues / epilogues of the bf16 step
int enqueue_step_quant(omx_qwen3 m, bool with_head) {
    const omx_qwen3_config& c = m->cfg;
    hipStream_t s = m->stream;
    const int hd = c.hidden_size, D = c.head_dim, bits = c.quant_bits, group = c.quant_group;
    const bool sf16 = c.quant_scales_f16 != 0;
    if (bits == 4) OMX_LAUNCH(qembed_kernel<4>, 4, 256, 0, s, m->h, m->q_embed.w, m->q_embed.scales, m->q_embed.biases, m->st, hd, group, m->step_seq,
                              m->rope_cur, m->rope_cos, m->rope_sin, D / 2, sf16);
    else OMX_LAUNCH(qembed_kernel<8>, 4, 256, 0, s, m->h, m->q_embed.w, m->q_embed.scales, m->q_embed.biases, m->st, hd, group, m->step_seq, m->rope_cur,
                    m->rope_cos, m->rope_sin, D / 2, sf16);
    OMX_LAUNCH_CHECK();
    bf16_t* h = m->h;
    bf16_t* hn = m->h2;
    // tensor parallel (round 4): q / k / v / gate / up / lm_head are this rank's packed rows, o / down its K slices (whole groups) --
    // their unrounded f32 row sums are all-reduced and folded into the residual by a launch of their own (the bf16 step folds them in
    // the next GEMV's prologue: the packed kernels' prologues are left alone)
    const bool tp = c.ep_size <= 1 && (c.tp_size > 1 || m->allreduce != nullptr);
    auto reduce_fold = [&](float* partial) -> int {
        OMX_REQUIRE(m->allreduce != nullptr, "tp_size > 1 but no communicator set (omx_qwen3_set_comm)");
        OMX_REQUIRE(m->allreduce(partial, partial, hd, kNcclFloat32, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
        ep_fold_kernel<<<8, 256, 0, s>>>(hn, h, partial, (int64_t)hd, sf16);
        OMX_LAUNCH_CHECK();
        bf16_t* t = h; h = hn; hn = t;
        return 0;
    };
    for (int l = 0; l < c.num_hidden_layers; ++l) {
        const LayerW& L = m->layers[l];
        const LayerQ& Q = m->qlayers[l];
        {   // [RMSNorm + QKV]
            QGemvArgs a = {};
            a.m[0] = Q.q; a.m[1] = Q.k; a.m[2] = Q.v;
            a.N = (m->H + 2 * m->Hkv) * D; a.K = hd; a.group = group;
            a.x = h; a.norm_w = L.in_ln; a.eps = c.rms_norm_eps; a.out = m->qkv; a.scales_f16 = sf16;
            if (launch_qgemv(a, bits, PRO_RMSNORM, EPI_STORE, s)) return 1;
        }
        const bool fused_o = !tp && attention_takes_oproj(m);
        // [q/k RMSNorm + RoPE + cache append + split-KV SDPA + merge]: the bf16 kernel (+ [O + residual] on the packed matrix)
        if (enqueue_attention(m, l, s, fused_o ? h : nullptr, fused_o ? hn : nullptr)) return 1;
        if (fused_o) {
            bf16_t* t = h; h = hn; hn = t;
        } else if (tp) {   // [O partial] [all-reduce] [+ residual]
            QGemvArgs a = {};
            a.m[0] = Q.o; a.N = hd; a.K = m->H * D; a.group = group;
            a.x = m->attn_out; a.out_f32 = m->partial_a; a.scales_f16 = sf16;
            if (launch_qgemv(a, bits, PRO_NONE, EPI_F32, s) || reduce_fold(m->partial_a)) return 1;
        } else {   // [O + residual]
            QGemvArgs a = {};
            a.m[0] = Q.o; a.N = hd; a.K = m->H * D; a.group = group;
            a.x = m->attn_out; a.resid = h; a.out = hn; a.scales_f16 = sf16;
            if (launch_qgemv(a, bits, PRO_NONE, EPI_RESIDUAL, s)) return 1;
            bf16_t* t = h; h = hn; hn = t;
        }
        if (c.num_experts > 0 && c.tp_size > 1) {
            // expert tensor parallel on packed stacks (round 5): [replicated packed router + this rank's columns of the routed experts]
            // [all-reduce of the slots' f32 partials] [weighted sum + residual with the single-device roundings]
            OMX_REQUIRE(m->allreduce != nullptr, "tp_size > 1 but no communicator set (omx_qwen3_set_comm)");
            if (omx_moe_block_partial_tp_q(m->moe_y, m->moe_inds, m->moe_scores, h, L.post_ln, c.rms_norm_eps, Q.moe_router.w, Q.moe_router.scales,
                                           Q.moe_router.biases, Q.moe_g.w, Q.moe_g.scales, Q.moe_g.biases, Q.moe_u.w, Q.moe_u.scales, Q.moe_u.biases,
                                           Q.moe_d.w, Q.moe_d.scales, Q.moe_d.biases, 1, hd, m->moe_I, c.num_experts, c.num_experts_per_tok, c.moe_mode,
                                           c.norm_topk_prob, group, bits, sf16 ? 1 : 0, s))
                return 1;
            OMX_REQUIRE(m->allreduce(m->moe_y, m->moe_y, (size_t)c.num_experts_per_tok * hd, kNcclFloat32, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
            if (omx_moe_combine_slots_ex(hn, m->moe_y, m->moe_scores, h, 1, hd, c.num_experts_per_tok, sf16 ? 1 : 0, s)) return 1;
            bf16_t* t = h; h = hn; hn = t;
            continue;
        }
        if (c.num_experts > 0 && c.ep_size > 1) {
            // expert parallel on packed stacks (round 5): this rank's experts only, the f32 partial all-reduced and folded into the residual
            const int el = c.num_experts / c.ep_size;
            if (omx_moe_block_partial_ep_q(m->partial_b, h, L.post_ln, c.rms_norm_eps, m->moe_xn, Q.moe_router.w, Q.moe_router.scales,
                                           Q.moe_router.biases, Q.moe_g.w, Q.moe_g.scales, Q.moe_g.biases, Q.moe_u.w, Q.moe_u.scales, Q.moe_u.biases,
                                           Q.moe_d.w, Q.moe_d.scales, Q.moe_d.biases, 1, hd, c.moe_intermediate_size, c.num_experts,
                                           c.num_experts_per_tok, c.moe_mode, c.norm_topk_prob, c.ep_rank * el, el, group, bits, sf16 ? 1 : 0, s))
                return 1;
            if (reduce_fold(m->partial_b)) return 1;
            continue;
        }
        if (c.num_experts > 0) {   // [RMSNorm + router] [selection] [RMSNorm + expert gate/up + SwiGLU] [expert down] [sum + residual]
            if (omx_moe_block_forward_q_ex(hn, h, h, L.post_ln, c.rms_norm_eps, m->moe_xn, Q.moe_router.w, Q.moe_router.scales,
                                           Q.moe_router.biases, Q.moe_g.w, Q.moe_g.scales, Q.moe_g.biases, Q.moe_u.w, Q.moe_u.scales,
                                           Q.moe_u.biases, Q.moe_d.w, Q.moe_d.scales, Q.moe_d.biases, 1, hd, c.moe_intermediate_size,
                                           c.num_experts, c.num_experts_per_tok, c.moe_mode, c.norm_topk_prob, group, bits, sf16 ? 1 : 0, s))
                return 1;
            bf16_t* t = h; h = hn; hn = t;
            continue;
        }
        {   // [RMSNorm + gate/up + SwiGLU]
            QGemvArgs a = {};
            a.m[0] = Q.gate; a.m[1] = Q.up; a.N = m->I; a.K = hd; a.group = group;
            a.x = h; a.norm_w = L.post_ln; a.eps = c.rms_norm_eps; a.out = m->act; a.scales_f16 = sf16;
            if (launch_qgemv(a, bits, PRO_RMSNORM, EPI_SWIGLU, s)) return 1;
        }
        if (tp) {   // [down partial] [all-reduce] [+ residual]
            QGemvArgs a = {};
            a.m[0] = Q.down; a.N = hd; a.K = m->I; a.group = group;
            a.x = m->act; a.out_f32 = m->partial_b; a.scales_f16 = sf16;
            if (launch_qgemv(a, bits, PRO_NONE, EPI_F32, s) || reduce_fold(m->partial_b)) return 1;
        } else {   // [down + residual]
            QGemvArgs a = {};
            a.m[0] = Q.down; a.N = hd; a.K = m->I; a.group = group;
            a.x = m->act; a.resid = h; a.out = hn; a.scales_f16 = sf16;
            if (launch_qgemv(a, bits, PRO_NONE, EPI_RESIDUAL, s)) return 1;
            bf16_t* t = h; h = hn; hn = t;
        }
    }
    if (with_head) {
        QGemvArgs a = {};
        a.m[0] = m->q_head; a.m[0].n = m->V; a.N = m->V; a.K = hd; a.group = group;
        a.x = h; a.norm_w = m->final_norm; a.eps = c.rms_norm_eps; a.out = m->logits; a.scales_f16 = sf16;
        a.argmax_slot = m->argmax_partials;
        a.row_offset = c.tp_rank * m->V;                 // this rank's vocabulary shard
        if (launch_qgemv(a, bits, PRO_RMSNORM, EPI_ARGMAX, s)) return 1;
        if (add_sampling_noise(m, s)) return 1;
        if (tp) {   // the ranks' packed (logit, index) keys: one unsigned max picks the token (enqueue_step_tail)
            sample_finalize_kernel<<<1, 256, 0, s>>>(m->st, m->argmax_partials, m->n_argmax_partials, m->out_ring, m->ring_cap, m->argmax_key);
            OMX_LAUNCH_CHECK();
            OMX_REQUIRE(m->allreduce(m->argmax_key, m->argmax_key, 1, kNcclUint64, kNcclMax, m->comm, s) == 0, "ncclAllReduce failed");
            apply_token_kernel<<<1, 1, 0, s>>>(m->st, m->argmax_key, m->out_ring, m->ring_cap);
            OMX_LAUNCH_CHECK();
        } else {
        OMX_LAUNCH(sample_finalize_kernel, 1, 256, 0, s, m->st, m->argmax_partials, m->n_argmax_partials, m->out_ring, m->ring_cap,
                   (unsigned long long*)nullptr);
        OMX_LAUNCH_CHECK();
        }
    } else {
        feed_prompt_kernel<<<1, 1, 0, s>>>(m->st, m->prompt_dev);
        OMX_LAUNCH_CHECK();
    }
    return 0;
}

// The layers of the step as ONE persistent launch (csrc/step_engine.hip): dense bf16 model on a single rank, a shape the engine's
// consumers reproduce bit for bit, every CU of the device free for one resident workgroup.  OMX_STEP_ENGINE=0 keeps one launch per op.
// OMX_STEP_ENGINE=2: the hybrid step -- attention + o stay their own launch (attn_step.hip: a chain of three all-to-all hand-offs that a
// launch boundary serves as well as granules do), everything between two attention launches ([gate/up] [down] [next layer's q/k/v]) is one
// segment of the engine: two launches per layer instead of four, the weight stream of three ops crossing its two edges without a stop.
int step_engine_mode(omx_qwen3 m) {
    const char* e = getenv("OMX_STEP_ENGINE");         // (read per call: tests flip it between engines of one process)
    const int mode = e ? atoi(e) : 0;
    const omx_qwen3_config& c = m->cfg;
    const bool ok = mode > 0 && !m->se_disabled && c.quant_bits == 0 && c.num_experts == 0 && c.tp_size == 1 && c.ep_size <= 1 && m->allreduce == nullptr &&
           !c.attention_bias && m->se_gran != nullptr && m->cus > 0 &&
           step_engine_ok(c.hidden_size, m->H, m->Hkv, c.head_dim, m->I, m->attn_nsplit, m->cus);
    return ok ? (mode == 2 ? 2 : 1) : 0;
}
bool step_engine_takes(omx_qwen3 m) { return step_engine_mode(m) != 0; }

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// seg_layer < 0: [embedding + every layer] in one launch; the residual stream after the last layer lands in m->h2.
// seg_layer = i: one segment of the hybrid step -- [gate/up, down] of layer i - 1 on the post-attention residual in m->h2, the new residual
// to m->h; [RMSNorm + q/k/v] of layer i into m->qkv
int enqueue_step_engine(omx_qwen3 m, hipStream_t s, int seg_layer = -1) {
    const omx_qwen3_config& c = m->cfg;
    const int D = c.head_dim, hd = c.hidden_size;
    StepEngineArgs a = {};
    a.layers = m->se_layers;
    a.L = c.num_hidden_layers; a.hidden = hd; a.H = m->H; a.Hkv = m->Hkv; a.D = D; a.I = m->I; a.cap = m->cap;
    a.eps = c.rms_norm_eps; a.scale = 1.0f / sqrtf((float)D);
    a.embed = m->embed; a.st = m->st; a.seq_ptr = m->step_seq;
    a.rope_cos = m->rope_cos; a.rope_sin = m->rope_sin;
    a.chunk = m->attn_chunk; a.nsplit = m->attn_nsplit;
    uint64_t* g = m->se_gran;
    a.g_x = g; g += hd / 2;
    a.g_x1 = g; g += hd / 2;
    a.g_qkv = g; g += (size_t)(m->H + 2 * m->Hkv) * D / 2;
    a.g_attn = g; g += (size_t)m->H * D / 2;
    a.g_act = g;
    a.g_part = m->attn_gran;
    a.h_out = seg_layer < 0 ? m->h2 : m->h;
    a.seg_layer = seg_layer;
    a.x_in = m->h; a.x1_in = m->h2; a.qkv_out = m->qkv;
    if (seg_layer > 0) a.seg_m = m->se_layers_host[seg_layer - 1];
    if (seg_layer >= 0 && seg_layer < c.num_hidden_layers) a.seg_a = m->se_layers_host[seg_layer];
    a.xcd_major = env_int("OMX_SE_XCD_MAJOR", 1);
    a.abort_flag = m->wait_abort;
    a.nsweep = env_int("OMX_SE_NSWEEP", 1);
    a.inflight = env_int("OMX_SE_INFLIGHT", 2);
    a.thin_gather = env_int("OMX_SE_THIN", 1);
    a.trace = m->se_trace;
    return launch_step_engine(a, m->cus, s);
}

// enqueue one decode step on m->stream.  with_head=false: prompt token whose logits nobody reads.
// tuning knob: OMX_GEMV_RPW_<QKV|O|GU|DOWN>=n overrides the rows-per-wave (per block for K-split kernels) heuristic of gemv.hip
static int rpw_env(const char* name) {
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
}

// kernel classes timed by omx_qwen3_time_step_kernels: an event pair armed for the launch that follows (launch_timing.hpp)
enum { KC_QKV = 0, KC_ATTN, KC_O, KC_GATE_UP, KC_DOWN, KC_HEAD, KC_ENGINE, KC_COUNT };
constexpr int kLayerClasses = KC_HEAD;
inline void time_next_launch(omx_qwen3 m, int layer, int cls) {
    if (g_launch_recorder) g_launch_recorder->next_tag = cls;
    if (!m->kernel_events) return;
    const size_t i = (cls == KC_HEAD ? (size_t)m->cfg.num_hidden_layers * kLayerClasses
                      : cls == KC_ENGINE ? (size_t)m->cfg.num_hidden_layers * kLayerClasses + 1 : (size_t)layer * kLayerClasses + cls) * 2;
    arm_launch_events((*m->kernel_events)[i], (*m->kernel_events)[i + 1]);
}

int enqueue_step_tail(omx_qwen3 m, bool with_head, const bf16_t* h, const float* pending = nullptr, int pending_n = 1, bool tp = false);

// the hybrid step (step_engine_mode == 2): per layer [engine segment] [attention + o]; m->h = residual entering a layer, m->h2 = after attention
int enqueue_step_hybrid(omx_qwen3 m, bool with_head) {
    const omx_qwen3_config& c = m->cfg;
    hipStream_t s = m->stream;
    const int hd = c.hidden_size, D = c.head_dim, L = c.num_hidden_layers;
    OMX_LAUNCH(embed_kernel, 2, 256, 0, s, m->h, m->embed, m->st, hd, m->step_seq, m->rope_cur, m->rope_cos, m->rope_sin, D / 2);
    OMX_LAUNCH_CHECK();
    for (int l = 0; l < L; ++l) {
        time_next_launch(m, l, KC_QKV);                      // (the segment's time is booked on the q/k/v class of the layer it ends in)
        if (enqueue_step_engine(m, s, l)) return 1;
        const bool fused_o = attention_takes_oproj(m);
        time_next_launch(m, l, KC_ATTN);
        if (enqueue_attention(m, l, s, fused_o ? m->h : nullptr, fused_o ? m->h2 : nullptr, nullptr)) return 1;
        if (!fused_o) {
            GemvArgs a = {};
            a.w0 = m->layers[l].o; a.n0 = hd; a.N = hd; a.K = m->H * D;
            a.x = m->attn_out; a.resid = m->h; a.out = m->h2;
            time_next_launch(m, l, KC_O);
            if (launch_gemv(a, PRO_NONE, EPI_RESIDUAL, s)) return 1;
        }
    }
    time_next_launch(m, L - 1, KC_DOWN);
    if (enqueue_step_engine(m, s, L)) return 1;
    return enqueue_step_tail(m, with_head, m->h);
}

int enqueue_step(omx_qwen3 m, bool with_head) {
    if (m->cfg.quant_bits) return enqueue_step_quant(m, with_head);
    const omx_qwen3_config& c = m->cfg;
    hipStream_t s = m->stream;
    const int hd = c.hidden_size, D = c.head_dim;
    const bool ep = c.ep_size > 1;                               // expert parallel: attention replicated, one all-reduce per MoE block
    const bool tp = !ep && (c.tp_size > 1 || m->allreduce != nullptr);   // a 1-rank communicator exercises the TP path
    const int engine_mode = step_engine_mode(m);
    if (engine_mode == 2) return enqueue_step_hybrid(m, with_head);
    const bool engine = engine_mode == 1;
    if (engine) {
        time_next_launch(m, 0, KC_ENGINE);
        if (enqueue_step_engine(m, s)) return 1;
    } else {
        OMX_LAUNCH(embed_kernel, 2, 256, 0, s, m->h, m->embed, m->st, hd, m->step_seq, m->rope_cur, m->rope_cos, m->rope_sin, D / 2);
        OMX_LAUNCH_CHECK();
    }
    bf16_t* h = engine ? m->h2 : m->h;      // residual stream entering the layer
    bf16_t* hn = engine ? m->h : m->h2;     // ping-pong partner
    const float* pending = nullptr;   // TP: all-reduced f32 partial not yet folded into h; MoE: the experts' weighted outputs
    int pending_n = 1;                //   ... how many f32 vectors `pending` holds (summed in order by the consumer's prologue)
    // MoE block without its weighted-sum launch (the next GEMV folds the experts' outputs in): every block of that GEMV reads top_k
    // extra f32 vectors, so by default only for top-2 routing (Mixtral-8x7B: 206.4 -> 207.9 tok/s); OMX_MOE_FOLD=1 forces it, 0 disables
    const char* fold_env = getenv("OMX_MOE_FOLD");
    const bool moe_fold = c.num_experts > 0 && !ep && !tp &&
                          (fold_env ? fold_env[0] == '1' : c.num_experts_per_tok <= 2);
    for (int l = 0; l < (engine ? 0 : c.num_hidden_layers); ++l) {
        const LayerW& L = m->layers[l];
        {   // [RMSNorm + QKV GEMV]  model.rs:168-170,324
            GemvArgs a = {};
            a.w0 = L.q; a.n0 = m->H * D;
            a.w1 = L.k; a.n1 = m->Hkv * D;
            a.w2 = L.v; a.n2 = m->Hkv * D;
            a.N = (m->H + 2 * m->Hkv) * D;
            a.K = hd;
            a.x = h; a.x_partial = pending; a.x_partial_n = pending_n; a.x_out = pending ? hn : nullptr;
            a.norm_w = L.in_ln; a.eps = c.rms_norm_eps;
            a.out = m->qkv;
            a.out_bias = L.qkv_bias;
            a.rows_per_wave = rpw_env("OMX_GEMV_RPW_QKV");
            time_next_launch(m, l, KC_QKV);
                if (launch_gemv(a, PRO_RMSNORM, EPI_STORE, s)) return 1;
            if (pending) { bf16_t* t = h; h = hn; hn = t; pending = nullptr; pending_n = 1; }
        }
        const bool fused_o = attention_takes_oproj(m);
        time_next_launch(m, l, KC_ATTN);
                if (enqueue_attention(m, l, s, fused_o ? h : nullptr, fused_o && !tp ? hn : nullptr, fused_o && tp ? m->partial_a : nullptr)) return 1;   // [q/k RMSNorm + RoPE + cache append + split-KV SDPA + merge]  model.rs:172-210
        if (fused_o && tp) {   // the rank's f32 partial of the O projection came out of the attention launch
            OMX_REQUIRE(m->allreduce != nullptr, "tp_size > 1 but no communicator set (omx_qwen3_set_comm)");
            OMX_REQUIRE(m->allreduce(m->partial_a, m->partial_a, hd, kNcclFloat32, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
            pending = m->partial_a;
        } else if (fused_o) {   // [O projection + residual] happened in the attention launch
            bf16_t* t = h; h = hn; hn = t;
        } else {   // [O GEMV + residual]  model.rs:214,325
            GemvArgs a = {};
            a.w0 = L.o; a.n0 = hd; a.N = hd; a.K = m->H * D;
            a.x = m->attn_out;
            a.rows_per_wave = rpw_env("OMX_GEMV_RPW_O");
            if (!tp) {
                a.resid = h; a.out = hn;
                time_next_launch(m, l, KC_O);
                if (launch_gemv(a, PRO_NONE, EPI_RESIDUAL, s)) return 1;
                bf16_t* t = h; h = hn; hn = t;
            } else {
                a.out = m->partial_a;
                a.peer = peer_fused_fits(m, a.N, a.K) ? m->peer_dev : nullptr;   // peer-store communicator: the rows are reduced over the ranks inside this launch
                if (launch_gemv(a, PRO_NONE, EPI_F32, s)) return 1;
                OMX_REQUIRE(m->allreduce != nullptr, "tp_size > 1 but no communicator set (omx_qwen3_set_comm)");
                if (!a.peer)
                    OMX_REQUIRE(m->allreduce(m->partial_a, m->partial_a, hd, kNcclFloat32, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
                pending = m->partial_a;
            }
        }
        if (c.num_experts > 0 && tp && c.tp_size > 1) {
            // expert tensor parallel: [fold the all-reduced O partial into the residual] [replicated router + this rank's columns of the
            // routed experts] [all-reduce of the slots' f32 partials] [weighted sum + residual with the single-device roundings]
            OMX_REQUIRE(m->allreduce != nullptr && pending != nullptr, "tp_size > 1 but no communicator set (omx_qwen3_set_comm)");
            ep_fold_kernel<<<8, 256, 0, s>>>(hn, h, pending, (int64_t)hd);
            OMX_LAUNCH_CHECK();
            { bf16_t* t = h; h = hn; hn = t; pending = nullptr; pending_n = 1; }
            if (omx_moe_block_partial_tp(m->moe_y, m->moe_inds, m->moe_scores, h, L.post_ln, c.rms_norm_eps, m->moe_xn, L.moe_gate, L.moe_wg,
                                         L.moe_wu, L.moe_wd, 1, hd, m->moe_I, c.num_experts, c.num_experts_per_tok, c.moe_mode, c.norm_topk_prob, s))
                return 1;
            OMX_REQUIRE(m->allreduce(m->moe_y, m->moe_y, (size_t)c.num_experts_per_tok * hd, kNcclFloat32, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
            if (omx_moe_combine_slots(hn, m->moe_y, m->moe_scores, h, 1, hd, c.num_experts_per_tok, s)) return 1;
            { bf16_t* t = h; h = hn; hn = t; }
            continue;
        }
        if (c.num_experts > 0 && ep) {
            // expert parallel: this rank's experts only, partial sums all-reduced, residual folded into the next prologue
            const int el = c.num_experts / c.ep_size;
            if (omx_moe_block_partial_ep(m->partial_b, h, L.post_ln, c.rms_norm_eps, m->moe_xn, L.moe_gate, L.moe_wg, L.moe_wu, L.moe_wd,
                                         1, hd, c.moe_intermediate_size, c.num_experts, c.num_experts_per_tok, c.moe_mode,
                                         c.norm_topk_prob, c.ep_rank * el, el, s))
                return 1;
            OMX_REQUIRE(m->allreduce != nullptr, "ep_size > 1 but no communicator set (omx_qwen3_set_comm)");
            OMX_REQUIRE(m->allreduce(m->partial_b, m->partial_b, hd, kNcclFloat32, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
            pending = m->partial_b;
            continue;
        }
        if (c.num_experts > 0) {
            // [RMSNorm] [router] [expert gate/up + SwiGLU] [expert down] [weighted sum] [+ residual]
            // (qwen3_moe.rs:475-503 / mixtral model.rs:296-308, :343-344)
            if (moe_fold) {   // ... without the last two launches: the down GEMVs store the weighted outputs, the next GEMV folds them in
                const int rc = omx_moe_block_partials(m->moe_partials, h, L.post_ln, c.rms_norm_eps, m->moe_xn, L.moe_gate, L.moe_wg, L.moe_wu,
                                                      L.moe_wd, hd, c.moe_intermediate_size, c.num_experts, c.num_experts_per_tok, c.moe_mode,
                                                      c.norm_topk_prob, s);
                if (rc == 0) { pending = m->moe_partials; pending_n = c.num_experts_per_tok; continue; }
                if (rc != 2) return 1;
            }
            if (omx_moe_block_forward(hn, h, h, L.post_ln, c.rms_norm_eps, m->moe_xn, L.moe_gate, L.moe_wg, L.moe_wu, L.moe_wd, 1, hd,
                                      c.moe_intermediate_size, c.num_experts, c.num_experts_per_tok, c.moe_mode, c.norm_topk_prob, s))
                return 1;
            bf16_t* t = h; h = hn; hn = t;
            continue;
        }
        {   // [RMSNorm + gate/up GEMV + SwiGLU]  model.rs:263-265,326
            GemvArgs a = {};
            a.w0 = L.gate; a.w1 = L.up; a.n0 = m->I; a.N = m->I; a.K = hd;
            a.x = h; a.x_partial = pending; a.x_partial_n = pending_n; a.x_out = pending ? hn : nullptr;
            a.norm_w = L.post_ln; a.eps = c.rms_norm_eps;
            a.out = m->act;
            a.rows_per_wave = rpw_env("OMX_GEMV_RPW_GU");
            time_next_launch(m, l, KC_GATE_UP);
                if (launch_gemv(a, PRO_RMSNORM, EPI_SWIGLU, s)) return 1;
            if (pending) { bf16_t* t = h; h = hn; hn = t; pending = nullptr; pending_n = 1; }
        }
        {   // [down GEMV + residual]  model.rs:266,327
            GemvArgs a = {};
            a.w0 = L.down; a.n0 = hd; a.N = hd; a.K = m->I;
            a.x = m->act;
            a.rows_per_wave = rpw_env("OMX_GEMV_RPW_DOWN");
            if (!tp) {
                a.resid = h; a.out = hn;
                time_next_launch(m, l, KC_DOWN);
                if (launch_gemv(a, PRO_NONE, EPI_RESIDUAL, s)) return 1;
                bf16_t* t = h; h = hn; hn = t;
            } else {
                a.out = m->partial_b;
                a.peer = peer_fused_fits(m, a.N, a.K) ? m->peer_dev : nullptr;
                if (launch_gemv(a, PRO_NONE, EPI_F32, s)) return 1;
                if (!a.peer)
                    OMX_REQUIRE(m->allreduce(m->partial_b, m->partial_b, hd, kNcclFloat32, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
                pending = m->partial_b;
            }
        }
    }
    return enqueue_step_tail(m, with_head, h, pending, pending_n, tp);
}

// [final RMSNorm + lm_head GEMV + argmax / sampling] on the residual stream `h` (plus a pending all-reduced partial)
int enqueue_step_tail(omx_qwen3 m, bool with_head, const bf16_t* h, const float* pending, int pending_n, bool tp) {
    const omx_qwen3_config& c = m->cfg;
    hipStream_t s = m->stream;
    const int hd = c.hidden_size;
    if (with_head) {   // [final RMSNorm + lm_head GEMV + argmax]  model.rs:423,480-489,733-735
        GemvArgs a = {};
        a.w0 = m->lm_head; a.n0 = m->V; a.N = m->V; a.K = hd;
        a.x = h; a.x_partial = pending; a.x_partial_n = pending_n; a.x_out = nullptr;
        a.norm_w = m->final_norm; a.eps = c.rms_norm_eps;
        a.out = m->logits;
        a.argmax_slot = m->argmax_partials;
        a.row_offset = c.tp_rank * m->V;
        time_next_launch(m, 0, KC_HEAD);
                if (launch_gemv(a, PRO_RMSNORM, EPI_ARGMAX, s)) return 1;
        if (add_sampling_noise(m, s)) return 1;
        if (!tp) {
            OMX_LAUNCH(sample_finalize_kernel, 1, 256, 0, s, m->st, m->argmax_partials, m->n_argmax_partials, m->out_ring, m->ring_cap,
                       (unsigned long long*)nullptr);
            OMX_LAUNCH_CHECK();
        } else {
            sample_finalize_kernel<<<1, 256, 0, s>>>(m->st, m->argmax_partials, m->n_argmax_partials, m->out_ring,
                                                     m->ring_cap, m->argmax_key);
            OMX_LAUNCH_CHECK();
            OMX_REQUIRE(m->allreduce(m->argmax_key, m->argmax_key, 1, kNcclUint64, kNcclMax, m->comm, s) == 0, "ncclAllReduce failed");
            apply_token_kernel<<<1, 1, 0, s>>>(m->st, m->argmax_key, m->out_ring, m->ring_cap);
            OMX_LAUNCH_CHECK();
        }
    } else {
        feed_prompt_kernel<<<1, 1, 0, s>>>(m->st, m->prompt_dev);
        OMX_LAUNCH_CHECK();
    }
    // the step must leave the residual stream roles as it found them for graph replay: h is
    // rewritten by embed_kernel at the start of every step, so no copy is needed.
    return 0;
}


// The with-head step as AQL packets on the engine's own queue (aql_step.hpp): OMX_STEP_AQL=1 (agent-scope fences on every packet, the
// semantics of the stream) or =2 (no fences between the packets of a replay -- the step's kernels hand every cross-kernel value over
// with write-through stores and coherent loads).  Greedy single-rank dense / packed steps only: the sampler's and the MoE block's
// launches are not recording sites.  Any failure leaves the engine on its hipGraph (and says why once on stderr with OMX_STEP_AQL_VERBOSE).
int step_aql_mode(omx_qwen3 m) {
    const char* e = getenv("OMX_STEP_AQL");
    const int mode = e ? atoi(e) : 0;
    const omx_qwen3_config& c = m->cfg;
    if (mode <= 0 || m->aql_disabled || m->eager || m->allreduce != nullptr || c.tp_size > 1 || c.ep_size > 1 || c.num_experts > 0 ||
        m->temperature != 0.f || step_engine_mode(m) != 0)
        return 0;
    return mode;
}

void build_aql(omx_qwen3 m) {
    const int mode = step_aql_mode(m);
    if (!mode || m->aql_full) return;
    // the launches are recorded under a stream capture as well: a launch site that is not a recording site would show up as a graph node
    LaunchRecorder rec;
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); m->aql_disabled = true; return; }
    g_launch_recorder = &rec;
    const int rc = enqueue_step(m, true);
    g_launch_recorder = nullptr;
    const hipError_t e = hipStreamEndCapture(m->stream, &g);
    size_t nodes = 0;
    if (g) { (void)hipGraphGetNodes(g, nullptr, &nodes); (void)hipGraphDestroy(g); }
    AqlProgram* p = nullptr;
    if (!rc && e == hipSuccess && nodes == 0) {
        const int fence = mode == 2 ? AQL_FENCE_NONE : mode == 3 ? AQL_FENCE_ACQUIRE : mode == 4 ? AQL_FENCE_RELEASE : AQL_FENCE_AGENT;
        p = aql_build(rec, fence);
    } else if (!rc && e == hipSuccess) {
        set_error("aql: %zu launches of the step are not recording sites", nodes);
    }
    if (!p) {
        if (getenv("OMX_STEP_AQL_VERBOSE")) fprintf(stderr, "omx: AQL step unavailable: %s\n", omx_last_error());
        (void)hipGetLastError();
        omx_clear_error();
        m->aql_disabled = true;
        return;
    }
    m->aql_full = p;
}

int build_graphs(omx_qwen3 m) {
    if (m->g_full || m->eager) return 0;
    if (resolve_weights(m)) return 1;
    const char* no_graph = getenv("OMX_NO_GRAPH");
    if (no_graph && no_graph[0] == '1') {
        m->eager = true;
        return 0;
    }
    for (int which = 0; which < 2; ++which) {
        hipGraph_t g = nullptr;
        OMX_HIP_CHECK(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue_step(m, which == 0);
        hipError_t e = hipStreamEndCapture(m->stream, &g);
        hipGraphExec_t ge = nullptr;
        if (!rc && e == hipSuccess) e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (rc || e != hipSuccess) {
            (void)hipGetLastError();
            if (m->allreduce != nullptr) {
                // a captured collective was refused: run the same launches eagerly instead
                if (m->g_full) { (void)hipGraphExecDestroy(m->g_full); m->g_full = nullptr; }
                m->eager = true;
                omx_clear_error();
                return 0;
            }
            return rc ? 1 : set_error("step graph capture failed: %s", hipGetErrorString(e));
        }
        (which == 0 ? m->g_full : m->g_nohead) = ge;
    }
    // the captured MoE block's scratch pointers must not move: the expert-parallel block and the expert-TENSOR-parallel one
    // (omx_moe_block_partial_tp bakes its get_workspace_aux pointer into the graph as well)
    if (m->cfg.num_experts > 0 && (m->cfg.ep_size > 1 || m->cfg.tp_size > 1)) workspace_aux_pin(m->stream, true);
    build_aql(m);
    return 0;
}

// Called with the position the next step will process.  The decode attention's split plan is part of the captured graph
// (attn_step.hip: fixed token ranges per split, so that nothing the kernel loads first depends on the position): one plan per
// context bucket of 1024 tokens (4096 beyond 8 k), the graphs are rebuilt when the position enters another bucket.
int prepare_step(omx_qwen3 m, int pos) {
    {
        const int tk = pos + 1, gran = tk <= 8192 ? 1024 : 4096;
        const int want = std::min(m->cap, (tk + gran - 1) / gran * gran);
        if (want != m->graph_tk_max) {
            drop_graphs(m);
            attn_step_plan(want, m->Hkv, m->H / m->Hkv, m->cfg.head_dim, &m->attn_chunk, &m->attn_nsplit);
            m->graph_tk_max = want;
        }
    }
    return build_graphs(m);
}

int run_step(omx_qwen3 m, bool with_head, int pos) {
    if (prepare_step(m, pos)) return 1;
    if (m->eager) return enqueue_step(m, with_head);
    OMX_HIP_CHECK(hipGraphLaunch(with_head ? m->g_full : m->g_nohead, m->stream));
    return 0;
}

// [final RMSNorm + lm_head + sampler] on one hidden row that is NOT the step graph's residual buffer: the last row of a batched
// prefill (model.rs:423, 480-489, 733-735).  Same kernels and state transition as the tail of a decode step.
int enqueue_head_on_row(omx_qwen3 m, const bf16_t* row, hipStream_t s) {
    const omx_qwen3_config& c = m->cfg;
    const int hd = c.hidden_size;
    // a tensor-parallel rank holds a vocabulary shard: its rows are numbered from its offset and the ranks' packed (logit, index) keys
    // meet in one unsigned max (enqueue_step_tail)
    const bool tp = m->allreduce != nullptr && c.ep_size <= 1;
    if (c.quant_bits > 0) {
        QGemvArgs a = {};
        a.m[0] = m->q_head; a.m[0].n = m->V; a.N = m->V; a.K = hd; a.group = c.quant_group;
        a.x = row; a.norm_w = m->final_norm; a.eps = c.rms_norm_eps; a.out = m->logits; a.scales_f16 = c.quant_scales_f16 != 0;
        a.argmax_slot = m->argmax_partials;
        a.row_offset = c.tp_rank * m->V;
        if (launch_qgemv(a, c.quant_bits, PRO_RMSNORM, EPI_ARGMAX, s)) return 1;
    } else {
        GemvArgs a = {};
        a.w0 = m->lm_head; a.n0 = m->V; a.N = m->V; a.K = hd;
        a.x = row; a.x_partial = nullptr; a.x_out = nullptr;
        a.norm_w = m->final_norm; a.eps = c.rms_norm_eps;
        a.out = m->logits;
        a.argmax_slot = m->argmax_partials;
        a.row_offset = c.tp_rank * m->V;
        if (launch_gemv(a, PRO_RMSNORM, EPI_ARGMAX, s)) return 1;
    }
    if (add_sampling_noise(m, s)) return 1;
    sample_finalize_kernel<<<1, 256, 0, s>>>(m->st, m->argmax_partials, m->n_argmax_partials, m->out_ring, m->ring_cap, tp ? m->argmax_key : nullptr);
    OMX_LAUNCH_CHECK();
    if (tp) {
        OMX_REQUIRE(m->allreduce(m->argmax_key, m->argmax_key, 1, kNcclUint64, kNcclMax, m->comm, s) == 0, "ncclAllReduce failed");
        apply_token_kernel<<<1, 1, 0, s>>>(m->st, m->argmax_key, m->out_ring, m->ring_cap);
        OMX_LAUNCH_CHECK();
    }
    return 0;
}

// Batched prefill of T prompt tokens (all but the last one, which goes through the decode step so
// that sampling stays in one place): fills the KV slabs of every layer.  Matrix-core path:
//   RMSNorm rows -> q/k/v GEMM -> [per-head norm + RoPE + cache scatter] -> flash attention
//   (causal, bottom-right aligned == the bool mask of utils.rs:134-153) -> o GEMM + residual ->
//   RMSNorm -> gate/up GEMM -> silu*up -> down GEMM + residual.        (model.rs:161-215,263-267,321-332)
// The last layer stops after its cache scatter: nothing downstream of it is consumed for these tokens.
// a device-wide barrier of the persistent kernel gave up (a block never arrived): the step's results are void
int step_gave_up(omx_qwen3 m, unsigned* code) {
    OMX_HIP_CHECK(hipMemcpy(code, m->wait_abort, 4, hipMemcpyDeviceToHost));
    return 0;
}
int step_health(omx_qwen3 m) {
    unsigned gave_up = 0;
    if (step_gave_up(m, &gave_up)) return 1;
    OMX_REQUIRE(gave_up == 0, "decode step: a wait inside a launch gave up (code 0x%x: a workgroup of the launch was not resident?)", gave_up);
    return 0;
}

// A launch whose workgroups wait on each other (the persistent step, the O projection inside the attention launch) gave up: some
// workgroup was not resident -- another process or stream holds CUs.  Drop to the next form that needs less co-residency, restore the
// step state the call started from and let the caller run the steps again (the KV rows they wrote are rewritten).  Returns 0 when a
// retry is possible.
int step_fallback(omx_qwen3 m, const StepState& st) {
    if (m->temperature != 0.f) return 1;                       // the sampler's key sequence advanced: no silent replay
    // more than one rank: a local replay would re-enqueue every all-reduce of the n steps (and the argmax one) on THIS rank only -- its
    // peers never issue them, so the collectives would pair with the peers' next call (wrong sums or a hang).  Report the abort instead.
    if (m->allreduce != nullptr || m->cfg.tp_size > 1 || m->cfg.ep_size > 1) return 1;
    if (!m->se_disabled && step_engine_takes(m)) m->se_disabled = true;
    else if (!m->oproj_disabled && attention_takes_oproj(m)) m->oproj_disabled = true;
    else return 1;
    drop_graphs(m);
    OMX_HIP_CHECK(hipMemsetAsync(m->wait_abort, 0, 4, m->stream));
    OMX_HIP_CHECK(hipMemcpyAsync(m->st, &st, sizeof(st), hipMemcpyHostToDevice, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    return 0;
}

// Text-encoder use of the same stack (flux-klein-mlx/src/qwen3_encoder.rs:141-224, 403-455): all T tokens through
// layers 0..last tap, hidden states copied out after the tapped layers, attention under an explicit additive mask.
struct EncodeOpts {
    const int* taps;          // ascending layer indices whose OUTPUT is extracted
    int n_taps;
    bf16_t* out;              // [T, n_taps * hidden]
    const bf16_t* mask;       // optional additive [T, T] (causal + padding), nullptr = causal
};

__global__ void copy_rows_strided_kernel(bf16_t* dst, int64_t dst_ld, const bf16_t* src, int64_t src_ld, int rows, int cols8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)rows * cols8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols8, c = i % cols8;
        *reinterpret_cast<u32x4*>(dst + r * dst_ld + c * 8) = *reinterpret_cast<const u32x4*>(src + r * src_ld + c * 8);
    }
}

// qwen3_encoder.rs:172-198: additive mask 0 where (j <= i and attention_mask[j]) else bf16(-1e9)
// expert-parallel batched prefill: h = bf16(resid + bf16(all-reduced partial))  (the residual of mixtral model.rs:343-344)

__global__ void encoder_mask_kernel(bf16_t* mask, const uint8_t* am, int T) {
    const bf16_t neg = f32_to_bf16(-1e9f);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)T * T; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i / T), k = (int)(i % T);
        mask[i] = (k <= q && am[k]) ? (bf16_t)0 : neg;
    }
}

// the prompt pass's row buffers (T rows each) and, for a packed model, the scratch its dequantised GEMM operands pass through: grown
// here, on the host, AHEAD of a prompt's device-timed region (a hipMalloc between the launches leaves the device idle for its duration)
int prefill_reserve(omx_qwen3 m, int T) {
    const omx_qwen3_config& c = m->cfg;
    const int hd = c.hidden_size, D = c.head_dim, H = m->H, Hkv = m->Hkv, I = m->I;
    if (T > m->pf_cap) {
        OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
        bf16_t** bufs[] = {&m->pf_h, &m->pf_h2, &m->pf_xn, &m->pf_q, &m->pf_k, &m->pf_v, &m->pf_qt, &m->pf_attn, &m->pf_g, &m->pf_u};
        const size_t sizes[] = {(size_t)hd, (size_t)hd, (size_t)hd, (size_t)H * D, (size_t)Hkv * D, (size_t)Hkv * D,
                                (size_t)H * D, (size_t)H * D, (size_t)I, (size_t)I};
        for (int i = 0; i < 10; ++i) {
            if (*bufs[i]) OMX_HIP_CHECK(hipFree(*bufs[i]));
            OMX_HIP_CHECK(hipMalloc((void**)bufs[i], sizes[i] * (size_t)T * 2));
        }
        m->pf_cap = T;
    }
    if (c.quant_bits != 0) {
        const size_t need = std::max((size_t)std::max(std::max(H * D, I), hd) * (size_t)std::max(hd, I),
                                     std::max((size_t)(H + 2 * Hkv) * D * hd, (size_t)2 * I * hd));
        if (need > m->dq_cap) {
            OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
            if (m->dq_buf) OMX_HIP_CHECK(hipFree(m->dq_buf));
            OMX_HIP_CHECK(hipMalloc((void**)&m->dq_buf, need * 2));
            m->dq_cap = need;
        }
    }
    return 0;
}

// OMX_DEQUANT_CACHE=1 / 0: keep / do not keep the dequantised matrices between prompts; default: keep them for a dense model when
// they take at most a quarter of the free HBM and 64 GB (Qwen3-8B: 13.7 GB; a sparse-MoE model's attention matrices only on request).
// ONE allocation, made once per model -- host time (~0.03 s per GB) that omx_qwen3_prefill spends ahead of its device-timed region.
void dq_cache_prepare(omx_qwen3 m) {
    if (m->dq_cache_mode >= 0) return;
    const omx_qwen3_config& c = m->cfg;
    const size_t hd = c.hidden_size, D = c.head_dim, H = m->H, Hkv = m->Hkv, I = m->I;
    const char* ce = getenv("OMX_DEQUANT_CACHE");
    const size_t per_layer = (H * D * hd * 2 + 2 * Hkv * D * hd + (c.num_experts == 0 ? 3 * I * hd : 0)) * 2;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const size_t need_b = per_layer * (size_t)c.num_hidden_layers;
    m->dq_cache_mode = ce ? (ce[0] == '1') : (c.num_experts == 0 && need_b <= free_b / 4 && need_b <= ((size_t)64 << 30));
    if (m->dq_cache_mode == 1 && !m->dq_slab) {
        m->dq_slab_bytes = need_b + (size_t)7 * c.num_hidden_layers * 256;      // (every matrix starts on a 256-byte boundary)
        if (hipMalloc((void**)&m->dq_slab, m->dq_slab_bytes) != hipSuccess) {
            (void)hipGetLastError();
            m->dq_slab = nullptr;
            m->dq_cache_mode = 0;
        }
    }
}

// out [T, H * D] = in [H, T, D] (16-bit elements): the attention output of the explicit SDPA form, token-major for the O projection
__global__ void heads_to_tokens_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ in, int H, int T, int D) {
    const int vpr = D / 8;
    const int64_t n = (int64_t)H * T * vpr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpr), t = (int)((i / vpr) % T), h = (int)(i / ((int64_t)vpr * T));
        reinterpret_cast<u32x4*>(out)[((int64_t)t * H + h) * vpr + v] = reinterpret_cast<const u32x4*>(in)[i];
    }
}

int prefill_prefix_batched(omx_qwen3 m, int T, int off, const EncodeOpts* enc = nullptr, bool full_last = false) {
    const omx_qwen3_config& c = m->cfg;
    // float16 checkpoints (round 4): the same pass in float16 -- weights dequantised to float16, the eight-wave GEMM kernel's float16
    // form, float16 norms / RoPE / slabs, the flash attention kernel's float16 form -- for
    // plain prompts of a dense model, also on tensor-parallel shards (each rank's float16 partial products summed in f32); encode /
    // verify and the expert forms stay bfloat16-only
    const bool f16 = c.quant_scales_f16 != 0;
    // (round 5: the encoder taps and passes of a handful of rows too -- a tap copies 16-bit rows whatever their format, and the 128-row
    //  float16 GEMM tile predicates its rows.  An encoder PADDING mask stays refused: the reference builds it as (1 - keep) * f16(-1e9) =
    //  0 * -inf = NaN on every kept key, flux-klein-mlx/src/qwen3_encoder.rs:196-198 -- there is no finite result to reproduce.)
    OMX_REQUIRE(!f16 || !(enc && enc->mask), "float16 encoder with an attention_mask: the reference's additive mask is 0 * f16(-1e9) = NaN in "
                "float16 (qwen3_encoder.rs:196-198); pass no mask (causal) or load the bfloat16 checkpoint");
    struct GemmF16Scope { bool on, was = false; explicit GemmF16Scope(bool o) : on(o) { if (on) was = gemm_set_f16(true); } ~GemmF16Scope() { if (on) gemm_set_f16(was); } } f16_scope(f16);
    const omx_dtype act_dt = f16 ? OMX_FLOAT16 : OMX_BFLOAT16;
    hipStream_t s = m->stream;
    const int hd = c.hidden_size, D = c.head_dim, H = m->H, Hkv = m->Hkv, I = m->I;
    if (prefill_reserve(m, T)) return 1;   // (omx_qwen3_prefill has called it ahead of its timed region already)
    // tensor parallel: q/k/v/gate/up are this rank's column shards (local H, Hkv, I), o / down are row
    // shards whose [T, hidden] bf16 partial sums are all-reduced -- two collectives per layer -- before the residual add
    const bool tp = m->allreduce != nullptr && c.ep_size <= 1;
    auto row_split = [&](bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* resid, int K) -> int {
        if (!tp) return launch_gemm_bf16_ex(out, x, w, nullptr, resid, T, hd, K, s);
        bf16_t* part = m->pf_xn;   // free between the projections that read it and the next norm that rewrites it
        if (launch_gemm_bf16(part, x, w, nullptr, T, hd, K, s)) return 1;
        if (f16) {
            // float16: the ranks' float16 partial products widened, summed in f32 by the collective (every communicator reduces f32; none
            // float16) and folded into the float16 residual with the decode step's two roundings
            if (!m->pf_ep_partial || m->pf_ep_cap < T) {
                OMX_HIP_CHECK(hipStreamSynchronize(s));
                if (m->pf_ep_partial) OMX_HIP_CHECK(hipFree(m->pf_ep_partial));
                OMX_HIP_CHECK(hipMalloc((void**)&m->pf_ep_partial, (size_t)std::max(T, m->pf_cap) * hd * sizeof(float)));
                m->pf_ep_cap = std::max(T, m->pf_cap);
            }
            f16_widen_kernel<<<1024, 256, 0, s>>>(m->pf_ep_partial, part, (int64_t)T * hd);
            OMX_LAUNCH_CHECK();
            OMX_REQUIRE(m->allreduce(m->pf_ep_partial, m->pf_ep_partial, (size_t)T * hd, kNcclFloat32, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
            ep_fold_kernel<<<1024, 256, 0, s>>>(out, resid, m->pf_ep_partial, (int64_t)T * hd, true);
            OMX_LAUNCH_CHECK();
            return 0;
        }
        OMX_REQUIRE(m->allreduce(part, part, (size_t)T * hd, kNcclBfloat16, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
        return omx_add(out, resid, part, (int64_t)T * hd, OMX_BFLOAT16, s);
    };
    const bool quant = c.quant_bits != 0;
    // quantized checkpoint: each weight is dequantised into one scratch matrix right before its GEMM (MLX's qmm does
    // the same per tile); K is the contraction width of that weight
    // `at`: element offset inside the scratch, so that the members of one segmented launch (q | k | v, gate | up) coexist
    if (quant) dq_cache_prepare(m);      // (omx_qwen3_prefill has called it ahead of its timed region already)
    auto W = [&](const bf16_t* dense, const QMat* qm, int K, size_t at = 0) -> const bf16_t* {
        if (!quant) return dense;
        if (m->dq_cache_mode == 1) {
            auto it = m->dq_cache.find(qm->w);
            if (it != m->dq_cache.end()) return it->second;
            const size_t bytes = ((size_t)qm->n * K * 2 + 255) & ~(size_t)255;
            if (m->dq_cache_bytes + bytes <= m->dq_slab_bytes) {
                bf16_t* keep = (bf16_t*)(m->dq_slab + m->dq_cache_bytes);
                if (launch_dequantize_bf16(keep, qm->w, qm->scales, qm->biases, qm->n, K, c.quant_group, c.quant_bits, f16, s, f16)) return nullptr;
                m->dq_cache[qm->w] = keep;
                m->dq_cache_bytes += bytes;
                return keep;
            }
            // (the slab is full -- matrices it was not sized for: those go through the scratch every time)
        }
        if (launch_dequantize_bf16(m->dq_buf + at, qm->w, qm->scales, qm->biases, qm->n, K, c.quant_group, c.quant_bits, f16, s, f16)) return nullptr;
        return m->dq_buf + at;
    };
    if (quant) {
        // QuantizedEmbedding::forward: gather the packed rows, dequantise (quantized.rs:192-203)
        const int wpr = hd * c.quant_bits / 32, gpr = hd / c.quant_group;
        uint32_t* rows_w = (uint32_t*)m->pf_xn;                       // scratch: [T, wpr] u32 fits in [T, hd] bf16
        bf16_t* rows_s = m->pf_h2;
        bf16_t* rows_b = m->pf_h2 + (size_t)T * gpr;
        if (omx_take_rows(rows_w, m->q_embed.w, m->prompt_dev, T, wpr, OMX_FLOAT32, s)) return 1;
        if (omx_take_rows(rows_s, m->q_embed.scales, m->prompt_dev, T, gpr, OMX_BFLOAT16, s)) return 1;
        if (omx_take_rows(rows_b, m->q_embed.biases, m->prompt_dev, T, gpr, OMX_BFLOAT16, s)) return 1;
        if (launch_dequantize_bf16(m->pf_h, (const uint32_t*)rows_w, rows_s, rows_b, T, hd, c.quant_group, c.quant_bits, f16, s, f16)) return 1;
    } else if (omx_take_rows(m->pf_h, m->embed, m->prompt_dev, T, hd, OMX_BFLOAT16, s)) {
        return 1;
    }
    bf16_t* h = m->pf_h;
    bf16_t* h2 = m->pf_h2;
    const float scale = 1.0f / sqrtf((float)D);
    const LayerQ no_q = {};
    const int n_run = enc ? enc->taps[enc->n_taps - 1] + 1 : c.num_hidden_layers;
    int next_tap = 0;
    const char* seg_env = getenv("OMX_PREFILL_SEGMENTED");   // 0: one launch per projection (A/B, tests)
    const bool seg_gemm = !(seg_env && seg_env[0] == '0');
    for (int l = 0; l < n_run; ++l) {
        const LayerW& L = m->layers[l];
        const LayerQ& Q = quant ? m->qlayers[l] : no_q;
        const bf16_t* w = nullptr;
        // q, k, v: one segmented launch over the three borrowed weights when the chip is filled that way (the separate k / v
        // grids are 128 tiles on 512 slots), else three launches
        GemmSegs qkv = {};
        qkv.n_plain = 3;
        qkv.plain[0] = {L.q, L.q_bias, m->pf_q, H * D, H * D, 0};
        qkv.plain[1] = {L.k, L.k_bias, m->pf_k, Hkv * D, Hkv * D, 0};
        qkv.plain[2] = {L.v, L.v_bias, m->pf_v, Hkv * D, Hkv * D, 0};
        // (a handful of rows: the weight-streaming launch normalises its staged copy of the rows itself -- no RMSNorm launch)
        const bool qkv_norm = !f16 && seg_gemm && gemm_segmented_preferred(T, hd, qkv) && gemv_rows_takes_norm(T, hd, qkv);
        if (!qkv_norm && omx_rms_norm(m->pf_xn, h, L.in_ln, T, hd, c.rms_norm_eps, act_dt, s)) return 1;
        if (qkv_norm) { qkv.pre_norm_w = L.in_ln; qkv.pre_norm_eps = c.rms_norm_eps; }
        if (f16 || (seg_gemm && gemm_segmented_preferred(T, hd, qkv))) {   // (float16: always the segmented 256-row kernel)
            if (quant) {   // the three dequantised matrices side by side in the scratch
                const size_t nq = (size_t)H * D * hd, nk = (size_t)Hkv * D * hd;
                if (!(qkv.plain[0].w = W(nullptr, &Q.q, hd, 0)) || !(qkv.plain[1].w = W(nullptr, &Q.k, hd, nq)) ||
                    !(qkv.plain[2].w = W(nullptr, &Q.v, hd, nq + nk)))
                    return 1;
            }
            if (launch_gemm_bf16_segmented(qkv_norm ? h : m->pf_xn, T, hd, qkv, s)) return 1;
        } else {
            if (!(w = W(L.q, &Q.q, hd)) || launch_gemm_bf16(m->pf_q, m->pf_xn, w, L.q_bias, T, H * D, hd, s)) return 1;
            if (!(w = W(L.k, &Q.k, hd)) || launch_gemm_bf16(m->pf_k, m->pf_xn, w, L.k_bias, T, Hkv * D, hd, s)) return 1;
            if (!(w = W(L.v, &Q.v, hd)) || launch_gemm_bf16(m->pf_v, m->pf_xn, w, L.v_bias, T, Hkv * D, hd, s)) return 1;
        }
        if (launch_qk_norm_rope_scatter(m->pf_q, m->pf_k, m->pf_v, L.q_norm, L.k_norm, m->rope_cos, m->rope_sin, m->pf_qt,
                                        m->kcache[l], m->vcache[l], T, H, Hkv, D, m->cap, off, c.rms_norm_eps, s, f16))
            return 1;
        if (!enc && !full_last && l == c.num_hidden_layers - 1) break;   // a prefix only has to leave its K/V rows behind
        const char* skv_env = getenv("OMX_PREFILL_SPLITKV");
        if (f16) {
            // float16: the flash kernel's float16 instantiation (f32 scores / softmax / accumulators, P rounded to float16 for the second
            // product, one rounding of the output) -- MLX's fast SDPA accumulates in f32 the same way
            // (OMX_F16_ATTN=explicit: f32 on widened copies through omx_sdpa, then heads back next to each other per token: the A/B form)
            const char* fe = getenv("OMX_F16_ATTN");
            if (fe && strcmp(fe, "explicit") == 0) {
                if (omx_sdpa(m->pf_q, m->pf_qt, m->kcache[l], m->vcache[l], 1, H, Hkv, T, off + T, D, 0, (int64_t)m->cap * D, scale, OMX_MASK_CAUSAL,
                             nullptr, OMX_FLOAT16, s))
                    return 1;
                heads_to_tokens_kernel<<<1024, 256, 0, s>>>(m->pf_attn, m->pf_q, H, T, D);
                OMX_LAUNCH_CHECK();
            } else if (launch_attn_prefill(m->pf_attn, m->pf_qt, m->kcache[l], m->vcache[l], 1, H, Hkv, T, off + T, D, 0, (int64_t)m->cap * D, scale,
                                           OMX_MASK_CAUSAL, nullptr, s, /*out_token_major=*/true, nullptr, /*f16=*/true))
                return 1;
        } else
        if (!enc && T <= 8 && H / Hkv <= 8 && !(skv_env && skv_env[0] == '0')) {
            // a handful of new rows over a long cache (speculative verify, a short follow-up prompt): the flash kernel gives them
            // H * ceil(T / 64) blocks that each walk all keys (53 us per layer for 5 rows at 2 k of context); the split-KV decode
            // kernel takes the T rows as batch entries over the ONE cache, row i seeing the first off + i + 1 keys
            AttnDecodeArgs a = {};
            a.q = m->pf_qt; a.q_bs = D; a.q_hs = (int64_t)T * D;                    // q_out[h][t][:] of the scatter kernel
            a.k = m->kcache[l]; a.v = m->vcache[l];
            a.kv_batch_stride = 0; a.kv_head_stride = (int64_t)m->cap * D;
            a.B = T; a.H = H; a.Hkv = Hkv; a.Tk = off + T;
            a.scale = scale; a.mask_mode = OMX_MASK_NONE; a.causal_tail = 1;
            a.nsplit = decode_nsplit(off + T, T * Hkv);
            void* aws = nullptr;
            if (get_workspace_aux(&aws, attn_decode_ws_bytes(T * H, a.nsplit, D), s)) return 1;
            a.ws_o = (float*)aws;
            a.ws_ml = a.ws_o + (size_t)T * H * a.nsplit * D;
            a.out = m->pf_attn;                                                       // [T, H * D]
            if (launch_attn_decode(a, D, s)) return 1;
        } else if (launch_attn_prefill(m->pf_attn, m->pf_qt, m->kcache[l], m->vcache[l], 1, H, Hkv, T, off + T, D, 0,
                                (int64_t)m->cap * D, scale, enc && enc->mask ? OMX_MASK_ADDITIVE : OMX_MASK_CAUSAL,
                                enc ? enc->mask : nullptr, s, /*out_token_major=*/true))
            return 1;
        if (!(w = W(L.o, &Q.o, H * D)) || row_split(h2, m->pf_attn, w, h, H * D)) return 1;
        GemmSegs gu = {};
        gu.w_gate = L.gate; gu.w_up = L.up; gu.out_act = m->pf_g; gu.half = I; gu.ld_act = I; gu.act_mode = 1;
        const bool gu_norm = !f16 && c.num_experts == 0 && seg_gemm && gemm_segmented_preferred(T, hd, gu) && gemv_rows_takes_norm(T, hd, gu);
        if (!gu_norm && omx_rms_norm(m->pf_xn, h2, L.post_ln, T, hd, c.rms_norm_eps, act_dt, s)) return 1;
        if (gu_norm) { gu.pre_norm_w = L.post_ln; gu.pre_norm_eps = c.rms_norm_eps; }
        if (c.num_experts > 0) {   // sparse-MoE feed-forward over all T rows (grouped MFMA GEMM route), then the residual
            if (quant && (c.ep_size > 1 || c.tp_size > 1)) {
                // packed stacks under expert parallelism / expert tensor parallelism (round 5): this rank's stacks dequantised per call, the
                // bf16 batched form, ONE all-reduce of the [T, hidden] f32 partial (the exchange combine stays a bf16-checkpoint path)
                const bool etp = c.tp_size > 1;
                const int el = etp ? c.num_experts : c.num_experts / c.ep_size;
                if (!m->pf_ep_partial || m->pf_ep_cap < T) {
                    OMX_HIP_CHECK(hipStreamSynchronize(s));
                    if (m->pf_ep_partial) OMX_HIP_CHECK(hipFree(m->pf_ep_partial));
                    OMX_HIP_CHECK(hipMalloc((void**)&m->pf_ep_partial, (size_t)std::max(T, m->pf_cap) * hd * sizeof(float)));
                    m->pf_ep_cap = std::max(T, m->pf_cap);
                }
                if (omx_moe_block_partial_ep_q(m->pf_ep_partial, h2, L.post_ln, c.rms_norm_eps, m->pf_xn, Q.moe_router.w, Q.moe_router.scales,
                                               Q.moe_router.biases, Q.moe_g.w, Q.moe_g.scales, Q.moe_g.biases, Q.moe_u.w, Q.moe_u.scales,
                                               Q.moe_u.biases, Q.moe_d.w, Q.moe_d.scales, Q.moe_d.biases, T, hd, etp ? m->moe_I : c.moe_intermediate_size,
                                               c.num_experts, c.num_experts_per_tok, c.moe_mode, c.norm_topk_prob, etp ? 0 : c.ep_rank * el, el,
                                               c.quant_group, c.quant_bits, f16 ? 1 : 0, s))
                    return 1;
                OMX_REQUIRE(m->allreduce != nullptr, "ep_size / tp_size > 1 but no communicator set (omx_qwen3_set_comm)");
                OMX_REQUIRE(m->allreduce(m->pf_ep_partial, m->pf_ep_partial, (size_t)T * hd, kNcclFloat32, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
                ep_fold_kernel<<<1024, 256, 0, s>>>(h, h2, m->pf_ep_partial, (int64_t)T * hd, f16);
                OMX_LAUNCH_CHECK();
            } else if (quant) {
                if (omx_moe_block_forward_q_ex(h, h2, h2, L.post_ln, c.rms_norm_eps, m->pf_xn, Q.moe_router.w, Q.moe_router.scales,
                                               Q.moe_router.biases, Q.moe_g.w, Q.moe_g.scales, Q.moe_g.biases, Q.moe_u.w, Q.moe_u.scales,
                                               Q.moe_u.biases, Q.moe_d.w, Q.moe_d.scales, Q.moe_d.biases, T, hd, c.moe_intermediate_size,
                                               c.num_experts, c.num_experts_per_tok, c.moe_mode, c.norm_topk_prob, c.quant_group,
                                               c.quant_bits, f16 ? 1 : 0, s))
                    return 1;
            } else if (c.ep_size > 1 || c.tp_size > 1) {
                // expert TENSOR parallel: the same launches over ALL experts at this rank's 1 / tp of their intermediate columns -- the f32
                // partial of every token's weighted sum is all-reduced like the expert-parallel one (each rank's partial products rounded
                // to bf16 before the sum: the dense model's row-split rounding, not the decode step's single-device one).
                // expert parallel: attention is replicated, so every rank already holds all T rows -- there is
                // nothing to dispatch.  Each rank routes all rows, multiplies the slots of ITS experts (grouped matrix-core GEMMs over
                // a device-side plan), and ONE all-reduce per layer sums the [T, hidden] f32 partials: the combine half of an
                // all-to-all exchange, with the reduction done by the collective.  (Until round 3 a prompt under EP was T decode steps.)
                const bool etp = c.tp_size > 1;
                const int el = etp ? c.num_experts : c.num_experts / c.ep_size;
                // expert parallel on the peer communicator's exchange path (round 4): the weighted sum as an all-to-all combine of the
                // routed slots' rows to their tokens' owners + an all-gather of the finished residual rows, in ONE kernel
                // (peer_allreduce.hip peer_moe_combine_kernel) -- a rank pushes ~T k / N + T (N - 1) / N rows of bf16 instead of
                // taking part in an all-reduce of [T, hidden] f32.  Same roundings (bit-identical for top-2).  OMX_EP_COMBINE=allreduce
                // keeps the all-reduce; any communicator without the exchange path does too.
                const char* cmb = getenv("OMX_EP_COMBINE");
                if (!etp && m->allreduce == (nccl_allreduce_fn)omx_peer_allreduce_fn() && omx_peer_comm_stage_bytes(m->comm) > 0 &&
                    T * c.num_experts_per_tok > 32 &&      // (a handful of rows takes the block's GEMV form, which has no slot tables)
                    !(cmb && strcmp(cmb, "allreduce") == 0)) {
                    omx_moe_ep_slots sl = {};
                    if (omx_moe_block_slots_ep(&sl, m->pf_xn, L.moe_gate, L.moe_wg, L.moe_wu, L.moe_wd, T, hd, c.moe_intermediate_size, c.num_experts,
                                               c.num_experts_per_tok, c.moe_mode, c.norm_topk_prob, c.ep_rank * el, el, s))
                        return 1;
                    const int rc = omx_peer_moe_combine(h, h2, &sl, T, hd, c.num_experts_per_tok, c.ep_rank * el, el, m->comm, s);
                    OMX_REQUIRE(rc == 0 || rc == 2, "expert-parallel combine over the peer communicator failed");
                    if (rc == 0) {
                        if (enc && next_tap < enc->n_taps && enc->taps[next_tap] == l) {
                            copy_rows_strided_kernel<<<1024, 256, 0, s>>>(enc->out + (size_t)next_tap * hd, (int64_t)enc->n_taps * hd, h, hd, T, hd / 8);
                            ++next_tap;
                        }
                        continue;
                    }
                    // (rc 2: this size does not fit the stages -- the slots were computed, the all-reduce form below recomputes them)
                }
                if (!m->pf_ep_partial || m->pf_ep_cap < T) {
                    OMX_HIP_CHECK(hipStreamSynchronize(s));
                    if (m->pf_ep_partial) OMX_HIP_CHECK(hipFree(m->pf_ep_partial));
                    OMX_HIP_CHECK(hipMalloc((void**)&m->pf_ep_partial, (size_t)std::max(T, m->pf_cap) * hd * sizeof(float)));
                    m->pf_ep_cap = std::max(T, m->pf_cap);
                }
                if (omx_moe_block_partial_ep(m->pf_ep_partial, m->pf_xn, nullptr, c.rms_norm_eps, nullptr, L.moe_gate, L.moe_wg, L.moe_wu, L.moe_wd,
                                             T, hd, etp ? m->moe_I : c.moe_intermediate_size, c.num_experts, c.num_experts_per_tok, c.moe_mode,
                                             c.norm_topk_prob, etp ? 0 : c.ep_rank * el, el, s))
                    return 1;
                OMX_REQUIRE(m->allreduce != nullptr, "ep_size / tp_size > 1 but no communicator set (omx_qwen3_set_comm)");
                OMX_REQUIRE(m->allreduce(m->pf_ep_partial, m->pf_ep_partial, (size_t)T * hd, kNcclFloat32, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
                ep_fold_kernel<<<1024, 256, 0, s>>>(h, h2, m->pf_ep_partial, (int64_t)T * hd);
                OMX_LAUNCH_CHECK();
            } else {
                if (omx_moe_forward(m->pf_attn, m->pf_xn, L.moe_gate, L.moe_wg, L.moe_wu, L.moe_wd, T, hd, c.moe_intermediate_size,
                                    c.num_experts, c.num_experts_per_tok, c.moe_mode, c.norm_topk_prob, nullptr, nullptr, s))
                    return 1;
                if (omx_add(h, h2, m->pf_attn, (int64_t)T * hd, OMX_BFLOAT16, s)) return 1;
            }
            if (enc && next_tap < enc->n_taps && enc->taps[next_tap] == l) {
                copy_rows_strided_kernel<<<1024, 256, 0, s>>>(enc->out + (size_t)next_tap * hd, (int64_t)enc->n_taps * hd, h, hd, T, hd / 8);
                ++next_tap;
            }
            continue;
        }
        // gate, up and nn::silu(gate) * up: one launch with the activation in the epilogue (768 tiles = 3 full rounds at
        // T = 2048 instead of 2 x 384), else two GEMMs + the elementwise kernel
        if (f16 || (seg_gemm && gemm_segmented_preferred(T, hd, gu))) {
            if (quant && (!(gu.w_gate = W(nullptr, &Q.gate, hd, 0)) || !(gu.w_up = W(nullptr, &Q.up, hd, (size_t)I * hd)))) return 1;
            if (launch_gemm_bf16_segmented(gu_norm ? h2 : m->pf_xn, T, hd, gu, s)) return 1;
        } else {
            if (!(w = W(L.gate, &Q.gate, hd)) || launch_gemm_bf16(m->pf_g, m->pf_xn, w, nullptr, T, I, hd, s)) return 1;
            if (!(w = W(L.up, &Q.up, hd)) || launch_gemm_bf16(m->pf_u, m->pf_xn, w, nullptr, T, I, hd, s)) return 1;
            if (launch_silu_mul(m->pf_g, m->pf_g, m->pf_u, (int64_t)T * I, s)) return 1;
        }
        if (!(w = W(L.down, &Q.down, I)) || row_split(h, m->pf_g, w, h2, I)) return 1;
        if (enc && next_tap < enc->n_taps && enc->taps[next_tap] == l) {   // raw hidden state, no final norm (:417-420)
            copy_rows_strided_kernel<<<1024, 256, 0, s>>>(enc->out + (size_t)next_tap * hd, (int64_t)enc->n_taps * hd, h, hd, T, hd / 8);
            OMX_LAUNCH_CHECK();
            ++next_tap;
        }
    }
    return 0;
}

}  // namespace

extern "C" {

int omx_fill_uniform_2d(void* dst, int64_t rows, int64_t cols, int64_t ld_full, int64_t row0, int64_t col0,
                        uint32_t seed, float amp, float offset, omx_dtype dtype, omx_stream stream);

int omx_qwen3_create(omx_qwen3* out, const omx_qwen3_config* cfg) {
    OMX_REQUIRE(out && cfg, "omx_qwen3_create: null argument");
    const omx_qwen3_config& c = *cfg;
    OMX_REQUIRE(c.tp_size >= 1 && c.tp_rank >= 0 && c.tp_rank < c.tp_size, "InvalidConfig: tp rank %d of %d", c.tp_rank, c.tp_size);
    OMX_REQUIRE(c.hidden_size > 0 && c.hidden_size % 64 == 0, "InvalidConfig: hidden_size %d must be a multiple of 64", c.hidden_size);
    OMX_REQUIRE(c.head_dim == 64 || c.head_dim == 128, "InvalidConfig: head_dim %d (64 or 128 supported)", c.head_dim);
    OMX_REQUIRE(c.num_attention_heads % c.num_key_value_heads == 0, "InvalidConfig: heads %d not a multiple of kv heads %d", c.num_attention_heads, c.num_key_value_heads);
    OMX_REQUIRE(c.num_attention_heads % c.tp_size == 0 && c.intermediate_size % c.tp_size == 0 && c.vocab_size % c.tp_size == 0,
                "InvalidConfig: heads %d / intermediate %d / vocab %d must divide by tp_size %d", c.num_attention_heads, c.intermediate_size, c.vocab_size, c.tp_size);
    // KV heads: split over the ranks, or -- with fewer KV heads than ranks -- replicated: tp_size / Hkv ranks share one head
    // (SURVEY.md 8e), which needs that many ranks to divide a query group
    OMX_REQUIRE(c.num_key_value_heads >= c.tp_size ? c.num_key_value_heads % c.tp_size == 0
                    : (c.tp_size % c.num_key_value_heads == 0 &&
                       (c.num_attention_heads / c.num_key_value_heads) % (c.tp_size / c.num_key_value_heads) == 0),
                "InvalidConfig: %d kv heads cannot be split or replicated over tp_size %d", c.num_key_value_heads, c.tp_size);
    OMX_REQUIRE(c.quant_bits == 0 || c.quant_bits == 4 || c.quant_bits == 8, "InvalidConfig: quantization bits %d (0 = bf16, 4, 8)", c.quant_bits);
    // (round 4) quantized checkpoints under tensor parallelism: the packed rows / K slices of the dense model; (round 5) also the packed
    // expert stacks of a sparse-MoE model, expert parallel or expert tensor parallel, with bf16 triplets
    omx_qwen3 m = new omx_qwen3_();
    m->cfg = c;
    if (m->cfg.rope_scale == 0.f) m->cfg.rope_scale = 1.f;
    if (m->cfg.ep_size < 1) m->cfg.ep_size = 1;
    if (m->cfg.quant_bits && m->cfg.quant_group == 0) m->cfg.quant_group = 64;     // nn/quantized.rs:330-333
    OMX_REQUIRE(!m->cfg.quant_bits || m->cfg.quant_group == 32 || m->cfg.quant_group == 64 || m->cfg.quant_group == 128,
                "InvalidConfig: quantization group_size %d (32, 64, 128)", m->cfg.quant_group);
    // (round 5: float16 sparse-MoE checkpoints also expert parallel / expert tensor parallel -- decode form; their prompts go token by token)
    OMX_REQUIRE(!m->cfg.quant_scales_f16 || (m->cfg.quant_bits && m->cfg.head_dim == 128),
                "InvalidConfig: a float16 checkpoint (quantization scales_dtype float16) runs as a packed model with head_dim 128");
    m->H = c.num_attention_heads / c.tp_size;
    m->Hkv = c.num_key_value_heads >= c.tp_size ? c.num_key_value_heads / c.tp_size : 1;
    m->I = c.intermediate_size / c.tp_size;
    m->V = c.vocab_size / c.tp_size;
    OMX_REQUIRE(c.num_experts > 0 || m->I % 64 == 0, "InvalidConfig: per-rank intermediate %d must be a multiple of 64", m->I);
    OMX_REQUIRE(!c.quant_bits || (c.hidden_size % 512 == 0 && (m->H * c.head_dim) % 512 == 0 && (c.num_experts > 0 || m->I % 512 == 0)),
                "InvalidConfig: quantized checkpoints need hidden %d, attention width %d and intermediate %d to be multiples of 512", c.hidden_size, m->H * c.head_dim, m->I);
    OMX_REQUIRE(!c.attention_bias || !c.quant_bits, "InvalidConfig: attention_bias (Qwen2) runs on bf16 checkpoints");
    const int step = 256;   // cache.rs:110-117
    m->cap = ((c.max_context > 0 ? c.max_context : 4096) + step - 1) / step * step;
    OMX_HIP_CHECK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    OMX_HIP_CHECK(hipEventCreate(&m->ev0));
    OMX_HIP_CHECK(hipEventCreate(&m->ev1));
    const int D = c.head_dim, L = c.num_hidden_layers;
    m->kcache.resize(L);
    m->vcache.resize(L);
    for (int l = 0; l < L; ++l) {
        if (dev_alloc(m, &m->kcache[l], (size_t)m->Hkv * m->cap * D)) return 1;
        if (dev_alloc(m, &m->vcache[l], (size_t)m->Hkv * m->cap * D)) return 1;
    }
    if (dev_alloc(m, &m->rope_cos, (size_t)m->cap * D / 2) || dev_alloc(m, &m->rope_sin, (size_t)m->cap * D / 2)) return 1;
    {
        const int n = m->cap * D / 2;
        rope_table_kernel<<<(n + 255) / 256, 256, 0, m->stream>>>(m->rope_cos, m->rope_sin, m->cap, D / 2,
                                                                  -log((double)m->cfg.rope_theta) / (double)(D / 2),
                                                                  (double)m->cfg.rope_scale);
        OMX_LAUNCH_CHECK();
    }
    if (c.num_experts > 0) {
        // tp_size > 1: expert TENSOR parallel -- attention sharded like the dense model, every expert's intermediate columns split over
        // the ranks (decode streams 1 / tp of the two routed experts on every rank; prompts: the expert-parallel batched form over all experts)
        // (packed stacks, round 5: a rank's columns are whole quantisation groups AND whole packed-GEMV steps: multiples of 512)
        OMX_REQUIRE(c.tp_size == 1 || (c.ep_size <= 1 && c.moe_intermediate_size % ((c.quant_bits ? 512 : 64) * c.tp_size) == 0),
                    "InvalidConfig: expert tensor parallelism (tp_size %d with experts) needs ep_size 1 and moe_intermediate_size %d divisible by %d * tp_size",
                    c.tp_size, c.moe_intermediate_size, c.quant_bits ? 512 : 64);
        m->moe_I = c.moe_intermediate_size / c.tp_size;
        OMX_REQUIRE(c.ep_size <= 1 || (c.ep_rank >= 0 && c.ep_rank < c.ep_size && c.num_experts % c.ep_size == 0),
                    "InvalidConfig: expert parallel rank %d of %d over %d experts", c.ep_rank, c.ep_size, c.num_experts);
        OMX_REQUIRE(!c.quant_bits || c.moe_intermediate_size % 512 == 0, "InvalidConfig: quantised experts need moe_intermediate_size %% 512 == 0 (%d)", c.moe_intermediate_size);
        OMX_REQUIRE(c.num_experts_per_tok >= 1 && c.num_experts_per_tok <= c.num_experts && c.moe_intermediate_size > 0 &&
                        c.moe_intermediate_size % 64 == 0 && (c.moe_mode == 0 || c.moe_mode == 1) && (c.tp_size > 1 || m->H * D >= c.hidden_size),
                    "InvalidConfig: experts %d top-%d moe_intermediate_size %d mode %d", c.num_experts, c.num_experts_per_tok,
                    c.moe_intermediate_size, c.moe_mode);
        // the MoE block takes its scratch from the library workspace: size it ONCE for the largest batch (a whole-context
        // prefill) so that the pointers captured in the step graph never move
        size_t need = 0;
        omx_moe_workspace_bytes(m->cap, c.hidden_size, c.moe_intermediate_size, c.num_experts, c.num_experts_per_tok, &need);
        void* ws = nullptr;
        if (get_workspace(&ws, need)) return 1;
        if (c.ep_size > 1 && get_workspace_aux(&ws, need, m->stream)) return 1;   // the expert-parallel block's per-stream scratch (moe.hip)
        if (c.tp_size > 1) {   // expert tensor parallel: the batched pass's plan + slot buffers at this rank's column count
            omx_moe_workspace_bytes(m->cap, c.hidden_size, m->moe_I, c.num_experts, c.num_experts_per_tok, &need);
            if (get_workspace_aux(&ws, need, m->stream)) return 1;
        }
        if (dev_alloc(m, &m->moe_xn, (size_t)c.hidden_size) || dev_alloc(m, &m->moe_out, (size_t)c.hidden_size)) return 1;
        if (c.tp_size > 1 && (dev_alloc(m, &m->moe_y, (size_t)c.num_experts_per_tok * c.hidden_size) ||
                              dev_alloc(m, &m->moe_inds, (size_t)c.num_experts_per_tok) || dev_alloc(m, &m->moe_scores, (size_t)c.num_experts_per_tok)))
            return 1;
    }
    if (dev_alloc(m, &m->rope_cur, (size_t)D) || dev_alloc(m, &m->attn_gran, attn_step_ws_granules(m->H, D)) ||
        dev_alloc(m, &m->attn_xg, (size_t)m->H * D / 2 + 8))
        return 1;
    if (dev_alloc(m, &m->step_seq, 16) || dev_alloc(m, &m->wait_abort, 16)) return 1;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        OMX_HIP_CHECK(hipGetDevice(&dev));
        OMX_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        m->cus = prop.multiProcessorCount;
        if (c.num_experts == 0 && !c.quant_bits && dev_alloc(m, &m->se_gran, step_engine_granules(c.hidden_size, m->H, m->Hkv, D, m->I))) return 1;
    }
    if (dev_alloc(m, &m->st, 1) || dev_alloc(m, &m->out_ring, (size_t)m->ring_cap) ||
        dev_alloc(m, &m->h, (size_t)c.hidden_size) || dev_alloc(m, &m->h2, (size_t)c.hidden_size) ||
        dev_alloc(m, &m->qkv, (size_t)(m->H + 2 * m->Hkv) * D) || dev_alloc(m, &m->attn_out, (size_t)m->H * D) ||
        dev_alloc(m, &m->act, (size_t)m->I) || dev_alloc(m, &m->logits, (size_t)m->V) ||
        dev_alloc(m, &m->partial_a, (size_t)c.hidden_size) || dev_alloc(m, &m->partial_b, (size_t)c.hidden_size) ||
        dev_alloc(m, &m->argmax_key, 1))
        return 1;
    if (c.num_experts > 0 && dev_alloc(m, &m->moe_partials, (size_t)c.num_experts_per_tok * c.hidden_size)) return 1;
    m->prompt_cap = m->cap;
    if (dev_alloc(m, &m->prompt_dev, (size_t)m->prompt_cap + 1)) return 1;
    m->n_argmax_partials = m->cfg.quant_bits ? qgemv_grid(m->V) : gemv_grid(m->V, c.hidden_size, EPI_ARGMAX, 0);
    if (dev_alloc(m, &m->argmax_partials, (size_t)m->n_argmax_partials)) return 1;
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    *out = m;
    return 0;
}

int omx_qwen3_destroy(omx_qwen3 m) {
    if (!m) return 0;
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    drop_graphs(m);
    for (const bf16_t* k : m->sb_keys) quant_unregister_sb(k);
    for (void* p : m->owned) (void)hipFree(p);
    if (m->dq_buf) (void)hipFree(m->dq_buf);
    if (m->dq_slab) (void)hipFree(m->dq_slab);
    if (m->verify_logits) (void)hipFree(m->verify_logits);
    if (m->verify_tokens) (void)hipFree(m->verify_tokens);
    if (m->pf_ep_partial) (void)hipFree(m->pf_ep_partial);
    for (bf16_t* p : {m->pf_h, m->pf_h2, m->pf_xn, m->pf_q, m->pf_k, m->pf_v, m->pf_qt, m->pf_attn, m->pf_g, m->pf_u})
        if (p) (void)hipFree(p);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->stream) {
        gemm_release_stream(m->stream);
        workspace_release_stream(m->stream);
        (void)hipStreamDestroy(m->stream);
    }
    delete m;
    return 0;
}

// bytes the forward will read behind checkpoint tensor `name` on THIS rank (after the TP / EP slicing), 0 for a name it does not
// use; mirrors resolve_weights above (and engine.py expected_shape, which reports the same thing as a shape)
static size_t expected_weight_bytes(omx_qwen3 m, const std::string& name) {
    const omx_qwen3_config& c = m->cfg;
    const size_t hd = c.hidden_size, D = c.head_dim, H = m->H, Hkv = m->Hkv, I = m->I, Im = c.tp_size > 1 && c.num_experts > 0 ? m->moe_I : c.moe_intermediate_size;   // (expert tensor parallel: this rank's columns)
    const size_t E = c.num_experts, El = c.ep_size > 1 ? E / c.ep_size : E;
    static const char* kSuffix[] = {".weight", ".scales", ".biases", ".bias"};
    int kind = -1;
    std::string stem;
    for (int i = 0; i < 4; ++i) {
        const size_t n = strlen(kSuffix[i]);
        if (name.size() > n && name.compare(name.size() - n, n, kSuffix[i]) == 0) { kind = i; stem = name.substr(0, name.size() - n); break; }
    }
    if (kind < 0) return 0;
    const bool quant = c.quant_bits != 0;
    // [n, k] Linear (x stack): dense bf16, or the packed triplet of a quantized checkpoint; bias [n]
    auto lin = [&](size_t n, size_t k, size_t stack = 1) -> size_t {
        if (kind == 3) return n * 2;
        if (!quant) return kind == 0 ? stack * n * k * 2 : 0;
        if (kind == 0) return stack * n * (k * c.quant_bits / 32) * 4;
        return stack * n * (k / c.quant_group) * 2;
    };
    auto vec = [&](size_t n) -> size_t { return kind == 0 ? n * 2 : 0; };
    if (stem == "model.embed_tokens") return lin(c.vocab_size, hd);
    if (stem == "lm_head") return lin(m->V, hd);
    if (stem == "model.norm") return vec(hd);
    if (stem.compare(0, 13, "model.layers.") != 0) return 0;
    const size_t dot = stem.find('.', 13);
    if (dot == std::string::npos) return 0;
    const std::string sub = stem.substr(dot + 1);
    if (sub == "self_attn.q_proj") return lin(H * D, hd);
    if (sub == "self_attn.k_proj" || sub == "self_attn.v_proj") return lin(Hkv * D, hd);
    if (sub == "self_attn.o_proj") return lin(hd, H * D);
    if (sub == "mlp.gate_proj" || sub == "mlp.up_proj") return lin(I, hd);
    if (sub == "mlp.down_proj") return lin(hd, I);
    if (sub == "input_layernorm" || sub == "post_attention_layernorm") return vec(hd);
    if (sub == "self_attn.q_norm" || sub == "self_attn.k_norm") return vec(D);
    if (E > 0)
        for (const char* mp : {"block_sparse_moe.", "mlp."}) {
            const std::string p = mp;
            if (sub == p + "gate") return lin(E, hd);
            if (sub == p + "switch_mlp.gate_proj" || sub == p + "switch_mlp.up_proj") return lin(Im, hd, El);
            if (sub == p + "switch_mlp.down_proj") return lin(hd, Im, El);
        }
    return 0;
}

int omx_qwen3_set_weight(omx_qwen3 m, const char* name, const void* ptr, size_t nbytes) {
    OMX_REQUIRE(m && name && ptr, "omx_qwen3_set_weight: null argument");
    // the engine reads raw device pointers: a tensor shorter than the config implies would be read past its end, so the size is part
    // of the call (the reference raises a shape error on load)
    const size_t want = expected_weight_bytes(m, name);
    OMX_REQUIRE(want == 0 || nbytes == want, "ShapeMismatch: %s holds %zu bytes, the config expects %zu", name, nbytes, want);
    OMX_REQUIRE(((uintptr_t)ptr & 15u) == 0, "omx_qwen3_set_weight: %s is not 16-byte aligned", name);
    OMX_REQUIRE(m->g_full == nullptr, "omx_qwen3_set_weight: weights are frozen once the decode step is built");
    m->named[name] = ptr;
    m->weights_resolved = false;
    return 0;
}

// device pointer of a registered (or synthesised) tensor by its checkpoint name: what a caller that ALSO drives the per-op mlx-c route on the
// same weights needs (omx_mlx_array_from_device wraps it; bench: per_op_route.hip)
int omx_qwen3_get_weight(omx_qwen3 m, const char* name, const void** ptr, size_t* nbytes) {
    OMX_REQUIRE(m && name && ptr, "omx_qwen3_get_weight: null argument");
    auto it = m->named.find(name);
    if (it == m->named.end()) return set_error("WeightNotFound: %s", name);
    *ptr = it->second;
    if (nbytes) *nbytes = expected_weight_bytes(m, name);
    return 0;
}

static int synth_weights_impl(omx_qwen3 m, uint32_t base_seed, bool peaked);
int omx_qwen3_synth_weights(omx_qwen3 m, uint32_t base_seed) { return synth_weights_impl(m, base_seed, false); }
/* The same synthetic checkpoint with PEAKED logits (parity at full size: i.i.d. weights give flat logits whose argmax flips on the last
 * bf16 bit, so token equality cannot be a hard assert): the embedding table is scaled to std 64 -- it dominates the ~10-rms sum of the
 * 36 layers' contributions -- and lm_head row v is row (v + 1) mod V of the SAME table at the usual std 0.02, so the greedy token after
 * token t is t - 1 with a top-1 margin of ~80 against a bf16 bound of ~0.5, while the other 151 935 logits still carry the layers'
 * arithmetic (std 0.2 of their 1.3).  oracle/ref_qwen3.py synth_weights(peaked=True) is the host twin. */
int omx_qwen3_synth_weights_peaked(omx_qwen3 m, uint32_t base_seed) { return synth_weights_impl(m, base_seed, true); }
static int synth_weights_impl(omx_qwen3 m, uint32_t base_seed, bool peaked) {
    OMX_REQUIRE(m, "omx_qwen3_synth_weights: null model");
    OMX_REQUIRE(!peaked || (m->cfg.quant_bits == 0 && !m->cfg.tie_word_embeddings), "omx_qwen3_synth_weights_peaked: bf16 checkpoints with an untied lm_head only");
    OMX_REQUIRE(!m->cfg.quant_scales_f16, "omx_qwen3_synth_weights: the device generator quantises in bf16; a float16-scale model takes uploaded triplets");
    const omx_qwen3_config& c = m->cfg;
    const int D = c.head_dim, hd = c.hidden_size, r = c.tp_rank;
    const float amp_w = (float)(0.02 * sqrt(3.0)), amp_n = (float)(0.01 * sqrt(3.0));   // == oracle/synth.py
    // logical tensor [rows_full, cols_full]; this rank holds rows [row0, row0+rows) x cols [col0, col0+cols)
    auto make = [&](const std::string& name, int64_t rows, int64_t cols, int64_t ld_full, int64_t row0, int64_t col0,
                    bool is_norm) -> int {
        bf16_t* p = nullptr;
        if (dev_alloc(m, &p, (size_t)rows * cols)) return 1;
        const uint32_t seed = base_seed ^ crc32_str(name.c_str());
        if (omx_fill_uniform_2d(p, rows, cols, ld_full, row0, col0, seed, is_norm ? amp_n : amp_w, is_norm ? 1.0f : 0.0f,
                                OMX_BFLOAT16, m->stream))
            return 1;
        m->named[name] = p;
        return 0;
    };
    const int Hq = m->H * D, Hk = m->Hkv * D;
    // first k / v row of this rank in the logical projection: its own KV heads, or the one head it shares with its neighbours
    const int kv_rep = c.num_key_value_heads >= c.tp_size ? 1 : c.tp_size / c.num_key_value_heads;
    const int64_t kv_row0 = (int64_t)(r / kv_rep) * Hk;
    if (c.quant_bits) {
        // the quantized model IS mlx quantize() of the synthetic bf16 model: generate each logical matrix into a scratch
        // buffer with the bf16 generator, quantise it on the device, keep only the (weight, scales, biases) triplet
        bf16_t* scratch = nullptr;
        size_t biggest = (size_t)std::max((int64_t)c.vocab_size, (int64_t)std::max(m->I, Hq)) * (size_t)std::max(hd, m->I);
        if (c.num_experts > 0) biggest = std::max(biggest, (size_t)c.num_experts * c.moe_intermediate_size * (size_t)hd);   // a whole expert stack
        OMX_HIP_CHECK(hipMalloc((void**)&scratch, biggest * 2));
        // (ld_full, row0, col0): this rank's window of the logical matrix -- quantisation is per group of one row, so the window's
        // triplet IS the slice of the whole matrix's triplet (K slices hold whole groups); seed_of: the logical tensor the values belong to
        auto makeq = [&](const std::string& prefix, int64_t rows, int64_t cols, int64_t ld_full = 0, int64_t row0 = 0, int64_t col0 = 0,
                         const char* seed_of = nullptr) -> int {
            const uint32_t seed = base_seed ^ crc32_str(seed_of ? seed_of : (prefix + ".weight").c_str());
            if (omx_fill_uniform_2d(scratch, rows, cols, ld_full ? ld_full : cols, row0, col0, seed, amp_w, 0.0f, OMX_BFLOAT16, m->stream)) return 1;
            uint32_t* pk = nullptr;
            bf16_t *sc = nullptr, *bi = nullptr;
            if (dev_alloc(m, &pk, (size_t)(rows * cols * c.quant_bits / 32)) || dev_alloc(m, &sc, (size_t)(rows * cols / c.quant_group)) ||
                dev_alloc(m, &bi, (size_t)(rows * cols / c.quant_group)))
                return 1;
            if (omx_quantize(pk, sc, bi, scratch, rows, (int)cols, c.quant_group, c.quant_bits, OMX_BFLOAT16, m->stream)) return 1;
            m->named[prefix + ".weight"] = pk;
            m->named[prefix + ".scales"] = sc;
            m->named[prefix + ".biases"] = bi;
            return 0;
        };
        int rc = 0;
        for (int i = 0; i < c.num_hidden_layers && !rc; ++i) {
            const std::string p = "model.layers." + std::to_string(i) + ".";
            rc = makeq(p + "self_attn.q_proj", Hq, hd, hd, (int64_t)r * Hq) || makeq(p + "self_attn.k_proj", Hk, hd, hd, kv_row0) ||
                 makeq(p + "self_attn.v_proj", Hk, hd, hd, kv_row0) ||
                 makeq(p + "self_attn.o_proj", hd, Hq, (int64_t)c.num_attention_heads * D, 0, (int64_t)r * Hq) || make(p + "input_layernorm.weight", 1, hd, hd, 0, 0, true) ||
                 make(p + "post_attention_layernorm.weight", 1, hd, hd, 0, 0, true);
            if (!rc && !c.no_qk_norm) rc = make(p + "self_attn.q_norm.weight", 1, D, D, 0, 0, true) || make(p + "self_attn.k_norm.weight", 1, D, D, 0, 0, true);
            if (!rc && c.num_experts > 0) {
                const std::string mp = p + (c.moe_mode == 0 ? "block_sparse_moe." : "mlp.");
                const int64_t E = c.num_experts, Im = c.moe_intermediate_size;
                if (c.tp_size > 1) {
                    // expert tensor parallel: rows [r I_l, + I_l) of every expert's gate / up (E strided windows of the logical stack, gathered
                    // into the scratch before ONE quantise call), the same columns -- whole groups -- of its down projection
                    const int64_t Il = m->moe_I;
                    auto makeq_rows = [&](const std::string& prefix) -> int {
                        const uint32_t seed = base_seed ^ crc32_str((prefix + ".weight").c_str());
                        for (int64_t e = 0; e < E; ++e)
                            if (omx_fill_uniform_2d(scratch + e * Il * hd, Il, hd, hd, e * Im + (int64_t)r * Il, 0, seed, amp_w, 0.0f, OMX_BFLOAT16, m->stream)) return 1;
                        uint32_t* pk = nullptr;
                        bf16_t *sc = nullptr, *bi = nullptr;
                        const int64_t rows = E * Il;
                        if (dev_alloc(m, &pk, (size_t)(rows * hd * c.quant_bits / 32)) || dev_alloc(m, &sc, (size_t)(rows * hd / c.quant_group)) ||
                            dev_alloc(m, &bi, (size_t)(rows * hd / c.quant_group)))
                            return 1;
                        if (omx_quantize(pk, sc, bi, scratch, rows, hd, c.quant_group, c.quant_bits, OMX_BFLOAT16, m->stream)) return 1;
                        m->named[prefix + ".weight"] = pk; m->named[prefix + ".scales"] = sc; m->named[prefix + ".biases"] = bi;
                        return 0;
                    };
                    rc = makeq(mp + "gate", E, hd) || makeq_rows(mp + "switch_mlp.gate_proj") || makeq_rows(mp + "switch_mlp.up_proj") ||
                         makeq(mp + "switch_mlp.down_proj", E * hd, Il, Im, 0, (int64_t)r * Il);
                } else {
                    const int64_t El = c.ep_size > 1 ? E / c.ep_size : E, e0 = c.ep_size > 1 ? c.ep_rank * El : 0;   // this rank's experts
                    rc = makeq(mp + "gate", E, hd) || makeq(mp + "switch_mlp.gate_proj", El * Im, hd, hd, e0 * Im) ||
                         makeq(mp + "switch_mlp.up_proj", El * Im, hd, hd, e0 * Im) || makeq(mp + "switch_mlp.down_proj", El * hd, Im, Im, e0 * hd);
                }
            } else if (!rc) {
                rc = makeq(p + "mlp.gate_proj", m->I, hd, hd, (int64_t)r * m->I) || makeq(p + "mlp.up_proj", m->I, hd, hd, (int64_t)r * m->I) ||
                     makeq(p + "mlp.down_proj", hd, m->I, c.intermediate_size, 0, (int64_t)r * m->I);
            }
        }
        rc = rc || makeq("model.embed_tokens", c.vocab_size, hd) || make("model.norm.weight", 1, hd, hd, 0, 0, true);
        if (!rc && !c.tie_word_embeddings) rc = makeq("lm_head", m->V, hd, hd, (int64_t)r * m->V);
        else if (!rc && c.tp_size > 1) rc = makeq("lm_head", m->V, hd, hd, (int64_t)r * m->V, 0, "model.embed_tokens.weight");   // tied: the table's shard
        (void)hipStreamSynchronize(m->stream);
        (void)hipFree(scratch);
        m->weights_resolved = false;
        return rc;
    }
    for (int i = 0; i < c.num_hidden_layers; ++i) {
        const std::string p = "model.layers." + std::to_string(i) + ".";
        if (make(p + "self_attn.q_proj.weight", Hq, hd, hd, (int64_t)r * Hq, 0, false) ||
            make(p + "self_attn.k_proj.weight", Hk, hd, hd, kv_row0, 0, false) ||
            make(p + "self_attn.v_proj.weight", Hk, hd, hd, kv_row0, 0, false) ||
            make(p + "self_attn.o_proj.weight", hd, Hq, (int64_t)c.num_attention_heads * D, 0, (int64_t)r * Hq, false) ||
            make(p + "input_layernorm.weight", 1, hd, hd, 0, 0, true) ||
            make(p + "post_attention_layernorm.weight", 1, hd, hd, 0, 0, true))
            return 1;
        if (!c.no_qk_norm && (make(p + "self_attn.q_norm.weight", 1, D, D, 0, 0, true) || make(p + "self_attn.k_norm.weight", 1, D, D, 0, 0, true)))
            return 1;
        // biases: this rank's columns of the logical [1, H_total * D] vector -- the same offsets as the rows of its projection
        const int64_t Hq_all = (int64_t)c.num_attention_heads * D, Hk_all = (int64_t)c.num_key_value_heads * D;
        if (c.attention_bias && (make(p + "self_attn.q_proj.bias", 1, Hq, Hq_all, 0, (int64_t)r * Hq, false) ||
                                 make(p + "self_attn.k_proj.bias", 1, Hk, Hk_all, 0, kv_row0, false) ||
                                 make(p + "self_attn.v_proj.bias", 1, Hk, Hk_all, 0, kv_row0, false)))
            return 1;
        if (c.num_experts > 0) {
            const std::string mp = p + (c.moe_mode == 0 ? "block_sparse_moe." : "mlp.");
            const int64_t E = c.num_experts, Im = c.moe_intermediate_size;
            const int64_t El = c.ep_size > 1 ? E / c.ep_size : E, e0 = c.ep_size > 1 ? c.ep_rank * El : 0;   // this rank's experts
            if (c.tp_size > 1) {
                // expert tensor parallel: rows [r I_l, +I_l) of every expert's gate / up, the same columns of its down projection
                const int64_t Il = m->moe_I;
                bf16_t *wg = nullptr, *wu = nullptr;
                if (dev_alloc(m, &wg, (size_t)(E * Il * hd)) || dev_alloc(m, &wu, (size_t)(E * Il * hd))) return 1;
                const uint32_t sg = base_seed ^ crc32_str((mp + "switch_mlp.gate_proj.weight").c_str());
                const uint32_t su = base_seed ^ crc32_str((mp + "switch_mlp.up_proj.weight").c_str());
                for (int64_t e = 0; e < E; ++e)
                    if (omx_fill_uniform_2d(wg + e * Il * hd, Il, hd, hd, e * Im + (int64_t)r * Il, 0, sg, amp_w, 0.f, OMX_BFLOAT16, m->stream) ||
                        omx_fill_uniform_2d(wu + e * Il * hd, Il, hd, hd, e * Im + (int64_t)r * Il, 0, su, amp_w, 0.f, OMX_BFLOAT16, m->stream))
                        return 1;
                m->named[mp + "switch_mlp.gate_proj.weight"] = wg;
                m->named[mp + "switch_mlp.up_proj.weight"] = wu;
                if (make(mp + "gate.weight", E, hd, hd, 0, 0, false) ||
                    make(mp + "switch_mlp.down_proj.weight", E * hd, Il, Im, 0, (int64_t)r * Il, false))
                    return 1;
            } else
            if (make(mp + "gate.weight", E, hd, hd, 0, 0, false) ||
                make(mp + "switch_mlp.gate_proj.weight", El * Im, hd, hd, e0 * Im, 0, false) ||
                make(mp + "switch_mlp.up_proj.weight", El * Im, hd, hd, e0 * Im, 0, false) ||
                make(mp + "switch_mlp.down_proj.weight", El * hd, Im, Im, e0 * hd, 0, false))
                return 1;
        } else if (make(p + "mlp.gate_proj.weight", m->I, hd, hd, (int64_t)r * m->I, 0, false) ||
                   make(p + "mlp.up_proj.weight", m->I, hd, hd, (int64_t)r * m->I, 0, false) ||
                   make(p + "mlp.down_proj.weight", hd, m->I, c.intermediate_size, 0, (int64_t)r * m->I, false)) {
            return 1;
        }
    }
    if (peaked) {
        const uint32_t seed = base_seed ^ crc32_str("model.embed_tokens.weight");
        bf16_t *e = nullptr, *hw = nullptr;
        if (dev_alloc(m, &e, (size_t)c.vocab_size * hd) || dev_alloc(m, &hw, (size_t)m->V * hd)) return 1;
        if (omx_fill_uniform_2d(e, c.vocab_size, hd, hd, 0, 0, seed, (float)(64.0 * sqrt(3.0)), 0.f, OMX_BFLOAT16, m->stream)) return 1;
        // this rank's head rows [r V_l, (r + 1) V_l) = table rows shifted by one, wrapping at the end of the vocabulary
        const int64_t first = (int64_t)r * m->V + 1, n_main = std::min<int64_t>(m->V, c.vocab_size - first);
        if (n_main > 0 && omx_fill_uniform_2d(hw, n_main, hd, hd, first, 0, seed, amp_w, 0.f, OMX_BFLOAT16, m->stream)) return 1;
        if (n_main < m->V && omx_fill_uniform_2d(hw + (size_t)std::max<int64_t>(n_main, 0) * hd, m->V - std::max<int64_t>(n_main, 0), hd, hd, 0, 0, seed,
                                                 amp_w, 0.f, OMX_BFLOAT16, m->stream))
            return 1;
        m->named["model.embed_tokens.weight"] = e;
        m->named["lm_head.weight"] = hw;
        if (make("model.norm.weight", 1, hd, hd, 0, 0, true)) return 1;
    } else {
    if (make("model.embed_tokens.weight", c.vocab_size, hd, hd, 0, 0, false) || make("model.norm.weight", 1, hd, hd, 0, 0, true))
        return 1;
    if (!c.tie_word_embeddings) {
        if (make("lm_head.weight", m->V, hd, hd, (int64_t)r * m->V, 0, false)) return 1;
    } else if (c.tp_size > 1) {
        // vocab shard of the tied table, same logical values as model.embed_tokens.weight
        bf16_t* p = nullptr;
        if (dev_alloc(m, &p, (size_t)m->V * hd)) return 1;
        const uint32_t seed = base_seed ^ crc32_str("model.embed_tokens.weight");
        if (omx_fill_uniform_2d(p, m->V, hd, hd, (int64_t)r * m->V, 0, seed, amp_w, 0.f, OMX_BFLOAT16, m->stream)) return 1;
        m->named["lm_head.weight"] = p;
    }
    }
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    m->weights_resolved = false;
    return 0;
}

int omx_qwen3_set_comm(omx_qwen3 m, void* comm, void* allreduce_fn) {
    OMX_REQUIRE(m, "omx_qwen3_set_comm: null model");
    OMX_REQUIRE(m->g_full == nullptr && !m->eager, "omx_qwen3_set_comm: communicator must be set before the first step");
    m->comm = comm;
    m->allreduce = (nccl_allreduce_fn)allreduce_fn;
    // OMX_PEER_FUSED=1: the O / down GEMVs reduce their own rows over the peers in their epilogue instead of a standalone
    // all-reduce kernel after them.  Opt-in: measured on one GPU (1-rank communicator, Qwen3-8B) the in-GEMV poll costs 6.5 us per
    // GEMV against 4.4 us for the extra launch -- its uncached loads queue behind the other waves' weight stream
    const char* fe = getenv("OMX_PEER_FUSED");
    m->peer_dev = (allreduce_fn == omx_peer_allreduce_fn() && fe && fe[0] == '1' && m->cfg.hidden_size <= kPeerMaxWords)
                      ? static_cast<const PeerDev*>(omx_peer_comm_device(comm)) : nullptr;
    return 0;
}

int omx_qwen3_set_sampler(omx_qwen3 m, float temperature, uint64_t seed) {
    OMX_REQUIRE(m, "omx_qwen3_set_sampler: null model");
    OMX_REQUIRE(temperature >= 0.f && temperature == temperature, "omx_qwen3_set_sampler: temperature %f must be >= 0", (double)temperature);
    if (!m->rng && dev_alloc(m, &m->rng, 4)) return 1;
    if (omx_random_key(m->rng, seed, (omx_stream)m->stream)) return 1;
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (temperature != m->temperature) {
        // 1/T is a launch argument inside the captured step: drop the graphs, the next step rebuilds them
        drop_graphs(m);
        m->eager = false;
        m->temperature = temperature;
    }
    return 0;
}

/* The sampler's key-sequence state (mlx-rs RandomState: two words).  The reference's speculative loop draws the draft's and the
 * target's tokens from ONE global sequence (speculative.rs:104-109: `categorical!` without a key); two engine models reproduce that by
 * handing this state back and forth.  set != 0 writes `state2` into the model, otherwise the model's state is read out. */
int omx_qwen3_sampler_state(omx_qwen3 m, uint32_t* state2, int set) {
    OMX_REQUIRE(m && state2, "omx_qwen3_sampler_state: null argument");
    OMX_REQUIRE(m->rng, "omx_qwen3_sampler_state: no sampler set (omx_qwen3_set_sampler)");
    if (set) OMX_HIP_CHECK(hipMemcpyAsync(m->rng, state2, 8, hipMemcpyHostToDevice, m->stream));
    else OMX_HIP_CHECK(hipMemcpyAsync(state2, m->rng, 8, hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    return 0;
}

int omx_qwen3_encode(omx_qwen3 m, const uint32_t* ids, int n, const uint8_t* attention_mask, const int* tap_layers, int n_taps,
                     void* out_dev) {
    OMX_REQUIRE(m && ids && tap_layers && out_dev, "omx_qwen3_encode: null argument");
    OMX_REQUIRE(n >= 1 && n <= m->cap, "omx_qwen3_encode: %d tokens exceed max_context %d", n, m->cap);
    OMX_REQUIRE(n_taps >= 1 && n_taps <= 16, "omx_qwen3_encode: %d taps (1..16)", n_taps);
    OMX_REQUIRE(m->cfg.tp_size == 1 && !m->allreduce, "omx_qwen3_encode: single-GPU only");
    for (int i = 0; i < n_taps; ++i)
        OMX_REQUIRE(tap_layers[i] >= 0 && tap_layers[i] < m->cfg.num_hidden_layers && (i == 0 || tap_layers[i] > tap_layers[i - 1]),
                    "omx_qwen3_encode: tap layers must be ascending and < %d", m->cfg.num_hidden_layers);
    if (resolve_weights(m)) return 1;
    hipStream_t s = m->stream;
    OMX_HIP_CHECK(hipMemcpyAsync(m->prompt_dev, ids, (size_t)n * 4, hipMemcpyHostToDevice, s));
    bf16_t* mask = nullptr;
    uint8_t* am = nullptr;
    if (attention_mask) {
        OMX_HIP_CHECK(hipMalloc((void**)&mask, (size_t)n * n * 2));
        OMX_HIP_CHECK(hipMalloc((void**)&am, (size_t)n));
        OMX_HIP_CHECK(hipMemcpyAsync(am, attention_mask, (size_t)n, hipMemcpyHostToDevice, s));
        encoder_mask_kernel<<<512, 256, 0, s>>>(mask, am, n);
    }
    const EncodeOpts enc = {tap_layers, n_taps, (bf16_t*)out_dev, mask};
    OMX_HIP_CHECK(hipEventRecord(m->ev0, s));
    const int rc = prefill_prefix_batched(m, n, 0, &enc);
    OMX_HIP_CHECK(hipEventRecord(m->ev1, s));
    OMX_HIP_CHECK(hipStreamSynchronize(s));
    if (mask) { (void)hipFree(mask); (void)hipFree(am); }
    if (rc) return 1;
    OMX_HIP_CHECK(hipEventElapsedTime(&m->last_prefill_ms, m->ev0, m->ev1));
    return 0;
}

int omx_qwen3_reset(omx_qwen3 m) {
    OMX_REQUIRE(m, "omx_qwen3_reset: null model");
    OMX_HIP_CHECK(hipMemsetAsync(m->st, 0, sizeof(StepState), m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    return 0;
}

int omx_qwen3_offset(omx_qwen3 m, int* offset) {
    OMX_REQUIRE(m && offset, "omx_qwen3_offset: null argument");
    StepState st;
    OMX_HIP_CHECK(hipMemcpyAsync(&st, m->st, sizeof(st), hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    *offset = st.pos;
    return 0;
}

int omx_qwen3_prefill(omx_qwen3 m, const uint32_t* prompt, int n_prompt, uint32_t* first_token) {
    OMX_REQUIRE(m && prompt && first_token, "omx_qwen3_prefill: null argument");
    OMX_REQUIRE(n_prompt >= 1, "omx_qwen3_prefill: empty prompt");
    int off = 0;
    if (omx_qwen3_offset(m, &off)) return 1;
    OMX_REQUIRE(off + n_prompt + 1 <= m->cap, "omx_qwen3_prefill: %d cached + %d prompt tokens exceed max_context %d", off, n_prompt, m->cap);
    for (int i = 0; i < n_prompt; ++i) OMX_REQUIRE(prompt[i] < (uint32_t)m->cfg.vocab_size, "omx_qwen3_prefill: token id %u out of range (vocab %d)", prompt[i], m->cfg.vocab_size);
    OMX_REQUIRE(n_prompt <= m->prompt_cap, "omx_qwen3_prefill: prompt of %d tokens exceeds max_context %d", n_prompt, m->prompt_cap);
    const char* serial_env = getenv("OMX_PREFILL_SERIAL");
    // tensor-parallel engines run the batched matrix-core prefill on their shards with two all-reduces per layer, expert-parallel ones
    // with one all-reduce of the MoE block's [T, hidden] partial per layer (round 3; token-serial before)
    // (float16 models: the batched pass exists for plain prompts -- dense and sparse-MoE models, on one rank or sharded (round 6: the
    //  float16 form of the sharded MoE block); short prompts go through the decode step)
    const bool f16_serial = m->cfg.quant_scales_f16 && n_prompt <= 16;
    const bool serial = (serial_env && serial_env[0] == '1') || n_prompt < 2 || f16_serial;
    if (prepare_step(m, serial ? off : off + n_prompt - 1)) return 1;   // the first step this call will run (graphs are per context bucket)
    if (!serial && m->cfg.quant_bits) dq_cache_prepare(m);             // (a once-per-model allocation: ahead of the timed region)
    if (!serial && prefill_reserve(m, n_prompt)) return 1;             // (row buffers / scratch of this prompt size: likewise)
    OMX_HIP_CHECK(hipMemcpyAsync(m->prompt_dev, prompt, (size_t)n_prompt * 4, hipMemcpyHostToDevice, m->stream));
    StepState st;
    OMX_HIP_CHECK(hipMemcpyAsync(&st, m->st, sizeof(st), hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    st.cur_token = prompt[0];
    st.prompt_idx = 0;
    const int count_before = st.out_count;
    OMX_HIP_CHECK(hipEventRecord(m->ev0, m->stream));
    bool batched_head = false;
    if (serial) {
        // token-serial prefill: identical arithmetic to n_prompt decode steps (the lm_head is skipped for
        // all but the last prompt position; the reference computes and discards those logits, model.rs:815)
        OMX_HIP_CHECK(hipMemcpyAsync(m->st, &st, sizeof(st), hipMemcpyHostToDevice, m->stream));
        for (int i = 0; i < n_prompt - 1; ++i)
            if (run_step(m, false, st.pos + i)) return 1;
    } else {
        // matrix-core prefill of ALL n tokens, then norm + lm_head + sampler on the last row (one more row in GEMMs whose
        // tile count does not change, instead of a 36-layer GEMV pass); OMX_PREFILL_TAIL_STEP=1: n-1 tokens batched and
        // the decode step for the last one
        const char* tail_env = getenv("OMX_PREFILL_TAIL_STEP");
        // (a tensor-parallel rank's vocabulary shard + the argmax all-reduce: enqueue_head_on_row does both since round 4 -- until then a
        //  TP prompt ended with a whole decode step, 2 ms of a 24 ms prompt at TP 2)
        const bool tail_step = tail_env && tail_env[0] == '1';
        const int nb = tail_step ? n_prompt - 1 : n_prompt;
        if (prefill_prefix_batched(m, nb, st.pos, nullptr, !tail_step)) return 1;
        st.pos += n_prompt - 1;
        st.prompt_idx = n_prompt - 1;
        st.cur_token = prompt[n_prompt - 1];
        OMX_HIP_CHECK(hipMemcpyAsync(m->st, &st, sizeof(st), hipMemcpyHostToDevice, m->stream));
        if (!tail_step) {
            if (enqueue_head_on_row(m, m->pf_h + (size_t)(n_prompt - 1) * m->cfg.hidden_size, m->stream)) return 1;
            batched_head = true;
        }
    }
    if (!batched_head && run_step(m, true, serial ? st.pos + n_prompt - 1 : st.pos)) return 1;
    OMX_HIP_CHECK(hipEventRecord(m->ev1, m->stream));
    OMX_HIP_CHECK(hipMemcpyAsync(first_token, m->out_ring + (count_before % m->ring_cap), 4, hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    OMX_HIP_CHECK(hipEventElapsedTime(&m->last_prefill_ms, m->ev0, m->ev1));
    return step_health(m);
}

/* Speculative decoding (mlx-rs-core/src/speculative.rs).  `verify` is verify_draft_tokens (:132-161): the target model runs ALL n
 * tokens [last accepted, draft 1 .. draft n-1] in one batched matrix-core pass on top of its cache (their K/V rows are appended) and
 * returns the greedy token of every position -- the lm_head is one [n, V] GEMM, the weights stream once for all rows.  Afterwards the
 * cache holds n more tokens and the next step's input token is greedy_out[n-1]; the caller then drops the rejected tail with
 * omx_qwen3_trim.  Single-rank bf16 weights (a vocabulary-sharded or quantized head has no batched form here). */
int omx_qwen3_verify(omx_qwen3 m, const uint32_t* tokens, int n, uint32_t* greedy_out) {
    OMX_REQUIRE(m && tokens && greedy_out, "omx_qwen3_verify: null argument");
    OMX_REQUIRE(n >= 1 && n <= 64, "omx_qwen3_verify: %d tokens (1..64 per call)", n);
    OMX_REQUIRE(m->allreduce == nullptr && m->cfg.quant_bits == 0, "omx_qwen3_verify: single-rank bf16 models only");
    for (int i = 0; i < n; ++i) OMX_REQUIRE(tokens[i] < (uint32_t)m->cfg.vocab_size, "omx_qwen3_verify: token id %u out of range (vocab %d)", tokens[i], m->cfg.vocab_size);
    StepState st;
    OMX_HIP_CHECK(hipMemcpyAsync(&st, m->st, sizeof(st), hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    OMX_REQUIRE(st.pos + n + 1 <= m->cap, "omx_qwen3_verify: %d cached + %d tokens exceed max_context %d", st.pos, n, m->cap);
    OMX_REQUIRE(n <= m->prompt_cap, "omx_qwen3_verify: %d tokens exceed the prompt buffer", n);
    if (resolve_weights(m)) return 1;          // (verify may be the first call on a fresh model)
    hipStream_t s = m->stream;
    const int hd = m->cfg.hidden_size, V = m->V;
    if (n > m->verify_cap) {
        OMX_HIP_CHECK(hipStreamSynchronize(s));
        if (m->verify_logits) OMX_HIP_CHECK(hipFree(m->verify_logits));
        if (m->verify_tokens) OMX_HIP_CHECK(hipFree(m->verify_tokens));
        m->verify_logits = nullptr; m->verify_tokens = nullptr; m->verify_cap = 0;
        const int cap = std::max(n, 16);
        OMX_HIP_CHECK(hipMalloc((void**)&m->verify_logits, (size_t)cap * V * sizeof(bf16_t)));
        OMX_HIP_CHECK(hipMalloc((void**)&m->verify_tokens, (size_t)cap * 4));
        m->verify_cap = cap;
    }
    OMX_HIP_CHECK(hipMemcpyAsync(m->prompt_dev, tokens, (size_t)n * 4, hipMemcpyHostToDevice, s));
    if (prefill_prefix_batched(m, n, st.pos, nullptr, /*full_last=*/true)) return 1;
    // [final RMSNorm rows] -> [lm_head GEMM, n x V] -> [argmax per row]   (model.rs:423, 480-489; sampler.rs:9-18 at temperature 0)
    if (omx_rms_norm(m->pf_xn, m->pf_h, m->final_norm, n, hd, m->cfg.rms_norm_eps, OMX_BFLOAT16, s)) return 1;
    if (launch_gemm_bf16(m->verify_logits, m->pf_xn, m->lm_head, nullptr, n, V, hd, s)) return 1;
    if (m->temperature == 0.f) {
        if (omx_argmax(m->verify_tokens, m->verify_logits, n, V, OMX_BFLOAT16, s)) return 1;
    } else {
        // speculative.rs:104-109 + :145-148: every position draws categorical(logits / T) with the NEXT key of the sequence -- one
        // [1, V] draw per row, exactly what a decode step does with its row
        for (int i = 0; i < n; ++i) {
            if (launch_rng_next(m->rng, s)) return 1;
            if (omx_random_categorical(m->verify_tokens + i, m->verify_logits + (size_t)i * V, 1, V, 1, 1.0f / m->temperature, m->rng + 2,
                                       OMX_BFLOAT16, (omx_stream)s))
                return 1;
        }
    }
    OMX_HIP_CHECK(hipMemcpyAsync(greedy_out, m->verify_tokens, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    OMX_HIP_CHECK(hipStreamSynchronize(s));
    st.pos += n;
    st.cur_token = greedy_out[n - 1];
    OMX_HIP_CHECK(hipMemcpyAsync(m->st, &st, sizeof(st), hipMemcpyHostToDevice, s));
    OMX_HIP_CHECK(hipStreamSynchronize(s));
    m->verify_rows = n;
    return 0;
}

/* bf16 logits [V] of row `row` of the last omx_qwen3_verify (the caller derives logprobs = logits - logsumexp, speculative.rs:150-152) */
int omx_qwen3_verify_logits(omx_qwen3 m, int row, void* host_bf16, int n) {
    OMX_REQUIRE(m && host_bf16, "omx_qwen3_verify_logits: null argument");
    OMX_REQUIRE(row >= 0 && row < m->verify_rows, "omx_qwen3_verify_logits: row %d of %d", row, m->verify_rows);
    OMX_REQUIRE(n == m->V, "omx_qwen3_verify_logits: expected %d entries, got %d", m->V, n);
    OMX_HIP_CHECK(hipMemcpyAsync(host_bf16, m->verify_logits + (size_t)row * m->V, (size_t)n * 2, hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    return 0;
}

/* KeyValueCache::trim -- the operation speculative.rs:165-169 notes the reference's cache trait lacks: forget the last n cached
 * tokens (their slab rows are simply overwritten by later appends) and make `next_token` the next step's input.  n = 0 only
 * replaces the pending input token. */
int omx_qwen3_trim(omx_qwen3 m, int n, uint32_t next_token) {
    OMX_REQUIRE(m, "omx_qwen3_trim: null argument");
    OMX_REQUIRE(next_token < (uint32_t)m->cfg.vocab_size, "omx_qwen3_trim: token id %u out of range (vocab %d)", next_token, m->cfg.vocab_size);
    StepState st;
    OMX_HIP_CHECK(hipMemcpyAsync(&st, m->st, sizeof(st), hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    OMX_REQUIRE(n >= 0 && n <= st.pos, "omx_qwen3_trim: cannot drop %d of %d cached tokens", n, st.pos);
    st.pos -= n;
    st.cur_token = next_token;
    OMX_HIP_CHECK(hipMemcpyAsync(m->st, &st, sizeof(st), hipMemcpyHostToDevice, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    return 0;
}

int omx_qwen3_last_prefill_ms(omx_qwen3 m, float* ms) {
    OMX_REQUIRE(m && ms, "omx_qwen3_last_prefill_ms: null argument");
    *ms = m->last_prefill_ms;
    return 0;
}

int omx_qwen3_decode(omx_qwen3 m, int n, uint32_t* tokens_out) {
    OMX_REQUIRE(m && tokens_out, "omx_qwen3_decode: null argument");
    OMX_REQUIRE(n >= 0 && n <= m->ring_cap, "omx_qwen3_decode: n=%d out of range (1..%d per call)", n, m->ring_cap);
    if (n == 0) return 0;
    StepState st;
    OMX_HIP_CHECK(hipMemcpyAsync(&st, m->st, sizeof(st), hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (prepare_step(m, st.pos)) return 1;
    OMX_REQUIRE(st.pos + n <= m->cap, "omx_qwen3_decode: %d cached + %d new tokens exceed max_context %d", st.pos, n, m->cap);
    for (;;) {
        // all n steps stay inside one split plan (one captured form)?  then the AQL program, if there is one, replays them
        bool aql = false;
        if (m->aql_full && step_aql_mode(m)) {
            const int tk = st.pos + n, gran = tk <= 8192 ? 1024 : 4096;
            aql = std::min(m->cap, (tk + gran - 1) / gran * gran) == m->graph_tk_max;
        }
        if (aql) {
            double ms = 0.0;
            if (aql_replay(m->aql_full, n, &ms)) {   // the queue is unusable: the error stands, later calls use the graph
                m->aql_disabled = true;
                return 1;
            }
            m->last_decode_ms = (float)ms;
        } else {
            OMX_HIP_CHECK(hipEventRecord(m->ev0, m->stream));
            for (int i = 0; i < n; ++i)
                if (run_step(m, true, st.pos + i)) return 1;
            OMX_HIP_CHECK(hipEventRecord(m->ev1, m->stream));
            OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
            OMX_HIP_CHECK(hipEventElapsedTime(&m->last_decode_ms, m->ev0, m->ev1));
        }
        unsigned gave_up = 0;
        if (step_gave_up(m, &gave_up)) return 1;
        if (!gave_up) break;
        if (step_fallback(m, st)) return step_health(m);   // no form with less co-residency left: report
        if (prepare_step(m, st.pos)) return 1;
    }
    std::vector<uint32_t> ring(m->ring_cap);
    OMX_HIP_CHECK(hipMemcpy(ring.data(), m->out_ring, (size_t)m->ring_cap * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) tokens_out[i] = ring[(st.out_count + i) % m->ring_cap];
    return 0;
}

int omx_qwen3_last_logits(omx_qwen3 m, void* host_bf16, int n) {
    OMX_REQUIRE(m && host_bf16, "omx_qwen3_last_logits: null argument");
    OMX_REQUIRE(n == m->V, "omx_qwen3_last_logits: expected %d entries, got %d", m->V, n);
    OMX_HIP_CHECK(hipMemcpyAsync(host_bf16, m->logits, (size_t)n * 2, hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    return 0;
}

/* test/debug hook: copy an internal bf16 buffer to the host ("h","h2","qkv","attn_out","act","k<l>","v<l>") */
int omx_qwen3_debug_read(omx_qwen3 m, const char* name, void* host, size_t n_elems) {
    OMX_REQUIRE(m && name && host, "omx_qwen3_debug_read: null argument");
    const void* src = nullptr;
    const std::string s(name);
    if (s == "h") src = m->h;
    else if (s == "h2") src = m->h2;
    else if (s == "qkv") src = m->qkv;
    else if (s == "attn_out") src = m->attn_out;
    else if (s == "act") src = m->act;
    else if (s.rfind("g_", 0) == 0 && m->se_gran) {   // granule buffers of the persistent step (8 bytes each: n_elems = 4 x granules)
        const size_t hd = m->cfg.hidden_size, D = m->cfg.head_dim;
        const uint64_t* g = m->se_gran;
        if (s == "g_x") src = g;
        else if (s == "g_x1") src = g + hd / 2;
        else if (s == "g_qkv") src = g + hd;
        else if (s == "g_attn") src = g + hd + (size_t)(m->H + 2 * m->Hkv) * D / 2;
        else if (s == "g_act") src = g + hd + (size_t)(m->H + 2 * m->Hkv) * D / 2 + (size_t)m->H * D / 2;
    }
    else if (s.size() > 1 && (s[0] == 'k' || s[0] == 'v')) {
        const int l = atoi(s.c_str() + 1);
        OMX_REQUIRE(l >= 0 && l < (int)m->kcache.size(), "omx_qwen3_debug_read: bad layer in %s", name);
        src = s[0] == 'k' ? m->kcache[l] : m->vcache[l];
    }
    OMX_REQUIRE(src != nullptr, "omx_qwen3_debug_read: unknown buffer %s", name);
    OMX_HIP_CHECK(hipMemcpyAsync(host, src, n_elems * 2, hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    return 0;
}

int omx_qwen3_last_decode_ms(omx_qwen3 m, float* ms) {
    OMX_REQUIRE(m && ms, "omx_qwen3_last_decode_ms: null argument");
    *ms = m->last_decode_ms;
    return 0;
}

int omx_qwen3_stream(omx_qwen3 m, omx_stream* s) {
    OMX_REQUIRE(m && s, "omx_qwen3_stream: null argument");
    *s = (omx_stream)m->stream;
    return 0;
}

/* debug hook (tools/attn_step_trace.py): run ONE decode step eagerly with the attention launches stamping the 100 MHz wall clock:
 * host receives [layers][attn splits][kv heads][8] = {block start, loads landed + q/k normed and roped, own chunk done, granules
 * stored, gathered and merged (consumer blocks), -, -, -}; *blocks = splits * kv heads */
int omx_qwen3_debug_trace_step(omx_qwen3 m, unsigned long long* host, size_t n_words, int* blocks) {
    OMX_REQUIRE(m && host && blocks, "omx_qwen3_debug_trace_step: null argument");
    StepState st;
    OMX_HIP_CHECK(hipMemcpyAsync(&st, m->st, sizeof(st), hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (prepare_step(m, st.pos)) return 1;
    const size_t per_layer = (size_t)m->attn_nsplit * m->Hkv * 8, need = per_layer * m->cfg.num_hidden_layers;
    OMX_REQUIRE(n_words >= need, "omx_qwen3_debug_trace_step: buffer of %zu words, need %zu", n_words, need);
    unsigned long long* dev = nullptr;
    OMX_HIP_CHECK(hipMalloc(&dev, need * 8));
    OMX_HIP_CHECK(hipMemsetAsync(dev, 0, need * 8, m->stream));
    m->attn_trace = dev;
    const int rc = enqueue_step(m, true);
    m->attn_trace = nullptr;
    if (!rc) {
        OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
        OMX_HIP_CHECK(hipMemcpy(host, dev, need * 8, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dev);
    *blocks = m->attn_nsplit * m->Hkv;
    return rc ? 1 : step_health(m);
}

/* measurement hook (bench.py roofline.achieved): runs `steps` REAL decode steps (they advance the context like any other) eagerly, each
 * launch of the five per-layer kernels and the lm_head carrying its own HIP event pair (hipExtLaunchKernelGGL start / stop events: the
 * dispatch's begin / end timestamps on the step's stream, launch_timing.hpp).  us[6] = average of {QKV GEMV, attention, O GEMV,
 * gate/up + SwiGLU GEMV, down GEMV, lm_head} over layers and steps.  Dense bf16 single-rank models only. */
int omx_qwen3_time_step_kernels(omx_qwen3 m, int steps, float* us) {
    OMX_REQUIRE(m && us && steps > 0, "omx_qwen3_time_step_kernels: bad arguments");
    OMX_REQUIRE(!m->cfg.quant_bits && m->cfg.num_experts == 0 && m->allreduce == nullptr, "omx_qwen3_time_step_kernels: dense bf16 single-rank models only");
    const int L = m->cfg.num_hidden_layers;
    std::vector<hipEvent_t> ev((size_t)(L * kLayerClasses + 2) * 2);
    for (auto& e : ev) OMX_HIP_CHECK(hipEventCreate(&e));
    double sum[KC_COUNT] = {};
    int rc = 0;
    bool fused_o = false, engine = false, hybrid = false;
    arm_launch_events(nullptr, nullptr);
    for (int it = 0; it < steps && !rc; ++it) {
        StepState st;
        OMX_HIP_CHECK(hipMemcpyAsync(&st, m->st, sizeof(st), hipMemcpyDeviceToHost, m->stream));
        OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
        if (st.pos + 1 > m->cap) { set_error("omx_qwen3_time_step_kernels: context full"); rc = 1; break; }
        if (prepare_step(m, st.pos)) { rc = 1; break; }
        fused_o = attention_takes_oproj(m);
        engine = step_engine_mode(m) == 1;
        hybrid = step_engine_mode(m) == 2;
        m->kernel_events = &ev;
        rc = enqueue_step(m, true);
        m->kernel_events = nullptr;
        if (rc) break;
        OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
        for (int l = (engine ? L : 0); l <= L; ++l)
            for (int k = 0; k < (l == L ? (engine ? 2 : 1) : kLayerClasses); ++k) {
                float ms = 0.f;
                const size_t i = ((size_t)l * kLayerClasses + k) * 2;
                if (l < L && k == KC_O && fused_o) continue;     // that pair was never armed
                if (hybrid && l < L && (k == KC_GATE_UP || (k == KC_DOWN && l != L - 1))) continue;   // one segment launch covers them
                OMX_HIP_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
                sum[l == L ? (k == 0 ? KC_HEAD : KC_ENGINE) : k] += ms * 1e3;
            }
        rc = step_health(m);
    }
    arm_launch_events(nullptr, nullptr);   // (a pair armed for a launch that never happened must not outlive its events)
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (rc) return 1;
    for (int k = 0; k < KC_COUNT; ++k) us[k] = (float)(sum[k] / ((k == KC_HEAD || k == KC_ENGINE ? 1.0 : (double)L) * steps));
    if (fused_o) us[KC_O] = 0.f;   // no separate launch: its work is inside the attention figure
    return 0;
}

/* debug hook (tools/step_engine_trace.py): ONE eager decode step on the persistent engine with per-CU wall-clock stamps:
 * host receives [CUs][64] words (12 per layer for the first four layers: layer start, x ready, qkv done, partials out, attention done,
 * attention vector ready, o done, x1 ready, gate/up done, act ready, down done); *cus_out = CUs */
int omx_qwen3_debug_trace_engine(omx_qwen3 m, unsigned long long* host, size_t n_words, int* cus_out) {
    OMX_REQUIRE(m && host && cus_out, "omx_qwen3_debug_trace_engine: null argument");
    StepState st;
    OMX_HIP_CHECK(hipMemcpyAsync(&st, m->st, sizeof(st), hipMemcpyDeviceToHost, m->stream));
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (prepare_step(m, st.pos)) return 1;
    OMX_REQUIRE(step_engine_takes(m), "omx_qwen3_debug_trace_engine: the persistent step is off or the model does not qualify");
    const size_t need = (size_t)m->cus * kStepEngineTraceWords;
    OMX_REQUIRE(n_words >= need, "omx_qwen3_debug_trace_engine: buffer of %zu words, need %zu", n_words, need);
    unsigned long long* dev = nullptr;
    OMX_HIP_CHECK(hipMalloc(&dev, need * 8));
    OMX_HIP_CHECK(hipMemsetAsync(dev, 0, need * 8, m->stream));
    m->se_trace = dev;
    const int rc = enqueue_step(m, true);
    m->se_trace = nullptr;
    if (!rc) {
        OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
        OMX_HIP_CHECK(hipMemcpy(host, dev, need * 8, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dev);
    *cus_out = m->cus;
    return rc ? 1 : step_health(m);
}

int omx_qwen3_decode_path(omx_qwen3 m, int* path) {
    OMX_REQUIRE(m && path, "omx_qwen3_decode_path: null argument");
    *path = m->eager ? 2 : (m->aql_full && step_aql_mode(m)) ? 3 : m->g_full ? 1 : 0;   // 3: AQL replay on the engine's own queue
    return 0;
}

int omx_qwen3_step_bytes(omx_qwen3 m, int ctx, double* bytes) {
    OMX_REQUIRE(m && bytes, "omx_qwen3_step_bytes: null argument");
    const omx_qwen3_config& c = m->cfg;
    const double D = c.head_dim, hd = c.hidden_size;
    // 2 B x [L (h H D + 2 h Hkv D + H D h + 3 h I) + V h] + ctx (2 L Hkv D 2 B) + KV write
    // sparse MoE: the router plus the top-k experts' three matrices are what one token streams
    const double moe_i = c.tp_size > 1 && c.num_experts > 0 ? m->moe_I : c.moe_intermediate_size;   // (expert tensor parallel: this rank's columns)
    const double ffn = c.num_experts > 0 ? hd * c.num_experts + 3.0 * hd * moe_i * c.num_experts_per_tok : 3.0 * hd * m->I;
    const double per_layer = hd * m->H * D + 2.0 * hd * m->Hkv * D + m->H * D * hd + ffn;
    // bytes per weight element: bf16 = 2; quantized = bits/8 packed + (scale + bias) bf16 per group
    const double bpe = c.quant_bits ? c.quant_bits / 8.0 + 4.0 / c.quant_group : 2.0;
    const double w = bpe * (c.num_hidden_layers * per_layer + (double)m->V * hd);
    const double kv = (double)ctx * (2.0 * c.num_hidden_layers * m->Hkv * D * 2.0) + 2.0 * c.num_hidden_layers * m->Hkv * D * 2.0;
    *bytes = w + kv;
    return 0;
}

}  // extern "C"
