// Attention launchers shared by the per-op ABI (sdpa.hip) and the decode engine.
#pragma once
#include "common.hpp"

namespace omx {

struct AttnDecodeArgs {
    const bf16_t* q;            // [B,H,1,D] bf16
    const bf16_t* k;            // [B,Hkv,*,D] with strides below (row stride = D)
    const bf16_t* v;
    int64_t kv_batch_stride, kv_head_stride;
    int B, H, Hkv, Tk;
    float scale;
    int mask_mode;              // OMX_MASK_NONE | OMX_MASK_BOOL ([Tk] u8) | OMX_MASK_ADDITIVE ([Tk] bf16)
    const void* mask;
    int nsplit;
    float* ws_o;                // [B*H, nsplit, D]
    float* ws_ml;               // [B*H, nsplit, 2]
    bf16_t* out;                // [B,H,1,D] == [B, H*D]
    // optional (0 = the defaults): element strides of q between batch entries and heads ([B,H,1,D]: H*D and D)
    int64_t q_bs, q_hs;
    // causal tail: batch entry b attends to the first Tk - (B - 1 - b) keys -- B consecutive new query rows over ONE shared cache
    // (kv_batch_stride 0), the last of them seeing all Tk keys (a short batched prefill / a speculative verify pass)
    int causal_tail;
};

// ---- the decode engine's attention launch (attn_step.hip): q/k norm + RoPE + cache append + split-KV SDPA + split merge ----
struct AttnStepArgs {
    const bf16_t* qkv;          // raw projections of the current token [H*D | Hkv*D | Hkv*D] (QKV GEMV output)
    bf16_t* k;                  // KV slabs [Hkv, cap, D]
    bf16_t* v;
    int64_t kv_head_stride;
    int H, Hkv, cap;
    float scale, eps;
    const bf16_t* q_norm_w;     // null: no q/k norm (Mixtral, Qwen2)
    const bf16_t* k_norm_w;
    const float* rope_cur;      // [D]: cos[D/2] | sin[D/2] of the CURRENT position (refreshed by the step's first kernel)
    const int* pos_ptr;         // device scalar: tokens already cached
    const unsigned* seq_ptr;    // step sequence number (never reset)
    unsigned tag_mul, tag_add;  // granule tag = *seq_ptr * tag_mul + tag_add: unique per (step, layer), 1 <= tag_add <= tag_mul
    int chunk;                  // tokens per split: a multiple of attn_step_block_tokens(D), fixed per captured graph
    int nsplit;                 // gridDim.y, G <= nsplit <= 48, nsplit * chunk >= every position the graph will see + 1
    uint64_t* ws;               // granules [H][nsplit][D + 2]  ({f32, tag} each: o[D], m, l)
    bf16_t* out;                // [H*D]
    unsigned* abort_flag;       // raised when a gather gave up (results void)
    unsigned long long* trace;  // optional: 8 wall-clock stamps per block [nsplit][Hkv][8] (tools/attn_step_trace.py)
    // ---- optional O projection in the same launch (o_w != null; attn_step_oproj_ok says whether the shape qualifies) ----
    const bf16_t* o_w;          // [o_rows, H*D] row-major (nn::Linear weight, model.rs:214)
    const bf16_t* o_resid;      // [o_rows] residual stream entering the layer
    bf16_t* o_out;              // [o_rows] = bf16(resid + bf16(o_w . attn))      (model.rs:325)
    float* o_out_f32;           // tensor parallel (o_w = this rank's columns): [o_rows] f32 partial o_w . attn instead, to be all-reduced
    int o_rows;
    int o_rpw, o_nhi;           // rows per wave of the non-consumer blocks: the first o_nhi waves hold o_rpw rows, the others o_rpw - 1
                                // (set by launch_attn_step)
    uint64_t* xg;               // granules [H*D/2]: {two packed bf16 of the attention vector, tag}
    // ---- ... on a 4-bit packed O matrix instead (o_wq != null, o_w == null; attn_step_oproj_q4_ok) ----
    const uint32_t* o_wq;       // [o_rows, H*D/8] MLX-packed nibbles
    const uint32_t* o_sb;       // [o_rows, H*D/group] scale | bias << 16 (QMat::sb)
    int o_group;
    int f16;                    // the model runs in float16 (a float16 MLX checkpoint): qkv, norm weights, K / V slabs and `out` hold float16
};
bool attn_step_oproj_ok(int H, int Hkv, int D, int nsplit, int o_rows);
bool attn_step_oproj_q4_ok(int H, int Hkv, int D, int nsplit, int o_rows, int group);
int attn_step_block_tokens(int D);
void attn_step_plan(int tk_max, int Hkv, int G, int D, int* chunk, int* nsplit);
size_t attn_step_ws_granules(int H, int D);
int launch_attn_step(const AttnStepArgs& a, int D, hipStream_t s);

size_t attn_decode_ws_bytes(int BH, int nsplit, int D);
int decode_nsplit(int Tk, int BHkv);   // sdpa.hip: splits of the per-op decode attention
int launch_attn_decode(const AttnDecodeArgs& a, int D, hipStream_t s);

}  // namespace omx
