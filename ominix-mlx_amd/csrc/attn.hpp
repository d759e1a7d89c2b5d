// Attention launchers shared by the per-op ABI (sdpa.hip) and the decode engine.
#pragma once
#include "common.hpp"

namespace omx {

struct AttnDecodeArgs {
    // generic path: q [B,H,1,D] bf16.  fused path: qkv = raw projections [H*D | Hkv*D | Hkv*D]
    const bf16_t* q;
    const bf16_t* qkv;
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
    // fused extras (qwen3-mlx/src/model.rs:172-196)
    const int* pos_ptr;         // device scalar: tokens already cached == RoPE offset
    const bf16_t* q_norm_w;
    const bf16_t* k_norm_w;
    const float* rope_cos;      // [max_pos, D/2]
    const float* rope_sin;
    float eps;
    // optional [B*Hkv * 16] zeroed arrival counters: the LAST split block of a KV head merges the splits itself
    // (same arithmetic as attn_combine_kernel) and no combine launch follows
    unsigned* arrive;
};

size_t attn_decode_ws_bytes(int BH, int nsplit, int D);
int launch_attn_decode(const AttnDecodeArgs& a, int D, bool fused, hipStream_t s);

}  // namespace omx
