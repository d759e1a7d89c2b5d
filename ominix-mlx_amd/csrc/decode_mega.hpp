// One persistent kernel per decode token (see decode_mega.hip).
#pragma once
#include "common.hpp"

namespace omx {

// step state lives in device memory so a step never needs host patching (engine.hip)
struct StepState {
    int pos;               // tokens in the cache == RoPE offset of the token being processed
    uint32_t cur_token;    // token fed to the embedding this step
    int out_count;         // tokens sampled so far
    int prompt_idx;        // next prompt token to feed during a token-serial prefill
};

struct MegaLayer {
    const bf16_t *q, *k, *v, *o, *gate, *up, *down, *q_norm, *k_norm, *in_ln, *post_ln;
    bf16_t *kc, *vc;       // KV slabs [Hkv, cap, D]
};

struct MegaArgs {
    const MegaLayer* layers;   // device array [n_layers]
    int n_layers, hidden, H, Hkv, I, V, cap;
    float eps, scale;
    const bf16_t *embed, *final_norm, *lm_head;
    const float *rope_cos, *rope_sin;
    StepState* st;
    bf16_t *h0, *h1, *qkv, *attn_out, *act, *logits;
    float *ws_o, *ws_ml;       // split-KV partials [H, nsplit, D], [H, nsplit, 2]
    int nsplit;                // the split count the chunk rule is derived from (same rule as attn_decode.hip)
    int attn_blocks;           // upper bound on blocks that take attention work
    unsigned* sync_words;      // gridsync.hpp layout, grid_sync_words(nblocks) words
    unsigned epoch0;           // first barrier epoch of this launch
    unsigned* kv_count;        // [Hkv * 16] arrival counters for the per-KV-head combine
    unsigned long long* argmax_partials;   // [nblocks]
    uint32_t* out_ring;
    int ring_cap;
    const uint32_t* prompt;    // token-serial prefill: next prompt tokens
    int with_head;             // 0: prompt token whose logits nobody reads
    // optional phase timeline (tools/mega_trace.py): thread 0 of every block stamps the 100 MHz wall clock at
    // the start (barrier passed) and end (before arrive) of each phase, plus attention sub-steps:
    // [layer][kTraceEvents][nblocks]
    unsigned long long* trace;
};
constexpr int kTraceEvents = 16;

// number of device-wide barriers one launch executes (the host advances epoch0 by this)
inline unsigned mega_barriers(int n_layers, int with_head) { return 5u * (unsigned)n_layers + (with_head ? 1u : 0u); }

// true when an instantiation exists for this shape (hidden, H*D, I in units of 512; D == 128; G <= 4)
bool mega_supported(int hidden, int attn_width, int inter, int head_dim, int group);
// co-resident block capacity of the instantiation (occupancy x CUs); 0 on error
int mega_capacity(int hidden, int attn_width, int inter, int* blocks);
int launch_decode_mega(const MegaArgs& a, int nblocks, hipStream_t s);

}  // namespace omx
