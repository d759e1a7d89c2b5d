// Persistent decode step: every layer of one token in ONE launch, on a loader / consumer engine per CU.
//   reference: the per-token body of Generate (qwen3-mlx/src/model.rs:314-340 decoder layer, :804-843 decode loop) -- the chain
//   [RMSNorm + q/k/v Linear] [q/k norm + RoPE + KV append + SDPA] [o Linear + residual] [RMSNorm + gate/up + silu*up] [down + residual]
//   that engine.hip otherwise enqueues as four launches per layer.
//
// Why: at batch 1 every launch of the step is a weight stream of 5-35 us, and each launch boundary costs ~2.3 us in which no weight
// byte moves (drain + dispatch + first-byte latency + ramp).  Weights depend on nothing, so a wave that ONLY loads never has to stop at
// a dependency: here one workgroup of 4 waves lives on every CU for the whole step,
//   * wave 0, the loader, streams the CU's share of every matrix (and its K/V chunk) HBM -> LDS with global_load_lds (nt) into a ring
//     of 16 KiB slots and runs ahead of the consumers across every dependency edge, bounded only by the ring;
//   * waves 1-3, the consumers, reduce the landed rows against the activation vector in LDS -- with the exact fma order, wave
//     reduction and rounding points of gemv.hip / attn_step.hip, so the hidden state is bit-identical to the launch-per-op step;
//   * an op's output vector travels to every CU as 8-byte {two bf16, tag} granules written with write-through stores and swept with
//     coherent loads (cdna_hip_programming.md Guideline 16, form R2: the data is the flag; tag = step sequence x layer, nothing is
//     reset between launches); no grid barrier anywhere.
// Follows the recipe of MI355X_MICROARCH.md "Persistent kernels" (rows engine-vs-launches, prefetch-credit, ldsdma-fill, nt-weights,
// gather-pass, polling-cost).
#pragma once
#include "common.hpp"
#include "step_state.hpp"

namespace omx {

struct StepEngineLayer {
    const bf16_t *q, *k, *v, *o, *gate, *up, *down, *in_ln, *post_ln, *q_norm, *k_norm;
    bf16_t *kc, *vc;   // KV slabs [Hkv, cap, D]
};

struct StepEngineArgs {
    const StepEngineLayer* layers;   // device array [L]
    int L, hidden, H, Hkv, D, I, cap;
    float eps, scale;
    const bf16_t* embed;             // [V, hidden]
    const StepState* st;
    unsigned* seq_ptr;               // step sequence number: read at entry (+1 = this step's), stored back by block 0 at exit
    const float *rope_cos, *rope_sin;   // [cap, D/2]
    int chunk, nsplit;               // split plan of the decode attention (attn_step_plan: the same one the launch-per-op step uses)
    uint64_t *g_x, *g_qkv, *g_part, *g_attn, *g_x1, *g_act;   // granule buffers, one per edge
    bf16_t* h_out;                   // [hidden] residual stream after the last layer (plain stores: read by the lm_head launch)
    unsigned* abort_flag;
    int nsweep;                      // consumer waves that sweep a hidden-sized edge (1 or 3)
    int inflight;                    // loader: fills in flight before it waits for the oldest (2 or 3)
    int thin_gather;                 // loader: one fill in flight while its CU sweeps granules (MI355X_MICROARCH.md gather-pass)
    unsigned long long* trace;       // optional [grid][kTraceWords] wall-clock stamps (tools/step_engine_trace.py)
    // segment mode (the hybrid step: attention + o stay their own launch, csrc/attn_step.hip): this launch runs [gate/up, down] of layer
    // seg_layer - 1 (skipped at 0) and [RMSNorm + q/k/v] of layer seg_layer (skipped at L); -1: the whole step
    int seg_layer;
    StepEngineLayer seg_m, seg_a;    // segment mode: layers seg_layer - 1 and seg_layer by value (set by the caller from its host table)
    int xcd_major;                   // shares of the rows XCD-major (see share_of)
    const bf16_t* x_in;              // seg_layer == 0: the embedding row (written by the step's first kernel)
    const bf16_t* x1_in;             // seg_layer > 0: residual stream after the attention launch of layer seg_layer - 1
    bf16_t* qkv_out;                 // raw projections for the attention launch of layer seg_layer; the new residual goes to h_out
    int xs_bytes, nslot;             // set by launch_step_engine
};

constexpr int kStepEngineTraceWords = 64;

// shape / mode check: dense bf16 single-rank model whose widths the consumers have a bit-identical reduction for and whose
// activation vectors fit the LDS next to the ring; `cus` = CUs of the device (one workgroup per CU, all resident)
bool step_engine_ok(int hidden, int H, int Hkv, int D, int I, int nsplit, int cus);
size_t step_engine_granules(int hidden, int H, int Hkv, int D, int I);   // total granules of the six edge buffers
int launch_step_engine(const StepEngineArgs& a, int cus, hipStream_t s);

}  // namespace omx
