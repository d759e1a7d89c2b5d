// Decode-time (M == 1) weight-streaming GEMV family for gfx950 -- the HBM-bound kernel that
// decides decode tokens/s.  Reference op: nn::Linear::forward = x . W^T
// (mlx-rs/src/nn/linear.rs:87-92 -> mlx_matmul, mlx-c ops.h:598-602) at M == 1, i.e. > 97 % of
// the bytes of a decode step.
//
// Design (MI355X_MICROARCH.md / cdna_hip_programming.md "GEMV / M <= 16 decode weights"):
//   * weights go HBM -> VGPR directly with global_load_dwordx4 (nt), never through LDS: each
//     weight byte is used once by one wave;
//   * one wave64 owns whole rows: lane l reads 16 B at (j*64 + l)*16 of the row, so every
//     wave-instruction is one fully coalesced 1 KiB burst;
//   * the activation vector is staged ONCE per block in LDS as bf16 (optionally RMS-normalised
//     on the way in: the fused prologue removes the separate norm launch and its round trip);
//   * a register double buffer keeps the next batch of rows in flight while the current one is
//     reduced, and the first batch is issued BEFORE the prologue so its latency hides the
//     x load + norm reduction;
//   * fused epilogues: bf16 store, residual add, SwiGLU (gate/up row pairs), f32 partial
//     (tensor-parallel row split), logits + greedy argmax.
#pragma once
#include "common.hpp"

namespace omx {

enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_ROUTE = 2 };
enum { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2, EPI_ARGMAX = 3, EPI_F32 = 4 };

struct GemvArgs {
    // up to three row-stacked weight matrices sharing K (q/k/v) -- or gate (w0) / up (w1) for SwiGLU
    const bf16_t* w0;
    const bf16_t* w1;
    const bf16_t* w2;
    int n0, n1, n2;
    int N;                      // logical output rows
    int K;
    const bf16_t* x;            // [K] activation (bf16)
    const uint32_t* x_row;      // optional device scalar: x += x_row[0] * K (embedding gather)
    const float* x_partial;     // optional [K] f32 added to x before use (TP: all-reduced partial sums)
    const bf16_t* norm_w;       // [K] RMSNorm weight (PRO_RMSNORM)
    float eps;
    const bf16_t* resid;        // [N] (EPI_RESIDUAL)
    void* out;                  // [N] bf16 (f32 for EPI_F32)
    bf16_t* x_out;              // optional [K]: block 0 writes the (x + x_partial) it consumed (residual stream)
    unsigned long long* argmax_slot;   // EPI_ARGMAX: [gridDim.x] per-block max of (orderable(logit)<<32 | ~row)
    int row_offset;             // EPI_ARGMAX: global row index offset (vocab shard)
    int rows_per_wave;
    // batched / expert-selected launch (MoE decode, gather_mm semantics): grid.y = n_batch
    int n_batch;                // 0 or 1: plain launch
    int x_div;                  // activation row of batch entry j is j / x_div (top-k slots share a token)
    size_t x_bstride;           // elements between activation rows
    size_t out_bstride_bytes;   // bytes between the output vectors of consecutive batch entries
    const uint32_t* w_sel;      // optional [n_batch] device array: expert id per batch entry
    size_t w_estride;           // elements between consecutive experts' matrices
    int swiglu_single_round;    // EPI_SWIGLU: fused_swiglu (one rounding) instead of nn::silu(g)*u (three)
    // expert parallelism: only experts [w_sel_lo, w_sel_lo + w_sel_n) live on this rank (w_sel_n == 0: all of them); a batch
    // entry routed elsewhere does no work (its output is never read: the combine skips it too)
    int w_sel_lo, w_sel_n;
    const bf16_t* out_bias;     // EPI_STORE: optional [N] added before the rounding (nn::Linear with bias = addmm, linear.rs:87-92)
    // optional timeline (tools/gemv_trace.py; only in -DOMX_GEMV_TRACE builds): thread 0 of block b stamps the 100 MHz wall clock into trace[b*4 + k] at
    // k = 0 first weight batch issued, 1 activation staged, 2 first batch reduced, 3 last store issued
    unsigned long long* trace;
    // EPI_F32 + peer: the output rows are reduced over the tensor-parallel ranks INSIDE this launch -- every row's partial goes to
    // all inboxes as a tagged granule, the wave then sums the ranks' words of its rows in rank order and stores the f32 total
    // (peer.hpp / peer_allreduce.hip: the same protocol and sequence number as the standalone all-reduce kernel, one launch less)
    const struct PeerDev* peer;
    // MoE decode without the weighted-sum launch: the expert down projections (batched, EPI_F32) store
    // z_j = bf16(bf16(y_j) * score_j) as f32 (out_scale: [n_batch] scores; out_scale_f is the kernel's own copy of its entry's), and the
    // next consumer folds x := bf16(x + bf16(z_0 + z_1 + ...)) -- x_partial_n vectors of K floats at x_partial, summed in order
    // (0 or 1: the one vector of the tensor-parallel step).  Same roundings and order as moe_combine_kernel (moe.hip).
    const bf16_t* out_scale;
    float out_scale_f;
    int x_partial_n;
    // PRO_ROUTE (MoE decode, one token, the experts' gate/up launch): x is the RAW row; every block normalises it (norm_w, eps) and
    // routes it itself -- router logits, top-k, scores with the arithmetic of moe_router_kernel launched for the same shape (moe.hip:
    // <= 8 experts over <= 4096 columns, a multiple of 512) -- then streams the expert of its batch entry.  No router launch; block
    // (0, 0) leaves the selection for the down projection / the combine in route_inds [top_k] / route_scores [top_k].
    const bf16_t* route_gate;   // [E, K]
    int route_E, route_k, route_mode, route_renorm;
    uint32_t* route_inds;
    bf16_t* route_scores;
};
bool gemv_route_supported(int K, int n_experts, int top_k);

int launch_gemv(const GemvArgs& a, int pro, int epi, hipStream_t s);
// number of blocks launch_gemv will use (== entries written to argmax_slot); resolves rows_per_wave
int gemv_grid(int N, int K, int epi, int rows_per_wave);
// true when launch_gemv has a kernel for contraction width K (K not a multiple of 512 only without a prologue)
bool gemv_k_supported(int K, bool needs_full_vectors);

}  // namespace omx
