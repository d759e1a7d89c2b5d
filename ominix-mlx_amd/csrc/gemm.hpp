// MFMA (matrix-core) kernels for the compute-bound side of the path: prefill / DiT GEMM and
// flash attention with Tq > 1.
#pragma once
#include "common.hpp"

namespace omx {

// Grouped (expert-segmented) GEMM descriptor; every pointer is DEVICE memory written by moe_plan_kernel.
struct GroupedDesc {
    const int* tile_expert = nullptr;   // [max_tiles] expert of each 128-row tile
    const int* tile_m0 = nullptr;       // [max_tiles] first row of the tile inside its expert segment
    const int* seg_start = nullptr;     // [E + 1] exclusive prefix sums of rows per expert
    const int* n_tiles = nullptr;       // [1] tiles actually in use
    const uint32_t* row_src = nullptr;  // optional [rows]: source activation row of each sorted position (gather)
    size_t w_estride = 0;               // elements between consecutive experts' [N, K] matrices
};

// out[M,N] = x[M,K] . W[N,K]^T (+ bias[N]); bf16 in/out, fp32 accumulate (nn::Linear, linear.rs:87-92)
int launch_gemm_bf16(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, int M, int N, int K,
                     hipStream_t s);

// same with an optional fused residual: out = bf16(resid + bf16(x.W^T (+bias)))
int launch_gemm_bf16_ex(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, const bf16_t* resid, int M,
                        int N, int K, hipStream_t s);
// out = bf16(max(0, x.W^T + bias))   (Paraformer FFN: Linear + ReLU, paraformer.rs:565-569)
int launch_gemm_bf16_bias_relu(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, int M, int N, int K,
                               hipStream_t s);
// gated residual epilogue of the DiT blocks: out = bf16(resid + (x.W^T) * gate[col])   (klein_model.rs:496-497, 922-925)
// tile preference of this thread's next plain GEMMs (gemm.hip g_tile_hint): 0 none, 128 = the 128 x 256 tile for a GEMM that shares the
// chip with another stream's grid
void gemm_tile_hint(int rows);
// float16 operands and results for this thread's next GEMMs (the bf16_t pointers then hold float16 bit patterns); a float16
// checkpoint's batched prompt pass switches it on around its launches
bool gemm_set_f16(bool on);   // returns the previous state
int launch_gemm_bf16_gated(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* resid, const bf16_t* gate, int M,
                           int N, int K, hipStream_t s);

// Segmented projection (256^2 kernel): ONE launch computes up to three plain Linears of the same input -- each with its own
// weight [cols, K], optional bias, output and row stride -- followed by an optional SwiGLU pair: column tiles of the last
// segment take 128 gate and the matching 128 up rows, and store act[m, c] = silu(bf16(x.Wg[c]^T)) * bf16(x.Wu[c]^T).
// Every output element is the same MFMA sequence as in the plain 256^2 kernel; what changes is that the launches' tile
// grids are concatenated (q/k/v: 128 + 32 + 32 tiles of 512 slots -> one 192-tile wave; gate/up: 2 x 384 -> 768 = 3 full
// rounds of 256) and the 2 * half wide gate/up intermediate never goes to HBM.
struct GemmSeg {
    const bf16_t* w;      // [cols, K]
    const bf16_t* bias;   // [cols] or null
    bf16_t* out;          // [M, cols] with row stride ld
    int cols, ld;
    int tile0;            // first column tile (filled by the launcher)
};
struct GemmSegs {
    GemmSeg plain[3];
    int n_plain;          // 0..3
    const bf16_t* w_gate; // [half, K]
    const bf16_t* w_up;   // [half, K]
    bf16_t* out_act;      // [M, half] with row stride ld_act
    int half, ld_act;     // half == 0: no SwiGLU segment
    int act_tile0;        // (filled by the launcher)
    int act_mode;         // 0: fused_swiglu, one rounding (metal_kernels.rs:188-236); 1: nn::silu(gate) * up with every
                          //    primitive rounded to bf16 (qwen3-mlx/src/model.rs:264-265)
    // implicit 3x3 convolution (launch_conv3x3_implicit): channels, output width, log2(channels / 64), shortcut
    int im_C, im_W, im_sh;
    const bf16_t* im_resid;
    // RMSNorm of the input rows inside the launch (the weight-streaming route for M <= 8 rows only, gemv_rows_takes_norm):
    // x is the raw residual stream, every block normalises its staged copy with the norm kernel's exact arithmetic
    const bf16_t* pre_norm_w;
    float pre_norm_eps;
};
// 3x3 convolution, stride 1, zero padding 1, as ONE GEMM over a zero-bordered NHWC activation [(H+2), (W+2), C] (no im2col
// matrix): out[(y*W + x), o] = bias[o] + sum_{tap, c} padded[(y + tap/3), (x + tap%3), c] * w[o, tap*C + c]  (+ resid).
// C = 64 * 2^j, >= 160 output tiles of 256^2.
bool conv3x3_implicit_supported(int H, int W, int C, int Cout);
int launch_conv3x3_implicit(bf16_t* out, const bf16_t* padded, const bf16_t* w, const bf16_t* bias, const bf16_t* resid, int H, int W,
                            int C, int Cout, hipStream_t s);
bool gemm_segmented_supported(int M, int K, const GemmSegs& segs);   // can one launch compute it
bool gemm_segmented_preferred(int M, int K, const GemmSegs& segs);   // ... and is that the faster schedule (callers with a fallback)
// the same launch over expert-sorted rows (MoE prefill): 256-row tiles from the device-built tile table (GroupedDesc, 256-row
// granularity), rows gathered through g.row_src, every weight pointer offset by expert * g.w_estride
int launch_gemm_bf16_segmented_grouped(const bf16_t* x, int max_rows, int K, const GemmSegs& segs, const GroupedDesc& g, int max_tiles,
                                       hipStream_t s);
int launch_gemm_bf16_segmented(const bf16_t* x, int M, int K, const GemmSegs& segs, hipStream_t s);

// projection with a SwiGLU segment (256^2 kernel): W = [n_plain plain rows | half gate rows | half up rows], all [., K].
//   out_plain[m, c] = bf16(x.W[c]^T) for c < n_plain (row stride ld_plain),
//   out_act[m, c]   = bf16(silu(g) * u), g = bf16(x.W[n_plain + c]^T), u = bf16(x.W[n_plain + half + c]^T)  (row stride ld_act)
// -- bit-identical to storing the whole projection and running fused_swiglu over it (klein_model.rs:489-493, 905-916).
bool gemm_swiglu_supported(int M, int n_plain, int half, int K);
bool gemm_swiglu_preferred(int M, int n_plain, int half, int K);
int launch_gemm_bf16_swiglu(bf16_t* out_plain, int ld_plain, bf16_t* out_act, int ld_act, const bf16_t* x, const bf16_t* w, int M,
                            int n_plain, int half, int K, hipStream_t s);

// rows sorted by expert: out[p, :] = x[row_src ? row_src[p] : p, :] . W[e(p)]^T, p in expert-sorted order
int launch_gemm_bf16_grouped(bf16_t* out, const bf16_t* x, const bf16_t* w, int max_rows, int N, int K,
                             const GroupedDesc& g, int max_tiles, hipStream_t s);

// float32 GEMM on the f32-input matrix cores (gemm_f32.hip): out[b] = alpha * A[b] . B[b] (+ bias) (relu) (+ resid), strided and
// batched; B is [N, K] row-major (b_nn = 0, nn::Linear) or [K, N] row-major (b_nn = 1)
struct GemmF32 {
    const float* a; const float* b; const float* bias; const float* resid; float* out;
    int M, N, K;
    int64_t lda, ldb, ldc, ldr;    // leading dimensions (ldr 0: = ldc)
    int64_t sa, sb, sc;            // batch strides
    int batch, relu, b_nn;
    float alpha;
    // round 6: the caller finishes the product itself (paraformer.hip: epilogue + FSMN + LayerNorm in ONE launch instead of reduce, add, norm).
    // Set both: the launch then applies NO epilogue (alpha, bias, relu, resid are the caller's) and reports where the raw product lies --
    // *defer_partial [splits][M][N] (the split-K scratch of the stream, valid until the stream's next GEMM) with *defer_splits >= 1
    const float** defer_partial = nullptr;
    int* defer_splits = nullptr;
};
int launch_gemm_f32(const GemmF32& p, hipStream_t s);

// the ring kernel keeps one fixed-size split-K scratch per stream; whoever destroys a stream hands it back first
void gemm_release_stream(hipStream_t s);

// element strides for SDPA operands that are not [B,H,T,D]-contiguous
struct AttnLayout {
    int64_t q_bs, q_hs, q_ts;   // queries: batch, head, token
    int64_t kv_ts;              // keys/values: token (head stride = kv_head_stride argument)
    int64_t o_bs, o_hs, o_ts;   // output
};

// SDPA with Tq > 1 (prefill / DiT joint attention): flash-attention forward on MFMA.
// float32 attention of the explicit form in one launch (attn_f32.hip): head width 128, no mask, Tk <= 512.  -1 = not this kernel's shape.
int launch_attn_f32(float* out, const float* q, const float* k, const float* v, int64_t ldq, int64_t ldkv, int64_t ldo, int Tq, int Tk, int heads,
                    float scale, hipStream_t s);
int launch_attn_prefill(bf16_t* out, const bf16_t* q, const bf16_t* k, const bf16_t* v, int B, int H, int Hkv, int Tq,
                        int Tk, int D, int64_t kv_batch_stride, int64_t kv_head_stride, float scale, int mask_mode,
                        const void* mask, hipStream_t s, bool out_token_major = false,
                        const AttnLayout* layout = nullptr, bool f16 = false);   // f16: float16 q / k / v / out (no mask or causal)

// the 4-wave persistent form (attn_flash4.hip): head_dim 128, bf16, no mask, Tk a multiple of 64
bool attn_flash4_supported(int B, int H, int Hkv, int Tq, int Tk, int D, int mask_mode, bool f16);
int launch_attn_flash4(bf16_t* out, const bf16_t* q, const bf16_t* k, const bf16_t* v, int B, int H, int Hkv, int Tq, int Tk,
                       int64_t kv_batch_stride, int64_t kv_head_stride, float scale, hipStream_t s, bool out_token_major,
                       const AttnLayout* layout);

// M <= 8 rows: weights streamed once against all rows (gemv_rows.hip); same epilogue semantics as the GEMM kernels
bool gemv_rows_supported(int M, int N, int K, const void* x, const void* w);
// ... and the segmented projection (GemmSegs above) in one such launch: q / k / v, or gate / up with the SwiGLU epilogue
bool gemv_rows_segmented_supported(int M, int K, const GemmSegs& segs);
int launch_gemv_rows_segmented(const bf16_t* x, int M, int K, const GemmSegs& segs, hipStream_t s);
// would launch_gemm_bf16_segmented(x, M, K, segs) take that route AND can it apply segs.pre_norm_w itself (K <= 4096: the whole
// row is staged at once)?  Callers then skip their RMSNorm launch and pass the raw rows.
bool gemv_rows_takes_norm(int M, int K, const GemmSegs& segs);
int launch_gemv_rows(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, const bf16_t* resid, const bf16_t* gate, int M, int N,
                     int K, int relu, hipStream_t s);

}  // namespace omx
