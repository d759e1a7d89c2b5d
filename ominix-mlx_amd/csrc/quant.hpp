// Packed-weight (MLX affine 4/8-bit) GEMV family: the decode-time Linear of a quantized checkpoint
// (nn::QuantizedLinear::forward, mlx-rs/src/nn/quantized.rs:361-385; quant.hip for the format).
#pragma once
#include "common.hpp"
#include "gemv.hpp"   // PRO_* / EPI_* codes shared with the bf16 GEMV family

namespace omx {

struct QMat {               // one member of a row-stacked weight (q | k | v) -- or gate (0) / up (1) for SwiGLU
    const uint32_t* w;      // [n, K*bits/32]
    const bf16_t* scales;   // [n, K/group]
    const bf16_t* biases;   // [n, K/group] or null
    int n;
    // optional repack built by the engine at load time: word g = scale[g] | bias[g] << 16, so that a lane fetches both with ONE
    // 4-byte load (two 2-byte loads per 16 bytes of weights cost the 4-bit GEMV 15 % of its streaming rate)
    const uint32_t* sb = nullptr;
    // optional second repack (round 6, qgemv_mfma.hip): the matrix in 9 KB tiles of 16 rows x 1 024 columns (words in the order the
    // matrix-core kernel's lanes consume them, the tile's scale | bias words behind them) -- 4-bit, group 64, K = 4096 / 12288
    const uint32_t* tiles = nullptr;
};
int launch_quant_interleave(uint32_t* sb, const bf16_t* scales, const bf16_t* biases, size_t n_groups, hipStream_t s);
// The raw-pointer C entry points (omx_moe_block_forward_q ...) receive the checkpoint's scales pointer; an engine that built the
// repack registers it under that pointer so those entry points find it (and removes it before freeing the repack).
void quant_register_sb(const bf16_t* scales, const uint32_t* sb);
void quant_unregister_sb(const bf16_t* scales);
const uint32_t* quant_find_sb(const bf16_t* scales);

struct QGemvArgs {
    QMat m[3];
    int N, K, group;
    const bf16_t* x;            // [n_x, K]
    const bf16_t* norm_w;       // PRO_RMSNORM
    float eps;
    const bf16_t* resid;        // EPI_RESIDUAL
    bf16_t* out;                // [n_batch, N]
    unsigned long long* argmax_slot;   // EPI_ARGMAX: one partial per block
    int rows_per_wave;
    int n_batch, x_div;         // batch entry j reads activation row j / x_div
    const uint32_t* w_sel;      // optional [n_batch] expert ids (gather_qmm)
    size_t w_estride, s_estride;    // words / groups between consecutive experts
    int swiglu_single_round;    // EPI_SWIGLU: fused_swiglu(up, gate) (one rounding, metal_kernels.rs:11-18) instead of nn::silu(g)*u
    int rolled_stage;           // A/B: stage the activation with the rolled loop (OMX_QGEMV_ROLLED_STAGE=1)
    int scales_f16;             // scales / biases (and QMat::sb's halves) hold float16 bit patterns: a float16 MLX checkpoint.  The
                                // activations stay bf16; every group's scale / bias enters the arithmetic as its exact float32 value
    // tensor parallel (round 4): EPI_F32 leaves the unrounded f32 row sums of this rank's K slice in out_f32 [N] (the all-reduce and the
    // fold into the residual follow as their own launches); EPI_ARGMAX numbers its rows from row_offset (this rank's vocabulary shard)
    float* out_f32;             // (batched: [n_batch, N])
    int row_offset;
    // expert parallel (round 5): only batch entries whose w_sel value lies in [w_sel_lo, w_sel_lo + w_sel_n) are computed, on expert
    // w_sel - w_sel_lo of this rank's stack; the others leave their output rows untouched (w_sel_n == 0: every entry, as before)
    int w_sel_lo, w_sel_n;
};
// packed [rows, cols*bits/32] -> bf16 [rows, cols]; scales_f16: scales / biases are float16 (engine-internal form of omx_dequantize)
int launch_dequantize_bf16(bf16_t* out, const uint32_t* packed, const void* scales, const void* biases, int64_t rows, int cols, int group_size,
                           int bits, bool scales_f16, hipStream_t s, bool out_f16 = false);   // out_f16: the result in float16 (a float16 model's prompt pass)

int launch_qgemv(const QGemvArgs& a, int bits, int pro, int epi, hipStream_t s);
// qgemv_mfma.hip (round 6): the dense 4-bit group-64 single-row forms on the matrix cores.  0 launched, -1 not its shape (take the VALU kernel), 1 error
int launch_qgemv4m(const QGemvArgs& a, int pro, int epi, hipStream_t s);
bool qgemv4m_shape_ok(int K, int group, int bits);
size_t qgemv4m_tile_words(int n, int K);            // u32 words of the tile form of an [n, K] matrix
int launch_qgemv4m_repack(uint32_t* tiles, const uint32_t* wq, const bf16_t* scales, const bf16_t* biases, int n, int K, hipStream_t s);
int qgemv_grid(int N);          // blocks launch_qgemv uses == argmax partials written

}  // namespace omx
