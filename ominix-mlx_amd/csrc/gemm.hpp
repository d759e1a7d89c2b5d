// MFMA (matrix-core) kernels for the compute-bound side of the path: prefill / DiT GEMM and
// flash attention with Tq > 1.
#pragma once
#include "common.hpp"

namespace omx {

// out[M,N] = x[M,K] . W[N,K]^T (+ bias[N]); bf16 in/out, fp32 accumulate (nn::Linear, linear.rs:87-92)
int launch_gemm_bf16(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, int M, int N, int K,
                     hipStream_t s);

// same with an optional fused residual: out = bf16(resid + bf16(x.W^T (+bias)))
int launch_gemm_bf16_ex(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, const bf16_t* resid, int M,
                        int N, int K, hipStream_t s);

// SDPA with Tq > 1 (prefill / DiT joint attention): flash-attention forward on MFMA.
int launch_attn_prefill(bf16_t* out, const bf16_t* q, const bf16_t* k, const bf16_t* v, int B, int H, int Hkv, int Tq,
                        int Tk, int D, int64_t kv_batch_stride, int64_t kv_head_stride, float scale, int mask_mode,
                        const void* mask, hipStream_t s, bool out_token_major = false);

}  // namespace omx
