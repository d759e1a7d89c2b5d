// Launchers of the batched-prefill glue kernels (prefill.hip) used by the decode engine.
#pragma once
#include "common.hpp"

namespace omx {

int launch_qk_norm_rope_scatter(const bf16_t* q_lin, const bf16_t* k_lin, const bf16_t* v_lin, const bf16_t* q_norm_w,
                                const bf16_t* k_norm_w, const float* rope_cos, const float* rope_sin, bf16_t* q_out,
                                bf16_t* kcache, bf16_t* vcache, int T, int H, int Hkv, int D, int cap, int offset,
                                float eps, hipStream_t s, bool f16 = false);   // f16: a float16 model's rows / slabs (head_dim 128)
int launch_silu_mul(bf16_t* out, const bf16_t* gate, const bf16_t* up, int64_t n, hipStream_t s);

}  // namespace omx
