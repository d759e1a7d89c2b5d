// Router selection shared by the router kernel (moe.hip) and the expert gate/up GEMV that routes in its own prologue (gemv.hip).
#pragma once
#include "common.hpp"
#include "act16.hpp"

namespace omx {

constexpr int kMaxExperts = 256;
constexpr int kMaxTopK = 8;

// one block per token: logits[e] = bf16(x . Wg[e]) ; mode 0: top-k of logits, softmax over the selected
// (precise) ; mode 1: softmax over all (precise, rounded to bf16), top-k, optional renormalisation
// softmax / top-k / renormalisation of one token's router logits by ONE wave (experts spread over the lanes)
// F16: a float16 model -- the scores are rounded to (and stored as) float16 where the bfloat16 model rounds to bfloat16
template <bool F16 = false>
__device__ __forceinline__ void route_from_logits(const float* s_logit, int t, int lane, int E, int k, int mode, int renorm,
                                                  uint32_t* __restrict__ inds, bf16_t* __restrict__ scores) {
    typedef Act16<F16> A16;
    constexpr int PER = kMaxExperts / 64;
    float v[PER];
    bool taken[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = lane + 64 * u;
        v[u] = e < E ? s_logit[e] : -INFINITY;
        taken[u] = e >= E;
    }
    if (mode == 1) {   // softmax over all experts first (qwen3_moe.rs:479)
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < PER; ++u) mx = fmaxf(mx, v[u]);
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < PER; ++u) sum += (lane + 64 * u < E) ? expf(v[u] - mx) : 0.f;
        sum = wave_sum(sum);
#pragma unroll
        for (int u = 0; u < PER; ++u) v[u] = (lane + 64 * u < E) ? A16::rnd(expf(v[u] - mx) / sum) : -INFINITY;
    }
    uint32_t sel[kMaxTopK];
    float selv[kMaxTopK];
    for (int j = 0; j < k; ++j) {   // descending, ties to the lower index: key = (orderable value, ~index)
        unsigned long long best = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (taken[u]) continue;
            uint32_t ub = __float_as_uint(v[u]);
            ub = (ub & 0x80000000u) ? ~ub : (ub | 0x80000000u);
            const unsigned long long key = ((unsigned long long)ub << 32) | (uint32_t)~(uint32_t)(lane + 64 * u);
            best = key > best ? key : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        const uint32_t e = ~(uint32_t)(best & 0xFFFFFFFFull);
        const uint32_t ub = (uint32_t)(best >> 32);
        sel[j] = e;
        selv[j] = __uint_as_float((ub & 0x80000000u) ? (ub & 0x7FFFFFFFu) : ~ub);
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if ((uint32_t)(lane + 64 * u) == e) taken[u] = true;
    }
    if (mode == 0) {   // softmax over the selected logits (model.rs:301-302)
        float mx = selv[0], sum = 0.f;
        for (int j = 1; j < k; ++j) mx = fmaxf(mx, selv[j]);
        for (int j = 0; j < k; ++j) sum += expf(selv[j] - mx);
        for (int j = 0; j < k; ++j) selv[j] = A16::rnd(expf(selv[j] - mx) / sum);
    } else if (renorm && k > 1) {
        float sum = 0.f;
        for (int j = 0; j < k; ++j) sum += selv[j];
        sum = A16::rnd(sum);
        for (int j = 0; j < k; ++j) selv[j] = A16::rnd(selv[j] / sum);
    }
    if (lane == 0) {
        for (int j = 0; j < k; ++j) {
            inds[(size_t)t * k + j] = sel[j];
            scores[(size_t)t * k + j] = A16::bits(selv[j]);
        }
    }
}

}  // namespace omx
