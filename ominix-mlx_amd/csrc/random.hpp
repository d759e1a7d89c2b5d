// Threefry-2x32 words, MLX layout (random.hip).  Device-inline so that samplers fuse the noise into their pass.
#pragma once
#include "common.hpp"

namespace omx {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// Threefry-2x32, 20 rounds (the generator behind mlx_random_bits)
__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
    const uint32_t ks[3] = {k0, k1, 0x1BD11BDAu ^ k0 ^ k1};
    uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
    constexpr int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
#pragma unroll
    for (int g = 0; g < 5; ++g) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x0 += x1;
            x1 = rotl32(x1, rot[g & 1][r]);
            x1 ^= x0;
        }
        x0 += ks[(g + 1) % 3];
        x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
    }
    o0 = x0;
    o1 = x1;
}

// word i of an n-word draw: block j = (j, j + ceil(n/2)) yields words j and j + ceil(n/2); an odd n has
// its unpaired middle word from block (n/2, 0)
__device__ __forceinline__ uint32_t random_word(uint32_t k0, uint32_t k1, uint64_t i, uint64_t n) {
    const uint64_t half = n >> 1, odd = n & 1;
    uint32_t a, b;
    if (i < half) {
        threefry2x32(k0, k1, (uint32_t)i, (uint32_t)(i + half + odd), a, b);
        return a;
    }
    if (odd && i == half) {
        threefry2x32(k0, k1, (uint32_t)half, 0u, a, b);
        return a;
    }
    threefry2x32(k0, k1, (uint32_t)(i - half - odd), (uint32_t)i, a, b);
    return b;
}

// uniform(0, 1): float(word) / float(2^32 - 1) (the divisor rounds to 2^32), clamped below 1
__device__ __forceinline__ float unit_from_word(uint32_t w) {
    return fminf(__uint2float_rn(w) * 2.3283064365386963e-10f, 0.99999994f);
}

// -log(-log(u)), each log correctly rounded to float32 (the oracle's rule, oracle/mlx_rng.py:_log32)
__device__ __forceinline__ float gumbel_from_word(uint32_t w) {
    const float l1 = (float)log((double)unit_from_word(w));
    return -(float)log((double)(-l1));
}

// float32 erfinv, the Giles / Juffa single-precision polynomial MLX evaluates (oracle/mlx_rng.py:erfinv32)
__device__ __forceinline__ float erfinv32(float a) {
    float t = __builtin_fmaf(a, 0.0f - a, 1.0f);
    t = (float)log((double)t);
    float p;
    if (fabsf(t) > 6.125f) {
        p = 3.03697567e-10f;
        p = __builtin_fmaf(p, t, 2.93243101e-8f);
        p = __builtin_fmaf(p, t, 1.22150334e-6f);
        p = __builtin_fmaf(p, t, 2.84108955e-5f);
        p = __builtin_fmaf(p, t, 3.93552968e-4f);
        p = __builtin_fmaf(p, t, 3.02698812e-3f);
        p = __builtin_fmaf(p, t, 4.83185798e-3f);
        p = __builtin_fmaf(p, t, -2.64646143e-1f);
        p = __builtin_fmaf(p, t, 8.40016484e-1f);
    } else {
        p = 5.43877832e-9f;
        p = __builtin_fmaf(p, t, 1.43285448e-7f);
        p = __builtin_fmaf(p, t, 1.22774793e-6f);
        p = __builtin_fmaf(p, t, 1.12963626e-7f);
        p = __builtin_fmaf(p, t, -5.61530760e-5f);
        p = __builtin_fmaf(p, t, -1.47697632e-4f);
        p = __builtin_fmaf(p, t, 2.31468678e-3f);
        p = __builtin_fmaf(p, t, 1.15392581e-2f);
        p = __builtin_fmaf(p, t, -2.32015476e-1f);
        p = __builtin_fmaf(p, t, 8.86226892e-1f);
    }
    return a * p;
}

// normal(0, 1): sqrt(2) * erfinv(uniform(nextafter(-1, 0), 1))
__device__ __forceinline__ float normal_from_word(uint32_t w) {
    const float lo = -0.99999994f;
    const float u = lo + (1.0f - lo) * unit_from_word(w);
    return 1.41421354f * erfinv32(u);
}

// orderable (value, first-index-wins) key, same packing as the greedy sampler
__device__ __forceinline__ unsigned long long sample_key(float v, uint32_t idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    if (v != v) u = 0;
    return ((unsigned long long)u << 32) | (uint32_t)(~idx);
}

// engine hooks
int launch_rng_next(uint32_t* state4, hipStream_t s);
int launch_sample_noise(unsigned long long* partials, int n_partials, const bf16_t* logits, const uint32_t* sub_key, int V_local,
                        int row_offset, int V_global, float inv_temp, bool logits_f16, hipStream_t s);

}  // namespace omx
