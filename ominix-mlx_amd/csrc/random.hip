// a10 (temperature branch): MLX's keyed random generator and the categorical sampler on gfx950.
//   reference: DefaultSampler::sample  mlx-rs-core/src/sampler.rs:9-18  (temp != 0:
//              categorical(logits * array!(1/temp)) -> mlx-rs/src/random.rs:456-497 -> mlx_random_categorical,
//              mlx/c/random.h:59-64); keys: random.rs:21-41 (RandomState), :98-115 (key, split);
//              KATs random.rs:549-562, 690-719 (tests/test_oracle_kats.py pins the oracle on them).
// The generator is counter based (Threefry-2x32, 20 rounds): word i of an n-word draw depends only on
// (key, i, n), so every kernel here computes its own words in registers -- no noise tensor goes through HBM,
// and the sampler is one pass over the logits (HBM-bound: V * 2 bytes per row).
#include "common.hpp"
#include "random.hpp"
#include "vec.hpp"

namespace omx {

namespace {

__global__ void key_kernel(uint32_t* out, uint32_t hi, uint32_t lo) {
    out[0] = hi;
    out[1] = lo;
}

__global__ void bits_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ key, uint64_t n) {
    const uint32_t k0 = key[0], k1 = key[1];
    const uint64_t half = n >> 1, odd = n & 1;
    // one Threefry block yields two words: i and i + half (+1 when n is odd)
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < half + odd; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t a, b;
        if (i < half) {
            threefry2x32(k0, k1, (uint32_t)i, (uint32_t)(i + half + odd), a, b);
            out[i] = a;
            out[i + half + odd] = b;
        } else {
            threefry2x32(k0, k1, (uint32_t)half, 0u, a, b);
            out[half] = a;
        }
    }
}

template <int KIND>   // 0 uniform(lo, lo + range), 1 gumbel, 2 normal * range + lo
__global__ void uniform_kernel(float* __restrict__ out, const uint32_t* __restrict__ key, uint64_t n, float lo, float range) {
    const uint32_t k0 = key[0], k1 = key[1];
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = random_word(k0, k1, i, n);
        if (KIND == 1) {
            out[i] = gumbel_from_word(w);
        } else if (KIND == 2) {
            float z = normal_from_word(w);
            if (range != 1.0f) z = z * range;
            if (lo != 0.0f) z = z + lo;
            out[i] = z;
        } else {
            out[i] = lo + range * unit_from_word(w);
        }
    }
}

// out[r, s] = argmax_v( f32(logits[r, v]) * inv_temp + gumbel(word (r*V + v)*S + s of an R*V*S-word draw) ),
// first index on ties.  One block per (r, s).
template <int DT>
__global__ __launch_bounds__(1024) void categorical_kernel(uint32_t* __restrict__ out, const typename Elem<DT>::T* __restrict__ logits,
                                                            const uint32_t* __restrict__ key, int V, int S, uint64_t n_words,
                                                            float inv_temp, int scale_first) {
    __shared__ unsigned long long red[16];
    const uint32_t k0 = key[0], k1 = key[1];
    const uint64_t r = blockIdx.x / S, s = blockIdx.x % S;
    const typename Elem<DT>::T* row = logits + r * (uint64_t)V;
    unsigned long long best = 0;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        float x = Elem<DT>::ld(row + v);
        if (scale_first) x = x * inv_temp;
        const float g = gumbel_from_word(random_word(k0, k1, (r * V + v) * S + s, n_words));
        const unsigned long long kx = sample_key(x + g, (uint32_t)v);
        best = kx > best ? kx : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (unsigned w = 1; w < blockDim.x / 64; ++w) best = red[w] > best ? red[w] : best;
        out[blockIdx.x] = ~(uint32_t)(best & 0xFFFFFFFFull);
    }
}

}  // namespace

// ---- engine hooks (random.hpp) ----
__global__ void rng_next_kernel(uint32_t* state /*[4]: state, sub-key*/) {
    // RandomState::next (random.rs:32-36): (state, sub) = split(state, 2)
    const uint32_t k0 = state[0], k1 = state[1];
    uint32_t a0, b0, a1, b1;
    threefry2x32(k0, k1, 0u, 2u, a0, b0);   // words 0 and 2
    threefry2x32(k0, k1, 1u, 3u, a1, b1);   // words 1 and 3
    state[0] = a0; state[1] = a1;
    state[2] = b0; state[3] = b1;
}

// F16: the logits row holds IEEE half words (a float16-scale checkpoint's activations), else bfloat16; both widen
// exactly to f32 before the 1/T product, as `logits * array!(1/T)` promotes to f32 in the reference.
template <bool F16>
__global__ __launch_bounds__(256) void sample_noise_kernel(unsigned long long* __restrict__ partials, const bf16_t* __restrict__ logits,
                                                           const uint32_t* __restrict__ sub_key, int V_local, int row_offset,
                                                           int V_global, float inv_temp) {
    __shared__ unsigned long long red[4];
    const uint32_t k0 = sub_key[0], k1 = sub_key[1];
    unsigned long long best = 0;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < V_local; v += gridDim.x * 256) {
        const int gv = v + row_offset;
        const float x = (F16 ? (float)reinterpret_cast<const _Float16*>(logits)[v] : bf16_to_f32(logits[v])) * inv_temp;
        const float g = gumbel_from_word(random_word(k0, k1, (uint64_t)gv, (uint64_t)V_global));
        const unsigned long long kx = sample_key(x + g, (uint32_t)gv);
        best = kx > best ? kx : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) best = red[w] > best ? red[w] : best;
        partials[blockIdx.x] = best;
    }
}

int launch_rng_next(uint32_t* state4, hipStream_t s) {
    rng_next_kernel<<<1, 1, 0, s>>>(state4);
    OMX_LAUNCH_CHECK();
    return 0;
}

int launch_sample_noise(unsigned long long* partials, int n_partials, const bf16_t* logits, const uint32_t* sub_key, int V_local,
                        int row_offset, int V_global, float inv_temp, bool logits_f16, hipStream_t s) {
    if (logits_f16) sample_noise_kernel<true><<<n_partials, 256, 0, s>>>(partials, logits, sub_key, V_local, row_offset, V_global, inv_temp);
    else sample_noise_kernel<false><<<n_partials, 256, 0, s>>>(partials, logits, sub_key, V_local, row_offset, V_global, inv_temp);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // namespace omx

extern "C" {

int omx_random_key(uint32_t* key, uint64_t seed, omx_stream stream) {
    OMX_REQUIRE(key, "omx_random_key: null key");
    omx::key_kernel<<<1, 1, 0, (hipStream_t)stream>>>(key, (uint32_t)(seed >> 32), (uint32_t)(seed & 0xFFFFFFFFull));
    OMX_LAUNCH_CHECK();
    return 0;
}

int omx_random_bits(uint32_t* out, const uint32_t* key, int64_t n, omx_stream stream) {
    OMX_REQUIRE(out && key, "omx_random_bits: null tensor");
    OMX_REQUIRE(n >= 0 && n <= 0x1FFFFFFFELL, "omx_random_bits: %lld words exceed the 2^32 counter space of one key", (long long)n);
    if (n == 0) return 0;
    const uint64_t work = ((uint64_t)n >> 1) + ((uint64_t)n & 1);
    const unsigned blocks = (unsigned)std::min<uint64_t>((work + 255) / 256, 65535);
    omx::bits_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(out, key, (uint64_t)n);
    OMX_LAUNCH_CHECK();
    return 0;
}

int omx_random_split(uint32_t* out, const uint32_t* key, int num, omx_stream stream) {
    OMX_REQUIRE(num >= 1, "omx_random_split: num=%d must be positive", num);
    return omx_random_bits(out, key, 2 * (int64_t)num, stream);
}

int omx_random_uniform(float* out, const uint32_t* key, int64_t n, float lo, float hi, omx_stream stream) {
    OMX_REQUIRE(out && key, "omx_random_uniform: null tensor");
    OMX_REQUIRE(n >= 0 && n <= 0x1FFFFFFFELL, "omx_random_uniform: %lld samples exceed the counter space of one key", (long long)n);
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 65535);
    omx::uniform_kernel<0><<<blocks, 256, 0, (hipStream_t)stream>>>(out, key, (uint64_t)n, lo, hi - lo);
    OMX_LAUNCH_CHECK();
    return 0;
}

int omx_random_gumbel(float* out, const uint32_t* key, int64_t n, omx_stream stream) {
    OMX_REQUIRE(out && key, "omx_random_gumbel: null tensor");
    OMX_REQUIRE(n >= 0 && n <= 0x1FFFFFFFELL, "omx_random_gumbel: %lld samples exceed the counter space of one key", (long long)n);
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 65535);
    omx::uniform_kernel<1><<<blocks, 256, 0, (hipStream_t)stream>>>(out, key, (uint64_t)n, 0.f, 1.f);
    OMX_LAUNCH_CHECK();
    return 0;
}

int omx_random_normal(float* out, const uint32_t* key, int64_t n, float loc, float scale, omx_stream stream) {
    OMX_REQUIRE(out && key, "omx_random_normal: null tensor");
    OMX_REQUIRE(n >= 0 && n <= 0x1FFFFFFFELL, "omx_random_normal: %lld samples exceed the counter space of one key", (long long)n);
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 65535);
    omx::uniform_kernel<2><<<blocks, 256, 0, (hipStream_t)stream>>>(out, key, (uint64_t)n, loc, scale);
    OMX_LAUNCH_CHECK();
    return 0;
}

int omx_random_categorical(uint32_t* out, const void* logits, int64_t rows, int n, int num_samples, float inv_temp,
                           const uint32_t* key, omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && logits && key, "omx_random_categorical: null tensor");
    OMX_REQUIRE(n > 0, "omx_random_categorical: empty distribution axis");
    OMX_REQUIRE(num_samples >= 1, "omx_random_categorical: num_samples=%d must be positive", num_samples);
    if (rows == 0) return 0;
    const uint64_t words = (uint64_t)rows * (uint64_t)n * (uint64_t)num_samples;
    OMX_REQUIRE(words <= 0x1FFFFFFFEULL, "omx_random_categorical: %llu noise words exceed the counter space of one key",
                (unsigned long long)words);
    OMX_REQUIRE(rows * num_samples <= 0x7FFFFFFFLL, "omx_random_categorical: too many rows");
    const int scale_first = inv_temp != 1.0f;
    OMX_DISPATCH_FLOAT(dtype, "omx_random_categorical",
                       (omx::categorical_kernel<DT><<<(unsigned)(rows * num_samples), 1024, 0, (hipStream_t)stream>>>(
                           out, (const omx::Elem<DT>::T*)logits, key, n, num_samples, words, inv_temp, scale_first)));
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
