// The 16-bit activation format of the decode-step kernels (packed-weight GEMVs, decode attention, embedding, MoE glue).
#pragma once
#include <hip/hip_fp16.h>

#include "common.hpp"

namespace omx {

// bfloat16, or float16: bfloat16, or float16 for a float16 checkpoint -- MLX runs such a model in
// float16 END TO END (nn/quantized.rs:361-385: the dequantised weight has the scales' dtype, the matmul its inputs'), so x, the
// RMSNorm output, every rounding point and the result are float16 there; the accumulation is float32 either way.
template <bool F16> struct Act16;
template <> struct Act16<false> {
    static constexpr uint32_t kMagicBytes = 0x43434343u;
    static constexpr float kMagic = 128.0f;      // 0x4300 | q stays 128 + q in the dot product; 128 * sum(x) is folded into the bias term
    static __device__ __forceinline__ uint32_t unmagic(uint32_t q) { return q; }
    static __device__ __forceinline__ float lo(uint32_t p) { return bf16lo(p); }
    static __device__ __forceinline__ float hi(uint32_t p) { return bf16hi(p); }
    static __device__ __forceinline__ float val(uint16_t b) { return bf16_to_f32(b); }
    static __device__ __forceinline__ uint16_t bits(float v) { return f32_to_bf16(v); }
    static __device__ __forceinline__ float rnd(float v) { return round_bf16(v); }
    static __device__ __forceinline__ uint32_t pack(float l, float h) { return pack_bf16(l, h); }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        typedef __bf16 v2 __attribute__((ext_vector_type(2)));
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
    }
};
template <> struct Act16<true> {
    static constexpr uint32_t kMagicBytes = 0x64646464u;
    // the 1024 is taken off again exactly (one packed float16 subtract per pair) before the dot product: left in, like the bf16 form's
    // 128, its 1024 * sum(x) excess would cost ~10 bits of the f32 accumulation against float16's 11-bit results
    static constexpr float kMagic = 0.0f;
    static __device__ __forceinline__ uint32_t unmagic(uint32_t q) {
        typedef _Float16 v2 __attribute__((ext_vector_type(2)));
        const v2 r = __builtin_bit_cast(v2, q) - v2{(_Float16)1024.0f, (_Float16)1024.0f};
        return __builtin_bit_cast(uint32_t, r);
    }
    static __device__ __forceinline__ float val(uint16_t b) { return __half2float(__ushort_as_half(b)); }
    static __device__ __forceinline__ float lo(uint32_t p) { return val((uint16_t)(p & 0xFFFFu)); }
    static __device__ __forceinline__ float hi(uint32_t p) { return val((uint16_t)(p >> 16)); }
    static __device__ __forceinline__ uint16_t bits(float v) { return __half_as_ushort(__float2half_rn(v)); }
    static __device__ __forceinline__ float rnd(float v) { return val(bits(v)); }
    static __device__ __forceinline__ uint32_t pack(float l, float h) { return (uint32_t)bits(l) | ((uint32_t)bits(h) << 16); }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        typedef _Float16 v2 __attribute__((ext_vector_type(2)));
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
    }
};


}  // namespace omx
