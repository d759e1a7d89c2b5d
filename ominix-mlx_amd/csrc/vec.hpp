// 16-byte vector access helpers: one global_load_dwordx4 / global_store_dwordx4 per lane.
#pragma once
#include "common.hpp"

namespace omx {

template <int DT> struct Vec16;

template <> struct Vec16<OMX_BFLOAT16> {
    static constexpr int N = 8;
    typedef bf16_t T;
    static __device__ __forceinline__ void ld(const T* p, float (&v)[8]) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = bf16lo(r[i]);
            v[2 * i + 1] = bf16hi(r[i]);
        }
    }
    static __device__ __forceinline__ void st(T* p, const float (&v)[8]) {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = pack_bf16(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<u32x4*>(p) = r;
    }
};

template <> struct Vec16<OMX_FLOAT16> {
    static constexpr int N = 8;
    typedef f16_t T;
    using h8 = __attribute__((ext_vector_type(8))) _Float16;
    static __device__ __forceinline__ void ld(const T* p, float (&v)[8]) {
        const h8 r = *reinterpret_cast<const h8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
    }
    static __device__ __forceinline__ void st(T* p, const float (&v)[8]) {
        h8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (_Float16)v[i];
        *reinterpret_cast<h8*>(p) = r;
    }
};

template <> struct Vec16<OMX_FLOAT32> {
    static constexpr int N = 4;
    typedef float T;
    static __device__ __forceinline__ void ld(const T* p, float (&v)[4]) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = r[i];
    }
    static __device__ __forceinline__ void st(T* p, const float (&v)[4]) {
        f32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = v[i];
        *reinterpret_cast<f32x4*>(p) = r;
    }
};

__host__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace omx

// dtype dispatch for host launchers
#define OMX_DISPATCH_FLOAT(dtype, NAME, ...)                                   \
    switch (dtype) {                                                           \
        case OMX_BFLOAT16: { constexpr int DT = OMX_BFLOAT16; __VA_ARGS__; break; } \
        case OMX_FLOAT16:  { constexpr int DT = OMX_FLOAT16;  __VA_ARGS__; break; } \
        case OMX_FLOAT32:  { constexpr int DT = OMX_FLOAT32;  __VA_ARGS__; break; } \
        default: return omx::set_error(NAME ": unsupported dtype %d", (int)(dtype)); \
    }
