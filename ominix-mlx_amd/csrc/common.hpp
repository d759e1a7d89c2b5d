// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wavefront = 64 lanes; every reduction below is written for that width.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/omx.h"

namespace omx {

typedef uint16_t bf16_t;   // raw bfloat16 bits
typedef _Float16 f16_t;

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kWave = 64;

// ---- bf16 <-> f32 ---------------------------------------------------------
__device__ __host__ __forceinline__ float bf16_to_f32(bf16_t b) {
    union { uint32_t u; float f; } c;
    c.u = (uint32_t)b << 16;
    return c.f;
}
// round-to-nearest-even, NaN kept quiet
__device__ __host__ __forceinline__ bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    // gfx950 converts in hardware (v_cvt_pk_bf16_f32, RNE); the software form below costs a compare and a
    // per-lane branch per element, which dominated the flash-attention softmax (64 conversions per tile)
    return __builtin_bit_cast(bf16_t, (__bf16)f);
#else
    union { uint32_t u; float f; } c;
    c.f = f;
    const uint32_t u = c.u;
    const uint32_t rne = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    const uint32_t nan = (u >> 16) | 0x0040u;
    return (bf16_t)(((u & 0x7FFFFFFFu) > 0x7F800000u) ? nan : rne);
#endif
}
__device__ __forceinline__ float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

// low / high bf16 of a packed dword as f32
__device__ __forceinline__ float bf16lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16hi(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// ---- dtype traits: element load/store as f32 --------------------------------
template <int DT> struct Elem;
template <> struct Elem<OMX_BFLOAT16> {
    typedef bf16_t T;
    static __device__ __forceinline__ float ld(const T* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(T* p, float v) { *p = f32_to_bf16(v); }
    static __device__ __forceinline__ float rnd(float v) { return round_bf16(v); }
};
template <> struct Elem<OMX_FLOAT16> {
    typedef f16_t T;
    static __device__ __forceinline__ float ld(const T* p) { return (float)*p; }
    static __device__ __forceinline__ void st(T* p, float v) { *p = (f16_t)v; }
    static __device__ __forceinline__ float rnd(float v) { return (float)(f16_t)v; }
};
template <> struct Elem<OMX_FLOAT32> {
    typedef float T;
    static __device__ __forceinline__ float ld(const T* p) { return *p; }
    static __device__ __forceinline__ void st(T* p, float v) { *p = v; }
    static __device__ __forceinline__ float rnd(float v) { return v; }
};

// ---- wave64 reductions -------------------------------------------------------
// Cross-lane traffic stays in the VALU: DPP inside a 16-lane row (quad_perm xor1 / xor2,
// row_half_mirror, row_mirror), v_readlane across the four rows.  The LDS-crossbar route
// (__shfl_xor -> ds_bpermute_b32) costs ~100+ cycles per dependent step and hipcc serialises
// every step behind an s_waitcnt: 144 of them were 8 us of a 9 us decode-attention kernel.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppRowMirror = 0x140;
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// sum over aligned groups of N lanes (N = 4, 8, 16); every lane of the group gets the result
template <int N>
__device__ __forceinline__ float group_sum(float v) {
    v += dpp_f<kDppXor1>(v);
    v += dpp_f<kDppXor2>(v);
    if (N >= 8) v += dpp_f<kDppHalfMirror>(v);
    if (N >= 16) v += dpp_f<kDppRowMirror>(v);
    return v;
}
template <int N>
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, dpp_f<kDppXor1>(v));
    v = fmaxf(v, dpp_f<kDppXor2>(v));
    if (N >= 8) v = fmaxf(v, dpp_f<kDppHalfMirror>(v));
    if (N >= 16) v = fmaxf(v, dpp_f<kDppRowMirror>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = group_sum<16>(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = group_max<16>(v);
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

// block reductions over NW waves through a small LDS scratch (NW floats)
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    if (NW == 1) return v;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += scratch[i];
    return s;
}

}  // namespace omx

// ---- host-side error plumbing (capi.cpp) ------------------------------------
namespace omx {
int set_error(const char* fmt, ...);   // formats into TLS slot, calls handler, returns 1
}
#define OMX_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return omx::set_error("%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define OMX_REQUIRE(cond, ...)                        \
    do {                                              \
        if (!(cond)) return omx::set_error(__VA_ARGS__); \
    } while (0)
#define OMX_LAUNCH_CHECK() OMX_HIP_CHECK(hipGetLastError())
