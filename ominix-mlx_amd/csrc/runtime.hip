// Library runtime: error slot + handler (mlx-c error.cpp:12-54 conventions), device
// memory, streams, synthetic fills.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.hpp"

namespace {
thread_local char g_err[1024] = {0};
omx_error_handler_func g_handler = nullptr;
void* g_handler_data = nullptr;
void (*g_handler_dtor)(void*) = nullptr;
std::mutex g_handler_mu;
}  // namespace

namespace omx {
int set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    omx_error_handler_func h;
    void* d;
    {
        std::lock_guard<std::mutex> lk(g_handler_mu);
        h = g_handler;
        d = g_handler_data;
    }
    if (h) h(g_err, d);
    return 1;
}
}  // namespace omx

extern "C" {

const char* omx_version(void) { return "omx-hip 0.1.0 (gfx950)"; }

// 1: this library was built with `make EXPERIMENTS=1` (the measured-negative engines of EXPERIMENTS.md are in it)
int omx_experiments_built(void) {
#ifdef OMX_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

void omx_set_error_handler(omx_error_handler_func handler, void* data, void (*dtor)(void*)) {
    std::lock_guard<std::mutex> lk(g_handler_mu);
    if (g_handler_dtor && g_handler_data) g_handler_dtor(g_handler_data);
    g_handler = handler;
    g_handler_data = data;
    g_handler_dtor = dtor;
}
const char* omx_last_error(void) { return g_err; }
void omx_clear_error(void) { g_err[0] = 0; }

int omx_device_count(int* count) {
    OMX_REQUIRE(count != nullptr, "omx_device_count: null out pointer");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        (void)hipGetLastError();
    }
    return 0;
}
int omx_device_name(char* buf, size_t buflen) {
    OMX_REQUIRE(buf && buflen > 0, "omx_device_name: bad buffer");
    hipDeviceProp_t p;
    int dev = 0;
    OMX_HIP_CHECK(hipGetDevice(&dev));
    OMX_HIP_CHECK(hipGetDeviceProperties(&p, dev));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return 0;
}
int omx_synchronize(omx_stream stream) {
    OMX_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int omx_malloc(void** ptr, size_t bytes) {
    OMX_REQUIRE(ptr != nullptr, "omx_malloc: null out pointer");
    OMX_HIP_CHECK(hipMalloc(ptr, bytes ? bytes : 16));
    return 0;
}
int omx_free(void* ptr) {
    if (ptr) OMX_HIP_CHECK(hipFree(ptr));
    return 0;
}
int omx_memcpy_h2d(void* dst, const void* src, size_t bytes, omx_stream s) {
    OMX_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    return 0;
}
int omx_memcpy_d2h(void* dst, const void* src, size_t bytes, omx_stream s) {
    OMX_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
    OMX_HIP_CHECK(hipStreamSynchronize((hipStream_t)s));
    return 0;
}
int omx_memcpy_d2d(void* dst, const void* src, size_t bytes, omx_stream s) {
    OMX_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return 0;
}
int omx_memset(void* dst, int value, size_t bytes, omx_stream s) {
    OMX_HIP_CHECK(hipMemsetAsync(dst, value, bytes, (hipStream_t)s));
    return 0;
}

}  // extern "C"

// ---- synthetic fill -----------------------------------------------------------
namespace {
__device__ __forceinline__ uint32_t hash_u32(uint64_t idx, uint32_t seed) {
    uint32_t x = (uint32_t)(idx * 0x9E3779B1ull + seed);
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
template <int DT>
__global__ void fill_uniform_kernel(typename omx::Elem<DT>::T* dst, size_t n, uint32_t seed, float amp, float offset) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        // NOTE: 2*u-1 and the final fma are written as separate roundings to match oracle/synth.py
        const float u = (float)(hash_u32(i, seed) >> 8) * (1.0f / 16777216.0f);
        const float t = __fsub_rn(__fmul_rn(2.0f, u), 1.0f);
        omx::Elem<DT>::st(dst + i, __fadd_rn(offset, __fmul_rn(amp, t)));
    }
}
template <int DT>
__global__ void fill_uniform_2d_kernel(typename omx::Elem<DT>::T* dst, int64_t rows, int64_t cols, int64_t ld_full,
                                       int64_t row0, int64_t col0, uint32_t seed, float amp, float offset) {
    const int64_t n = rows * cols;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int64_t r = i / cols, c = i % cols;
        const uint64_t logical = (uint64_t)((row0 + r) * ld_full + col0 + c);
        const float u = (float)(hash_u32(logical, seed) >> 8) * (1.0f / 16777216.0f);
        const float t = __fsub_rn(__fmul_rn(2.0f, u), 1.0f);
        omx::Elem<DT>::st(dst + i, __fadd_rn(offset, __fmul_rn(amp, t)));
    }
}
}  // namespace

extern "C" int omx_fill_uniform_2d(void* dst, int64_t rows, int64_t cols, int64_t ld_full, int64_t row0, int64_t col0,
                                   uint32_t seed, float amp, float offset, omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(dst && rows >= 0 && cols >= 0 && ld_full >= cols, "omx_fill_uniform_2d: bad arguments");
    const int64_t n = rows * cols;
    if (n == 0) return 0;
    int64_t want = (n + 255) / 256;
    const int blocks = (int)(want < 8192 ? want : 8192);
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case OMX_BFLOAT16:
            fill_uniform_2d_kernel<OMX_BFLOAT16><<<blocks, 256, 0, s>>>((omx::bf16_t*)dst, rows, cols, ld_full, row0, col0, seed, amp, offset);
            break;
        case OMX_FLOAT32:
            fill_uniform_2d_kernel<OMX_FLOAT32><<<blocks, 256, 0, s>>>((float*)dst, rows, cols, ld_full, row0, col0, seed, amp, offset);
            break;
        default:
            return omx::set_error("omx_fill_uniform_2d: unsupported dtype %d", (int)dtype);
    }
    OMX_LAUNCH_CHECK();
    return 0;
}

extern "C" int omx_fill_uniform(void* dst, size_t n, uint32_t seed, float amp, float offset, omx_dtype dtype,
                                omx_stream stream) {
    if (n == 0) return 0;
    const int threads = 256;
    size_t want = (n + threads - 1) / threads;
    const int blocks = (int)(want < 4096 ? want : 4096);
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case OMX_BFLOAT16:
            fill_uniform_kernel<OMX_BFLOAT16><<<blocks, threads, 0, s>>>((omx::bf16_t*)dst, n, seed, amp, offset);
            break;
        case OMX_FLOAT16:
            fill_uniform_kernel<OMX_FLOAT16><<<blocks, threads, 0, s>>>((omx::f16_t*)dst, n, seed, amp, offset);
            break;
        case OMX_FLOAT32:
            fill_uniform_kernel<OMX_FLOAT32><<<blocks, threads, 0, s>>>((float*)dst, n, seed, amp, offset);
            break;
        default:
            return omx::set_error("omx_fill_uniform: unsupported dtype %d", (int)dtype);
    }
    OMX_LAUNCH_CHECK();
    return 0;
}
