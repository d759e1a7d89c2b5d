// Elementwise / small glue kernels of the per-op ABI: RoPE, fused_swiglu, add, embedding
// gather, argmax.  All HBM-bound: 16-B vector access where the layout allows it.
#include <math.h>

#include "vec.hpp"

namespace omx {

// ---------------------------------------------------------------------------------
// RoPE  (mlx_fast_rope, mlx-c fast.h:169-178; mlx-rs/src/fast.rs:15-46)
// x,out [batch, T, D]; thread = (position t, pair i): the angle and its sin/cos are
// evaluated ONCE in fp64 and reused across the batch(=B*H) loop, so trig cost is
// amortised and the angle keeps full precision at long offsets.
// ---------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void rope_kernel(typename Elem<DT>::T* __restrict__ out,
                                                   const typename Elem<DT>::T* __restrict__ x, int64_t batch, int T,
                                                   int D, int dims, int traditional, double neg_log_base_over_half,
                                                   double scale, int offset, const float* __restrict__ freqs = nullptr) {
    typedef typename Elem<DT>::T T_;
    const int half = dims >> 1;
    const int per_row = half + (D - dims);   // rotating pairs + pass-through columns
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)T * per_row) return;
    const int t = (int)(idx / per_row);
    const int j = (int)(idx % per_row);
    if (j >= half) {   // columns >= dims are copied unchanged
        const int col = dims + (j - half);
        for (int64_t b = 0; b < batch; ++b) out[(b * T + t) * D + col] = x[(b * T + t) * D + col];
        return;
    }
    // custom frequencies (fast.rs:15-46 `freqs`): the angle of pair j is position / freqs[j] instead of position * base^(-j / half)
    const double ang = ((double)(offset + t) * scale) * (freqs ? 1.0 / (double)freqs[j] : exp((double)j * neg_log_base_over_half));
    double sd, cd;
    sincos(ang, &sd, &cd);
    const float c = (float)cd, s = (float)sd;
    const int i0 = traditional ? 2 * j : j;
    const int i1 = traditional ? 2 * j + 1 : j + half;
    for (int64_t b = 0; b < batch; ++b) {
        const int64_t base = (b * T + t) * D;
        const float x1 = Elem<DT>::ld(x + base + i0);
        const float x2 = Elem<DT>::ld(x + base + i1);
        Elem<DT>::st((T_*)out + base + i0, x1 * c - x2 * s);
        Elem<DT>::st((T_*)out + base + i1, x1 * s + x2 * c);
    }
}

// ---------------------------------------------------------------------------------
// fused_swiglu (mlx-rs-core/src/metal_kernels.rs:11-18): out = silu(gate) * x
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + __expf(-g)); }

template <int DT, int OP>   // OP 0: swiglu(x,gate)  1: add(a,b)
__global__ __launch_bounds__(256) void binary_kernel(typename Elem<DT>::T* __restrict__ out,
                                                     const typename Elem<DT>::T* __restrict__ a,
                                                     const typename Elem<DT>::T* __restrict__ b, int64_t n, bool vec) {
    constexpr int N = Vec16<DT>::N;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if (vec) {
        const int64_t nv = n / N;
        for (int64_t i = tid; i < nv; i += stride) {
            float va[N], vb[N];
            Vec16<DT>::ld(a + i * N, va);
            Vec16<DT>::ld(b + i * N, vb);
#pragma unroll
            for (int j = 0; j < N; ++j) va[j] = (OP == 0) ? silu_f(vb[j]) * va[j] : va[j] + vb[j];
            Vec16<DT>::st(out + i * N, va);
        }
        done = nv * N;
    }
    for (int64_t i = done + tid; i < n; i += stride) {
        const float x = Elem<DT>::ld(a + i), y = Elem<DT>::ld(b + i);
        Elem<DT>::st(out + i, (OP == 0) ? silu_f(y) * x : x + y);
    }
}

template <int DT, int OP>
static int launch_binary(void* out, const void* a, const void* b, int64_t n, hipStream_t s) {
    typedef typename Elem<DT>::T T;
    if (n == 0) return 0;
    const bool vec = aligned16(out) && aligned16(a) && aligned16(b);
    int64_t work = vec ? (n + Vec16<DT>::N - 1) / Vec16<DT>::N : n;
    int64_t blocks = (work + 255) / 256;
    if (blocks > 2048) blocks = 2048;   // grid-stride beyond 8 blocks/CU
    binary_kernel<DT, OP><<<(unsigned)blocks, 256, 0, s>>>((T*)out, (const T*)a, (const T*)b, n, vec);
    OMX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------
// embedding gather: out[r,:] = table[ids[r],:]
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void take_rows_kernel(uint8_t* __restrict__ out, const uint8_t* __restrict__ table,
                                                        const uint32_t* __restrict__ ids, int64_t row_bytes, bool vec) {
    const int64_t r = blockIdx.x;
    const uint8_t* src = table + (int64_t)ids[r] * row_bytes;
    uint8_t* dst = out + r * row_bytes;
    if (vec) {
        for (int64_t i = threadIdx.x * 16; i < row_bytes; i += 256 * 16)
            *reinterpret_cast<u32x4*>(dst + i) = *reinterpret_cast<const u32x4*>(src + i);
    } else {
        for (int64_t i = threadIdx.x; i < row_bytes; i += 256) dst[i] = src[i];
    }
}

// ---------------------------------------------------------------------------------
// argmax over the last axis, first index on ties (sampler.rs:9-12)
// key = (orderable(value) << 32) | ~index  -> a plain u64 max picks max value, min index
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t argmax_key(float v, uint32_t idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    if (v != v) u = 0;   // NaN never wins
    return ((uint64_t)u << 32) | (uint32_t)(~idx);
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t other = __shfl_xor(k, o, 64);
        k = other > k ? other : k;
    }
    return k;
}

template <int DT>
__global__ __launch_bounds__(1024) void argmax_kernel(uint32_t* __restrict__ out,
                                                      const typename Elem<DT>::T* __restrict__ logits, int n) {
    __shared__ uint64_t part[16];
    const int64_t r = blockIdx.x;
    const typename Elem<DT>::T* row = logits + r * (int64_t)n;
    uint64_t best = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t k = argmax_key(Elem<DT>::ld(row + i), (uint32_t)i);
        best = k > best ? k : best;
    }
    best = wave_max_u64(best);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x < 64) {
        uint64_t k = threadIdx.x < (blockDim.x >> 6) ? part[threadIdx.x] : 0;
        k = wave_max_u64(k);
        if (threadIdx.x == 0) out[r] = ~(uint32_t)(k & 0xFFFFFFFFu);
    }
}

}  // namespace omx

extern "C" {

int omx_rope(void* out, const void* x, int64_t batch, int T, int D, int dims, int traditional, float base, float scale,
             int offset, omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && x, "omx_rope: null tensor");
    OMX_REQUIRE(dims > 0 && dims <= D && (dims % 2) == 0, "omx_rope: dims=%d must be even and <= D=%d", dims, D);
    OMX_REQUIRE(base > 0.f, "omx_rope: base must be positive");
    if (batch == 0 || T == 0) return 0;
    const int half = dims / 2;
    const int64_t total = (int64_t)T * (half + (D - dims));
    const unsigned blocks = (unsigned)((total + 255) / 256);
    const double nl = -log((double)base) / (double)half;
    OMX_DISPATCH_FLOAT(dtype, "omx_rope",
                       (omx::rope_kernel<DT><<<blocks, 256, 0, (hipStream_t)stream>>>(
                           (omx::Elem<DT>::T*)out, (const omx::Elem<DT>::T*)x, batch, T, D, dims, traditional, nl,
                           (double)scale, offset)));
    OMX_LAUNCH_CHECK();
    return 0;
}

/* fast::rope with `freqs` (fast.rs:15-46; mlx/c/fast.h mlx_fast_rope): freqs [dims / 2] float32 on the device replace the base */
int omx_rope_freqs(void* out, const void* x, int64_t batch, int T, int D, int dims, int traditional, const float* freqs, float scale,
                   int offset, omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && x && freqs, "omx_rope_freqs: null tensor");
    OMX_REQUIRE(dims > 0 && dims <= D && (dims % 2) == 0, "omx_rope_freqs: dims=%d must be even and <= D=%d", dims, D);
    if (batch == 0 || T == 0) return 0;
    const int half = dims / 2;
    const int64_t total = (int64_t)T * (half + (D - dims));
    const unsigned blocks = (unsigned)((total + 255) / 256);
    OMX_DISPATCH_FLOAT(dtype, "omx_rope_freqs",
                       (omx::rope_kernel<DT><<<blocks, 256, 0, (hipStream_t)stream>>>(
                           (omx::Elem<DT>::T*)out, (const omx::Elem<DT>::T*)x, batch, T, D, dims, traditional, 0.0,
                           (double)scale, offset, freqs)));
    OMX_LAUNCH_CHECK();
    return 0;
}

int omx_fused_swiglu(void* out, const void* x, const void* gate, int64_t n, omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && x && gate, "omx_fused_swiglu: null tensor");
    OMX_DISPATCH_FLOAT(dtype, "omx_fused_swiglu", return (omx::launch_binary<DT, 0>(out, x, gate, n, (hipStream_t)stream)));
    return 0;
}

int omx_add(void* out, const void* a, const void* b, int64_t n, omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && a && b, "omx_add: null tensor");
    OMX_DISPATCH_FLOAT(dtype, "omx_add", return (omx::launch_binary<DT, 1>(out, a, b, n, (hipStream_t)stream)));
    return 0;
}

int omx_take_rows(void* out, const void* table, const uint32_t* ids, int64_t n_ids, int dim, omx_dtype dtype,
                  omx_stream stream) {
    OMX_REQUIRE(out && table && ids, "omx_take_rows: null tensor");
    if (n_ids == 0 || dim == 0) return 0;
    size_t es = (dtype == OMX_FLOAT32 || dtype == OMX_UINT32 || dtype == OMX_INT32) ? 4
                : (dtype == OMX_BFLOAT16 || dtype == OMX_FLOAT16)                   ? 2
                                                                                    : 0;
    OMX_REQUIRE(es != 0, "omx_take_rows: unsupported dtype %d", (int)dtype);
    const int64_t row_bytes = (int64_t)dim * es;
    const bool vec = (row_bytes % 16 == 0) && omx::aligned16(out) && omx::aligned16(table);
    omx::take_rows_kernel<<<(unsigned)n_ids, 256, 0, (hipStream_t)stream>>>((uint8_t*)out, (const uint8_t*)table, ids,
                                                                             row_bytes, vec);
    OMX_LAUNCH_CHECK();
    return 0;
}

int omx_argmax(uint32_t* out, const void* logits, int64_t rows, int n, omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && logits, "omx_argmax: null tensor");
    OMX_REQUIRE(n > 0, "omx_argmax: empty reduction axis");
    if (rows == 0) return 0;
    OMX_DISPATCH_FLOAT(dtype, "omx_argmax",
                       (omx::argmax_kernel<DT><<<(unsigned)rows, 1024, 0, (hipStream_t)stream>>>(
                           out, (const omx::Elem<DT>::T*)logits, n)));
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
