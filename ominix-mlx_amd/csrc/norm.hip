// Row normalisations: RMSNorm, LayerNorm, fused_modulate.
//   reference: mlx_fast_rms_norm / mlx_fast_layer_norm (mlx-c fast.h:163-168, 93-99;
//   mlx-rs/src/fast.rs:165-219) and mlx-rs-core/src/metal_kernels.rs:28-94 (fused_modulate).
// One wave64 per row: 16-B loads, fp32 accumulation, cross-lane reduction by shuffles only
// (no LDS).  A row is read from HBM once; the second/third pass hits the CU's L1/L2.
// These standalone kernels serve the per-op ABI; the decode engine fuses RMSNorm into the
// GEMV prologue (gemv.hip) and never launches them.
#include "vec.hpp"

namespace omx {

enum { NORM_RMS = 0, NORM_LAYER = 1, NORM_MODULATE = 2 };

// MODE NORM_MODULATE: w = scale[B,H], b = shift[B,H] broadcast over S rows per batch:
//   out = (1 + scale) * LN(x) + shift
// KEEP: 16-byte vectors a lane holds of a row that is read once (8: rows up to 4096 bf16 / 2048 f32 elements; 6: up to 3072 / 1536 -- the
// DiT's hidden size: 48 instead of 64 row registers put a fifth wave on every SIMD, and 4 608 rows then fit the chip in one round)
template <int DT, int MODE, bool VEC, int KEEP = 8>
__global__ __launch_bounds__(256) void rownorm_kernel(typename Elem<DT>::T* __restrict__ out,
                                                      const typename Elem<DT>::T* __restrict__ x,
                                                      const typename Elem<DT>::T* __restrict__ w,
                                                      const typename Elem<DT>::T* __restrict__ b, int64_t rows,
                                                      int dim, float eps, int rows_per_batch, bool keep) {
    typedef typename Elem<DT>::T T;
    constexpr int N = Vec16<DT>::N;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * dim;
    T* orow = out + row * dim;
    const T* wr = w;
    const T* br = b;
    if (MODE == NORM_MODULATE) {
        const int64_t bi = row / rows_per_batch;
        wr = w + bi * dim;
        br = b + bi * dim;
    }
    // rows that fit a lane's registers (<= KEEP 16-byte vectors per lane: 4096 bf16 / 2048 f32 elements): ONE read of the row -- the
    // three passes of a LayerNorm (mean, variance, output) were three dependent global round trips, 26.6 us for the DiT's
    // [4608, 3072] modulate (56 MB: 2.1 TB/s).  Same sums in the same order: bit-identical.
    if (VEC && keep && dim <= KEEP * 64 * N) {
        float v[KEEP][N];
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int i = lane * N + k * 64 * N;
            if (i < dim) Vec16<DT>::ld(xr + i, v[k]);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KEEP; ++k)
            if (lane * N + k * 64 * N < dim) {
#pragma unroll
                for (int j = 0; j < N; ++j) s += (MODE == NORM_RMS) ? v[k][j] * v[k][j] : v[k][j];
            }
        s = wave_sum(s);
        float mean = 0.f, rstd;
        if (MODE == NORM_RMS) {
            rstd = 1.0f / sqrtf(s / (float)dim + eps);
        } else {
            mean = s / (float)dim;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < KEEP; ++k)
                if (lane * N + k * 64 * N < dim) {
#pragma unroll
                    for (int j = 0; j < N; ++j) q += (v[k][j] - mean) * (v[k][j] - mean);
                }
            q = wave_sum(q);
            rstd = 1.0f / sqrtf(q / (float)dim + eps);
        }
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int i = lane * N + k * 64 * N;
            if (i < dim) {
                float wv[N], bv[N];
                if (wr) Vec16<DT>::ld(wr + i, wv);
                if (br) Vec16<DT>::ld(br + i, bv);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float y = (v[k][j] - mean) * rstd;
                    if (MODE == NORM_MODULATE) {
                        y = (1.0f + wv[j]) * y + bv[j];
                    } else {
                        if (wr) y *= wv[j];
                        if (MODE == NORM_LAYER && br) y += bv[j];
                    }
                    v[k][j] = y;
                }
                Vec16<DT>::st(orow + i, v[k]);
            }
        }
        return;
    }
    // (unroll 8: with a handful of rows -- a decode-sized batch -- nothing hides a load's latency but the row's other loads; the rolled
    // loop waited for each 16-byte load before issuing the next: 7.4 us for 5 rows of 4096)
    float s = 0.f;
    if (VEC) {
#pragma unroll 8
        for (int i = lane * N; i < dim; i += 64 * N) {
            float v[N];
            Vec16<DT>::ld(xr + i, v);
#pragma unroll
            for (int j = 0; j < N; ++j) s += (MODE == NORM_RMS) ? v[j] * v[j] : v[j];
        }
    } else {
        for (int i = lane; i < dim; i += 64) {
            const float v = Elem<DT>::ld(xr + i);
            s += (MODE == NORM_RMS) ? v * v : v;
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (MODE == NORM_RMS) {
        rstd = 1.0f / sqrtf(s / (float)dim + eps);
    } else {
        mean = s / (float)dim;
        float q = 0.f;
        if (VEC) {
    #pragma unroll 8
        for (int i = lane * N; i < dim; i += 64 * N) {
                float v[N];
                Vec16<DT>::ld(xr + i, v);
#pragma unroll
                for (int j = 0; j < N; ++j) q += (v[j] - mean) * (v[j] - mean);
            }
        } else {
            for (int i = lane; i < dim; i += 64) {
                const float v = Elem<DT>::ld(xr + i) - mean;
                q += v * v;
            }
        }
        q = wave_sum(q);
        rstd = 1.0f / sqrtf(q / (float)dim + eps);
    }
    if (VEC) {
#pragma unroll 8
        for (int i = lane * N; i < dim; i += 64 * N) {
            float v[N], wv[N], bv[N];
            Vec16<DT>::ld(xr + i, v);
            if (wr) Vec16<DT>::ld(wr + i, wv);
            if (br) Vec16<DT>::ld(br + i, bv);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float y = (v[j] - mean) * rstd;
                if (MODE == NORM_MODULATE) {
                    y = (1.0f + wv[j]) * y + bv[j];
                } else {
                    if (wr) y *= wv[j];
                    if (MODE == NORM_LAYER && br) y += bv[j];
                }
                v[j] = y;
            }
            Vec16<DT>::st(orow + i, v);
        }
    } else {
        for (int i = lane; i < dim; i += 64) {
            float y = (Elem<DT>::ld(xr + i) - mean) * rstd;
            if (MODE == NORM_MODULATE) {
                y = (1.0f + Elem<DT>::ld(wr + i)) * y + Elem<DT>::ld(br + i);
            } else {
                if (wr) y *= Elem<DT>::ld(wr + i);
                if (MODE == NORM_LAYER && br) y += Elem<DT>::ld(br + i);
            }
            Elem<DT>::st(orow + i, y);
        }
    }
}

template <int DT, int MODE>
static int launch_rownorm(void* out, const void* x, const void* w, const void* b, int64_t rows, int dim, float eps,
                          int rows_per_batch, hipStream_t s) {
    typedef typename Elem<DT>::T T;
    if (rows == 0 || dim == 0) return 0;
    const bool vec = (dim % Vec16<DT>::N == 0) && aligned16(out) && aligned16(x) && (!w || aligned16(w)) &&
                     (!b || aligned16(b));
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const char* ke = getenv("OMX_NORM_KEEP");      // 0: the three-pass form (A/B)
    const bool keep = !(ke && ke[0] == '0');
    if (vec && dim <= 6 * 64 * Vec16<DT>::N)
        rownorm_kernel<DT, MODE, true, 6><<<grid, block, 0, s>>>((T*)out, (const T*)x, (const T*)w, (const T*)b, rows, dim,
                                                                 eps, rows_per_batch, keep);
    else if (vec)
        rownorm_kernel<DT, MODE, true><<<grid, block, 0, s>>>((T*)out, (const T*)x, (const T*)w, (const T*)b, rows, dim,
                                                              eps, rows_per_batch, keep);
    else
        rownorm_kernel<DT, MODE, false><<<grid, block, 0, s>>>((T*)out, (const T*)x, (const T*)w, (const T*)b, rows,
                                                               dim, eps, rows_per_batch, keep);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // namespace omx

extern "C" {

int omx_rms_norm(void* out, const void* x, const void* weight, int64_t rows, int dim, float eps, omx_dtype dtype,
                 omx_stream stream) {
    OMX_REQUIRE(out && x, "omx_rms_norm: null tensor");
    OMX_REQUIRE(rows >= 0 && dim >= 0, "omx_rms_norm: negative shape");
    OMX_DISPATCH_FLOAT(dtype, "omx_rms_norm",
                       return (omx::launch_rownorm<DT, omx::NORM_RMS>(out, x, weight, nullptr, rows, dim, eps, 1,
                                                                      (hipStream_t)stream)));
    return 0;
}

int omx_layer_norm(void* out, const void* x, const void* weight, const void* bias, int64_t rows, int dim, float eps,
                   omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && x, "omx_layer_norm: null tensor");
    OMX_REQUIRE(rows >= 0 && dim >= 0, "omx_layer_norm: negative shape");
    OMX_DISPATCH_FLOAT(dtype, "omx_layer_norm",
                       return (omx::launch_rownorm<DT, omx::NORM_LAYER>(out, x, weight, bias, rows, dim, eps, 1,
                                                                        (hipStream_t)stream)));
    return 0;
}

int omx_fused_modulate(void* out, const void* x, const void* shift, const void* scale, int B, int S, int H, float eps,
                       omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && x && shift && scale, "omx_fused_modulate: null tensor");
    OMX_REQUIRE(B >= 0 && S >= 0 && H >= 0, "omx_fused_modulate: negative shape");
    OMX_DISPATCH_FLOAT(dtype, "omx_fused_modulate",
                       return (omx::launch_rownorm<DT, omx::NORM_MODULATE>(out, x, scale, shift, (int64_t)B * S, H, eps,
                                                                           S > 0 ? S : 1, (hipStream_t)stream)));
    return 0;
}

}  // extern "C"
