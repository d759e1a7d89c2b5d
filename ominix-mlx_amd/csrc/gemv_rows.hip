// A Linear over a HANDFUL of activation rows (M <= 8) as an HBM stream: out[t, n] = sum_k x[t, k] W[n, k] (+ bias) (relu) (+ resid / gate).
// The matrix-core GEMMs spend such calls on 64-row tiles that are 90 % padding (a 5-row verify pass of speculative decoding ran the
// Qwen3-8B projections at 1.0-3.6 TB/s); here every weight row is read ONCE, 16 bytes per lane straight to registers (non-temporal),
// and multiplied against all M activation rows, which sit in LDS one 4096-element K chunk at a time.
//   block = 4 waves, a wave owns RPW = 4 consecutive output rows; per K chunk it issues all 4 x 8 weight vectors, then for each
//   (vector j, activation row t) reads x once from LDS and feeds the 4 rows' accumulators (four v_dot2c_f32_bf16 per 16 bytes, f32), DPP wave sums at the end, lane 0 applies the GEMM kernels' epilogue (gemm.hip: bias, relu, gated / plain residual with
//   the same rounding points).  Bytes: 2 N K per call whatever M; VALU: M x 4 dot2 per 16-byte load per lane.
#include "gemm.hpp"
#include "launch_timing.hpp"

namespace omx {
namespace {

constexpr int kRowsRPW = 4, kRowsChunk = 4096, kRowsNV = kRowsChunk / 512;   // 8 vectors of 16 B per lane and row per chunk

struct RowsArgs {
    const bf16_t* x; const bf16_t* w; const bf16_t* bias; const bf16_t* resid; const bf16_t* gate; bf16_t* out;
    int M, N, K, relu;
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_rows;
// 8 exact bf16 products accumulated in f32 by four v_dot2c_f32_bf16: no unpacking of either operand (with fma chains on unpacked
// halves the kernel was VALU-bound at T = 8: a 5-row pass ran slower than the matrix-core route).  The pairs are taken with
// shufflevector from the whole vector (attn_step.hip explains why).
__device__ __forceinline__ float dot8_rows(const u32x4 w, const u32x4 xp, float acc) {
    const bf16x8_rows A = __builtin_bit_cast(bf16x8_rows, w), B = __builtin_bit_cast(bf16x8_rows, xp);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A, A, 0, 1), __builtin_shufflevector(B, B, 0, 1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A, A, 2, 3), __builtin_shufflevector(B, B, 2, 3), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A, A, 4, 5), __builtin_shufflevector(B, B, 4, 5), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A, A, 6, 7), __builtin_shufflevector(B, B, 6, 7), acc, false);
    return acc;
}

template <int T>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rows_smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(rows_smem);                       // [T][512] vectors of the current K chunk
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * kRowsRPW;
    float acc[kRowsRPW][T];
#pragma unroll
    for (int r = 0; r < kRowsRPW; ++r)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[r][t] = 0.f;
    const int kvec = a.K / 8;                                             // 16-byte vectors per row
    for (int k0 = 0; k0 < a.K; k0 += kRowsChunk) {
        const int v0 = k0 / 8, nv = min(512, kvec - v0);                  // vectors of this chunk
        // the activation chunk first (small, from L2): its loads are AHEAD of the weights in the wave's in-order return queue
        __syncthreads();                                                   // the previous chunk's reads are done
        for (int i = threadIdx.x; i < T * 512; i += 256) {
            const int t = i >> 9, v = i & 511;
            // (clamped address + select: a predicated LOAD makes hipcc branch and drain the queue per load)
            const u32x4 xv = *(reinterpret_cast<const u32x4*>(a.x + (size_t)min(t, a.M - 1) * a.K) + v0 + min(v, nv - 1));
            xs[i] = (v < nv && t < a.M) ? xv : u32x4{0u, 0u, 0u, 0u};
        }
        // weights in the order they are consumed (vector j of every row before vector j + 1): the multiply starts on the first
        // vectors while the later ones are still in flight
        u32x4 w[kRowsNV][kRowsRPW];
#pragma unroll
        for (int j = 0; j < kRowsNV; ++j)
#pragma unroll
            for (int r = 0; r < kRowsRPW; ++r) {
                const u32x4* p = reinterpret_cast<const u32x4*>(a.w + (size_t)min(row0 + r, a.N - 1) * a.K) + v0;
                w[j][r] = __builtin_nontemporal_load(p + min(j * 64 + lane, nv - 1));     // lanes past the row's end: zeroed x, any w
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kRowsNV; ++j)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const u32x4 xp = xs[t * 512 + j * 64 + lane];
#pragma unroll
                for (int r = 0; r < kRowsRPW; ++r) acc[r][t] = dot8_rows(w[j][r], xp, acc[r][t]);
            }
    }
#pragma unroll
    for (int r = 0; r < kRowsRPW; ++r)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[r][t] = wave_sum(acc[r][t]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < kRowsRPW; ++r) {
            const int col = row0 + r;
            if (col >= a.N) continue;
            const float bv = a.bias ? bf16_to_f32(a.bias[col]) : 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (t >= a.M) continue;
                const size_t o = (size_t)t * a.N + col;
                float v = acc[r][t] + bv;
                if (a.relu) v = fmaxf(v, 0.f);
                if (a.gate) v = bf16_to_f32(a.resid[o]) + v * bf16_to_f32(a.gate[col]);
                else if (a.resid) v = bf16_to_f32(a.resid[o]) + round_bf16(v);
                a.out[o] = f32_to_bf16(v);
            }
        }
    }
}

}  // namespace

bool gemv_rows_supported(int M, int N, int K, const void* x, const void* w) {
    return M >= 1 && M <= 8 && N >= 1 && K >= 8 && K % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15u) == 0;
}

int launch_gemv_rows(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, const bf16_t* resid, const bf16_t* gate, int M, int N,
                     int K, int relu, hipStream_t s) {
    OMX_REQUIRE(gemv_rows_supported(M, N, K, x, w), "gemv_rows: unsupported shape M=%d N=%d K=%d", M, N, K);
    const RowsArgs a = {x, w, bias, resid, gate, out, M, N, K, relu};
    const dim3 grid((N + 4 * kRowsRPW - 1) / (4 * kRowsRPW)), block(256);
#define OMX_ROWS_CASE(TT)                                                                                            \
    {                                                                                                                \
        const size_t shmem = (size_t)TT * 512 * 16;                                                                  \
        if (shmem > 48 * 1024)                                                                                       \
            OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemv_rows_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
        gemv_rows_kernel<TT><<<grid, block, shmem, s>>>(a);                                                          \
        OMX_LAUNCH_CHECK();                                                                                          \
        return 0;                                                                                                    \
    }
    // one instantiation per row count: the staging traffic (M x 8 KB per block and chunk) and the multiply both scale with it
    switch (M) {
        case 1: OMX_ROWS_CASE(1)
        case 2: OMX_ROWS_CASE(2)
        case 3: OMX_ROWS_CASE(3)
        case 4: OMX_ROWS_CASE(4)
        case 5: OMX_ROWS_CASE(5)
        case 6: OMX_ROWS_CASE(6)
        case 7: OMX_ROWS_CASE(7)
        default: OMX_ROWS_CASE(8)
    }
#undef OMX_ROWS_CASE
}

}  // namespace omx
