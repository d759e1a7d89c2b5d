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

constexpr int kRowsChunk = 4096, kRowsNV = kRowsChunk / 512;   // 8 vectors of 16 B per lane and row per chunk

struct RowsArgs {
    const bf16_t* x; const bf16_t* w; const bf16_t* bias; const bf16_t* resid; const bf16_t* gate; bf16_t* out;
    int M, N, K, relu;
    GemmSegs sg;          // SEG launches: up to three plain Linears and a SwiGLU pair of the same input (gemm.hpp)
    int plain_rows;       // sum of the plain segments' cols; N = plain_rows + 2 * sg.half virtual rows
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

// SEG: the virtual row space is the plain segments' rows one after the other (cols % 4 == 0: a wave's four rows stay inside one
// segment), then the SwiGLU pair in groups of four = gate rows 2p, 2p + 1 and up rows 2p, 2p + 1, so that the wave that owns a group
// holds both factors of two activation columns.
// RPW: output rows per wave.  4 where that still gives every CU two blocks or more; 2 for the narrow outputs (N = hidden: 256 blocks
// of 4 would leave one block per CU with nothing to overlap its load -> multiply phases; 5-row verify pass 5.67 -> 5.26 ms together with the unrolled RMSNorm)
// (Blocks of 8 waves for the wide launches -- gate / up as 768 resident blocks instead of 1.5 rounds of 1536, the activation rows staged
//  half as often -- measured 2-3 % SLOWER on the verify pass: 3.91 / 5.43 / 6.87 ms against 3.84 / 5.31 / 6.66 at 2 / 5 / 8 rows.)
template <int T, bool SEG, int kRowsRPW>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const RowsArgs a) {
    static_assert(!SEG || kRowsRPW == 4 || kRowsRPW == 2, "segments are cut on multiples of the wave's rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char rows_smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(rows_smem);                       // [T][512] vectors of the current K chunk
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * kRowsRPW;
    const bf16_t* wrow[kRowsRPW];
    int seg = -1, scol = 0;                                              // SEG: plain segment (3 = the SwiGLU pair) and first column in it
    if constexpr (SEG) {
        const int v = min(row0, a.N - kRowsRPW);                         // waves past the end re-read the last group and store nothing
        if (kRowsRPW == 4 && v >= a.plain_rows) {
            seg = 3; scol = (v - a.plain_rows) / 2;
#pragma unroll
            for (int r = 0; r < kRowsRPW; ++r) wrow[r] = (r < 2 ? a.sg.w_gate : a.sg.w_up) + (size_t)(scol + (r & 1)) * a.K;
        } else {
            int b = 0;
            seg = 0;
            if (a.sg.n_plain > 1 && v >= a.sg.plain[0].cols) { seg = 1; b = a.sg.plain[0].cols; }
            if (a.sg.n_plain > 2 && v >= a.sg.plain[0].cols + a.sg.plain[1].cols) { seg = 2; b = a.sg.plain[0].cols + a.sg.plain[1].cols; }
            scol = v - b;
            const bf16_t* wb = seg == 0 ? a.sg.plain[0].w : seg == 1 ? a.sg.plain[1].w : a.sg.plain[2].w;
#pragma unroll
            for (int r = 0; r < kRowsRPW; ++r) wrow[r] = wb + (size_t)(scol + r) * a.K;
        }
    } else {
#pragma unroll
        for (int r = 0; r < kRowsRPW; ++r) wrow[r] = a.w + (size_t)min(row0 + r, a.N - 1) * a.K;
    }
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
        u32x4 nw[2] = {};                                                  // SEG + pre_norm_w: the norm weights of this thread's two vector columns
        if (SEG && a.sg.pre_norm_w) {
#pragma unroll
            for (int h = 0; h < 2; ++h) nw[h] = *(reinterpret_cast<const u32x4*>(a.sg.pre_norm_w) + min((int)threadIdx.x + 256 * h, nv - 1));
        }
        // weights in the order they are consumed (vector j of every row before vector j + 1): the multiply starts on the first
        // vectors while the later ones are still in flight
        u32x4 w[kRowsNV][kRowsRPW];
#pragma unroll
        for (int j = 0; j < kRowsNV; ++j)
#pragma unroll
            for (int r = 0; r < kRowsRPW; ++r) {
                const u32x4* p = reinterpret_cast<const u32x4*>(wrow[r]) + v0;
                w[j][r] = __builtin_nontemporal_load(p + min(j * 64 + lane, nv - 1));     // lanes past the row's end: zeroed x, any w
            }
        __syncthreads();
        if (SEG && a.sg.pre_norm_w) {
            // RMSNorm of the staged rows (K <= 4096: this chunk IS the row), the arithmetic of rownorm_kernel<bf16, RMS> (norm.hip) to the
            // bit: one wave per row, lane l sums the squares of elements [8 l + 512 k, +8) for k = 0.. in order, DPP wave sum,
            // rstd = 1 / sqrt(s / K + eps), y = bf16((x * rstd) * w)
            float* rs = reinterpret_cast<float*>(rows_smem + (size_t)T * 512 * 16);
            for (int t = wave; t < T; t += 4) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < kRowsNV; ++k) {
                    const u32x4 xv = xs[t * 512 + k * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = bf16lo(xv[e]), hi = bf16hi(xv[e]);
                        s += lo * lo;
                        s += hi * hi;
                    }
                }
                s = wave_sum(s);
                if (lane == 0) rs[t] = 1.0f / sqrtf(s / (float)a.K + a.sg.pre_norm_eps);
            }
            __syncthreads();
            for (int i = threadIdx.x; i < T * 512; i += 256) {
                const int t = i >> 9;
                const u32x4 xv = xs[i], wv = nw[(i >> 8) & 1];
                const float rstd = rs[t];
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y0 = bf16lo(xv[e]) * rstd, y1 = bf16hi(xv[e]) * rstd;
                    y0 *= bf16lo(wv[e]);
                    y1 *= bf16hi(wv[e]);
                    o[e] = pack_bf16(y0, y1);
                }
                xs[i] = o;
            }
            __syncthreads();
        }
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
    if constexpr (SEG) {
        if (lane != 0 || row0 >= a.N) return;
        if (kRowsRPW == 4 && seg == 3) {   // act[t, c] = silu(bf16(x.Wg[c])) * bf16(x.Wu[c]): the two roundings of the 256^2 kernel's SwiGLU epilogue (gemm.hip)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if (t >= a.M) continue;
                    const float gt = round_bf16(acc[c][t]), up = round_bf16(acc[(2 + c) % kRowsRPW][t]);
                    float v;
                    if (a.sg.act_mode == 1) {
                        const float sg = round_bf16(1.0f / (1.0f + expf(-gt)));
                        v = round_bf16(gt * sg) * up;
                    } else {
                        v = gt / (1.0f + expf(-gt)) * up;
                    }
                    a.sg.out_act[(size_t)t * a.sg.ld_act + scol + c] = f32_to_bf16(v);
                }
            return;
        }
        const GemmSeg& S = seg == 0 ? a.sg.plain[0] : seg == 1 ? a.sg.plain[1] : a.sg.plain[2];
#pragma unroll
        for (int r = 0; r < kRowsRPW; ++r) {
            const float bv = S.bias ? bf16_to_f32(S.bias[scol + r]) : 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (t < a.M) S.out[(size_t)t * S.ld + scol + r] = f32_to_bf16(acc[r][t] + bv);
        }
        return;
    }
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

// one instantiation per row count: the staging traffic (M x 8 KB per block and chunk) and the multiply both scale with it
template <bool SEG, int kRowsRPW>
int launch_rows_t(const RowsArgs& a, hipStream_t s) {
    const dim3 grid((a.N + 4 * kRowsRPW - 1) / (4 * kRowsRPW)), block(256);
#define OMX_ROWS_CASE(TT)                                                                                            \
    {                                                                                                                \
        const size_t shmem = (size_t)TT * 512 * 16 + 64;                                                                 \
        if (shmem > 48 * 1024)                                                                                       \
            OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemv_rows_kernel<TT, SEG, kRowsRPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
        gemv_rows_kernel<TT, SEG, kRowsRPW><<<grid, block, shmem, s>>>(a);                                                     \
        OMX_LAUNCH_CHECK();                                                                                          \
        return 0;                                                                                                    \
    }
    switch (a.M) {
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
int launch_rows(const RowsArgs& a, bool seg, hipStream_t s) {
    // two rows per wave while four would give fewer than three blocks per CU (a SwiGLU pair needs its four: 2 gate + 2 up rows)
    const bool narrow = a.N < 768 * 16 && !(seg && a.sg.half > 0);
    // (one row per wave for N = 4096, 4 blocks per CU: -2 % at 5 rows, +11 % at 8 -- every wave re-reads all activation rows from LDS)
    if (seg) return narrow ? launch_rows_t<true, 2>(a, s) : launch_rows_t<true, 4>(a, s);
    return narrow ? launch_rows_t<false, 2>(a, s) : launch_rows_t<false, 4>(a, s);
}

}  // namespace

bool gemv_rows_supported(int M, int N, int K, const void* x, const void* w) {
    return M >= 1 && M <= 8 && N >= 1 && K >= 8 && K % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15u) == 0;
}

int launch_gemv_rows(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, const bf16_t* resid, const bf16_t* gate, int M, int N,
                     int K, int relu, hipStream_t s) {
    OMX_REQUIRE(gemv_rows_supported(M, N, K, x, w), "gemv_rows: unsupported shape M=%d N=%d K=%d", M, N, K);
    RowsArgs a = {};
    a.x = x; a.w = w; a.bias = bias; a.resid = resid; a.gate = gate; a.out = out; a.M = M; a.N = N; a.K = K; a.relu = relu;
    return launch_rows(a, false, s);
}

bool gemv_rows_segmented_supported(int M, int K, const GemmSegs& g) {
    if (M < 1 || M > 8 || K < 8 || K % 8 != 0 || g.n_plain < 0 || g.n_plain > 3 || g.im_C != 0) return false;
    int rows = 0;
    for (int i = 0; i < g.n_plain; ++i) {
        if (g.plain[i].cols < 4 || g.plain[i].cols % 4 != 0) return false;
        rows += g.plain[i].cols;
    }
    if (g.half < 0 || g.half % 2 != 0) return false;
    return rows + 2 * g.half >= 4;
}

int launch_gemv_rows_segmented(const bf16_t* x, int M, int K, const GemmSegs& segs, hipStream_t s) {
    OMX_REQUIRE(gemv_rows_segmented_supported(M, K, segs), "gemv_rows: unsupported segmented shape M=%d K=%d", M, K);
    RowsArgs a = {};
    a.x = x; a.M = M; a.K = K; a.sg = segs;
    uintptr_t align = reinterpret_cast<uintptr_t>(x);
    for (int i = 0; i < segs.n_plain; ++i) {
        OMX_REQUIRE(segs.plain[i].w && segs.plain[i].out, "gemv_rows: null weight / output in segment %d", i);
        align |= reinterpret_cast<uintptr_t>(segs.plain[i].w);
        a.plain_rows += segs.plain[i].cols;
    }
    if (segs.half > 0) {
        OMX_REQUIRE(segs.w_gate && segs.w_up && segs.out_act, "gemv_rows: null gate / up / activation pointer");
        align |= reinterpret_cast<uintptr_t>(segs.w_gate) | reinterpret_cast<uintptr_t>(segs.w_up);
    }
    if (segs.pre_norm_w) {
        OMX_REQUIRE(K <= kRowsChunk, "gemv_rows: the in-launch RMSNorm needs the whole row staged at once (K = %d > %d)", K, kRowsChunk);
        align |= reinterpret_cast<uintptr_t>(segs.pre_norm_w);
    }
    OMX_REQUIRE((align & 15u) == 0, "gemv_rows: operands must be 16-byte aligned");
    a.N = a.plain_rows + 2 * segs.half;
    return launch_rows(a, true, s);
}

}  // namespace omx
