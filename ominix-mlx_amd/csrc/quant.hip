// MLX affine group quantisation on gfx950 (the reference's flagship checkpoint format).
//   reference: mlx_rs::ops::{quantize, dequantize, quantized_matmul, gather_qmm}
//              (mlx-rs/src/ops/quantization.rs:41-153, 226-279) -> mlx_quantize / mlx_dequantize /
//              mlx_quantized_matmul / mlx_gather_qmm (mlx-c ops.h:356-365, 471-484, 793-810);
//              nn::QuantizedLinear::forward (mlx-rs/src/nn/quantized.rs:361-385).
// Format: w [N, K] -> packed u32 [N, K*bits/32] (element j of a row = the `bits`-wide field at bit
// (j*bits) mod 32 of word floor(j*bits/32), LSB first), scales / biases [N, K/group] in the activation
// dtype; w ~= q * scale + bias.  bits 4 or 8, group 32 / 64 / 128.
//
// quantized_matmul (transpose = true: x . dequant(W)^T):
//   * M <= 16 (decode): weight-streaming GEMV that reads the PACKED weights -- a quarter (4-bit) of the
//     bf16 bytes.  One wave per row, each lane owns W words per step; per lane  acc += scale * sum(x_i q_i)
//     + bias * sum(x_i), the per-chunk sum(x_i) being shared by all rows (computed once per block into LDS).
//     grid.y walks the activation rows / expert-selected batch entries (gather_qmm).
//   * M > 16 (prefill): dequantise W once into the workspace, then the bf16 MFMA GEMM (gemm.hip).
#include <hip/hip_fp16.h>
#include "common.hpp"
#include "gemm.hpp"
#include <mutex>
#include <unordered_map>

#include "quant.hpp"
#include "act16.hpp"
#include "launch_timing.hpp"
#include "vec.hpp"
#include "workspace.hpp"

namespace omx {

namespace {

// ---- quantize: one wave per group of 32/64/128 elements (MLX affine_quantize) ----
template <int BITS, int DT = OMX_BFLOAT16>
__global__ __launch_bounds__(256) void quantize_kernel(uint32_t* __restrict__ packed, typename Elem<DT>::T* __restrict__ scales,
                                                       typename Elem<DT>::T* __restrict__ biases, const typename Elem<DT>::T* __restrict__ w,
                                                       int64_t n_groups, int group) {
    constexpr int EPW = 32 / BITS;
    constexpr float n_bins = (float)((1 << BITS) - 1);
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const typename Elem<DT>::T* src = w + g * group;
    const int per_lane = group / 64 > 0 ? group / 64 : 1;   // 128 -> 2, 64 -> 1, 32 -> 1 (upper half idle)
    float v[2] = {0.f, 0.f};
    float mx = -INFINITY, mn = INFINITY;
    for (int i = 0; i < per_lane; ++i) {
        const int e = lane * per_lane + i;
        if (e < group) {
            v[i] = Elem<DT>::ld(src + e);
            mx = fmaxf(mx, v[i]);
            mn = fminf(mn, v[i]);
        }
    }
    mx = wave_max(mx);
    mn = -wave_max(-mn);
    float scale = fmaxf((mx - mn) / n_bins, 1e-7f);
    const bool side = fabsf(mn) > fabsf(mx);
    scale = side ? scale : -scale;
    const float edge = side ? mn : mx;
    const float q0 = rintf(edge / scale);
    const bool at_zero = q0 == 0.f;
    scale = at_zero ? scale : edge / q0;
    const float bias = at_zero ? 0.f : edge;
    if (lane == 0) {
        Elem<DT>::st(scales + g, scale);
        Elem<DT>::st(biases + g, bias);
    }
    // pack: element e goes to word e / EPW at bit (e % EPW) * BITS; the EPW elements of a word sit in
    // EPW / per_lane consecutive lanes
    uint32_t word = 0;
    for (int i = 0; i < per_lane; ++i) {
        const int e = lane * per_lane + i;
        if (e < group) {
            const float q = fminf(fmaxf(rintf((v[i] - bias) / scale), 0.f), n_bins);
            word |= (uint32_t)q << ((e % EPW) * BITS);
        }
    }
    constexpr int kLanesPerWordMax = EPW;   // per_lane == 1
    const int lanes_per_word = EPW / per_lane;
    for (int o = 1; o < kLanesPerWordMax; o <<= 1)
        if (o < lanes_per_word) word |= __shfl_xor(word, o, 64);
    const int e0 = lane * per_lane;
    if (e0 < group && (lane % lanes_per_word) == 0) packed[(g * group + e0) / EPW] = word;
}

// a 16-bit scale / bias pattern as float32: bfloat16, or float16 for a float16 checkpoint
template <bool F16>
__device__ __forceinline__ float scale_to_f32(uint16_t bits) {
    if (F16) return __half2float(__ushort_as_half(bits));
    return bf16_to_f32((bf16_t)bits);
}

template <int BITS>
__global__ __launch_bounds__(256) void dequantize_kernel(bf16_t* __restrict__ out, const uint32_t* __restrict__ packed,
                                                         const bf16_t* __restrict__ scales, const bf16_t* __restrict__ biases,
                                                         int64_t n_words, int group, bool scales_f16 = false, bool out_f16 = false) {
    constexpr int EPW = 32 / BITS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t wd = packed[i];
        const int64_t g = i * EPW / group;
        const float s = scales_f16 ? scale_to_f32<true>(scales[g]) : bf16_to_f32(scales[g]);
        const float b = biases ? (scales_f16 ? scale_to_f32<true>(biases[g]) : bf16_to_f32(biases[g])) : 0.f;
        bf16_t o[EPW];
#pragma unroll
        for (int e = 0; e < EPW; ++e) {
            const float v = (float)((wd >> (e * BITS)) & ((1u << BITS) - 1u)) * s + b;
            o[e] = out_f16 ? (bf16_t)__half_as_ushort(__float2half(v)) : f32_to_bf16(v);
        }
        if (EPW == 8) *reinterpret_cast<u32x4*>(out + i * 8) = *reinterpret_cast<const u32x4*>(o);
        else *reinterpret_cast<u32x2*>(out + i * 4) = *reinterpret_cast<const u32x2*>(o);
    }
}

// any width that divides 32 and any float dtype (the result has the scales' dtype: ops/quantization.rs:118-153); one element per store
template <int BITS, int DT>
__global__ __launch_bounds__(256) void dequantize_any_kernel(typename Elem<DT>::T* __restrict__ out, const uint32_t* __restrict__ packed,
                                                             const typename Elem<DT>::T* __restrict__ scales,
                                                             const typename Elem<DT>::T* __restrict__ biases, int64_t n_words, int group) {
    constexpr int EPW = 32 / BITS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t wd = packed[i];
        const int64_t g = i * EPW / group;
        const float sc = Elem<DT>::ld(scales + g), b = biases ? Elem<DT>::ld(biases + g) : 0.f;
#pragma unroll
        for (int e = 0; e < EPW; ++e) Elem<DT>::st(out + i * EPW + e, (float)((wd >> (e * BITS)) & ((1u << BITS) - 1u)) * sc + b);
    }
}

__device__ __forceinline__ uint64_t qargmax_key(float v, uint32_t idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    if (v != v) u = 0;
    return ((uint64_t)u << 32) | (uint32_t)(~idx);
}

// W = u32 words per lane per step; a lane's W*EPW elements lie inside one group.
// PRO / EPI as in gemv.hip (same arithmetic and rounding points): RMSNorm prologue; store, residual add, SwiGLU
// over (gate, up) row pairs, logits + greedy-argmax partial.
// SB: scales and biases come interleaved from QMat::sb (one load per row and step instead of two)
template <int BITS, int W, int PRO, int EPI, int RB, bool SB = false, bool F16S = false>
__global__ __launch_bounds__(256) void qgemv_kernel(const QGemvArgs a) {
    typedef Act16<F16S> A16;                                // activations / outputs: bfloat16, or float16 for a float16 checkpoint (F16S)
    constexpr int EPW = 32 / BITS, EPL = W * EPW;          // elements per lane per step
    constexpr int LR = (EPI == EPI_SWIGLU) ? 2 : 1;         // physical rows per logical row
    constexpr int NR = RB * LR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                       // [K]
    float* xsum = reinterpret_cast<float*>(smem + (size_t)a.K * 2);     // [K / EPL]
    float* red = xsum + a.K / EPL;                                      // [8]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int by = blockIdx.y;
    const bf16_t* xg = a.x + (size_t)(by / a.x_div) * a.K;
    size_t e = a.w_sel ? a.w_sel[by] : 0;
    if (a.w_sel_n > 0) {      // expert parallel: a slot routed to another rank's expert (block-uniform: before any barrier)
        if (e < (size_t)a.w_sel_lo || e >= (size_t)(a.w_sel_lo + a.w_sel_n)) return;
        e -= (size_t)a.w_sel_lo;
    }
    bf16_t* out = a.out + (size_t)by * a.N;

    const int steps = a.K / (64 * EPL);
    const int words_per_row = a.K / EPW, groups_per_row = a.K / a.group;
    const int row_begin = (blockIdx.x * 4 + wave) * a.rows_per_wave;
    const int row_end = min(row_begin + a.rows_per_wave, a.N);
    uint64_t best = 0;
    // physical row pr of the batch: which member matrix, which row inside it
    auto locate = [&](int pr, const uint32_t*& wq, const bf16_t*& sc, const bf16_t*& bi, const uint32_t*& sbp) {
        int mi, row;
        if (EPI == EPI_SWIGLU) {
            mi = pr & 1;
            row = min(pr >> 1, a.N - 1);
        } else {
            row = min(pr, a.N - 1);
            mi = 0;
            if (row >= a.m[0].n) { row -= a.m[0].n; mi = 1; if (row >= a.m[1].n) { row -= a.m[1].n; mi = 2; } }
        }
        const QMat& M = a.m[mi];
        wq = M.w + e * a.w_estride + (size_t)row * words_per_row;
        sc = M.scales + e * a.s_estride + (size_t)row * groups_per_row;
        bi = M.biases ? M.biases + e * a.s_estride + (size_t)row * groups_per_row : nullptr;
        sbp = SB ? M.sb + e * a.s_estride + (size_t)row * groups_per_row : nullptr;
    };
    // A "unit" = one K step of one batch of RB logical rows (NR physical rows): NR x W words + NR scales + NR biases per
    // lane.  Units of consecutive steps / batches are streamed through TWO register sets: the loads of unit f+1 are in
    // flight while unit f is multiplied (the weights are read once, straight to registers, non-temporal).
    struct Unit {
        uint32_t wd[NR][W];
        bf16_t sc[NR], bi[NR];
        uint32_t sbv[NR];
    };
    const int nbatch = (row_end - row_begin + RB - 1) / RB;
    const int nunits = nbatch > 0 ? nbatch * steps : 0;
    const uint32_t* rw[NR];      // row pointers of the batch being ISSUED (issue order is monotonic in f)
    const bf16_t* rs[NR];
    const bf16_t* rb[NR];
    const uint32_t* rsb[NR];
    auto issue = [&](Unit& u, int f) {
        const int st = f % steps;
        if (st == 0) {
            const int r0 = row_begin + (f / steps) * RB;
#pragma unroll
            for (int r = 0; r < NR; ++r) locate(EPI == EPI_SWIGLU ? 2 * (r0 + r / 2) + (r & 1) : r0 + r, rw[r], rs[r], rb[r], rsb[r]);
        }
        const int chunk = st * 64 + lane;
        const int g = chunk * EPL / a.group;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const uint32_t* p = rw[r] + (size_t)chunk * W;
            if (W == 4) {
                const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
                u.wd[r][0] = v[0]; u.wd[r][W > 1 ? 1 : 0] = v[1]; u.wd[r][W > 2 ? 2 : 0] = v[2]; u.wd[r][W > 3 ? 3 : 0] = v[3];
            } else if (W == 2) {
                const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
                u.wd[r][0] = v[0]; u.wd[r][W > 1 ? 1 : 0] = v[1];
            } else {
                u.wd[r][0] = __builtin_nontemporal_load(p);
            }
            if (SB) {
                u.sbv[r] = rsb[r][g];
            } else {
                u.sc[r] = rs[r][g];
                u.bi[r] = rb[r] ? rb[r][g] : (bf16_t)0;
            }
        }
    };
    float acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.f;
    auto consume = [&](const Unit& u, int f) {
        const int r0 = row_begin + (f / steps) * RB, st = f % steps;
        const int chunk = st * 64 + lane;
        uint32_t xp[EPL / 2];   // the lane's activations, still packed bf16 pairs
#pragma unroll
        for (int j = 0; j < EPL / 8; ++j) {
            const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + (size_t)chunk * EPL + j * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) xp[j * 4 + q] = xv[q];
        }
        const float xsm = xsum[chunk];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float d = 0.f;
            const float scl = SB ? (F16S ? scale_to_f32<true>((uint16_t)u.sbv[r]) : bf16lo(u.sbv[r])) : scale_to_f32<F16S>(u.sc[r]);
            float bia = SB ? (F16S ? scale_to_f32<true>((uint16_t)(u.sbv[r] >> 16)) : bf16hi(u.sbv[r])) : scale_to_f32<F16S>(u.bi[r]);
#pragma unroll
            for (int wi = 0; wi < W; ++wi) {
                const uint32_t wdw = u.wd[r][wi];
                if (BITS == 4) {
                    // nibbles -> bf16 pairs by bit assembly: 0x4300 | q is the bf16 value 128 + q, so each v_dot2c
                    // accumulates x . (128 + q); the 128 * sum(x) excess is folded into the bias term below
                    // One v_perm per pair: the 0x43 exponent byte comes from the second source, the two nibble bytes from
                    // the same masked word -- so a pair is (q0, q2), (q4, q6) of the even nibbles or (q1, q3), (q5, q7) of the
                    // odd ones, and the activations were stored in LDS in that order (put() below).
                    const uint32_t lo = wdw & 0x0F0F0F0Fu, hi = (wdw >> 4) & 0x0F0F0F0Fu;
                    const uint32_t c43 = A16::kMagicBytes;   // bf16: 0x4300 | q = 128 + q; float16: 0x6400 | q = 1024 + q
                    const uint32_t q0 = __builtin_amdgcn_perm(c43, lo, 0x04010400u), q1 = __builtin_amdgcn_perm(c43, lo, 0x04030402u);
                    const uint32_t q2 = __builtin_amdgcn_perm(c43, hi, 0x04010400u), q3 = __builtin_amdgcn_perm(c43, hi, 0x04030402u);
                    d = A16::dot2(xp[wi * 4 + 0], A16::unmagic(q0), d);
                    d = A16::dot2(xp[wi * 4 + 1], A16::unmagic(q1), d);
                    d = A16::dot2(xp[wi * 4 + 2], A16::unmagic(q2), d);
                    d = A16::dot2(xp[wi * 4 + 3], A16::unmagic(q3), d);
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t xw = xp[wi * 2 + (b >> 1)];
                        d = fmaf((b & 1) ? A16::hi(xw) : A16::lo(xw), (float)((wdw >> (8 * b)) & 0xFFu), d);
                    }
                }
            }
            if (BITS == 4) bia = fmaf(-A16::kMagic, scl, bia);
            acc[r] = fmaf(scl, d, acc[r]);
            acc[r] = fmaf(bia, xsm, acc[r]);
        }
        if (st == steps - 1) {   // the batch's rows are complete: reduce, epilogue, restart the accumulators
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = wave_sum(acc[r]);
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const int row = r0 + r;
                    if (row >= row_end) break;
                    const float v0 = acc[LR * r], v1 = acc[LR * r + (LR - 1)];
                    if (EPI == EPI_STORE) {
                        out[row] = A16::bits(v0);
                    } else if (EPI == EPI_F32) {
                        a.out_f32[(size_t)by * a.N + row] = v0;
                    } else if (EPI == EPI_RESIDUAL) {
                        out[row] = A16::bits(A16::val(a.resid[row]) + A16::rnd(v0));
                    } else if (EPI == EPI_SWIGLU) {
                        // nn::silu(gate) * up, every primitive's result held in bf16 (qwen3-mlx/src/model.rs:264-265)
                        const float g = A16::rnd(v0), uu = A16::rnd(v1);
                        if (a.swiglu_single_round) {
                            out[row] = A16::bits(g / (1.0f + expf(-g)) * uu);   // mlx_rs_core::fused_swiglu(up, gate)
                        } else {
                            const float sg = A16::rnd(1.0f / (1.0f + expf(-g)));
                            out[row] = A16::bits(A16::rnd(g * sg) * uu);
                        }
                    } else if (EPI == EPI_ARGMAX) {
                        const bf16_t lb = A16::bits(v0);
                        out[row] = lb;
                        const uint64_t key = qargmax_key(A16::val(lb), (uint32_t)(row + a.row_offset));
                        best = key > best ? key : best;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = 0.f;
        }
    };
    // the first two units go out before the activation is even loaded: they depend on the weights only
    Unit uA, uB;
    if (nunits > 0) issue(uA, 0);
    if (nunits > 1) issue(uB, 1);

    // ---- prologue: x -> LDS as bf16 (RMS-normalised on the way in) and, in the same pass, the per-chunk sums
    //      sum(x_i) that every row's bias term shares (EPL elements = EPL/8 consecutive threads, reduced by DPP) ----
    static_assert(EPL >= 8, "a lane chunk must cover at least one 16-byte activation vector");
    auto put = [&](int i, const u32x4 o) {
        if (BITS == 4) {
            // 8 consecutive activations are kept as (x0,x2) (x4,x6) (x1,x3) (x5,x7): the pairing of the one-perm nibble unpack
            u32x4 t;
            t[0] = __builtin_amdgcn_perm(o[1], o[0], 0x05040100u); t[1] = __builtin_amdgcn_perm(o[3], o[2], 0x05040100u);
            t[2] = __builtin_amdgcn_perm(o[1], o[0], 0x07060302u); t[3] = __builtin_amdgcn_perm(o[3], o[2], 0x07060302u);
            *reinterpret_cast<u32x4*>(xs + i) = t;
        } else {
            *reinterpret_cast<u32x4*>(xs + i) = o;
        }
        float sv = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) sv += A16::lo(o[q]) + A16::hi(o[q]);
        if (EPL >= 16) sv += dpp_f<kDppXor1>(sv);
        if (EPL >= 32) sv += dpp_f<kDppXor2>(sv);
        if (((i >> 3) & (EPL / 8 - 1)) == 0) xsum[i / EPL] = sv;
    };
    if (PRO == PRO_RMSNORM && a.K <= 4096) {
        // the hidden-sized prologues (q/k/v, gate/up, lm_head: K <= 4096 = two vectors per thread): the row and the norm weights stay in
        // registers between the two passes -- one global round trip instead of two in a launch that is a chain of them.  Same sums.
        u32x4 raw[2], nwv[2];
        float ss = 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = threadIdx.x * 8 + it * 2048;
            if (i < a.K) {
                raw[it] = *reinterpret_cast<const u32x4*>(xg + i);
                nwv[it] = *reinterpret_cast<const u32x4*>(a.norm_w + i);
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (threadIdx.x * 8 + it * 2048 < a.K) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ss = fmaf(A16::lo(raw[it][q]), A16::lo(raw[it][q]), ss);
                    ss = fmaf(A16::hi(raw[it][q]), A16::hi(raw[it][q]), ss);
                }
            }
        }
        ss = block_sum<4>(ss, red);
        const float rstd = 1.0f / sqrtf(ss / (float)a.K + a.eps);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = threadIdx.x * 8 + it * 2048;
            if (i < a.K) {
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q] = A16::pack(A16::lo(raw[it][q]) * rstd * A16::lo(nwv[it][q]), A16::hi(raw[it][q]) * rstd * A16::hi(nwv[it][q]));
                put(i, o);
            }
        }
    } else if (PRO == PRO_RMSNORM) {
        float ss = 0.f;
        for (int i = threadIdx.x * 8; i < a.K; i += 256 * 8) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(xg + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ss = fmaf(A16::lo(raw[q]), A16::lo(raw[q]), ss);
                ss = fmaf(A16::hi(raw[q]), A16::hi(raw[q]), ss);
            }
        }
        ss = block_sum<4>(ss, red);
        const float rstd = 1.0f / sqrtf(ss / (float)a.K + a.eps);
        for (int i = threadIdx.x * 8; i < a.K; i += 256 * 8) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(xg + i);
            const u32x4 nw = *reinterpret_cast<const u32x4*>(a.norm_w + i);
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                o[q] = A16::pack(A16::lo(raw[q]) * rstd * A16::lo(nw[q]), A16::hi(raw[q]) * rstd * A16::hi(nw[q]));
            put(i, o);
        }
    } else if (a.K <= 8 * 2048 && !a.rolled_stage) {
        // all of the row's vectors of this thread in flight at once (the rolled loop below waits for each 16-byte load before it
        // issues the next: six dependent L2 round trips in the down projection's prologue, K = 12288)
        u32x4 v[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int i = threadIdx.x * 8 + it * 2048;
            if (i < a.K) v[it] = *reinterpret_cast<const u32x4*>(xg + i);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int i = threadIdx.x * 8 + it * 2048;
            if (i < a.K) put(i, v[it]);
        }
    } else {
        for (int i = threadIdx.x * 8; i < a.K; i += 256 * 8) put(i, *reinterpret_cast<const u32x4*>(xg + i));
    }
    __syncthreads();

    for (int f = 0; f < nunits; f += 2) {
        if (f > 0 && f + 1 < nunits) issue(uB, f + 1);
        consume(uA, f);
        if (f + 1 >= nunits) break;
        if (f + 2 < nunits) issue(uA, f + 2);
        consume(uB, f + 1);
    }
    if (EPI == EPI_ARGMAX) {
        uint64_t* bred = reinterpret_cast<uint64_t*>(red);
        __syncthreads();
        if (lane == 0) bred[wave] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t b = bred[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) b = bred[w] > b ? bred[w] : b;
            a.argmax_slot[blockIdx.x] = b;
        }
    }
}

template <int BITS, int W>
int launch_qgemv_w(const QGemvArgs& a, int pro, int epi, hipStream_t s) {
    constexpr int EPW = 32 / BITS;
    const int groups = (a.N + a.rows_per_wave - 1) / a.rows_per_wave;
    const dim3 grid((groups + 3) / 4, a.n_batch > 1 ? a.n_batch : 1), block(256);
    const size_t shmem = (size_t)a.K * 2 + (size_t)(a.K / (W * EPW)) * 4 + 64;
    // RB = logical rows per unit: 4 for long matrices, 2 when the matrix is small enough that wave count matters more
    // (rows_per_wave == RB there: one batch per wave, twice the waves) and for SwiGLU row pairs
    // interleaved scale/bias words: the engine's K % 2048 == 0 matrices (every member of the stack must carry them)
    bool sb = W == 4;
    for (int i = 0; i < 3 && sb; ++i)
        if (a.m[i].w && !a.m[i].sb) sb = false;
#define OMX_QGEMV_LAUNCH(P, E, SBF, F16)                                                       \
    {                                                                                         \
        if (E == EPI_SWIGLU || a.rows_per_wave == 2) OMX_LAUNCH((qgemv_kernel<BITS, W, P, E, 2, SBF, F16>), grid, block, shmem, s, a); \
        else OMX_LAUNCH((qgemv_kernel<BITS, W, P, E, 4, SBF, F16>), grid, block, shmem, s, a);          \
        OMX_LAUNCH_CHECK();                                                                   \
        return 0;                                                                             \
    }
#define OMX_QGEMV_CASE(P, E)                                                                  \
    if (pro == P && epi == E) {                                                               \
        if constexpr (W == 4) {                                                               \
            if (sb) {                                                                         \
                if (a.scales_f16) OMX_QGEMV_LAUNCH(P, E, true, true)                          \
                OMX_QGEMV_LAUNCH(P, E, true, false)                                           \
            }                                                                                 \
        }                                                                                     \
        if (a.scales_f16) OMX_QGEMV_LAUNCH(P, E, false, true)                                 \
        OMX_QGEMV_LAUNCH(P, E, false, false)                                                  \
    }
    OMX_QGEMV_CASE(PRO_NONE, EPI_STORE)
    OMX_QGEMV_CASE(PRO_RMSNORM, EPI_STORE)
    OMX_QGEMV_CASE(PRO_NONE, EPI_RESIDUAL)
    OMX_QGEMV_CASE(PRO_RMSNORM, EPI_SWIGLU)
    OMX_QGEMV_CASE(PRO_NONE, EPI_SWIGLU)
    OMX_QGEMV_CASE(PRO_RMSNORM, EPI_ARGMAX)
    OMX_QGEMV_CASE(PRO_NONE, EPI_F32)
#undef OMX_QGEMV_CASE
#undef OMX_QGEMV_LAUNCH
    return set_error("quantized gemv: unsupported prologue/epilogue combination %d/%d", pro, epi);
}

template <int BITS>
int launch_qgemv_bits(const QGemvArgs& a_in, int pro, int epi, hipStream_t s) {
    QGemvArgs a = a_in;
    constexpr int EPW = 32 / BITS;
    int W = 4;
    while (W * EPW > 8 && (a.K % (64 * W * EPW) != 0 || W * EPW > a.group)) W >>= 1;
    OMX_REQUIRE(a.K % (64 * W * EPW) == 0 && W * EPW <= a.group && W * EPW >= 8, "quantized_matmul: K=%d unsupported for %d-bit group %d (K must be a multiple of %d)",
                a.K, BITS, a.group, 64 * EPW);
    if (a.n_batch < 1) a.n_batch = 1;
    if (a.x_div < 1) a.x_div = 1;
    // long streams for the vocabulary matrix, one batch per wave otherwise; small matrices: two rows per wave
    a.rows_per_wave = a.N >= 65536 ? 16 : (a.N <= 8192 && epi != EPI_SWIGLU) ? 2 : 4;
    if (const char* e = getenv("OMX_QGEMV_RPW_SMALL"))   // tuning knob: rows per wave of the small matrices (2 or 4)
        if (a.N <= 8192 && epi != EPI_SWIGLU && (atoi(e) == 2 || atoi(e) == 4)) a.rows_per_wave = atoi(e);
    if (const char* e = getenv("OMX_QGEMV_ROLLED_STAGE")) a.rolled_stage = e[0] == '1';
    if (const char* e = getenv("OMX_QGEMV_RPW_LONGK"))   // ... of the small matrices with a long row (K > 8192: the down projection)
        if (a.N <= 8192 && a.K > 8192 && epi != EPI_SWIGLU && (atoi(e) == 2 || atoi(e) == 4 || atoi(e) == 8)) a.rows_per_wave = atoi(e);
    if (const char* e = getenv("OMX_QGEMV_RPW_GU"))      // ... of the gate/up pair launch (logical rows: 2, 4, 8)
        if (epi == EPI_SWIGLU && a.N < 65536 && (atoi(e) == 2 || atoi(e) == 4 || atoi(e) == 8)) a.rows_per_wave = atoi(e);
    if (W == 4) return launch_qgemv_w<BITS, 4>(a, pro, epi, s);
    if (W == 2) return launch_qgemv_w<BITS, 2>(a, pro, epi, s);
    if constexpr (BITS == 4) return launch_qgemv_w<BITS, 1>(a, pro, epi, s);
    return set_error("quantized gemv: K=%d too small for %d-bit weights", a.K, BITS);
}

// quantize / dequantize alone take what mlx_rs::ops::quantize takes: 2, 4 or 8 bits on bfloat16 / float16 / float32 (the reference's
// own value test loops [2, 4, 8] on float32: ops/quantization.rs:289-305); the matmul kernels stay 4 / 8 bit (check_format)
int check_format_qdq(const char* who, int K, int group, int bits, int dtype) {
    OMX_REQUIRE(dtype == OMX_BFLOAT16 || dtype == OMX_FLOAT16 || dtype == OMX_FLOAT32, "%s: bf16 / f16 / f32 only (got dtype %d)", who, dtype);
    OMX_REQUIRE(bits == 2 || bits == 4 || bits == 8, "%s: bits must be 2, 4 or 8 (got %d; the 3 / 5 / 6-bit MLX packings are not built)", who, bits);
    OMX_REQUIRE(group == 32 || group == 64 || group == 128, "%s: group_size must be 32, 64 or 128 (got %d)", who, group);
    OMX_REQUIRE(K > 0 && K % group == 0, "%s: the last dimension (%d) must be divisible by the group size (%d)", who, K, group);
    return 0;
}

// dtype: OMX_BFLOAT16, or OMX_FLOAT16 where `f16_scales_ok` -- scales / biases of a float16 checkpoint (activations stay bf16)
int check_format(const char* who, int K, int group, int bits, int dtype, bool f16_scales_ok = false) {
    OMX_REQUIRE(dtype == OMX_BFLOAT16 || (f16_scales_ok && dtype == OMX_FLOAT16), "%s: bf16 activations / scales only (got dtype %d)", who, dtype);
    OMX_REQUIRE(bits == 4 || bits == 8, "%s: bits must be 4 or 8 (got %d)", who, bits);
    OMX_REQUIRE(group == 32 || group == 64 || group == 128, "%s: group_size must be 32, 64 or 128 (got %d)", who, group);
    OMX_REQUIRE(K > 0 && K % group == 0, "%s: the last dimension (%d) must be divisible by the group size (%d)", who, K, group);
    return 0;
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void quant_interleave_kernel(uint32_t* __restrict__ sb, const bf16_t* __restrict__ scales,
                                                               const bf16_t* __restrict__ biases, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        sb[i] = (uint32_t)scales[i] | ((uint32_t)(biases ? biases[i] : (bf16_t)0) << 16);
}
}  // namespace

int launch_quant_interleave(uint32_t* sb, const bf16_t* scales, const bf16_t* biases, size_t n_groups, hipStream_t s) {
    OMX_REQUIRE(sb && scales, "quant interleave: null tensor");
    if (n_groups == 0) return 0;
    quant_interleave_kernel<<<(unsigned)std::min<size_t>((n_groups + 255) / 256, 4096), 256, 0, s>>>(sb, scales, biases, n_groups);
    OMX_LAUNCH_CHECK();
    return 0;
}

namespace {
std::mutex g_sb_mu;
std::unordered_map<const void*, const uint32_t*> g_sb_of_scales;
}  // namespace
void quant_register_sb(const bf16_t* scales, const uint32_t* sb) {
    std::lock_guard<std::mutex> lk(g_sb_mu);
    g_sb_of_scales[scales] = sb;
}
void quant_unregister_sb(const bf16_t* scales) {
    std::lock_guard<std::mutex> lk(g_sb_mu);
    g_sb_of_scales.erase(scales);
}
const uint32_t* quant_find_sb(const bf16_t* scales) {
    std::lock_guard<std::mutex> lk(g_sb_mu);
    auto it = g_sb_of_scales.find(scales);
    return it == g_sb_of_scales.end() ? nullptr : it->second;
}

int qgemv_grid(int N) {
    const int rpw = N >= 65536 ? 16 : N <= 8192 ? 2 : 4;
    return ((N + rpw - 1) / rpw + 3) / 4;
}

int launch_qgemv(const QGemvArgs& a, int bits, int pro, int epi, hipStream_t s) {
    OMX_REQUIRE(bits == 4 || bits == 8, "quantized gemv: bits must be 4 or 8 (got %d)", bits);
    if (bits == 4) {
        const int r = launch_qgemv4m(a, pro, epi, s);
        if (r >= 0) return r;
    }
    return bits == 4 ? launch_qgemv_bits<4>(a, pro, epi, s) : launch_qgemv_bits<8>(a, pro, epi, s);
}

}  // namespace omx

using namespace omx;

extern "C" int omx_quantize(void* packed, void* scales, void* biases, const void* w, int64_t rows, int cols, int group_size,
                            int bits, omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(packed && scales && biases && w, "omx_quantize: null tensor");
    if (check_format_qdq("omx_quantize", cols, group_size, bits, dtype)) return 1;
    const int64_t n_groups = rows * (cols / group_size);
    if (n_groups == 0) return 0;
    const unsigned blocks = (unsigned)((n_groups + 3) / 4);
#define OMX_Q_CASE(B, D)                                                                                                     \
    if (bits == B && dtype == D) {                                                                                           \
        typedef Elem<D>::T T;                                                                                                \
        quantize_kernel<B, D><<<blocks, 256, 0, (hipStream_t)stream>>>((uint32_t*)packed, (T*)scales, (T*)biases, (const T*)w, n_groups, group_size); \
    }
    OMX_Q_CASE(2, OMX_BFLOAT16) OMX_Q_CASE(4, OMX_BFLOAT16) OMX_Q_CASE(8, OMX_BFLOAT16)
    OMX_Q_CASE(2, OMX_FLOAT16) OMX_Q_CASE(4, OMX_FLOAT16) OMX_Q_CASE(8, OMX_FLOAT16)
    OMX_Q_CASE(2, OMX_FLOAT32) OMX_Q_CASE(4, OMX_FLOAT32) OMX_Q_CASE(8, OMX_FLOAT32)
#undef OMX_Q_CASE
    OMX_LAUNCH_CHECK();
    return 0;
}

static int launch_dequantize_any(void* out, const uint32_t* packed, const void* scales, const void* biases, int64_t rows, int cols,
                                 int group_size, int bits, bool scales_f16, bool out_f16, hipStream_t s) {
    OMX_REQUIRE(out && packed && scales, "omx_dequantize: null tensor");
    const int64_t n_words = rows * cols * bits / 32;
    if (n_words == 0) return 0;
    const unsigned blocks = (unsigned)((n_words + 255) / 256 < 16384 ? (n_words + 255) / 256 : 16384);
    if (bits == 4) dequantize_kernel<4><<<blocks, 256, 0, s>>>((bf16_t*)out, packed, (const bf16_t*)scales, (const bf16_t*)biases, n_words, group_size, scales_f16, out_f16);
    else dequantize_kernel<8><<<blocks, 256, 0, s>>>((bf16_t*)out, packed, (const bf16_t*)scales, (const bf16_t*)biases, n_words, group_size, scales_f16, out_f16);
    OMX_LAUNCH_CHECK();
    return 0;
}
int omx::launch_dequantize_bf16(bf16_t* out, const uint32_t* packed, const void* scales, const void* biases, int64_t rows, int cols,
                                int group_size, int bits, bool scales_f16, hipStream_t s, bool out_f16) {
    return launch_dequantize_any(out, packed, scales, biases, rows, cols, group_size, bits, scales_f16, out_f16, s);
}

/* mlx_rs::ops::dequantize (ops/quantization.rs:118-153): the result has the dtype of scales / biases -- OMX_BFLOAT16, or OMX_FLOAT16 for
 * a float16 checkpoint's triplets (each element one fma in float32 from the exact scale / bias, one rounding) */
extern "C" int omx_dequantize(void* out, const void* packed, const void* scales, const void* biases, int64_t rows, int cols,
                              int group_size, int bits, omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && packed && scales, "omx_dequantize: null tensor");
    if (check_format_qdq("omx_dequantize", cols, group_size, bits, dtype)) return 1;
    if (bits != 2 && dtype != OMX_FLOAT32) {   // the 16-bit forms the matmul paths share (vector stores)
        const bool f16 = dtype == OMX_FLOAT16;
        return launch_dequantize_any(out, (const uint32_t*)packed, scales, biases, rows, cols, group_size, bits, f16, f16, (hipStream_t)stream);
    }
    const int64_t n_words = rows * cols * bits / 32;
    if (n_words == 0) return 0;
    const unsigned blocks = (unsigned)((n_words + 255) / 256 < 16384 ? (n_words + 255) / 256 : 16384);
#define OMX_DQ_CASE(B, D)                                                                                                    \
    if (bits == B && dtype == D) {                                                                                           \
        typedef Elem<D>::T T;                                                                                                \
        dequantize_any_kernel<B, D><<<blocks, 256, 0, (hipStream_t)stream>>>((T*)out, (const uint32_t*)packed, (const T*)scales, (const T*)biases, n_words, group_size); \
    }
    OMX_DQ_CASE(2, OMX_BFLOAT16) OMX_DQ_CASE(2, OMX_FLOAT16) OMX_DQ_CASE(2, OMX_FLOAT32) OMX_DQ_CASE(4, OMX_FLOAT32) OMX_DQ_CASE(8, OMX_FLOAT32)
#undef OMX_DQ_CASE
    OMX_LAUNCH_CHECK();
    return 0;
}

/* out [M, N] = x [M, K] . dequant(W [N, K])^T   (nn::QuantizedLinear::forward, quantized.rs:366-375) */
extern "C" int omx_quantized_matmul(void* out, const void* x, const void* packed, const void* scales, const void* biases, int M,
                                    int N, int K, int group_size, int bits, omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && x && packed && scales, "omx_quantized_matmul: null tensor");
    // dtype = the dtype of x, out, scales and biases alike: bfloat16, or float16 (a float16 MLX checkpoint runs in float16 end to end)
    if (check_format("omx_quantized_matmul", K, group_size, bits, dtype, true)) return 1;
    OMX_REQUIRE(M >= 0 && N > 0, "omx_quantized_matmul: bad shape");
    if (M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == OMX_FLOAT16 && !(M <= 16 && K % 512 == 0)) {
        // many float16 rows: weights dequantised to float16 (one fma + one rounding per element, like MLX's qmm tile), both operands widened
        // to float32 and multiplied on the f32 matrix cores (exact products of float16 values, f32 accumulation), one rounding to float16
        void* ws = nullptr;
        const size_t nw = (size_t)N * K, nx = (size_t)M * K, no = (size_t)M * N;
        if (get_workspace(&ws, nw * 2 + (nw + nx + no) * 4 + 1024)) return 1;
        char* p = (char*)ws;
        f16_t* w16 = (f16_t*)p; p += (nw * 2 + 255) & ~(size_t)255;
        float* w32 = (float*)p; p += nw * 4;
        float* x32 = (float*)p; p += nx * 4;
        float* o32 = (float*)p;
        if (launch_dequantize_any(w16, (const uint32_t*)packed, scales, biases, N, K, group_size, bits, true, true, s)) return 1;
        if (omx_cast(w32, OMX_FLOAT32, w16, OMX_FLOAT16, (int64_t)nw, stream) || omx_cast(x32, OMX_FLOAT32, x, OMX_FLOAT16, (int64_t)nx, stream)) return 1;
        GemmF32 g = {};
        g.a = x32; g.b = w32; g.out = o32; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.batch = 1; g.alpha = 1.0f;
        if (launch_gemm_f32(g, s)) return 1;
        return omx_cast(out, OMX_FLOAT16, o32, OMX_FLOAT32, (int64_t)no, stream);
    }
    if (M <= 16 && K % 512 == 0) {
        QGemvArgs a = {};
        a.m[0] = QMat{(const uint32_t*)packed, (const bf16_t*)scales, (const bf16_t*)biases, N};
        a.x = (const bf16_t*)x; a.out = (bf16_t*)out; a.N = N; a.K = K; a.group = group_size;
        a.n_batch = M; a.x_div = 1; a.scales_f16 = dtype == OMX_FLOAT16;
        return launch_qgemv(a, bits, PRO_NONE, EPI_STORE, s);
    }
    void* ws = nullptr;
    if (get_workspace(&ws, (size_t)N * K * 2)) return 1;
    if (launch_dequantize_bf16((bf16_t*)ws, (const uint32_t*)packed, scales, biases, N, K, group_size, bits, dtype == OMX_FLOAT16, s)) return 1;
    return launch_gemm_bf16((bf16_t*)out, (const bf16_t*)x, (const bf16_t*)ws, nullptr, M, N, K, s);
}

/* out [n, N] = x [n / x_div, K] . dequant(W[rhs_indices[i]])^T : SwitchLinear on expert-stacked quantized weights
 * (mixtral-mlx/src/model.rs:195-201 -> gather_qmm, ops/quantization.rs:226-279); packed [E, N, K*bits/32] */
extern "C" int omx_gather_qmm(void* out, const void* x, const void* packed, const void* scales, const void* biases,
                              const uint32_t* rhs_indices, int n_rows, int x_div, int N, int K, int n_experts, int group_size,
                              int bits, omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && x && packed && scales && rhs_indices, "omx_gather_qmm: null tensor");
    if (check_format("omx_gather_qmm", K, group_size, bits, dtype, true)) return 1;
    OMX_REQUIRE(n_rows >= 0 && x_div >= 1 && N > 0 && n_experts >= 1, "omx_gather_qmm: bad shape");
    OMX_REQUIRE(K % 512 == 0, "omx_gather_qmm: K=%d must be a multiple of 512", K);
    if (n_rows == 0) return 0;
    QGemvArgs a = {};
    a.m[0] = QMat{(const uint32_t*)packed, (const bf16_t*)scales, (const bf16_t*)biases, N};
    a.x = (const bf16_t*)x; a.out = (bf16_t*)out; a.N = N; a.K = K; a.group = group_size;
    a.n_batch = n_rows; a.x_div = x_div; a.w_sel = rhs_indices; a.scales_f16 = dtype == OMX_FLOAT16;
    a.w_estride = (size_t)N * K * bits / 32; a.s_estride = (size_t)N * (K / group_size);
    return launch_qgemv(a, bits, PRO_NONE, EPI_STORE, (hipStream_t)stream);
}
