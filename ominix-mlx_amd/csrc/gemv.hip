// See gemv.hpp for the design notes.
#include "moe_route.hpp"
#include "gemv.hpp"
#include "peer.hpp"
#include "launch_timing.hpp"

#include <stdlib.h>

namespace omx {

namespace {

constexpr int kBlock = 256;   // 4 waves
constexpr int kWaves = 4;

__device__ __forceinline__ u32x4 ld_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ const bf16_t* row_ptr(const GemvArgs& a, int row) {
    // wave-uniform: which of the stacked matrices owns this row
    if (row < a.n0) return a.w0 + (size_t)row * a.K;
    row -= a.n0;
    if (row < a.n1) return a.w1 + (size_t)row * a.K;
    row -= a.n1;
    return a.w2 + (size_t)row * a.K;
}

__device__ __forceinline__ float dot8(const u32x4 w, const float (&xf)[8], float acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc = fmaf(bf16lo(w[i]), xf[2 * i], acc);
        acc = fmaf(bf16hi(w[i]), xf[2 * i + 1], acc);
    }
    return acc;
}

__device__ __forceinline__ uint64_t argmax_key(float v, uint32_t idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    if (v != v) u = 0;
    return ((uint64_t)u << 32) | (uint32_t)(~idx);
}

template <int EPI>
__device__ __forceinline__ void epilogue(const GemvArgs& a, int row, float v0, float v1, uint64_t& best) {
    if (EPI == EPI_STORE) {
        reinterpret_cast<bf16_t*>(a.out)[row] = f32_to_bf16(a.out_bias ? v0 + bf16_to_f32(a.out_bias[row]) : v0);
    } else if (EPI == EPI_F32) {
        // (`best` carries the call's tag, read ONCE before the first weight load: a load here would wait behind -- drain -- the
        //  next batch's prefetch; the total is stored by peer_finish_rows)
        if (a.peer) peer_store_word(a.peer, (unsigned)best, row, __float_as_uint(v0));
        else reinterpret_cast<float*>(a.out)[row] = a.out_scale ? round_bf16(round_bf16(v0) * a.out_scale_f) : v0;
    } else if (EPI == EPI_RESIDUAL) {
        reinterpret_cast<bf16_t*>(a.out)[row] = f32_to_bf16(bf16_to_f32(a.resid[row]) + round_bf16(v0));
    } else if (EPI == EPI_SWIGLU) {
        // nn::silu(gate) * up, every primitive's result held in bf16
        // (qwen3-mlx/src/model.rs:264-265; mlx-rs/src/nn/activation.rs:876-880)
        const float g = round_bf16(v0);
        const float u = round_bf16(v1);
        if (a.swiglu_single_round) {
            // mlx_rs_core::fused_swiglu(up, gate) (metal_kernels.rs:11-18): one kernel, one rounding
            reinterpret_cast<bf16_t*>(a.out)[row] = f32_to_bf16(g / (1.0f + expf(-g)) * u);
        } else {
            const float sg = round_bf16(1.0f / (1.0f + expf(-g)));
            reinterpret_cast<bf16_t*>(a.out)[row] = f32_to_bf16(round_bf16(g * sg) * u);
        }
    } else if (EPI == EPI_ARGMAX) {
        const bf16_t lb = f32_to_bf16(v0);
        reinterpret_cast<bf16_t*>(a.out)[row] = lb;
        const uint64_t key = argmax_key(bf16_to_f32(lb), (uint32_t)(row + a.row_offset));
        best = key > best ? key : best;
    }
}

// EPI_F32 + peer, after a wave (or the block's reducing threads) issued the stores of rows [begin, end): lane / thread `idx` of
// `stride` sums the ranks' words of its rows; then the block reports in (the last block of the launch hands the sequence number on)
__device__ __forceinline__ void peer_finish_rows(const GemvArgs& a, unsigned tag, int begin, int end, int idx, int stride) {
    for (int row = begin + idx; row < end; row += stride) reinterpret_cast<float*>(a.out)[row] = peer_poll_sum_f32(a.peer, tag, row);
    __syncthreads();
    if (threadIdx.x == 0) peer_block_done(a.peer, tag, gridDim.x * gridDim.y);
}

// NVW    = 16-byte vectors per lane per row per wave (compile-time, fully unrolled)
// KSPLIT = waves sharing one row (1: a wave owns whole rows; 4: each wave owns a K quarter)
// RB     = logical rows per register batch; LR physical rows per logical row (2 for SwiGLU)
template <int NVW, int KSPLIT, int RB, int PRO, int EPI, bool TAIL = false>
#ifndef OMX_GEMV_MINWAVES
#define OMX_GEMV_MINWAVES 1   // (tuning builds: make VARIANT=w3 VARIANT_FLAGS=-DOMX_GEMV_MINWAVES=3 asks hipcc for <= 168 VGPRs)
#endif
__global__ __launch_bounds__(kBlock, OMX_GEMV_MINWAVES) void gemv_kernel(const GemvArgs a_in) {
    // batched / expert-selected form (MoE decode): blockIdx.y picks the activation row, the output row
    // block and, through a device index array, the expert whose weights are streamed
    GemvArgs a = a_in;
    if (EPI == EPI_F32) {
        if (a_in.out_scale) a.out_scale_f = bf16_to_f32(a_in.out_scale[blockIdx.y]);
    }
    if (a_in.n_batch > 1 || a_in.w_sel) {
        const int by = blockIdx.y;
        a.x = a_in.x + (size_t)(by / a_in.x_div) * a_in.x_bstride;
        a.out = reinterpret_cast<char*>(a_in.out) + (size_t)by * a_in.out_bstride_bytes;
        if (a_in.w_sel) {
            size_t e = a_in.w_sel[by];
            if (a_in.w_sel_n > 0) {   // expert-parallel shard: block-uniform early exit for experts of other ranks
                if (e < (size_t)a_in.w_sel_lo || e >= (size_t)(a_in.w_sel_lo + a_in.w_sel_n)) return;
                e -= (size_t)a_in.w_sel_lo;
            }
            a.w0 = a_in.w0 + e * a_in.w_estride;
            if (a_in.w1) a.w1 = a_in.w1 + e * a_in.w_estride;
        }
    }
    constexpr int LR = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int NV = NVW * KSPLIT;            // vectors per lane for the whole row
    constexpr int NR = RB * LR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem);                            // [NV*64] packed bf16 activation
    float* red = reinterpret_cast<float*>(smem + (size_t)NV * 64 * 16);    // [4] block-reduce scratch
    float* part = red + 8;                                                 // KSPLIT>1: [rows][LR][KSPLIT]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int K = a.K;
    const int rpw = a.rows_per_wave;
    // KSPLIT==1: every wave has its own rows.  KSPLIT==4: the block's waves share the rows.
    const int row_begin = (KSPLIT == 1 ? (blockIdx.x * kWaves + wave) : blockIdx.x) * rpw;
    const int row_end = min(row_begin + rpw, a.N);
    const bool active = row_begin < a.N;
    const int koff = (KSPLIT == 1) ? 0 : wave * NVW * 64;   // first vector of this wave's K slice
    const int kvec = K / 8;                                   // 16-byte vectors per row
    // TAIL (compile time: a runtime select around the loads would make hipcc branch and drain per load): K does not fill the
    // instantiation's vector rows -- lanes beyond K read zero weights and zero activations
    constexpr bool tail = TAIL;

    u32x4 wA[NR][NVW], wB[NR][NVW];
    uint64_t best = 0;   // EPI_ARGMAX: running (orderable logit << 32 | ~row) of this thread; EPI_F32 + peer: the call's tag
    if (EPI == EPI_F32) {
        if (a.peer) best = peer_tag(a.peer);
    }

#define OMX_ISSUE(WB, R0)                                                                          \
    {                                                                                              \
        _Pragma("unroll") for (int r = 0; r < RB; ++r) {                                           \
            const int row = min((R0) + r, a.N - 1); /* clamp: tail re-reads a valid row */         \
            if (EPI == EPI_SWIGLU) {                                                               \
                const u32x4* g = reinterpret_cast<const u32x4*>(a.w0 + (size_t)row * K) + koff;    \
                const u32x4* u = reinterpret_cast<const u32x4*>(a.w1 + (size_t)row * K) + koff;    \
                _Pragma("unroll") for (int j = 0; j < NVW; ++j) {                                  \
                    const bool in = !tail || koff + j * 64 + lane < kvec;                          \
                    WB[LR * r][j] = in ? ld_nt(g + j * 64 + lane) : u32x4{0u, 0u, 0u, 0u};         \
                    WB[LR * r + (LR - 1)][j] = in ? ld_nt(u + j * 64 + lane) : u32x4{0u, 0u, 0u, 0u}; \
                }                                                                                  \
            } else {                                                                               \
                const u32x4* p = reinterpret_cast<const u32x4*>(row_ptr(a, row)) + koff;           \
                _Pragma("unroll") for (int j = 0; j < NVW; ++j)                                    \
                    WB[r][j] = (!tail || koff + j * 64 + lane < kvec) ? ld_nt(p + j * 64 + lane) : u32x4{0u, 0u, 0u, 0u}; \
            }                                                                                      \
        }                                                                                          \
    }

#define OMX_COMPUTE(WB, R0)                                                                        \
    {                                                                                              \
        float acc[NR];                                                                             \
        _Pragma("unroll") for (int r = 0; r < NR; ++r) acc[r] = 0.f;                               \
        _Pragma("unroll") for (int j = 0; j < NVW; ++j) {                                          \
            const u32x4 xp = xs[koff + j * 64 + lane];                                             \
            float xf[8];                                                                           \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                        \
                xf[2 * q] = bf16lo(xp[q]);                                                         \
                xf[2 * q + 1] = bf16hi(xp[q]);                                                     \
            }                                                                                      \
            _Pragma("unroll") for (int r = 0; r < NR; ++r) acc[r] = dot8(WB[r][j], xf, acc[r]);    \
        }                                                                                          \
        _Pragma("unroll") for (int r = 0; r < NR; ++r) acc[r] = wave_sum(acc[r]);                  \
        if (lane == 0) {                                                                           \
            _Pragma("unroll") for (int r = 0; r < RB; ++r) {                                       \
                const int row = (R0) + r;                                                          \
                if (row < row_end) {                                                               \
                    if (KSPLIT == 1) {                                                             \
                        epilogue<EPI>(a, row, acc[LR * r], acc[LR * r + (LR - 1)], best);          \
                    } else {                                                                       \
                        const int lr = row - row_begin;                                            \
                        part[(lr * LR) * KSPLIT + wave] = acc[LR * r];                             \
                        if (LR == 2) part[(lr * LR + 1) * KSPLIT + wave] = acc[LR * r + (LR - 1)]; \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }

    // ---- first weight batch goes out before the activation is even loaded ----
    // (round 2 measured the alternatives in the step: the activation / norm weight / residual loads FIRST, the whole first
    //  round as straight-line code with counted waits, either load order -- each 4-5 % slower per token than this form, whose
    //  conservative vmcnt(0) before the prologue lets a wave's first 16 KiB land before it asks for more)
    // (timeline stamps only in -DOMX_GEMV_TRACE builds, `make VARIANT=trace VARIANT_FLAGS=-DOMX_GEMV_TRACE`: the four stores
    //  change hipcc's register allocation of the whole kernel, 221 -> 157 VGPRs, and cost 2 % of a decode step)
#ifdef OMX_GEMV_TRACE
    unsigned long long* const tr = a.trace ? a.trace + (size_t)blockIdx.x * 4 : nullptr;
#else
    constexpr unsigned long long* tr = nullptr;
#endif
    if (tr && threadIdx.x == 0) tr[0] = wall_clock64();
    if (PRO != PRO_ROUTE) {   // (PRO_ROUTE: which expert's rows these are is only known after the prologue)
        if (active) OMX_ISSUE(wA, row_begin);
    }

    if constexpr (PRO == PRO_ROUTE) {
        // ---- prologue with routing.  moe_router_kernel at this shape runs 512 threads, thread v owning vector v of the row: its sum of
        //      squares per thread, wave sums, a serial sum over the 8 waves; then expert e's logit by ONE wave (lane l: elements
        //      it * 512 + l * 8, one fma chain, wave sum, bf16).  Thread t stands in for router threads t and t + 256. ----
        float* s_red8 = red;                                   // [8]
        float* s_logit = red + 8;                              // [route_E <= 8]
        uint32_t* s_sel = reinterpret_cast<uint32_t*>(red + 16);   // [kMaxTopK]
        bf16_t* s_score = reinterpret_cast<bf16_t*>(red + 24);     // [kMaxTopK]
        const bf16_t* xg = a.x;
        constexpr int PV = (NV * 64 + kBlock - 1) / kBlock;
        static_assert(PV <= 2 && KSPLIT == 1, "routing prologue: rows of at most 4096 elements");
        u32x4 xv[PV], nwv[PV];
#pragma unroll
        for (int i = 0; i < PV; ++i) {
            const int v = threadIdx.x + i * kBlock;
            float ss = 0.f;
            if (v < kvec) {
                xv[i] = *(reinterpret_cast<const u32x4*>(xg) + v);
                nwv[i] = *(reinterpret_cast<const u32x4*>(a.norm_w) + v);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ss = fmaf(bf16lo(xv[i][q]), bf16lo(xv[i][q]), ss);
                    ss = fmaf(bf16hi(xv[i][q]), bf16hi(xv[i][q]), ss);
                }
            }
            ss = wave_sum(ss);
            if (lane == 0) s_red8[wave + 4 * i] = ss;
        }
        if (PV < 2 && threadIdx.x < 4) s_red8[4 + threadIdx.x] = 0.f;
        __syncthreads();
        float tot = 0.f;
        for (int w = 0; w < 8; ++w) tot += s_red8[w];
        const float rstd = 1.0f / sqrtf(tot / (float)K + a.eps);
#pragma unroll
        for (int i = 0; i < PV; ++i) {
            const int v = threadIdx.x + i * kBlock;
            if (v < NV * 64) {
                u32x4 o = {0u, 0u, 0u, 0u};
                if (v < kvec) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[q] = pack_bf16(bf16lo(xv[i][q]) * rstd * bf16lo(nwv[i][q]), bf16hi(xv[i][q]) * rstd * bf16hi(nwv[i][q]));
                }
                xs[v] = o;
            }
        }
        __syncthreads();
        for (int e = wave; e < a.route_E; e += kWaves) {
            const u32x4* g = reinterpret_cast<const u32x4*>(a.route_gate + (size_t)e * K);
            float acc = 0.f;
            for (int it = 0; it * 64 < kvec; ++it) {
                const u32x4 xa = xs[it * 64 + lane], gb = g[it * 64 + lane];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc = fmaf(bf16lo(xa[q]), bf16lo(gb[q]), acc);
                    acc = fmaf(bf16hi(xa[q]), bf16hi(gb[q]), acc);
                }
            }
            acc = wave_sum(acc);
            if (lane == 0) s_logit[e] = round_bf16(acc);
        }
        __syncthreads();
        if (wave == 0) route_from_logits(s_logit, 0, lane, a.route_E, a.route_k, a.route_mode, a.route_renorm, s_sel, s_score);
        __syncthreads();
        size_t e = s_sel[blockIdx.y];
        if (blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < a.route_k) {
            a.route_inds[threadIdx.x] = s_sel[threadIdx.x];
            a.route_scores[threadIdx.x] = s_score[threadIdx.x];
        }
        if (a_in.w_sel_n > 0) {   // expert-parallel shard: block-uniform early exit for experts of other ranks
            if (e < (size_t)a_in.w_sel_lo || e >= (size_t)(a_in.w_sel_lo + a_in.w_sel_n)) return;
            e -= (size_t)a_in.w_sel_lo;
        }
        a.w0 = a_in.w0 + e * a_in.w_estride;
        if (a_in.w1) a.w1 = a_in.w1 + e * a_in.w_estride;
        if (active) OMX_ISSUE(wA, row_begin);
    } else
    // ---- prologue: stage x (bf16) in LDS; optionally x := bf16(x + bf16(partial)); RMS-normalise ----
    {
        const bf16_t* xg = a.x + (a.x_row ? (size_t)a.x_row[0] * K : 0);
        float ss = 0.f;
        constexpr int PV = (NV * 64 + kBlock - 1) / kBlock;   // vectors per thread
        u32x4 xv[PV], nwv[PV];
#pragma unroll
        for (int i = 0; i < PV; ++i) {
            const int v = threadIdx.x + i * kBlock;
            if (v < NV * 64) {
                u32x4 raw = (!TAIL || v < kvec) ? *(reinterpret_cast<const u32x4*>(xg) + v) : u32x4{0u, 0u, 0u, 0u};
                if (PRO == PRO_RMSNORM) nwv[i] = *(reinterpret_cast<const u32x4*>(a.norm_w) + v);
                if (a.x_partial) {
                    const f32x4 p0 = *(reinterpret_cast<const f32x4*>(a.x_partial) + 2 * v);
                    const f32x4 p1 = *(reinterpret_cast<const f32x4*>(a.x_partial) + 2 * v + 1);
                    float pp[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
                    for (int j = 1; j < a.x_partial_n; ++j) {   // MoE: the experts' weighted outputs, summed in slot order
                        const f32x4* pj = reinterpret_cast<const f32x4*>(a.x_partial + (size_t)j * K) + 2 * v;
                        const f32x4 q0 = pj[0], q1 = pj[1];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { pp[e] += q0[e]; pp[4 + e] += q1[e]; }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        raw[q] = pack_bf16(bf16lo(raw[q]) + round_bf16(pp[2 * q]),
                                           bf16hi(raw[q]) + round_bf16(pp[2 * q + 1]));
                    if (a.x_out && blockIdx.x == 0) *(reinterpret_cast<u32x4*>(a.x_out) + v) = raw;
                }
                xv[i] = raw;
                if (PRO == PRO_RMSNORM) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = bf16lo(raw[q]), hi = bf16hi(raw[q]);
                        ss = fmaf(lo, lo, ss);
                        ss = fmaf(hi, hi, ss);
                    }
                }
            }
        }
        if (PRO == PRO_RMSNORM) {
            ss = block_sum<kWaves>(ss, red);
            const float rstd = 1.0f / sqrtf(ss / (float)K + a.eps);
#pragma unroll
            for (int i = 0; i < PV; ++i) {
                const int v = threadIdx.x + i * kBlock;
                if (v < NV * 64) {
                    const u32x4 nw = nwv[i];
                    u32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[q] = pack_bf16(bf16lo(xv[i][q]) * rstd * bf16lo(nw[q]),
                                         bf16hi(xv[i][q]) * rstd * bf16hi(nw[q]));
                    xs[v] = o;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < PV; ++i) {
                const int v = threadIdx.x + i * kBlock;
                if (v < NV * 64) xs[v] = xv[i];
            }
        }
        __syncthreads();
    }
    if (tr && threadIdx.x == 0) tr[1] = wall_clock64();

    // ---- stream rows: batch b is reduced from one register set while batch b+1 is in flight ----
    if (active) {
        for (int r0 = row_begin; r0 < row_end; r0 += 2 * RB) {
            if (r0 + RB < row_end) OMX_ISSUE(wB, r0 + RB);
            OMX_COMPUTE(wA, r0);
            if (tr && threadIdx.x == 0 && r0 == row_begin) tr[2] = wall_clock64();
            if (r0 + RB >= row_end) break;
            if (r0 + 2 * RB < row_end) OMX_ISSUE(wA, r0 + 2 * RB);
            OMX_COMPUTE(wB, r0 + RB);
        }
    }
    if (KSPLIT > 1) {
        __syncthreads();
        const int lr = threadIdx.x;
        if (lr < row_end - row_begin) {
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int w = 0; w < KSPLIT; ++w) {
                v0 += part[(lr * LR) * KSPLIT + w];
                if (LR == 2) v1 += part[(lr * LR + 1) * KSPLIT + w];
            }
            epilogue<EPI>(a, row_begin + lr, v0, v1, best);
        }
    }
    if (EPI == EPI_F32) {
        if (a.peer) {   // (uniform over the launch)
            if (KSPLIT > 1) peer_finish_rows(a, (unsigned)best, row_begin, row_end, threadIdx.x, kBlock);
            else peer_finish_rows(a, (unsigned)best, row_begin, row_end, lane, 64);
        }
    }
    if (EPI == EPI_ARGMAX) {
        // one partial per block (no same-address atomics: 150k of them serialise at ~12 ns each);
        // argmax_finalize in engine.hip reduces the partials
        uint64_t* bred = reinterpret_cast<uint64_t*>(red);   // 4 x u64 = first 32 B of the scratch
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        __syncthreads();
        if (lane == 0) bred[wave] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t b = bred[0];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) b = bred[w] > b ? bred[w] : b;
            a.argmax_slot[blockIdx.x] = b;
        }
    }
    if (tr && threadIdx.x == 0) tr[3] = wall_clock64();
#undef OMX_ISSUE
#undef OMX_COMPUTE
}

// ---- any contraction width (K a multiple of 8): the fallback for shapes without a tuned instantiation (Qwen2.5-7B:
//      K = 3584 / 18944).  Same prologues and epilogues, a runtime loop over the 16-byte vectors of a row, RB rows share
//      each activation vector; one register set (no double buffer), so it streams at roughly 2/3 of the tuned kernels. ----
template <int PRO, int EPI>
__global__ __launch_bounds__(kBlock) void gemv_generic_kernel(const GemvArgs a_in) {
    GemvArgs a = a_in;
    if (EPI == EPI_F32) {
        if (a_in.out_scale) a.out_scale_f = bf16_to_f32(a_in.out_scale[blockIdx.y]);
    }
    if (a_in.n_batch > 1 || a_in.w_sel) {
        const int by = blockIdx.y;
        a.x = a_in.x + (size_t)(by / a_in.x_div) * a_in.x_bstride;
        a.out = reinterpret_cast<char*>(a_in.out) + (size_t)by * a_in.out_bstride_bytes;
        if (a_in.w_sel) {
            size_t e = a_in.w_sel[by];
            if (a_in.w_sel_n > 0) {
                if (e < (size_t)a_in.w_sel_lo || e >= (size_t)(a_in.w_sel_lo + a_in.w_sel_n)) return;
                e -= (size_t)a_in.w_sel_lo;
            }
            a.w0 = a_in.w0 + e * a_in.w_estride;
            if (a_in.w1) a.w1 = a_in.w1 + e * a_in.w_estride;
        }
    }
    constexpr int LR = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int RB = (EPI == EPI_SWIGLU) ? 2 : 4;
    constexpr int NR = RB * LR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = a.K, kvec = K / 8;
    u32x4* xs = reinterpret_cast<u32x4*>(smem);                              // [kvec]
    float* red = reinterpret_cast<float*>(smem + (size_t)kvec * 16);         // [8]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const bf16_t* xg = a.x + (a.x_row ? (size_t)a.x_row[0] * K : 0);
        float ss = 0.f;
        for (int v = threadIdx.x; v < kvec; v += kBlock) {
            u32x4 raw = *(reinterpret_cast<const u32x4*>(xg) + v);
            if (a.x_partial) {
                const f32x4 p0 = *(reinterpret_cast<const f32x4*>(a.x_partial) + 2 * v);
                const f32x4 p1 = *(reinterpret_cast<const f32x4*>(a.x_partial) + 2 * v + 1);
                float pp[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
                for (int j = 1; j < a.x_partial_n; ++j) {
                    const f32x4* pj = reinterpret_cast<const f32x4*>(a.x_partial + (size_t)j * K) + 2 * v;
                    const f32x4 q0 = pj[0], q1 = pj[1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pp[e] += q0[e]; pp[4 + e] += q1[e]; }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    raw[q] = pack_bf16(bf16lo(raw[q]) + round_bf16(pp[2 * q]), bf16hi(raw[q]) + round_bf16(pp[2 * q + 1]));
                if (a.x_out && blockIdx.x == 0) *(reinterpret_cast<u32x4*>(a.x_out) + v) = raw;
            }
            xs[v] = raw;
            if (PRO == PRO_RMSNORM) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float lo = bf16lo(raw[q]), hi = bf16hi(raw[q]);
                    ss = fmaf(lo, lo, ss);
                    ss = fmaf(hi, hi, ss);
                }
            }
        }
        if (PRO == PRO_RMSNORM) {
            ss = block_sum<kWaves>(ss, red);
            const float rstd = 1.0f / sqrtf(ss / (float)K + a.eps);
            for (int v = threadIdx.x; v < kvec; v += kBlock) {
                const u32x4 raw = xs[v];
                const u32x4 nw = *(reinterpret_cast<const u32x4*>(a.norm_w) + v);
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q] = pack_bf16(bf16lo(raw[q]) * rstd * bf16lo(nw[q]), bf16hi(raw[q]) * rstd * bf16hi(nw[q]));
                xs[v] = o;
            }
        }
        __syncthreads();
    }
    const int rpw = a.rows_per_wave;
    const int row_begin = (blockIdx.x * kWaves + wave) * rpw;
    const int row_end = min(row_begin + rpw, a.N);
    uint64_t best = 0;
    if (EPI == EPI_F32) {
        if (a.peer) best = peer_tag(a.peer);
    }
    for (int r0 = row_begin; r0 < row_end; r0 += RB) {
        const u32x4* rows[NR];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = min(r0 + r, a.N - 1);
            if (EPI == EPI_SWIGLU) {
                rows[LR * r] = reinterpret_cast<const u32x4*>(a.w0 + (size_t)row * K);
                rows[LR * r + (LR - 1)] = reinterpret_cast<const u32x4*>(a.w1 + (size_t)row * K);
            } else {
                rows[r] = reinterpret_cast<const u32x4*>(row_ptr(a, row));
            }
        }
        float acc[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r] = 0.f;
        for (int v = lane; v < kvec; v += 64) {
            u32x4 w[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) w[r] = ld_nt(rows[r] + v);
            const u32x4 xp = xs[v];
            float xf[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { xf[2 * q] = bf16lo(xp[q]); xf[2 * q + 1] = bf16hi(xp[q]); }
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = dot8(w[r], xf, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (r0 + r < row_end) epilogue<EPI>(a, r0 + r, acc[LR * r], acc[LR * r + (LR - 1)], best);
        }
    }
    if (EPI == EPI_F32) {
        if (a.peer) peer_finish_rows(a, (unsigned)best, row_begin, row_end, lane, 64);
    }
    if (EPI == EPI_ARGMAX) {
        uint64_t* bred = reinterpret_cast<uint64_t*>(red);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        __syncthreads();
        if (lane == 0) bred[wave] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t b = bred[0];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) b = bred[w] > b ? bred[w] : b;
            a.argmax_slot[blockIdx.x] = b;
        }
    }
}

int launch_generic(const GemvArgs& a, int pro, int epi, hipStream_t s) {
    const int groups = (a.N + a.rows_per_wave - 1) / a.rows_per_wave;
    const dim3 grid((groups + kWaves - 1) / kWaves, a.n_batch > 1 ? a.n_batch : 1), block(kBlock);
    const size_t shmem = (size_t)(a.K / 8) * 16 + 64;
#define OMX_GEN_CASE(P, E)                                                                                         \
    if (pro == P && epi == E) {                                                                                    \
        if (shmem > 48 * 1024)                                                                                     \
            OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemv_generic_kernel<P, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
        OMX_LAUNCH_TIMED((gemv_generic_kernel<P, E>), grid, block, shmem, s, a);                                         \
        OMX_LAUNCH_CHECK();                                                                                        \
        return 0;                                                                                                  \
    }
    OMX_GEN_CASE(PRO_NONE, EPI_STORE)
    OMX_GEN_CASE(PRO_RMSNORM, EPI_STORE)
    OMX_GEN_CASE(PRO_NONE, EPI_RESIDUAL)
    OMX_GEN_CASE(PRO_RMSNORM, EPI_SWIGLU)
    OMX_GEN_CASE(PRO_NONE, EPI_SWIGLU)
    OMX_GEN_CASE(PRO_RMSNORM, EPI_ARGMAX)
    OMX_GEN_CASE(PRO_NONE, EPI_F32)
#undef OMX_GEN_CASE
    return set_error("gemv: unsupported prologue/epilogue combination %d/%d", pro, epi);
}

template <int NVW, int KSPLIT, int RB>
int launch_nv(const GemvArgs& a, int pro, int epi, hipStream_t s) {
    const int groups = (a.N + a.rows_per_wave - 1) / a.rows_per_wave;   // row groups (waves or blocks)
    const dim3 grid(KSPLIT == 1 ? (groups + kWaves - 1) / kWaves : groups, a.n_batch > 1 ? a.n_batch : 1), block(kBlock);
    const size_t shmem = (size_t)NVW * KSPLIT * 64 * 16 + (pro == PRO_ROUTE ? 128 : 32) + (KSPLIT > 1 ? (size_t)a.rows_per_wave * 2 * KSPLIT * 4 : 0);
    const bool tail = a.K / 8 < NVW * KSPLIT * 64;
#define OMX_GEMV_CASE(P, E)                                                                          \
    if (pro == P && epi == E) {                                                                      \
        if (tail && P == PRO_NONE)                                                                   \
            OMX_LAUNCH_TIMED((gemv_kernel<NVW, KSPLIT, (E == EPI_SWIGLU ? (RB > 1 ? RB / 2 : 1) : RB), PRO_NONE, E, true>), grid, block, shmem, s, a); \
        else                                                                                         \
            OMX_LAUNCH_TIMED((gemv_kernel<NVW, KSPLIT, (E == EPI_SWIGLU ? (RB > 1 ? RB / 2 : 1) : RB), P, E>), grid, block, shmem, s, a); \
        OMX_LAUNCH_CHECK();                                                                          \
        return 0;                                                                                    \
    }
    OMX_GEMV_CASE(PRO_NONE, EPI_STORE)
    OMX_GEMV_CASE(PRO_RMSNORM, EPI_STORE)
    OMX_GEMV_CASE(PRO_NONE, EPI_RESIDUAL)
    OMX_GEMV_CASE(PRO_RMSNORM, EPI_SWIGLU)
    OMX_GEMV_CASE(PRO_NONE, EPI_SWIGLU)
    OMX_GEMV_CASE(PRO_RMSNORM, EPI_ARGMAX)
    OMX_GEMV_CASE(PRO_NONE, EPI_F32)
    if constexpr (KSPLIT == 1 && NVW <= 8) {
        if (pro == PRO_ROUTE && epi == EPI_SWIGLU && !tail) {
            OMX_LAUNCH_TIMED((gemv_kernel<NVW, KSPLIT, (RB > 1 ? RB / 2 : 1), PRO_ROUTE, EPI_SWIGLU>), grid, block, shmem, s, a);
            OMX_LAUNCH_CHECK();
            return 0;
        }
    }
#undef OMX_GEMV_CASE
    return set_error("gemv: unsupported prologue/epilogue combination %d/%d", pro, epi);
}

}  // namespace

// Row groups are sized so that the WHOLE grid is co-resident in one round: these kernels are
// register-heavy (two in-flight register sets), 2 waves/SIMD for the 8-vector variants (1 for the
// SwiGLU pair kernel), and a second round of blocks pays the cold-start latency again.
// vectors-per-row (K/512, rounded up) of the tuned instantiation that serves K, or 0 (generic kernel).  A row whose last
// vector rows are partly or wholly empty (K not a multiple of 512, or no instantiation for exactly K/512) is served by the
// next larger instantiation with masked loads -- only without a prologue (plain activation staging).
static int tuned_nv(int K, bool plain_prologue) {
    static const int kSizes[] = {1, 2, 3, 4, 6, 7, 8, 12, 16, 24, 28, 32, 40};
    if (K <= 0 || K % 8 != 0) return 0;
    const int nv = (K + 511) / 512;
    for (int sz : kSizes) {
        if (sz == nv && (K % 512 == 0 || plain_prologue)) return sz;
        if (sz > nv) return (plain_prologue && sz * 3 <= nv * 4) ? sz : 0;   // at most a third of the lanes idle
    }
    return 0;
}
static bool tuned(int K, bool plain_prologue) { return tuned_nv(K, plain_prologue) != 0; }

static int resolve_rpw(int N, int K, int epi, int rpw, bool is_tuned = true) {
    const bool split = is_tuned && (K + 511) / 512 > 8;
    if (rpw <= 0) {
        // measured on MI355X (tools/gemv_sweep.py): short row groups in whole double-buffer rounds win;
        // small matrices want every CU busy (>= ~1500 waves), the vocabulary-sized one longer streams
        if (split) rpw = 8;
        else if (N >= 65536) rpw = 8;
        else rpw = (N / 4 >= 1536) ? 4 : 2;
        (void)epi;
    }
    if (split && rpw > 256) rpw = 256;
    return rpw;
}

// the routing prologue reproduces moe_router_kernel's 512-thread form: <= 8 experts, rows of <= 4096 elements in whole 512-element
// vectors rows (K % 512 == 0 -> a tuned instantiation without a tail)
bool gemv_route_supported(int K, int n_experts, int top_k) {
    return K > 0 && K <= 4096 && K % 512 == 0 && tuned(K, false) && n_experts >= 1 && n_experts <= 8 && top_k >= 1 && top_k <= n_experts;
}

int gemv_grid(int N, int K, int epi, int rows_per_wave) {
    const bool t = tuned(K, false);   // the callers that need the grid (argmax partials) launch with the RMSNorm prologue
    const int rpw = resolve_rpw(N, K, epi, rows_per_wave, t);
    const int groups = (N + rpw - 1) / rpw;
    return (t && ((K + 511) / 512) > 8) ? groups : (groups + kWaves - 1) / kWaves;
}

bool gemv_k_supported(int K, bool needs_full_vectors) {
    (void)needs_full_vectors;
    return K > 0 && K % 8 == 0 && K <= 65536;   // tuned kernels where they exist, the generic one otherwise
}

int launch_gemv(const GemvArgs& a_in, int pro, int epi, hipStream_t s) {
    GemvArgs a = a_in;
    OMX_REQUIRE(a.K > 0 && a.K % 8 == 0 && a.K <= 65536, "gemv: K=%d must be a positive multiple of 8 (at most 65536)", a.K);
    OMX_REQUIRE(a.N > 0, "gemv: N must be positive");
    const int nv = tuned_nv(a.K, pro == PRO_NONE && !a.x_partial);
    const bool t = nv != 0;
    a.rows_per_wave = resolve_rpw(a.N, a.K, epi, a.rows_per_wave, t);
    if (!t) return launch_generic(a, pro, epi, s);
    switch (nv) {
        // RB*NVW ~ 16 x 1-KiB loads in flight per wave per register set
        case 1: return launch_nv<1, 1, 8>(a, pro, epi, s);
        case 2: return launch_nv<2, 1, 8>(a, pro, epi, s);
        case 3: return launch_nv<3, 1, 4>(a, pro, epi, s);
        case 4: return launch_nv<4, 1, 4>(a, pro, epi, s);
        case 6: return launch_nv<6, 1, 2>(a, pro, epi, s);
        case 7: return launch_nv<7, 1, 2>(a, pro, epi, s);
        case 8: return launch_nv<8, 1, 2>(a, pro, epi, s);
        case 12: return launch_nv<3, 4, 4>(a, pro, epi, s);
        case 16: return launch_nv<4, 4, 4>(a, pro, epi, s);
        case 24: return launch_nv<6, 4, 2>(a, pro, epi, s);
        case 28: return launch_nv<7, 4, 2>(a, pro, epi, s);
        case 32: return launch_nv<8, 4, 2>(a, pro, epi, s);
        case 40: return launch_nv<10, 4, 2>(a, pro, epi, s);
        default: return set_error("gemv: K=%d (K/512=%d) has no instantiated kernel", a.K, nv);
    }
}

}  // namespace omx
