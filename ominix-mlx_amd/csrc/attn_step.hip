// Attention of ONE decode step inside the engine's step graph (Tq == 1, batch 1): q/k RMSNorm + RoPE + KV-cache append +
// split-KV SDPA + split merge in a single launch whose dependent chain is as short as the hardware allows.
//   reference: Attention::forward of qwen3-mlx/src/model.rs:161-215 (q_norm/k_norm :172-181, rope at cache.offset() :186-194,
//   cache.update_and_fetch :196, scaled_dot_product_attention :198-210 -> mlx-rs-core/src/utils.rs:191-209);
//   Mixtral / Qwen2 wiring without q/k norm (mixtral-mlx/src/model.rs:96-160, qwen3-mlx/src/qwen2.rs:100-160).
//
// Why a second decode-attention kernel next to attn_decode.hip (which stays the per-op SDPA of the mlx-c route): at batch 1 and
// a 2 k context a layer's KV is 9 MB -- 1.5 us of HBM time -- yet the round-1 kernel took 14.5 us, because it was a chain of
// eight dependent memory round trips (position -> K/V -> ... -> partial store -> ack -> counter atomic -> (m,l) -> partials ->
// output).  This kernel removes five of them:
//   * NOTHING it loads at the start depends on the position: the split a block owns is a FIXED token range
//     [split*chunk, +chunk) chosen when the step graph is captured (the engine re-captures when the context outgrows the
//     bucket), and the RoPE row of the current position sits in a small `rope_cur` buffer that the step's first kernel
//     refreshes.  Position, raw q/k/v, norm weights, RoPE row and the first K/V rows are therefore ONE round of loads;
//   * the split partials travel as data-tagged 8-byte granules {f32 value, tag} written with write-through (sc1) stores
//     (cdna_hip_programming.md Guideline 16, form R2): no store-ack wait, no arrival counter, no flag.  The tag is the step
//     sequence number x layer, so nothing has to be reset between launches;
//   * the blocks with split < G are also the consumers: after their own chunk, waves 0..D/64-1 of block (kvh, j) gather the
//     granules of head kvh*G + j with coherent loads -- all splits in flight at once, re-read until every tag matches --
//     merge them (max, rescale, sum, one division, one rounding) and store the bf16 head output.
// Every block of the launch is co-resident (one block per CU by construction, __launch_bounds__(512, 1), grid <= 256), spins are
// bounded and a wait that gives up raises `abort_flag` (the host reports it) instead of hanging the GPU.
//
// O projection in the same launch (NVW > 0; model.rs:214, 325).  The attention phase is a chain of latencies during which the HBM
// idles, and the O-projection weights (33 MB at Qwen3-8B shapes) depend on nothing: every wave issues the loads of ITS two weight
// rows right after its first K/V loads (in-order return: nothing the attention waits for is queued behind them) and keeps the rows
// in registers.  The consumers publish the merged head outputs a second time as granules {two packed bf16, tag}; every block
// sweeps those K/2 granules into LDS (all 512 threads, one coherent round per pass) and finishes with the exact arithmetic of the
// separate O GEMV (gemv_kernel<NVW, 1, 2, PRO_NONE, EPI_RESIDUAL>: same lane-to-element map, same fma order, same wave
// reduction, same roundings) -- bit-identical hidden state, one launch boundary and one weight-stream ramp less per layer.
#include <algorithm>

#include "attn.hpp"
#include "act16.hpp"
#include "launch_timing.hpp"

namespace omx {

namespace {

constexpr int kBlock = 512;
constexpr int kWaves = 8;
constexpr unsigned kSpinLimit = 1u << 15;   // gather passes (~1 us each) before a consumer gives up

typedef __attribute__((address_space(1))) unsigned long long gu64;

__device__ __forceinline__ void st_granule_u32(uint64_t* p, unsigned tag, unsigned v) {
    __hip_atomic_store((gu64*)p, ((unsigned long long)tag << 32) | (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(uint64_t* p, unsigned tag, float v) { st_granule_u32(p, tag, __float_as_uint(v)); }
__device__ __forceinline__ unsigned long long ld_granule(const uint64_t* p) {
    return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int N>
__device__ __forceinline__ float swap_halves(float v) {     // value of lane (l ^ N/2) of the aligned N-lane group
    if (N == 16) return dpp_f<0x128>(v);                       // row_ror:8
    return dpp_f<0x1B>(dpp_f<kDppHalfMirror>(v));              // (7 - i) then quad reverse == i ^ 4
}

template <bool F16 = false>
__device__ __forceinline__ void unpack8(const u32x4 r, float (&x)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x[2 * e] = Act16<F16>::lo(r[e]);
        x[2 * e + 1] = Act16<F16>::hi(r[e]);
    }
}

// merge the granules of one head: `NB` batches of 16 splits, every load of every live batch in flight before the first wait
template <int D, int NB, bool F16 = false>
__device__ __forceinline__ void gather_head(const AttnStepArgs& a, const uint64_t* base, int n_active, unsigned tag, int lane,
                                            int dim, bf16_t* out, uint64_t* xg_head) {
    constexpr int STRIDE = D + 2;
    unsigned long long ml0 = 0, ml1 = 0, og[NB][16];
    bool ok_ml = false, ok_b[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) ok_b[b] = b * 16 >= n_active;   // batches past the live splits are never read
    const int sl = min(lane, n_active - 1);
    for (unsigned spins = 0;; ++spins) {
        if (!ok_ml) {
            ml0 = ld_granule(base + (size_t)sl * STRIDE + D);
            ml1 = ld_granule(base + (size_t)sl * STRIDE + D + 1);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (ok_b[b]) continue;
#pragma unroll
            for (int j = 0; j < 16; ++j) og[b][j] = ld_granule(base + (size_t)min(b * 16 + j, n_active - 1) * STRIDE + dim);
        }
        bool all = true;
        if (!ok_ml) ok_ml = __all((unsigned)(ml0 >> 32) == tag && (unsigned)(ml1 >> 32) == tag);
        all = ok_ml;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (!ok_b[b]) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 16; ++j) ok &= (unsigned)(og[b][j] >> 32) == tag;
                ok_b[b] = __all(ok);
            }
            all = all && ok_b[b];
        }
        if (all) break;
        if (spins >= kSpinLimit) {   // a producer never showed up: void result, loud flag, no hang
            if (lane == 0) __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    const float m = lane < n_active ? __uint_as_float((unsigned)ml0) : -INFINITY;
    const float l = lane < n_active ? __uint_as_float((unsigned)ml1) : 0.f;
    const float M = wave_max(m);
    const float f = (m == -INFINITY) ? 0.f : __expf(m - M);
    const float L = wave_sum(f * l);
    float acc = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b * 16 >= n_active) break;
#pragma unroll
        for (int j = 0; j < 16; ++j)   // splits past n_active re-read the last live one and carry f == 0
            acc = fmaf(readlane_f(f, b * 16 + j), __uint_as_float((unsigned)og[b][j]), acc);
    }
    const bf16_t r = Act16<F16>::bits(acc / L);     // (16-bit pattern: bfloat16, or float16 in a float16 model)
    out[dim] = r;
    if (xg_head) {   // the same value once more, for the O-projection phase of every block: one granule per dim pair
        const float nb = dpp_f<kDppXor1>(bf16_to_f32(r));
        if (!(lane & 1)) st_granule_u32(xg_head + dim / 2, tag, (unsigned)r | ((unsigned)f32_to_bf16(nb) << 16));
    }
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
// sum of 8 exact bf16 products, f32 accumulation (v_dot2c_f32_bf16).  The pairs are taken with shufflevector from the whole
// 8-element vector: bit-casting the dwords of a u32x4 one by one made hipcc (ROCm 7.2) feed all four dot2 instructions the
// SAME register pair -- 4x the first product, silently wrong scores.
__device__ __forceinline__ float dot8_bf16(const u32x4 a, const u32x4 b) {
    const bf16x8_t A = __builtin_bit_cast(bf16x8_t, a), B = __builtin_bit_cast(bf16x8_t, b);
    float d = 0.f;
    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A, A, 0, 1), __builtin_shufflevector(B, B, 0, 1), d, false);
    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A, A, 2, 3), __builtin_shufflevector(B, B, 2, 3), d, false);
    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A, A, 4, 5), __builtin_shufflevector(B, B, 4, 5), d, false);
    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A, A, 6, 7), __builtin_shufflevector(B, B, 6, 7), d, false);
    return d;
}

// One block per CU: 8 waves share a split's token range in units of one wave-instruction (TPW token rows), so the serial
// instruction chain of a wave is a quarter of what four waves with four rows each had, two waves per SIMD cover each other's
// latencies, and the q/k RMSNorm + RoPE is done ONCE per block (wave g: query head g, wave GT % 8: the new key row) and shared
// through LDS instead of five times per wave.  attn_step_plan keeps the grid <= 256 blocks.
constexpr int kKU = 3;   // units in flight per wave

// ... and of float16 rows (a float16 checkpoint's K cache): v_dot2_f32_f16, same order
template <bool F16>
__device__ __forceinline__ float dot8_16(const u32x4 a, const u32x4 b) {
    if constexpr (!F16) return dot8_bf16(a, b);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) d = Act16<true>::dot2(a[e], b[e], d);
    return d;
}

// the O GEMV's accumulation (gemv.hip dot8): lo then hi of each dword, one fma chain
__device__ __forceinline__ float dot8_chain(const u32x4 w, const u32x4 xp, float acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc = fmaf(bf16lo(w[i]), bf16lo(xp[i]), acc);
        acc = fmaf(bf16hi(w[i]), bf16hi(xp[i]), acc);
    }
    return acc;
}

constexpr int kORows = 4;   // most O-projection rows a wave holds (attn_step_oproj_ok)

// the merged attention vector: awaited, then swept from its granules into LDS (natural element order); a block-wide step
template <int NVW>
__device__ __forceinline__ void oproj_await_vector(const AttnStepArgs& a, u32x4* sm_x, unsigned tag, int lane, int wave) {
    // wait politely: ONE wave watches one granule per consumer wave (the last of its 32) with a sleep between looks; only then does
    // the block read the whole vector -- normally once (measured: sweep done at 11.3 us with the watch, 11.9 us without)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == 0) {
        constexpr int SEGS = NVW * 8;                                  // consumer waves: H * D / 64
        const uint64_t* probe = a.xg + (size_t)min(lane, SEGS - 1) * 32 + 31;
        for (unsigned spins = 0; spins < kSpinLimit; ++spins) {
            const unsigned long long g = ld_granule(probe);
            if (__all((unsigned)(g >> 32) == tag)) break;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    // sweep the granules into LDS: thread t owns granules [t * GP, + GP) of NVW * 256
    constexpr int GP = (NVW + 1) / 2;      // NVW * 256 granules over 512 threads: 1, 1, 2, 4 (NVW = 7: 448 threads), 4
    static_assert((NVW * 256) % GP == 0, "a thread takes all of its granules or none");
    bool done = (int)threadIdx.x * GP >= NVW * 256;
    unsigned* sx = reinterpret_cast<unsigned*>(sm_x);
    for (unsigned spins = 0;; ++spins) {
        if (!done) {
            unsigned long long g[GP];
#pragma unroll
            for (int i = 0; i < GP; ++i) g[i] = ld_granule(a.xg + (size_t)threadIdx.x * GP + i);
            bool ok = true;
#pragma unroll
            for (int i = 0; i < GP; ++i) ok &= (unsigned)(g[i] >> 32) == tag;
            if (ok) {
#pragma unroll
                for (int i = 0; i < GP; ++i) sx[threadIdx.x * GP + i] = (unsigned)g[i];
                done = true;
            }
        }
        if (__syncthreads_and(done)) break;                          // also orders the LDS writes before the reads below
        if (spins >= kSpinLimit) {
            if (threadIdx.x == 0) __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// The O-projection phase of a non-consumer block (attn_step_kernel, NVW > 0): R weight rows per wave go out NOW -- the block's
// latency-critical work is over and its attention registers are free -- then the merged attention vector is awaited and swept into
// LDS, then the separate O GEMV's arithmetic (gemv.hip gemv_kernel<NVW, 1, 2, PRO_NONE, EPI_RESIDUAL>) runs on the held rows.
// Measured alternatives at Qwen3-8B shapes, ctx 2 k: weight loads issued with the first K/V round -> 33 MB of requests fill the HBM
// queues ahead of the q / K / V rows (they land at 6.7 us instead of 2.1); issued after that round landed, in every block -> a wave's
// memory instructions queue in order behind its own prefetch, the partial stores left 1.5 us and the gather 0.5 us later (kernel
// 13.8 us, still 2.4 us/layer better than two launches).
template <int NVW, int R>
__device__ __forceinline__ void oproj_phase(const AttnStepArgs& a, u32x4* sm_x, unsigned tag, int lane, int wave, int o_row0,
                                            unsigned long long* tr) {
    u32x4 ow[R][NVW];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = min(o_row0 + r, a.o_rows - 1);                           // clamp: surplus waves re-read a valid row
        const u32x4* p = reinterpret_cast<const u32x4*>(a.o_w + (size_t)row * (NVW * 512));
#pragma unroll
        for (int j = 0; j < NVW; ++j) ow[r][j] = __builtin_nontemporal_load(p + j * 64 + lane);
    }
    bf16_t o_res[R];
#pragma unroll
    for (int r = 0; r < R; ++r) o_res[r] = a.o_resid[min(o_row0 + r, a.o_rows - 1)];
    __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE: the scheduler would sink them to their use
    oproj_await_vector<NVW>(a, sm_x, tag, lane, wave);
    if (tr && threadIdx.x == 0) tr[5] = wall_clock64();
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < NVW; ++j) {
        const u32x4 xp = sm_x[j * 64 + lane];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = dot8_chain(ow[r][j], xp, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (o_row0 + r < a.o_rows) {
                if (a.o_out_f32) a.o_out_f32[o_row0 + r] = acc[r];   // the O GEMV's EPI_F32: this rank's partial, un-rounded
                else a.o_out[o_row0 + r] = f32_to_bf16(bf16_to_f32(o_res[r]) + round_bf16(acc[r]));
            }
    }
    if (tr && threadIdx.x == 0) tr[6] = wall_clock64();
}


// The same phase on a 4-bit packed O matrix (quant.hip qgemv_kernel<4, 4, PRO_NONE, EPI_RESIDUAL, 2, true>: 32 elements per lane and
// step, interleaved scale | bias words): a row is NVW / 4 steps of one 16-byte weight load + one 4-byte scale/bias load per lane.
// The attention vector is re-paired in LDS the way that kernel's prologue stores it ((x0,x2) (x4,x6) (x1,x3) (x5,x7) per 8
// elements) with the per-32 sums next to it; nibble unpack, v_dot2c order and the scale / bias fmas are that kernel's, so the
// residual stream is bit-identical to the two-launch step.
template <int NVW, int R>
__device__ __forceinline__ void oproj_phase_q4(const AttnStepArgs& a, u32x4* sm_x, float* sm_xsum, unsigned tag, int lane, int wave,
                                               int o_row0, unsigned long long* tr) {
    static_assert(NVW % 4 == 0, "K must be a multiple of 2048");
    constexpr int ST = NVW / 4;
    u32x4 ow[R][ST];
    uint32_t sbv[R][ST];
    const int words_per_row = NVW * 64, groups_per_row = NVW * 512 / a.o_group;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = min(o_row0 + r, a.o_rows - 1);
        const uint32_t* p = a.o_wq + (size_t)row * words_per_row;
        const uint32_t* psb = a.o_sb + (size_t)row * groups_per_row;
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            const int chunk = st * 64 + lane;
            ow[r][st] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + (size_t)chunk * 4));
            sbv[r][st] = psb[chunk * 32 / a.o_group];
        }
    }
    bf16_t o_res[R];
#pragma unroll
    for (int r = 0; r < R; ++r) o_res[r] = a.o_resid[min(o_row0 + r, a.o_rows - 1)];
    __builtin_amdgcn_sched_barrier(0);
    oproj_await_vector<NVW>(a, sm_x, tag, lane, wave);
    // re-pair + per-chunk sums: thread t owns the 8 elements of vector t (whole waves: NVW * 64 threads)
    if ((int)threadIdx.x < NVW * 64) {
        const u32x4 o = sm_x[threadIdx.x];
        u32x4 t;
        t[0] = __builtin_amdgcn_perm(o[1], o[0], 0x05040100u); t[1] = __builtin_amdgcn_perm(o[3], o[2], 0x05040100u);
        t[2] = __builtin_amdgcn_perm(o[1], o[0], 0x07060302u); t[3] = __builtin_amdgcn_perm(o[3], o[2], 0x07060302u);
        sm_x[threadIdx.x] = t;
        float sv = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) sv += bf16lo(o[q]) + bf16hi(o[q]);
        sv += dpp_f<kDppXor1>(sv);
        sv += dpp_f<kDppXor2>(sv);
        if ((threadIdx.x & 3) == 0) sm_xsum[threadIdx.x >> 2] = sv;
    }
    __syncthreads();
    if (tr && threadIdx.x == 0) tr[5] = wall_clock64();
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int st = 0; st < ST; ++st) {
        const int chunk = st * 64 + lane;
        uint32_t xp[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 xv = sm_x[chunk * 4 + j];
#pragma unroll
            for (int q = 0; q < 4; ++q) xp[j * 4 + q] = xv[q];
        }
        const float xsm = sm_xsum[chunk];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float d = 0.f;
            const float scl = bf16lo(sbv[r][st]);
            float bia = bf16hi(sbv[r][st]);
#pragma unroll
            for (int wi = 0; wi < 4; ++wi) {
                const uint32_t wdw = ow[r][st][wi];
                const uint32_t lo = wdw & 0x0F0F0F0Fu, hi = (wdw >> 4) & 0x0F0F0F0Fu;
                const uint32_t c43 = 0x43434343u;
                const uint32_t q0 = __builtin_amdgcn_perm(c43, lo, 0x04010400u), q1 = __builtin_amdgcn_perm(c43, lo, 0x04030402u);
                const uint32_t q2 = __builtin_amdgcn_perm(c43, hi, 0x04010400u), q3 = __builtin_amdgcn_perm(c43, hi, 0x04030402u);
                d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xp[wi * 4 + 0]), __builtin_bit_cast(bf16x2_t, q0), d, false);
                d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xp[wi * 4 + 1]), __builtin_bit_cast(bf16x2_t, q1), d, false);
                d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xp[wi * 4 + 2]), __builtin_bit_cast(bf16x2_t, q2), d, false);
                d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xp[wi * 4 + 3]), __builtin_bit_cast(bf16x2_t, q3), d, false);
            }
            bia = fmaf(-128.0f, scl, bia);
            acc[r] = fmaf(scl, d, acc[r]);
            acc[r] = fmaf(bia, xsm, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (o_row0 + r < a.o_rows) a.o_out[o_row0 + r] = f32_to_bf16(bf16_to_f32(o_res[r]) + round_bf16(acc[r]));
    }
    if (tr && threadIdx.x == 0) tr[6] = wall_clock64();
}


template <int D, int GT, bool TRACE, int NVW, bool F16 = false>
__global__ __launch_bounds__(kBlock, 1) void attn_step_kernel(const AttnStepArgs a) {
    static_assert(!F16 || NVW == 0, "the O projection rides in the launch for bfloat16 models only");
    typedef Act16<F16> A16;
    constexpr int LPR = D / 8;            // lanes per K/V row
    constexpr int TPW = 64 / LPR;         // token rows per wave-instruction == one unit
    constexpr bool QO = NVW >= 100;       // 100 + n: the O matrix is 4-bit packed (oproj_phase_q4)
    constexpr int NV = QO ? NVW - 100 : NVW;
    constexpr bool OPROJ = NV > 0;        // K = H * D = NV * 512
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sm_o = reinterpret_cast<float*>(smem);                 // [kWaves][TPW][GT][D]
    float* sm_m = sm_o + kWaves * TPW * GT * D;                   // [kWaves][GT]
    float* sm_l = sm_m + kWaves * GT;                             // [kWaves][GT]
    u32x4* sm_q = reinterpret_cast<u32x4*>(sm_l + kWaves * GT);   // [GT + 1][LPR]: roped q heads and the new k row, packed bf16
    u32x4* sm_x = sm_q + (GT + 1) * LPR;                          // OPROJ: [NV * 64] the attention vector, packed bf16
    float* sm_xsum = reinterpret_cast<float*>(sm_x + NV * 64);    // QO: [NV * 16] sums of 32 consecutive elements

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane % LPR;             // 8-element chunk of the head dim owned by this lane
    const int sg = lane / LPR;            // token row inside the unit
    const int kvh = blockIdx.x, split = blockIdx.y;
    const int G = a.H / a.Hkv;
    unsigned long long* tr = TRACE ? a.trace + ((size_t)split * a.Hkv + kvh) * 8 : nullptr;
    if (TRACE && threadIdx.x == 0) tr[0] = wall_clock64();

    bf16_t* Kb = a.k + (size_t)kvh * a.kv_head_stride;
    bf16_t* Vb = a.v + (size_t)kvh * a.kv_head_stride;
    const int t_begin = split * a.chunk;
    const int n_units = a.chunk / TPW;

    // ---- ONE round of loads: nothing below depends on the position ----
    const bf16_t* kraw = a.qkv + (size_t)a.H * D + (size_t)kvh * D;
    const bool does_q = wave < GT, does_k = wave == GT % kWaves;
    // the row this wave normalises: query head `wave` (clamped) or, for the key wave without a query head, the new key row
    const bf16_t* my_raw = does_q ? a.qkv + (size_t)(kvh * G + min(wave, G - 1)) * D : kraw;
    const u32x4 raw0 = *reinterpret_cast<const u32x4*>(my_raw + c * 8);
    const u32x4 raw1 = *reinterpret_cast<const u32x4*>(kraw + c * 8);                        // (key wave that also owns a head)
    const u32x4 vnew = *reinterpret_cast<const u32x4*>(kraw + (size_t)a.Hkv * D + c * 8);
    const int i0 = (c % (LPR / 2)) * 8;
    const f32x4* cp = reinterpret_cast<const f32x4*>(a.rope_cur + i0);
    const f32x4* sp = reinterpret_cast<const f32x4*>(a.rope_cur + D / 2 + i0);
    const f32x4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
    u32x4 wq_raw = {0, 0, 0, 0}, wk_raw = {0, 0, 0, 0};
    if (a.q_norm_w) {
        wq_raw = *reinterpret_cast<const u32x4*>(a.q_norm_w + c * 8);
        wk_raw = *reinterpret_cast<const u32x4*>(a.k_norm_w + c * 8);
    }
    u32x4 kr[kKU], vr[kKU];
    auto issue_kv = [&](int u0) {
#pragma unroll
        for (int u = 0; u < kKU; ++u) {
            const int tc = min(t_begin + (u0 + u * kWaves) * TPW + sg, a.cap - 1);   // rows past the position are masked below
            kr[u] = *reinterpret_cast<const u32x4*>(Kb + (size_t)tc * D + c * 8);
            vr[u] = *reinterpret_cast<const u32x4*>(Vb + (size_t)tc * D + c * 8);
        }
    };
    issue_kv(wave);
    const int pos = *a.pos_ptr;                                    // tokens already cached == RoPE offset (model.rs:186-194)
    const unsigned tag = *a.seq_ptr * a.tag_mul + a.tag_add;

    // ---- per-head RMSNorm (optional) + RoPE with the reference's bf16 roundings: one row per wave, shared through LDS ----
    {
        float cs[8], sn[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cs[e] = c0[e]; cs[4 + e] = c1[e];
            sn[e] = s0[e]; sn[4 + e] = s1[e];
        }
        const bool first_half = c < LPR / 2;
        auto norm_rope = [&](const u32x4 raw, const u32x4 w_raw) -> u32x4 {
            float x[8], w[8];
            unpack8<F16>(raw, x);
            unpack8<F16>(w_raw, w);
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = fmaf(x[e], x[e], ss);
            ss = group_sum<LPR>(ss);
            const float rstd = a.q_norm_w ? 1.0f / sqrtf(ss / (float)D + a.eps) : 1.0f;   // no q/k norm (Mixtral, Qwen2): x goes to RoPE as it is
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xn = a.q_norm_w ? A16::rnd(x[e] * rstd * w[e]) : x[e];   // RMSNorm output is rounded to the model's dtype
                const float other = swap_halves<LPR>(xn);                               // element i +- D/2
                y[e] = first_half ? xn * cs[e] - other * sn[e] : other * sn[e] + xn * cs[e];
            }
            u32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = A16::pack(y[2 * e], y[2 * e + 1]);     // RoPE output likewise
            return out;
        };
        if (does_q && sg == 0) sm_q[wave * LPR + c] = norm_rope(raw0, wq_raw);
        if (does_k && sg == 0) sm_q[GT * LPR + c] = norm_rope(does_q ? raw1 : raw0, wk_raw);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: the K/V loads stay in flight across it
    u32x4 q[GT];
#pragma unroll
    for (int g = 0; g < GT; ++g) q[g] = sm_q[g * LPR + c];
    const u32x4 knew = sm_q[GT * LPR + c];

    const int Tk = pos + 1;
    const int n_active = (Tk + a.chunk - 1) / a.chunk;                  // splits that own at least one token
    const int t_end = min(Tk, t_begin + a.chunk);
    const bool active = split < n_active;
    if (TRACE && threadIdx.x == 0) tr[1] = wall_clock64();

    float m[GT], l[GT];
    f32x2 o[GT][4];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[g][e] = f32x2{0.f, 0.f};
    }
    // one round = the kKU units a wave holds in registers.  The FIRST round is straight-line code ahead of the loop: hipcc's waitcnt
    // pass counts loads exactly there, so its waits on the K/V rows do not cover the O-projection weights issued after them (inside
    // the loop it falls back to small counts that drain everything older: own-chunk time went from 1.2 to 2.9 us)
    auto round = [&](const int u0) {
        float s[kKU][GT];
        f32x2 vf[kKU][4];
#pragma unroll
        for (int u = 0; u < kKU; ++u) {
            const int unit = u0 + u * kWaves;
            const int tok = t_begin + unit * TPW + sg;
            const bool live = unit < n_units && tok < t_end;
            u32x4 kp = kr[u], vp = vr[u];
            if (tok == pos && unit < n_units) {
                // this lane group owns the NEW token: use the row built above and append it to the cache (cache.rs:183-188)
                kp = knew;
                vp = vnew;
                *reinterpret_cast<u32x4*>(Kb + (size_t)pos * D + c * 8) = kp;
                *reinterpret_cast<u32x4*>(Vb + (size_t)pos * D + c * 8) = vp;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)   // a row past the position: its p is 0, but 0 * garbage must stay 0
                vf[u][e] = live ? f32x2{A16::lo(vp[e]), A16::hi(vp[e])} : f32x2{0.f, 0.f};
#pragma unroll
            for (int g = 0; g < GT; ++g) {
                const float d = group_sum<LPR>(dot8_16<F16>(q[g], kp)) * a.scale;
                s[u][g] = live ? d : -INFINITY;
            }
        }
        if (u0 + kWaves * kKU < n_units && t_begin + (u0 + kWaves * kKU) * TPW < t_end) issue_kv(u0 + kWaves * kKU);
#pragma unroll
        for (int g = 0; g < GT; ++g) {                  // one running max per head for the whole wave
            float mx = s[0][g];
#pragma unroll
            for (int u = 1; u < kKU; ++u) mx = fmaxf(mx, s[u][g]);
            float wmx = readlane_f(mx, 0);
#pragma unroll
            for (int r = 1; r < TPW; ++r) wmx = fmaxf(wmx, readlane_f(mx, r * LPR));
            const float mn = fmaxf(m[g], wmx);
            const float alpha = (mn == -INFINITY) ? 1.f : __expf(m[g] - mn);
            m[g] = mn;
            l[g] *= alpha;
            const f32x2 al = {alpha, alpha};
#pragma unroll
            for (int e = 0; e < 4; ++e) o[g][e] *= al;
#pragma unroll
            for (int u = 0; u < kKU; ++u) {
                const float p = (mn == -INFINITY) ? 0.f : __expf(s[u][g] - mn);
                l[g] += p;
                const f32x2 pp = {p, p};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[g][e] = __builtin_elementwise_fma(pp, vf[u][e], o[g][e]);
            }
        }
    };
    if (wave < n_units && t_begin + wave * TPW < t_end) round(wave);
    for (int u0 = wave + kWaves * kKU; u0 < n_units && t_begin + u0 * TPW < t_end; u0 += kWaves * kKU) round(u0);
    if (TRACE && threadIdx.x == 0) tr[2] = wall_clock64();

    if (active) {
        // ---- every token row of every wave parks its partial in LDS (same m inside a wave: plain sums) ----
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            float* dst = sm_o + (((size_t)(wave * TPW + sg) * GT + g) * D + c * 8);
            *reinterpret_cast<f32x4*>(dst) = f32x4{o[g][0][0], o[g][0][1], o[g][1][0], o[g][1][1]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[g][2][0], o[g][2][1], o[g][3][0], o[g][3][1]};
            float lw = readlane_f(l[g], 0);   // the LPR lanes of a row hold identical l
#pragma unroll
            for (int r = 1; r < TPW; ++r) lw += readlane_f(l[g], r * LPR);
            if (lane == 0) {
                sm_m[wave * GT + g] = m[g];
                sm_l[wave * GT + g] = lw;
            }
        }
        __syncthreads();
        if (TRACE && threadIdx.x == 0) tr[7] = wall_clock64();
        // ---- merge the 8 waves x TPW rows; the split's partial leaves as tagged granules ----
        for (int idx = threadIdx.x; idx < G * D; idx += kBlock) {
            const int g = idx / D, d = idx % D;
            float M = sm_m[g];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) M = fmaxf(M, sm_m[w * GT + g]);
            float L = 0.f, O = 0.f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                const float mw = sm_m[w * GT + g];
                const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
                float ow = 0.f;
#pragma unroll
                for (int r = 0; r < TPW; ++r) ow += sm_o[((size_t)(w * TPW + r) * GT + g) * D + d];
                L = fmaf(f, sm_l[w * GT + g], L);
                O = fmaf(f, ow, O);
            }
            uint64_t* gr = a.ws + ((size_t)(kvh * G + g) * a.nsplit + split) * (D + 2);
            st_granule(gr + d, tag, O);
            if (d == 0) {
                st_granule(gr + D, tag, M);
                st_granule(gr + D + 1, tag, L);
            }
        }
    }
    if (TRACE && threadIdx.x == 0) tr[3] = wall_clock64();

    // ---- OPROJ: the blocks that are not consumers (split >= G) share the output rows; rows per wave is a compile-time constant of the
    //      phase (a runtime-predicated load block per row made hipcc drain the queue between rows: three serial round trips) ----
    if (OPROJ && split >= G) {
        // wave g of the (blocks - H) * 8 producer waves: the first o_nhi waves take o_rpw rows, the others o_rpw - 1 -- together exactly
        // o_rows (with o_rpw rows everywhere the surplus waves re-read the last row: 44 MB of traffic for a 33.6 MB matrix)
        const int g = ((split - G) * (int)gridDim.x + kvh) * kWaves + wave;
        const bool hi = g < a.o_nhi;
        const int rows = hi ? a.o_rpw : a.o_rpw - 1;
        const int o_row0 = hi ? g * a.o_rpw : a.o_nhi * a.o_rpw + (g - a.o_nhi) * (a.o_rpw - 1);
        unsigned long long* otr = TRACE ? tr : nullptr;
        if constexpr (QO) {
            if (rows <= 1) oproj_phase_q4<NV, 1>(a, sm_x, sm_xsum, tag, lane, wave, o_row0, otr);
            else if (rows == 2) oproj_phase_q4<NV, 2>(a, sm_x, sm_xsum, tag, lane, wave, o_row0, otr);
            else if (rows == 3) oproj_phase_q4<NV, 3>(a, sm_x, sm_xsum, tag, lane, wave, o_row0, otr);
            else oproj_phase_q4<NV, 4>(a, sm_x, sm_xsum, tag, lane, wave, o_row0, otr);
        } else {
            if (rows <= 1) oproj_phase<(NV > 0 ? NV : 1), 1>(a, sm_x, tag, lane, wave, o_row0, otr);
            else if (rows == 2) oproj_phase<(NV > 0 ? NV : 1), 2>(a, sm_x, tag, lane, wave, o_row0, otr);
            else if (rows == 3) oproj_phase<(NV > 0 ? NV : 1), 3>(a, sm_x, tag, lane, wave, o_row0, otr);
            else oproj_phase<(NV > 0 ? NV : 1), 4>(a, sm_x, tag, lane, wave, o_row0, otr);
        }
        return;
    }

    // ---- consumers: block (kvh, j < G) merges head kvh*G + j ----
    if (split < G && wave < D / 64) {
        const int head = kvh * G + split;
        const uint64_t* base = a.ws + (size_t)head * a.nsplit * (D + 2);
        bf16_t* out = a.out + (size_t)head * D;
        const int dim = wave * 64 + lane;
        const int nb = (a.nsplit + 15) / 16;
        uint64_t* xg_head = OPROJ ? a.xg + (size_t)head * (D / 2) : nullptr;
        if (nb <= 1) gather_head<D, 1, F16>(a, base, n_active, tag, lane, dim, out, xg_head);
        else if (nb == 2) gather_head<D, 2, F16>(a, base, n_active, tag, lane, dim, out, xg_head);
        else gather_head<D, 3, F16>(a, base, n_active, tag, lane, dim, out, xg_head);
        if (TRACE && threadIdx.x == 0) tr[4] = wall_clock64();
    }

}

}  // namespace

int attn_step_block_tokens(int D) { return 64 / (D / 8); }   // granularity of a split's token range: one wave-instruction

// token range per split and split count for a context bucket of `tk_max` tokens: one block per CU (<= 256 blocks: every block
// of the launch is resident, the consumers cannot starve a producer), at most kMaxSplits splits (the consumer gathers 3
// batches of 16), at least G (one consumer block per query head of a KV group)
constexpr int kMaxSplits = 48;
void attn_step_plan(int tk_max, int Hkv, int G, int D, int* chunk, int* nsplit) {
    const int unit = attn_step_block_tokens(D);
    int target = std::max(1, std::min(kMaxSplits, 256 / std::max(Hkv, 1)));
    if (const char* v = getenv("OMX_ATTN_STEP_SPLITS")) target = std::max(1, std::min(kMaxSplits, atoi(v)));
    int ch = unit * ((tk_max + unit * target - 1) / (unit * target));
    if (ch < unit) ch = unit;
    int ns = (tk_max + ch - 1) / ch;
    if (ns < G) ns = G;
    *chunk = ch;
    *nsplit = ns;
}

size_t attn_step_ws_granules(int H, int D) { return (size_t)H * kMaxSplits * (D + 2); }

// rows per wave when the waves of the non-consumer blocks (split >= G: blocks - H of them) share the O projection's output rows
static int oproj_rows_per_wave(int H, int Hkv, int nsplit, int o_rows) {
    const int waves = (Hkv * nsplit - H) * kWaves;
    return waves > 0 ? (o_rows + waves - 1) / waves : 1 << 20;
}
// the O projection can ride in the launch when K = H * D is one of the register layouts (NVW x 512, NVW in {1, 2, 4, 7, 8}: the
// GEMV instantiations gemv_kernel<NVW, 1, ...> whose arithmetic the phase reproduces; 7 = Qwen2.5-7B's 28 heads of 128) and at most
// kORows rows per wave of the non-consumer blocks cover the output
bool attn_step_oproj_ok(int H, int Hkv, int D, int nsplit, int o_rows) {
    const int K = H * D, nvw = K / 512;
    return K % 512 == 0 && (nvw == 1 || nvw == 2 || nvw == 4 || nvw == 7 || nvw == 8) && oproj_rows_per_wave(H, Hkv, nsplit, o_rows) <= kORows;
}

// ... on a 4-bit packed matrix: K a multiple of 2048 (quant.hip's four-word lane chunk) with the register layouts 4 and 8
bool attn_step_oproj_q4_ok(int H, int Hkv, int D, int nsplit, int o_rows, int group) {
    const int K = H * D;
    return (K == 2048 || K == 4096) && (group == 32 || group == 64 || group == 128) && attn_step_oproj_ok(H, Hkv, D, nsplit, o_rows);
}

int launch_attn_step(const AttnStepArgs& a_in, int D, hipStream_t s) {
    AttnStepArgs a = a_in;
    const int G = a.H / a.Hkv;
    OMX_REQUIRE(a.H % a.Hkv == 0 && G >= 1 && G <= 8, "decode attention: %d query heads over %d KV heads unsupported (group of at most 8)", a.H, a.Hkv);
    OMX_REQUIRE(a.nsplit >= G && a.nsplit <= kMaxSplits && a.chunk > 0 && a.chunk % attn_step_block_tokens(D) == 0 && a.Hkv * a.nsplit <= 256,
                "decode attention: bad split plan (chunk %d, %d splits, group %d, %d kv heads)", a.chunk, a.nsplit, G, a.Hkv);
    OMX_REQUIRE(a.ws && a.rope_cur && a.pos_ptr && a.seq_ptr && a.abort_flag && a.tag_mul > a.tag_add - 1u && a.tag_add >= 1u,
                "decode attention: missing step state");
    int nvw = 0;
    const bool qo = a.o_wq != nullptr;
    if (qo)
        OMX_REQUIRE(!a.o_w && a.o_sb && a.o_out && !a.o_out_f32 && attn_step_oproj_q4_ok(a.H, a.Hkv, D, a.nsplit, a.o_rows, a.o_group),
                    "decode attention + packed O projection: shape does not qualify (H*D = %d, group %d)", a.H * D, a.o_group);
    if (a.o_w || qo) {
        OMX_REQUIRE(a.o_resid && (a.o_out || a.o_out_f32) && a.xg && attn_step_oproj_ok(a.H, a.Hkv, D, a.nsplit, a.o_rows),
                    "decode attention + O projection: shape does not qualify (H*D = %d, %d rows, %d blocks)", a.H * D, a.o_rows, a.Hkv * a.nsplit);
        nvw = a.H * D / 512;
        a.o_rpw = oproj_rows_per_wave(a.H, a.Hkv, a.nsplit, a.o_rows);
        const int waves = (a.Hkv * a.nsplit - a.H) * kWaves;
        // o_nhi waves with o_rpw rows + the rest with o_rpw - 1 = o_rows; with one row per wave at most (o_rpw == 1) every wave keeps
        // its (possibly clamped) row: a wave of the phase always holds at least one
        a.o_nhi = a.o_rpw > 1 ? a.o_rows - waves * (a.o_rpw - 1) : waves;
    }
    const dim3 grid(a.Hkv, a.nsplit), block(kBlock);
    const int gt = G <= 1 ? 1 : G <= 2 ? 2 : G <= 4 ? 4 : 8;
#define OMX_ATTN_LAUNCH(DD, GG, TT, NN)                                                                                  \
    {                                                                                                                    \
        if (shmem > 48 * 1024)                                                                                           \
            OMX_HIP_CHECK(hipFuncSetAttribute((const void*)attn_step_kernel<DD, GG, TT, NN>,                            \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));                  \
        OMX_LAUNCH_TIMED((attn_step_kernel<DD, GG, TT, NN>), grid, block, shmem, s, a);                                  \
        OMX_LAUNCH_CHECK();                                                                                              \
        return 0;                                                                                                        \
    }
#define OMX_ATTN_STEP_CASE(DD, GG)                                                                                       \
    if (D == DD && gt == GG) {                                                                                           \
        const size_t shmem = ((size_t)kWaves * (64 / (DD / 8)) * GG * DD + 2 * kWaves * GG) * sizeof(float) +            \
                             (size_t)(GG + 1) * (DD / 8) * 16 + (size_t)nvw * 64 * 16 + (qo ? (size_t)nvw * 64 : 0);     \
        if (a.trace && !qo) {   /* timeline builds: the plain kernel and the widest fused one */                        \
            if (nvw == 0) OMX_ATTN_LAUNCH(DD, GG, true, 0)                                                               \
            if (nvw == 8) OMX_ATTN_LAUNCH(DD, GG, true, 8)                                                               \
            return set_error("decode attention: no traced instantiation for H*D = %d", a.H * DD);                       \
        }                                                                                                                \
        if (qo) {                                                                                                        \
            if (nvw == 4) OMX_ATTN_LAUNCH(DD, GG, false, 104)                                                            \
            OMX_ATTN_LAUNCH(DD, GG, false, 108)                                                                          \
        }                                                                                                                \
        if (a.f16) {   /* a float16 checkpoint's model: float16 q / k / v, cache and output; D = 128, no O projection in the launch */ \
            if constexpr (DD == 128) {                                                                                   \
                if (nvw == 0) {                                                                                          \
                    if (shmem > 48 * 1024)                                                                               \
                        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)attn_step_kernel<DD, GG, false, 0, true>,        \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));      \
                    OMX_LAUNCH_TIMED((attn_step_kernel<DD, GG, false, 0, true>), grid, block, shmem, s, a);              \
                    OMX_LAUNCH_CHECK();                                                                                  \
                    return 0;                                                                                            \
                }                                                                                                        \
            }                                                                                                            \
            return set_error("decode attention: float16 models take head_dim 128 without the O projection in the launch"); \
        }                                                                                                                \
        if (nvw == 0) OMX_ATTN_LAUNCH(DD, GG, false, 0)                                                                  \
        if (nvw == 1) OMX_ATTN_LAUNCH(DD, GG, false, 1)                                                                  \
        if (nvw == 2) OMX_ATTN_LAUNCH(DD, GG, false, 2)                                                                  \
        if (nvw == 4) OMX_ATTN_LAUNCH(DD, GG, false, 4)                                                                  \
        if (nvw == 7) OMX_ATTN_LAUNCH(DD, GG, false, 7)                                                                  \
        OMX_ATTN_LAUNCH(DD, GG, false, 8)                                                                                \
    }
    OMX_ATTN_STEP_CASE(128, 1) OMX_ATTN_STEP_CASE(128, 2) OMX_ATTN_STEP_CASE(128, 4) OMX_ATTN_STEP_CASE(128, 8)
    OMX_ATTN_STEP_CASE(64, 1) OMX_ATTN_STEP_CASE(64, 2) OMX_ATTN_STEP_CASE(64, 4) OMX_ATTN_STEP_CASE(64, 8)
#undef OMX_ATTN_STEP_CASE
#undef OMX_ATTN_LAUNCH
    return set_error("decode attention: head_dim %d unsupported (64 or 128)", D);
}

}  // namespace omx
