// SDPA with Tq > 1 on the gfx950 matrix cores: flash-attention forward (prefill, DiT joint
// attention, Paraformer-sized encoders).
//   reference: mlx_fast_scaled_dot_product_attention (mlx-c fast.h:189-198; mlx-rs/src/fast.rs:121-151)
//   as called by mlx_rs_core::scaled_dot_product_attention (utils.rs:191-209) with mask = explicit
//   bool array at prefill (utils.rs:134-153, qwen3-mlx/src/model.rs:401), "causal", or none
//   (FLUX joint attention, flux-klein-mlx/src/klein_model.rs:474-483, which the reference computes by
//   materialising the [H,S,S] scores).  Scores never leave registers here.
//
// Mapping (wave64, MFMA 16x16x32 bf16, fp32 accumulate):
//   * block = 4 waves = 64 query rows (16 per wave) of one (batch, head); KV tiles of 64 keys staged
//     in LDS by LDS-DMA (global_load_lds: no VGPR round trip), both row-major with an XOR swizzle applied on
//     the SOURCE side (the DMA image is lane-linear): K in 16-B chunks for ds_read_b128, V in 32-B blocks for
//     ds_read_b64_tr_b16, the hardware transpose read that hands each lane 4 keys of ONE head-dim column --
//     the A operand of the second product -- out of a [4 keys][16 dims] block;
//   * "swapped" products: S^T = K Q^T and O^T = V^T P^T.  In both C layouts a lane's column is its
//     query row (lane & 15), so the online-softmax state (m, l) and the O rescale are lane-local;
//     row max / sum across the 4 lanes that share a query use v_permlane16/32_swap (no LDS);
//   * P goes from the S^T accumulator straight into the B operand of the second product: the MFMA
//     contraction index is permuted identically on the V^T side, so no transpose of P is needed;
//   * softmax in fp32 (fast.rs:116); P is rounded to bf16 for the second product.
#include <stdlib.h>

#include "gemm.hpp"
#include "act16.hpp"
#include <type_traits>
#include "gridsync.hpp"
#include <map>
#include <mutex>
#include <thread>

namespace omx {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4v = __attribute__((ext_vector_type(4))) float;
using u32x2v = __attribute__((ext_vector_type(2))) uint32_t;

constexpr int KB = 64;    // keys per LDS tile

struct PrefillArgs {
    const bf16_t *q, *k, *v;
    bf16_t* out;
    int B, H, Hkv, Tq, Tk;
    int64_t kv_batch_stride, kv_head_stride;
    float scale;
    int mask_mode;
    const void* mask;
    // element strides of q and out: [B,H,Tq,D] by default; the engine writes out as [Tq, H*D]
    int64_t q_bs, q_hs, q_ts, o_bs, o_hs, o_ts;
    int64_t kv_ts;   // elements between consecutive key rows (D when K/V are [.., T, D] contiguous)
    unsigned long long* trace;   // two-phase kernel, timeline build: [block][wave][8] cycle sums (tools/attn_pp_trace.py)
    float* sk_ws;                // stream-K form: parked pieces [share boundary][2][512 lanes][QW * (NDT * 4 + 2)] f32
    unsigned* sk_cnt;            // ... arrival counters [share boundary] (zero between launches)
};

// max / sum over the 4 lanes {l, l^16, l^32, l^48} that hold the same query column
__device__ __forceinline__ float quad_rows_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_rows_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// QW = 16-row query sub-tiles per wave (2: a wave owns 32 query rows and every K / V^T fragment read from
// LDS feeds two MFMAs).  Next tile's K/V are fetched into registers while the current tile is multiplied
// (issue-early / write-late staging, cdna_hip_programming.md T14).
// F16: float16 operands and result (a float16 checkpoint's prompt pass): the same data movement -- every element is 16 bits -- with
// the float16 MFMA, P rounded to float16 for the second product and the output rounded once to float16.
template <int D, int QW, int MASK, bool F16 = false>
__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(const PrefillArgs a) {
    typedef Act16<F16> A16;
    using h8 = __attribute__((ext_vector_type(8))) _Float16;
    typedef typename std::conditional<F16, h8, bf16x8>::type p8;     // P fragments in the operand type
    typedef typename std::conditional<F16, _Float16, __bf16>::type pe;
    auto mfma = [](bf16x8 x, p8 y, f32x4v c) {
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, x), y, c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
    };
    constexpr int DC = D / 8;       // 16-B chunks per K row
    constexpr int NI = D / 32;      // MFMA k-steps over the head dim
    constexpr int NDT = D / 16;     // 16-wide output tiles over the head dim
    constexpr int QBLK = 64 * QW;   // query rows per block
    constexpr int KCH = KB * DC / 256;          // K chunks staged per thread
    // two buffers each: tile t+1 is staged while tile t is multiplied, ONE barrier per tile
    __shared__ __attribute__((aligned(16))) bf16_t sK2[2][KB * D];
    __shared__ __attribute__((aligned(16))) bf16_t sV2[2][KB * D];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qcol = lane & 15, rg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kvh = h / (a.H / a.Hkv);
    // causal: the last query blocks see the most keys.  Two blocks share a CU, and the dispatcher hands block L and block
    // L + 256 to the same CU: rows of the grid that start in an even 256-block chunk run longest-first, rows in an odd chunk
    // shortest-first, so a CU's two blocks add up to the same number of key tiles (16 + 1, 15 + 2, ...) instead of 2 x 16 on
    // one CU and 2 x 1 on another; a multi-round grid still starts with long blocks.
    int qt = blockIdx.x;
    if (MASK == OMX_MASK_CAUSAL) {
        const int row_start = (int)gridDim.x * (int)(blockIdx.y + gridDim.y * blockIdx.z);
        if (((row_start >> 8) & 1) == 0) qt = (int)gridDim.x - 1 - qt;
    }
    const int q0 = qt * QBLK;
    const int shift = a.Tk - a.Tq;                       // causal: query i sees keys <= i + shift

    const bf16_t* Kb = a.k + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;
    const bf16_t* Vb = a.v + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;

    int qrow[QW], qrow_c[QW];
    bf16x8 qf[QW][NI];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        qrow[w] = q0 + (wave * QW + w) * 16 + qcol;      // this lane's query row in sub-tile w
        qrow_c[w] = min(qrow[w], a.Tq - 1);
        const bf16_t* Qp = a.q + (size_t)b * a.q_bs + (size_t)h * a.q_hs + (size_t)qrow_c[w] * a.q_ts;
#pragma unroll
        for (int i = 0; i < NI; ++i) qf[w][i] = *reinterpret_cast<const bf16x8*>(Qp + i * 32 + rg * 8);
    }

    f32x4v o[QW][NDT];
    float m_run[QW], l_run[QW];   // l_run: this lane's PARTIAL sum (reduced at the end)
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        m_run[w] = -INFINITY;
        l_run[w] = 0.f;
#pragma unroll
        for (int t = 0; t < NDT; ++t) o[w][t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }

    int kv_end = a.Tk;
    if (MASK == OMX_MASK_CAUSAL) kv_end = max(0, min(a.Tk, q0 + QBLK + shift));

    // ---- staging.  K: global_load_lds (no VGPR round trip; the LDS image is lane-linear, so the bank-conflict
    //      swizzle -- 16-B chunk index ^= key & (DC-1) -- is applied to the per-lane SOURCE address).
    //      V: (key 2p, key 2p+1) chunk pairs through registers, written transposed ([d][key]). ----
    // per-thread source offsets are loop invariants (elements from the tile's first key row); a tile that lies
    // inside the sequence adds a wave-uniform base (SALU), only the last partial tile recomputes clamped rows
    constexpr int VSH = (D == 128) ? 0 : 1;           // keys per 256 B of LDS = 1 << VSH
    constexpr int VBM = D / 16 - 1;                   // 32-B blocks per row - 1
    uint32_t koff[KCH], voff[KCH];
#pragma unroll
    for (int it = 0; it < KCH; ++it) {
        const int ci = threadIdx.x + it * 256;
        const int row = ci / DC;
        koff[it] = (uint32_t)(row * a.kv_ts + ((ci % DC) ^ (row & (DC - 1))) * 8);
        voff[it] = (uint32_t)(row * a.kv_ts + ((ci % DC) ^ (((row >> VSH) & VBM) << 1)) * 8);
    }
    // K: 16-B chunk index ^= key & (DC-1) (ds_read_b128 of 16 consecutive key rows).  V: 32-B block index ^= the key
    // row's position among the 8 rows that two 16-lane groups of one ds_read_b64_tr_b16 touch
    // The DMA goes out through an asm statement: hipcc counts a builtin LDS-DMA like a load and puts s_waitcnt vmcnt(0) in front of
    // the next ds_read it cannot prove disjoint -- the first K fragment read of the CURRENT tile waited for the NEXT tile to land
    // (cdna_hip_programming.md 5.7 item 1).  The waits are ours: vmcnt(0) ahead of the barrier that ends the tile.
    const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma16 = [&](const bf16_t* gsrc, unsigned lds_dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    };
    auto stage = [&](int k0, const bf16_t* src, const uint32_t (&off)[KCH], bf16_t* dst, bool is_v) {
        const unsigned d0 = (unsigned)(uintptr_t)dst + wave_u * 1024u;
        if (k0 + KB <= a.Tk) {
            const bf16_t* base = src + (size_t)k0 * a.kv_ts;
#pragma unroll
            for (int it = 0; it < KCH; ++it) dma16(base + off[it], d0 + it * 4096u);
            return;
        }
#pragma unroll
        for (int it = 0; it < KCH; ++it) {
            const int ci = threadIdx.x + it * 256;
            const int row = ci / DC;
            const int ch = (ci % DC) ^ (is_v ? (((row >> VSH) & VBM) << 1) : (row & (DC - 1)));
            const int key = min(k0 + row, a.Tk - 1);
            dma16(src + (size_t)key * a.kv_ts + ch * 8, d0 + it * 4096u);
        }
    };
    auto stage_k = [&](int k0, bf16_t* sK) { stage(k0, Kb, koff, sK, false); };
    auto stage_v = [&](int k0, bf16_t* sV) { stage(k0, Vb, voff, sV, true); };
    // transpose-read addressing: lane (i = lane & 15, rg) of fragment (key group j2, d-tile t) supplies the address of
    // V[j2*16 + rg*4 + i/4][t*16 + (i&3)*4 ..+3]; the hardware returns V[j2*16 + rg*4 + 0..3][t*16 + i] to it
    const int v_key = rg * 4 + (qcol >> 2);
    const int v_sw = (v_key >> VSH) & VBM;
    const int v_lane_off = v_key * D + (qcol & 3) * 4;   // elements
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

    if (kv_end > 0) {
        stage_k(0, sK2[0]);
        stage_v(0, sV2[0]);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    int buf = 0;
    for (int k0 = 0; k0 < kv_end; k0 += KB, buf ^= 1) {
        const bf16_t* sK = sK2[buf];
        const bf16_t* sV = sV2[buf];
        const bool more = k0 + KB < kv_end;
        if (more) {   // next tile: K and V straight into the other LDS buffers, in flight under both products
            stage_k(k0 + KB, sK2[buf ^ 1]);
            stage_v(k0 + KB, sV2[buf ^ 1]);
        }

        // ---- S^T = K Q^T for 4 key tiles of 16, both query sub-tiles share each K fragment ----
        f32x4v s[QW][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int w = 0; w < QW; ++w) s[w][kt] = f32x4v{0.f, 0.f, 0.f, 0.f};
        // head-dim step outermost: 4 x QW independent accumulators sit between two MFMAs on the same one; the K
        // fragments of step i+1 are read while step i is multiplied (explicit two-deep register pipeline)
        bf16x8 kf[2][4];
        auto read_k = [&](int i, bf16x8 (&dst)[4]) {
            const int ch = i * 4 + rg;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int row = kt * 16 + qcol;   // A operand: lane & 15 indexes the key row
                dst[kt] = *reinterpret_cast<const bf16x8*>(&sK[(row * DC + (ch ^ (row & (DC - 1)))) * 8]);
            }
        };
        read_k(0, kf[0]);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + 1 < NI) read_k(i + 1, kf[(i + 1) & 1]);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int w = 0; w < QW; ++w) s[w][kt] = mfma(kf[i & 1][kt], __builtin_bit_cast(p8, qf[w][i]), s[w][kt]);
        }
        // ---- online softmax in the base-2 domain: p = 2^(s*c - m), c = scale * log2(e), m = running max of s*c.
        //      (lane: query qcol of each sub-tile, keys kt*16 + rg*4 + r).  Tiles that need no masking -- all keys
        //      valid and, under a causal mask, entirely below every query row of the block -- take a path without
        //      any select; the normaliser sums the fp32 probabilities (P is rounded to bf16 only for the MFMA). ----
        p8 pf[QW][2];
        const float c2 = a.scale * 1.44269504088896340736f;
        bool plain = (k0 + KB <= a.Tk) && (MASK == OMX_MASK_NONE || (MASK == OMX_MASK_CAUSAL && k0 + KB - 1 <= q0 + shift));
        if (plain) {
#pragma unroll
            for (int w = 0; w < QW; ++w) {
                float mx = s[w][0][0];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[w][kt][r]);
                mx = quad_rows_max(mx) * c2;
                const float m_new = fmaxf(m_run[w], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run[w] - m_new);   // 2^(-inf) = 0 on the first tile
                m_run[w] = m_new;
                l_run[w] *= alpha;
#pragma unroll
                for (int t = 0; t < NDT; ++t) o[w][t] *= alpha;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(s[w][2 * j + (e >> 2)][e & 3], c2, -m_new));
                        l_run[w] += p;
                        pf[w][j][e] = (pe)p;
                    }
            }
        } else {
#pragma unroll
            for (int w = 0; w < QW; ++w) {
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + kt * 16 + rg * 4 + r;
                        // (scores stay raw unless an additive mask joins them: the exponent below is then the plain path's fma, so a
                        //  tile gives the same bits whichever path a block's geometry sends it down)
                        float v = MASK == OMX_MASK_ADDITIVE ? s[w][kt][r] * c2 : s[w][kt][r];
                        bool keep = key < a.Tk;
                        // MASK is a compile-time mode; the mask bytes are loaded unconditionally from a clamped address
                        if (MASK == OMX_MASK_CAUSAL) keep = keep && (key <= qrow[w] + shift);
                        if (MASK == OMX_MASK_BOOL) {
                            const uint8_t mb = reinterpret_cast<const uint8_t*>(a.mask)[(size_t)qrow_c[w] * a.Tk + min(key, a.Tk - 1)];
                            keep = keep & (mb != 0);
                        }
                        if (MASK == OMX_MASK_ADDITIVE)
                            v += 1.44269504088896340736f * A16::val(reinterpret_cast<const bf16_t*>(a.mask)[(size_t)qrow_c[w] * a.Tk + min(key, a.Tk - 1)]);
                        v = keep ? v : -INFINITY;
                        s[w][kt][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = quad_rows_max(mx);
                if (MASK != OMX_MASK_ADDITIVE) mx *= c2;
                const float m_new = fmaxf(m_run[w], mx);
                const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run[w] - m_new);
                m_run[w] = m_new;
                l_run[w] *= alpha;
#pragma unroll
                for (int t = 0; t < NDT; ++t) o[w][t] *= alpha;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float sv = s[w][2 * j + (e >> 2)][e & 3];
                        const float p = (m_new == -INFINITY) ? 0.f
                                        : __builtin_amdgcn_exp2f(MASK == OMX_MASK_ADDITIVE ? sv - m_new : fmaf(sv, c2, -m_new));
                        l_run[w] += p;
                        pf[w][j][e] = (pe)p;
                    }
            }
        }
        // ---- O^T += V^T P^T : k-slot (rg*8 + e) <-> key (2j + (e>>2))*16 + rg*4 + (e&3) on BOTH operands.
        //      16 V^T fragments (2 key halves x NDT d-tiles), each two transpose reads, four ahead of their MFMAs ----
        {
            constexpr int NF = 2 * NDT, AHEAD = 4;
            u32x4 vf[AHEAD];
            auto read_v = [&](int idx) {
                const int j = idx / NDT, t = idx % NDT;
                const bf16_t* p0 = sV + v_lane_off + ((t ^ v_sw) * 16) + (2 * j) * 16 * D;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)p0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(p0 + 16 * D));
                const u32x2v l2 = __builtin_bit_cast(u32x2v, lo), h2 = __builtin_bit_cast(u32x2v, hi);
                return u32x4{l2[0], l2[1], h2[0], h2[1]};
            };
#pragma unroll
            for (int idx = 0; idx < AHEAD; ++idx) vf[idx] = read_v(idx);
#pragma unroll
            for (int idx = 0; idx < NF; ++idx) {
                const int j = idx / NDT, t = idx % NDT;
                const u32x4 cur = vf[idx % AHEAD];
                if (idx + AHEAD < NF) vf[idx % AHEAD] = read_v(idx + AHEAD);
#pragma unroll
                for (int w = 0; w < QW; ++w)
                    o[w][t] = mfma(__builtin_bit_cast(bf16x8, cur), pf[w][j], o[w][t]);
            }
            }
        __builtin_amdgcn_s_waitcnt(0);   // the K tile of the next iteration has landed (vmcnt), our LDS traffic retired (lgkmcnt)
        __syncthreads();
    }

    // ---- normalise and store: lane holds out[qrow][t*16 + rg*4 + 0..3] ----
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        const float l_tot = quad_rows_sum(l_run[w]);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (qrow[w] < a.Tq) {
            bf16_t* op = a.out + (size_t)b * a.o_bs + (size_t)h * a.o_hs + (size_t)qrow[w] * a.o_ts;
#pragma unroll
            for (int t = 0; t < NDT; ++t) {
                u32x2v wv = {A16::pack(o[w][t][0] * inv, o[w][t][1] * inv), A16::pack(o[w][t][2] * inv, o[w][t][3] * inv)};
                *reinterpret_cast<u32x2v*>(op + t * 16 + rg * 4) = wv;
            }
        }
    }
}


// ---- Two-phase ("ping-pong") form for long sequences: ONE block of 8 waves per CU, 256 query rows, two waves per SIMD that run
//      in opposite phases.  A wave's tile step is split into X = {S_i = K_i Q^T, O += V_{i-1}^T P_{i-1}} (64 MFMAs, LDS reads) and
//      Y = {online softmax of S_i -> P_i, O rescale} (VALU only); waves 0-3 run X while waves 4-7 run Y and vice versa, every
//      phase ends in an s_barrier that all 8 waves meet.  Without it the co-resident waves of a SIMD drift into the same phase:
//      both queue for the matrix pipe, then both for the VALU (64 MFMAs = 1024 matrix cycles and ~1300 VALU cycles per tile and
//      wave ran as their SUM, MfmaUtil 0.32; see EXPERIMENTS.md R3-4).  MI355X_MICROARCH.md "Two waves per SIMD".
//      K_{i+1} and V_i are fetched by LDS-DMA at the start of even phases into the buffers whose last readers met the previous
//      barrier, and awaited (vmcnt) ahead of the barrier that ends the following odd phase: two phases of flight. ----
// One unit of the two-phase kernel: 256 query rows of head h, key tiles [t_first, t_last) (the whole unit unless SK).  Called by all 8
// waves of the block.  SK: a piece of a unit parks its un-normalised state and the unit's last piece merges (see below).
template <int D, int MASK, bool TR, bool SK>
__device__ __forceinline__ void pp_unit(const PrefillArgs& a, bf16_t* sK, bf16_t* sV, int b, int h, int qt, int cut, int t_first, int t_last) {
    constexpr int QW = 2;
    constexpr int DC = D / 8, NI = D / 32, NDT = D / 16;
    constexpr int QBLK = 8 * 16 * QW;
    constexpr int KCH = KB * DC / 512;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool late = wave_u >= 4;
    const int qcol = lane & 15, rg = lane >> 4;
    const int kvh = h / (a.H / a.Hkv);
    const int q0 = qt * QBLK;
    const int shift = a.Tk - a.Tq;

    const bf16_t* Kb = a.k + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;
    const bf16_t* Vb = a.v + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;

    int qrow[QW], qrow_c[QW];
    bf16x8 qf[QW][NI];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        qrow[w] = q0 + (wave * QW + w) * 16 + qcol;
        qrow_c[w] = min(qrow[w], a.Tq - 1);
        const bf16_t* Qp = a.q + (size_t)b * a.q_bs + (size_t)h * a.q_hs + (size_t)qrow_c[w] * a.q_ts;
#pragma unroll
        for (int i = 0; i < NI; ++i) qf[w][i] = *reinterpret_cast<const bf16x8*>(Qp + i * 32 + rg * 8);
    }
    f32x4v o[QW][NDT];
    float m_run[QW], l_run[QW];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        m_run[w] = -INFINITY;
        l_run[w] = 0.f;
#pragma unroll
        for (int t = 0; t < NDT; ++t) o[w][t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    int kv_end = a.Tk;
    if (MASK == OMX_MASK_CAUSAL) kv_end = max(0, min(a.Tk, q0 + QBLK + shift));
    const int nt = (kv_end + KB - 1) / KB;
    const int ta = SK ? t_first : 0, tb = SK ? t_last : nt;     // this call's tile range (stream-K: a piece of the unit)

    constexpr int VSH = (D == 128) ? 0 : 1;
    constexpr int VBM = D / 16 - 1;
    uint32_t koff[KCH], voff[KCH];
#pragma unroll
    for (int it = 0; it < KCH; ++it) {
        const int ci = threadIdx.x + it * 512;
        const int row = ci / DC;
        koff[it] = (uint32_t)(row * a.kv_ts + ((ci % DC) ^ (row & (DC - 1))) * 8);
        voff[it] = (uint32_t)(row * a.kv_ts + ((ci % DC) ^ (((row >> VSH) & VBM) << 1)) * 8);
    }
    auto dma16 = [&](const bf16_t* gsrc, unsigned lds_dst) {   // (asm: see attn_prefill_kernel)
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    };
    auto stage = [&](int k0, const bf16_t* src, const uint32_t (&off)[KCH], bf16_t* dst, bool is_v) {
        const unsigned d0 = (unsigned)(uintptr_t)dst + wave_u * 1024u;
        if (k0 + KB <= a.Tk) {
            const bf16_t* base = src + (size_t)k0 * a.kv_ts;
#pragma unroll
            for (int it = 0; it < KCH; ++it) dma16(base + off[it], d0 + it * 8192u);
            return;
        }
#pragma unroll
        for (int it = 0; it < KCH; ++it) {
            const int ci = threadIdx.x + it * 512;
            const int row = ci / DC;
            const int ch = (ci % DC) ^ (is_v ? (((row >> VSH) & VBM) << 1) : (row & (DC - 1)));
            const int key = min(k0 + row, a.Tk - 1);
            dma16(src + (size_t)key * a.kv_ts + ch * 8, d0 + it * 8192u);
        }
    };
    // tile step i, even phase: K_{i+1} and V_i set out
    auto fetch = [&](int i) {
        if (i + 1 < tb) stage((i + 1) * KB, Kb, koff, sK + ((i + 1) & 1) * (KB * D), false);
        if (i < tb) stage(i * KB, Vb, voff, sV + (i & 1) * (KB * D), true);
    };
    const int v_key = rg * 4 + (qcol >> 2);
    const int v_sw = (v_key >> VSH) & VBM;
    const int v_lane_off = v_key * D + (qcol & 3) * 4;
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

    f32x4v sc[QW][4];      // S_i^T, from X(i) to Y(i)
    bf16x8 pf[QW][2];      // P_i, from Y(i) to X(i + 1)
    const float c2 = a.scale * 1.44269504088896340736f;

    auto qk = [&](const bf16_t* sK) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int w = 0; w < QW; ++w) sc[w][kt] = f32x4v{0.f, 0.f, 0.f, 0.f};
        bf16x8 kf[2][4];
        auto read_k = [&](int i, bf16x8 (&dst)[4]) {
            const int ch = i * 4 + rg;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int row = kt * 16 + qcol;
#ifdef OMX_PP_NOLDS   /* experiment: the X phase without its LDS reads (results are garbage) */
                dst[kt] = qf[0][(i + kt) & (NI - 1)];
                (void)row; (void)ch;
#else
                dst[kt] = *reinterpret_cast<const bf16x8*>(&sK[(row * DC + (ch ^ (row & (DC - 1)))) * 8]);
#endif
            }
        };
        read_k(0, kf[0]);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + 1 < NI) read_k(i + 1, kf[(i + 1) & 1]);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int w = 0; w < QW; ++w) sc[w][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i & 1][kt], qf[w][i], sc[w][kt], 0, 0, 0);
        }
    };
    auto pv = [&](const bf16_t* sV) {
        constexpr int NF = 2 * NDT, AHEAD = 4;
        u32x4 vf[AHEAD];
        auto read_v = [&](int idx) {
            const int j = idx / NDT, t = idx % NDT;
            const bf16_t* p0 = sV + v_lane_off + ((t ^ v_sw) * 16) + (2 * j) * 16 * D;
#ifdef OMX_PP_NOLDS
            (void)p0;
            return __builtin_bit_cast(u32x4, qf[1][(j + t) & (NI - 1)]);
#else
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)p0);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(p0 + 16 * D));
            const u32x2v l2 = __builtin_bit_cast(u32x2v, lo), h2 = __builtin_bit_cast(u32x2v, hi);
            return u32x4{l2[0], l2[1], h2[0], h2[1]};
#endif
        };
#pragma unroll
        for (int idx = 0; idx < AHEAD; ++idx) vf[idx] = read_v(idx);
#pragma unroll
        for (int idx = 0; idx < NF; ++idx) {
            const int j = idx / NDT, t = idx % NDT;
            const u32x4 cur = vf[idx % AHEAD];
            if (idx + AHEAD < NF) vf[idx % AHEAD] = read_v(idx + AHEAD);
#pragma unroll
            for (int w = 0; w < QW; ++w)
                o[w][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur), pf[w][j], o[w][t], 0, 0, 0);
        }
    };
    // online softmax of tile i in the base-2 domain (attn_prefill_kernel's, same order of operations)
    auto softmax = [&](int i) {
        const int k0 = i * KB;
        const bool plain = (k0 + KB <= a.Tk) && (MASK == OMX_MASK_NONE || (MASK == OMX_MASK_CAUSAL && k0 + KB - 1 <= q0 + shift));
        if (plain) {
#pragma unroll
            for (int w = 0; w < QW; ++w) {
                float mx = sc[w][0][0];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[w][kt][r]);
                mx = quad_rows_max(mx) * c2;
                // (a deferred / skipped rescale -- cdna_hip_programming.md T13 -- was tried in three forms: any branch or asm region around
                //  the 64 accumulators made hipcc spill them, 172-204 B of scratch and a 2x slower kernel; EXPERIMENTS.md R3-4.
                //  Round 4: the max / sum as four partial chains instead of one of 16 dependent operations -- 346 -> 363 us for the max
                //  alone, 473 us with the sums: this kernel sits on the schedule hipcc finds for exactly this form; R4-14)
                const float m_new = fmaxf(m_run[w], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run[w] - m_new);
                m_run[w] = m_new;
                l_run[w] *= alpha;
#pragma unroll
                for (int t = 0; t < NDT; ++t) o[w][t] *= alpha;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(sc[w][2 * j + (e >> 2)][e & 3], c2, -m_new));
                        l_run[w] += p;
                        pf[w][j][e] = (__bf16)p;
                    }
            }
        } else {
#pragma unroll
            for (int w = 0; w < QW; ++w) {
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + kt * 16 + rg * 4 + r;
                        // (scores stay raw unless an additive mask joins them: the exponent below is then the plain path's fma, so a
                        //  tile gives the same bits whichever path a block's geometry sends it down)
                        float v = MASK == OMX_MASK_ADDITIVE ? sc[w][kt][r] * c2 : sc[w][kt][r];
                        bool keep = key < a.Tk;
                        if (MASK == OMX_MASK_CAUSAL) keep = keep && (key <= qrow[w] + shift);
                        if (MASK == OMX_MASK_BOOL) {
                            const uint8_t mb = reinterpret_cast<const uint8_t*>(a.mask)[(size_t)qrow_c[w] * a.Tk + min(key, a.Tk - 1)];
                            keep = keep & (mb != 0);
                        }
                        if (MASK == OMX_MASK_ADDITIVE)
                            v += 1.44269504088896340736f * bf16_to_f32(reinterpret_cast<const bf16_t*>(a.mask)[(size_t)qrow_c[w] * a.Tk + min(key, a.Tk - 1)]);
                        v = keep ? v : -INFINITY;
                        sc[w][kt][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = quad_rows_max(mx);
                if (MASK != OMX_MASK_ADDITIVE) mx *= c2;
                const float m_new = fmaxf(m_run[w], mx);
                const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run[w] - m_new);
                m_run[w] = m_new;
                l_run[w] *= alpha;
#pragma unroll
                for (int t = 0; t < NDT; ++t) o[w][t] *= alpha;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float sv = sc[w][2 * j + (e >> 2)][e & 3];
                        const float p = (m_new == -INFINITY) ? 0.f
                                        : __builtin_amdgcn_exp2f(MASK == OMX_MASK_ADDITIVE ? sv - m_new : fmaf(sv, c2, -m_new));
                        l_run[w] += p;
                        pf[w][j][e] = (__bf16)p;
                    }
            }
        }
    };
    // phase ends.  Even: nothing of ours may still be reading LDS when the partner's next DMA overwrites it (lgkmcnt); odd: the
    // DMA this wave issued one phase ago has landed (vmcnt) before anyone reads it
    auto end_even = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto end_odd = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // timeline build: cycles spent in X work, Y work, and from "work done" to "barrier passed" of even / odd phases
    unsigned long long tw[5] = {0, 0, 0, 0, 0}, tc = 0;
    auto clk = [&]() { return TR ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull; };
    auto lap = [&](int slot) {
        if (TR) {
            const unsigned long long t = clk();
            tw[slot] += t - tc;
            tc = t;
        }
    };
    if (tb > ta) {
        stage(ta * KB, Kb, koff, sK + (ta & 1) * (KB * D), false);
        // (the builtin, not asm: hipcc must KNOW that the Q loads above have landed -- otherwise it puts its own vmcnt(N) waits for them
        //  inside the loop, where the hardware counter also holds our DMA, and every X phase waits for the tile it just requested)
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_s_barrier();
        tc = clk();
        const unsigned long long t_begin = tc;
        // (measured: the first / last step peeled out of the loops and PV ahead of QK^T -- 16 fewer live registers -- ran 9 % SLOWER)
        if (!late) {
#ifdef OMX_PP_PRIO
            if (OMX_PP_PRIO == 0) __builtin_amdgcn_s_setprio(1);
#endif
            for (int i = ta; i <= tb; ++i) {
                fetch(i);
                if (i < tb) qk(sK + (i & 1) * (KB * D));
                if (i > ta) pv(sV + ((i - 1) & 1) * (KB * D));
                if (TR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                lap(0); end_even(); lap(2);
                if (i < tb) softmax(i);
                lap(1); end_odd(); lap(3);
            }
        } else {
#ifdef OMX_PP_PRIO
            if (OMX_PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);   // static priority for the younger half (MI355X_MICROARCH.md, item 4)
#endif
            for (int i = ta; i <= tb; ++i) {
                fetch(i);
                if (i > ta) softmax(i - 1);
                lap(1); end_even(); lap(2);
                if (i < tb) qk(sK + (i & 1) * (KB * D));
                if (i > ta) pv(sV + ((i - 1) & 1) * (KB * D));
                if (TR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                lap(0); end_odd(); lap(3);
            }
        }
        if (TR && a.trace && lane == 0) {
            const int L = (int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z);
            unsigned long long* tr = a.trace + ((size_t)L * 8 + wave) * 8;
            tr[0] = tw[0]; tr[1] = tw[1]; tr[2] = tw[2]; tr[3] = tw[3]; tr[4] = clk() - t_begin; tr[5] = (unsigned long long)nt;
        }
    }

    if constexpr (SK) {
        if (!(ta == 0 && tb == nt)) {
            // a piece of the unit: park (O, m, l) -- un-normalised, per lane as held -- and let the unit's LAST piece to arrive merge.
            // At most two pieces per unit (a block's share is at least one unit long); part 0 is the one that starts at tile 0;
            // `cut` names the boundary between two shares that runs through this unit (one parking slot per boundary).
            __shared__ unsigned s_last;
            const int part = ta == 0 ? 0 : 1;
            constexpr int PER_LANE = QW * (NDT * 4 + 2);
            float* mine = a.sk_ws + (((size_t)cut * 2 + part) * 512 + threadIdx.x) * PER_LANE;
#pragma unroll
            for (int w = 0; w < QW; ++w) {
#pragma unroll
                for (int t = 0; t < NDT; ++t) {
                    st_coh64(mine + (w * NDT + t) * 4, (uint64_t)__float_as_uint(o[w][t][0]) | ((uint64_t)__float_as_uint(o[w][t][1]) << 32));
                    st_coh64(mine + (w * NDT + t) * 4 + 2, (uint64_t)__float_as_uint(o[w][t][2]) | ((uint64_t)__float_as_uint(o[w][t][3]) << 32));
                }
                st_coh64(mine + QW * NDT * 4 + w * 2, (uint64_t)__float_as_uint(m_run[w]) | ((uint64_t)__float_as_uint(l_run[w]) << 32));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                s_last = __hip_atomic_fetch_add(a.sk_cnt + cut, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u ? 1u : 0u;
            __syncthreads();
            const bool last = s_last != 0;
            __syncthreads();              // (s_last is reused by this block's next piece)
            if (!last) return;
            const float* theirs = a.sk_ws + (((size_t)cut * 2 + (1 - part)) * 512 + threadIdx.x) * PER_LANE;
#pragma unroll
            for (int w = 0; w < QW; ++w) {
                const u32x4 ml = ld_coh128(theirs + QW * NDT * 4 + (w & ~1) * 2);      // (m, l) pairs of two sub-tiles per 16 bytes
                const float m2 = __uint_as_float(ml[(w & 1) * 2]), l2 = __uint_as_float(ml[(w & 1) * 2 + 1]);
                const float M = fmaxf(m_run[w], m2);
                // two products and one sum per value, no fma: the result does not depend on which piece arrived last
                const float fa = (M == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run[w] - M);
                const float fb = (M == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - M);
                l_run[w] = l_run[w] * fa + l2 * fb;
#pragma unroll
                for (int t = 0; t < NDT; ++t) {
                    const u32x4 v = ld_coh128(theirs + (w * NDT + t) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[w][t][e] = o[w][t][e] * fa + __uint_as_float(v[e]) * fb;
                }
            }
            if (threadIdx.x == 0) __hip_atomic_store(a.sk_cnt + cut, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        const float l_tot = quad_rows_sum(l_run[w]);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (qrow[w] < a.Tq) {
            bf16_t* op = a.out + (size_t)b * a.o_bs + (size_t)h * a.o_hs + (size_t)qrow[w] * a.o_ts;
#pragma unroll
            for (int t = 0; t < NDT; ++t) {
                u32x2v wv = {pack_bf16(o[w][t][0] * inv, o[w][t][1] * inv), pack_bf16(o[w][t][2] * inv, o[w][t][3] * inv)};
                *reinterpret_cast<u32x2v*>(op + t * 16 + rg * 4) = wv;
            }
        }
    }

}

template <int D, int MASK, bool TR = false>
__global__ __launch_bounds__(512, 1) void attn_prefill_pp_kernel(const PrefillArgs a) {
    constexpr int QW = 2;
    constexpr int QBLK = 8 * 16 * QW;           // 256 query rows per block
    __shared__ __attribute__((aligned(16))) bf16_t sK2[2][KB * D];
    __shared__ __attribute__((aligned(16))) bf16_t sV2[2][KB * D];
    (void)QBLK;
    // blocks of one head on one XCD (its L2 then holds that head's K / V once, not eight times): linear block id L runs on XCD
    // L % 8; when the head count is a multiple of 8, XCD x works through heads x, x + 8, ...
    int qt = blockIdx.x, h = blockIdx.y;
    const int b = blockIdx.z;
    if ((a.H & 7) == 0) {
        const int L = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
        const int xcd = L & 7, idx = L >> 3;
        h = (idx / (int)gridDim.x) * 8 + xcd;
        qt = idx % (int)gridDim.x;
    }
    if (MASK == OMX_MASK_CAUSAL) qt = (int)gridDim.x - 1 - qt;   // longest blocks first
    pp_unit<D, MASK, TR, false>(a, &sK2[0][0], &sV2[0][0], b, h, qt, 0, 0, 0);
}

#ifdef OMX_EXPERIMENTS   // measured negatives (EXPERIMENTS.md R4-14, R3-4): `make EXPERIMENTS=1`
// ---- The two-phase kernel on 32x32x16 MFMAs (round 4; OMX_ATTN_PP32=1, head_dim 128): the same block (8 waves, 256 query rows, K / V
//      tiles of 64 keys by LDS-DMA, waves 0-3 and 4-7 in opposite phases), but a wave's 32 query rows are ONE 32-wide MFMA column block:
//      S^T tile (32 keys x 32 queries) = 8 MFMAs over the head dim, O^T tile (32 dims x 32 queries) += V^T P^T = 4 MFMAs over the tile's
//      keys -- 32 MFMA issues per tile step instead of 64 for the same flops (the loop is issue-bound, EXPERIMENTS.md R3-4), and a
//      lane holds 32 scores of ONE query, so the row max / sum need one cross-lane step (lane ^ 32) instead of two.
//      Lane layout of a 32x32 result: column (query) = lane & 31, rows 8 g + 4 (lane >> 5) + r for g, r = 0..3.  Operand k-slot
//      8 (lane >> 5) + e  <->  key 8 (2 j + (e >> 2)) + 4 (lane >> 5) + (e & 3) of the 16-key step j on BOTH operands of the second
//      product, so that P's fragment is eight consecutive accumulator registers of the lane.  Not bit-identical to the 16x16x32 kernels
//      (other summation order inside the MFMAs). ----
#ifndef OMX_PP32_KAHEAD
#define OMX_PP32_KAHEAD 1
#endif
#ifndef OMX_PP32_CHAINS
#define OMX_PP32_CHAINS 2
#endif
template <int D, int MASK>
__device__ __forceinline__ void pp32_unit(const PrefillArgs& a, bf16_t* sK, bf16_t* sV, int b, int h, int qt) {
    using f32x16v = __attribute__((ext_vector_type(16))) float;
    constexpr int DC = D / 8, NI = D / 16, NT32 = D / 32;
    constexpr int QBLK = 256;
    constexpr int KCH = KB * DC / 512;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool late = wave_u >= 4;
    const int q32 = lane & 31, hh = lane >> 5;
    const int kvh = h / (a.H / a.Hkv);
    const int q0 = qt * QBLK;
    const int shift = a.Tk - a.Tq;
    const bf16_t* Kb = a.k + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;
    const bf16_t* Vb = a.v + (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;

    const int qrow = q0 + wave * 32 + q32, qrow_c = min(qrow, a.Tq - 1);
    bf16x8 qf[NI];
    {
        const bf16_t* Qp = a.q + (size_t)b * a.q_bs + (size_t)h * a.q_hs + (size_t)qrow_c * a.q_ts;
#pragma unroll
        for (int i = 0; i < NI; ++i) qf[i] = *reinterpret_cast<const bf16x8*>(Qp + i * 16 + hh * 8);
    }
    f32x16v o[NT32];
#pragma unroll
    for (int t = 0; t < NT32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    int kv_end = a.Tk;
    if (MASK == OMX_MASK_CAUSAL) kv_end = max(0, min(a.Tk, q0 + QBLK + shift));
    const int nt = (kv_end + KB - 1) / KB;

    constexpr int VSH = (D == 128) ? 0 : 1;
    constexpr int VBM = D / 16 - 1;
    uint32_t koff[KCH], voff[KCH];
#pragma unroll
    for (int it = 0; it < KCH; ++it) {
        const int ci = threadIdx.x + it * 512;
        const int row = ci / DC;
        koff[it] = (uint32_t)(row * a.kv_ts + ((ci % DC) ^ (row & (DC - 1))) * 8);
        voff[it] = (uint32_t)(row * a.kv_ts + ((ci % DC) ^ (((row >> VSH) & VBM) << 1)) * 8);
    }
    auto dma16 = [&](const bf16_t* gsrc, unsigned lds_dst) {   // (asm: see attn_prefill_kernel)
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    };
    auto stage = [&](int k0, const bf16_t* src, const uint32_t (&off)[KCH], bf16_t* dst, bool is_v) {
        const unsigned d0 = (unsigned)(uintptr_t)dst + wave_u * 1024u;
        if (k0 + KB <= a.Tk) {
            const bf16_t* base = src + (size_t)k0 * a.kv_ts;
#pragma unroll
            for (int it = 0; it < KCH; ++it) dma16(base + off[it], d0 + it * 8192u);
            return;
        }
#pragma unroll
        for (int it = 0; it < KCH; ++it) {
            const int ci = threadIdx.x + it * 512;
            const int row = ci / DC;
            const int ch = (ci % DC) ^ (is_v ? (((row >> VSH) & VBM) << 1) : (row & (DC - 1)));
            const int key = min(k0 + row, a.Tk - 1);
            dma16(src + (size_t)key * a.kv_ts + ch * 8, d0 + it * 8192u);
        }
    };
    auto fetch = [&](int i) {
        if (i + 1 < nt) stage((i + 1) * KB, Kb, koff, sK + ((i + 1) & 1) * (KB * D), false);
        if (i < nt) stage(i * KB, Vb, voff, sV + (i & 1) * (KB * D), true);
    };
    // V^T fragment (32 dims x 16 keys of step j, d-tile t): the 16-lane group g4 = lane >> 4 reads keys 16 j + 4 (g4 >> 1) + 0..3 (+ 8 for
    // the second half of the k-slots) at dims 32 t + 16 (g4 & 1) + (lane & 15) through the transposing read
    const int l16 = lane & 15, g4 = lane >> 4;
    const int v_key = 4 * (g4 >> 1) + (l16 >> 2);            // key row (mod 16) whose address this lane supplies
    const int v_lane_off = v_key * D + (l16 & 3) * 4;
    const int v_blk = g4 & 1;                                 // which 16-dim half of the 32-dim tile
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

    f32x16v sc[2];         // S_i^T: key tiles of 32, from X(i) to Y(i)
    bf16x8 pf[4];          // P_i by 16-key step, from Y(i) to X(i + 1)
    const float c2 = a.scale * 1.44269504088896340736f;

    auto qk = [&](const bf16_t* sKt) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
        // (a K fragment feeds ONE MFMA here -- 64 matrix cycles per head-dim step -- so the reads run KAH steps ahead, not one)
        constexpr int KAH = OMX_PP32_KAHEAD;
        bf16x8 kf[KAH + 1][2];
        auto read_k = [&](int i, bf16x8 (&dst)[2]) {
            const int ch = i * 2 + hh;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int row = kt * 32 + q32;     // A operand: lane & 31 indexes the key row
                dst[kt] = *reinterpret_cast<const bf16x8*>(&sKt[(row * DC + (ch ^ (row & (DC - 1)))) * 8]);
            }
        };
#pragma unroll
        for (int i = 0; i < KAH; ++i) read_k(i, kf[i]);
#if OMX_PP32_CHAINS == 4
        f32x16v sd[2];     // four accumulator chains (even / odd head-dim steps of each key tile), summed at the end
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sd[kt][r] = 0.f;
#endif
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + KAH < NI) read_k(i + KAH, kf[(i + KAH) % (KAH + 1)]);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#if OMX_PP32_CHAINS == 4
                if (i & 1) sd[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i % (KAH + 1)][kt], qf[i], sd[kt], 0, 0, 0);
                else
#endif
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i % (KAH + 1)][kt], qf[i], sc[kt], 0, 0, 0);
            }
        }
#if OMX_PP32_CHAINS == 4
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) sc[kt] += sd[kt];
#endif
    };
    auto pv = [&](const bf16_t* sVt) {
        constexpr int NF = 4 * NT32, AHEAD = 4;       // fragment idx = j * NT32 + t
        u32x4 vf[AHEAD];
        auto read_v = [&](int idx) {
            const int j = idx / NT32, t = idx % NT32;
            const int key0 = 16 * j + v_key;                                  // (+ 8 for the second read)
            const int blk = 2 * t + v_blk;
            const bf16_t* p0 = sVt + (16 * j) * D + v_lane_off + ((blk ^ ((key0 >> VSH) & VBM)) * 16);
            const bf16_t* p1 = sVt + (16 * j + 8) * D + v_lane_off + ((blk ^ (((key0 + 8) >> VSH) & VBM)) * 16);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)p0);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)p1);
            const u32x2v l2 = __builtin_bit_cast(u32x2v, lo), h2 = __builtin_bit_cast(u32x2v, hi);
            return u32x4{l2[0], l2[1], h2[0], h2[1]};
        };
#pragma unroll
        for (int idx = 0; idx < AHEAD; ++idx) vf[idx] = read_v(idx);
#pragma unroll
        for (int idx = 0; idx < NF; ++idx) {
            const int j = idx / NT32, t = idx % NT32;
            const u32x4 cur = vf[idx % AHEAD];
            if (idx + AHEAD < NF) vf[idx % AHEAD] = read_v(idx + AHEAD);
            o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur), pf[j], o[t], 0, 0, 0);
        }
    };
    auto pair_max = [&](float v) {   // over the two lanes (l, l ^ 32) that hold one query
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    auto softmax = [&](int i) {
        const int k0 = i * KB;
        const bool plain = (k0 + KB <= a.Tk) && (MASK == OMX_MASK_NONE || (MASK == OMX_MASK_CAUSAL && k0 + KB - 1 <= q0 + shift));
        float mx;
        if (plain) {
            // (a lane holds all 32 scores of its query: four independent chains instead of one of 32 dependent operations)
            float m4[4] = {sc[0][0], sc[0][1], sc[0][2], sc[0][3]};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) m4[r & 3] = fmaxf(m4[r & 3], sc[kt][r]);
            mx = pair_max(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]))) * c2;
        } else {
            mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kt * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
                    float v = MASK == OMX_MASK_ADDITIVE ? sc[kt][r] * c2 : sc[kt][r];
                    bool keep = key < a.Tk;
                    if (MASK == OMX_MASK_CAUSAL) keep = keep && (key <= qrow + shift);
                    if (MASK == OMX_MASK_BOOL) {
                        const uint8_t mb = reinterpret_cast<const uint8_t*>(a.mask)[(size_t)qrow_c * a.Tk + min(key, a.Tk - 1)];
                        keep = keep & (mb != 0);
                    }
                    if (MASK == OMX_MASK_ADDITIVE)
                        v += 1.44269504088896340736f * bf16_to_f32(reinterpret_cast<const bf16_t*>(a.mask)[(size_t)qrow_c * a.Tk + min(key, a.Tk - 1)]);
                    v = keep ? v : -INFINITY;
                    sc[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = pair_max(mx);
            if (MASK != OMX_MASK_ADDITIVE) mx *= c2;
        }
        const float m_new = fmaxf(m_run, mx);
        const bool dead = !plain && m_new == -INFINITY;
        const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int t = 0; t < NT32; ++t) o[t] *= alpha;
        float l4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sv = sc[kt][8 * j + e];
                    const float p = dead ? 0.f : __builtin_amdgcn_exp2f((!plain && MASK == OMX_MASK_ADDITIVE) ? sv - m_new : fmaf(sv, c2, -m_new));
                    l4[e & 3] += p;
                    pf[kt * 2 + j][e] = (__bf16)p;
                }
        l_run += (l4[0] + l4[1]) + (l4[2] + l4[3]);
    };
    auto end_even = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto end_odd = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if (nt > 0) {
        stage(0, Kb, koff, sK, false);
        __builtin_amdgcn_s_waitcnt(0);     // (the builtin: hipcc must know the Q loads have landed -- see pp_unit)
        __builtin_amdgcn_s_barrier();
        if (!late) {
            for (int i = 0; i <= nt; ++i) {
                fetch(i);
                if (i < nt) qk(sK + (i & 1) * (KB * D));
                if (i > 0) pv(sV + ((i - 1) & 1) * (KB * D));
                end_even();
                if (i < nt) softmax(i);
                end_odd();
            }
        } else {
            for (int i = 0; i <= nt; ++i) {
                fetch(i);
                if (i > 0) softmax(i - 1);
                end_even();
                if (i < nt) qk(sK + (i & 1) * (KB * D));
                if (i > 0) pv(sV + ((i - 1) & 1) * (KB * D));
                end_odd();
            }
        }
    }
    // normalise and store: lane holds out[qrow][32 t + 8 g + 4 hh + 0..3]
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        const float l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (qrow < a.Tq) {
            bf16_t* op = a.out + (size_t)b * a.o_bs + (size_t)h * a.o_hs + (size_t)qrow * a.o_ts;
#pragma unroll
            for (int t = 0; t < NT32; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2v wv = {pack_bf16(o[t][4 * g] * inv, o[t][4 * g + 1] * inv), pack_bf16(o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv)};
                    *reinterpret_cast<u32x2v*>(op + t * 32 + 8 * g + 4 * hh) = wv;
                }
        }
    }
}

template <int D, int MASK>
__global__ __launch_bounds__(512, 1) void attn_prefill_pp32_kernel(const PrefillArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t sK2[2][KB * D];
    __shared__ __attribute__((aligned(16))) bf16_t sV2[2][KB * D];
    int qt = blockIdx.x, h = blockIdx.y;
    const int b = blockIdx.z;
    if ((a.H & 7) == 0) {     // blocks of one head on one XCD (attn_prefill_pp_kernel)
        const int L = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
        const int xcd = L & 7, idx = L >> 3;
        h = (idx / (int)gridDim.x) * 8 + xcd;
        qt = idx % (int)gridDim.x;
    }
    if (MASK == OMX_MASK_CAUSAL) qt = (int)gridDim.x - 1 - qt;
    pp32_unit<D, MASK>(a, &sK2[0][0], &sV2[0][0], b, h, qt);
}

// Stream-K form (no mask): the grid is one block per CU and the units x key tiles are cut into EQUAL shares -- 432 units on 256 CUs run
// as 1.69 rounds of whole blocks (the second one 69 % full) but as 121.5 tiles per CU here.  A share covers the tail of one unit, whole
// units, and the head of another; a cut unit is merged by whichever of its two pieces finishes last.  XCD x owns a contiguous eighth
// of the units (its L2 then sees ~3 heads' K / V), block index c -> logical share (c % 8) * (G / 8) + c / 8.
template <int D>
__global__ __launch_bounds__(512, 1) void attn_prefill_sk_kernel(const PrefillArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t sK2[2][KB * D];
    __shared__ __attribute__((aligned(16))) bf16_t sV2[2][KB * D];
    const int gx = (a.Tq + 255) / 256, nt = (a.Tk + KB - 1) / KB;
    const int G = (int)gridDim.x;
    const int c = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);       // (G is a multiple of 8)
    const long long total = (long long)gx * a.H * a.B * nt;
    long long g = total * c / G;
    const long long g_end = total * (c + 1) / G;
    while (g < g_end) {
        const int unit = (int)(g / nt), t0 = (int)(g % nt);
        const int t1 = (int)min((long long)nt, t0 + (g_end - g));
        const int qt = unit % gx, h = (unit / gx) % a.H, b = unit / (gx * a.H);
        const int cut = t0 > 0 ? c - 1 : c;       // tail piece: the boundary with the previous share; head piece: with the next
        pp_unit<D, OMX_MASK_NONE, false, true>(a, &sK2[0][0], &sV2[0][0], b, h, qt, cut, t0, t1);
        g += t1 - t0;
        __syncthreads();       // the next piece re-stages the LDS tiles this one may still be reading
    }
}

#endif   // OMX_EXPERIMENTS

}  // namespace

#ifdef OMX_EXPERIMENTS
namespace {
// parking slots + arrival counters of the stream-K kernel, one set per stream (allocated on first use, 70 MiB: 256 boundaries x 2 pieces)
constexpr int kSkBlocks = 256;
constexpr size_t kSkFloats = (size_t)kSkBlocks * 2 * 512 * 68;
struct SkWs { float* ws = nullptr; unsigned* cnt = nullptr; };
std::mutex g_sk_mu;
std::map<std::pair<hipStream_t, std::thread::id>, SkWs> g_sk_ws;
int sk_workspace(hipStream_t s, float** ws, unsigned** cnt) {
    std::lock_guard<std::mutex> lk(g_sk_mu);
    SkWs& w = g_sk_ws[{s, s ? std::thread::id() : std::this_thread::get_id()}];
    if (!w.ws) {
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        OMX_HIP_CHECK(hipThreadExchangeStreamCaptureMode(&mode));
        hipError_t e = hipMalloc((void**)&w.ws, kSkFloats * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&w.cnt, kSkBlocks * 4);
        if (e == hipSuccess) e = hipMemset(w.cnt, 0, kSkBlocks * 4);       // (null stream, waited for below: also right under a capture of `s`)
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (e != hipSuccess) {
            if (w.ws) (void)hipFree(w.ws);
            if (w.cnt) (void)hipFree(w.cnt);
            w = SkWs{};
            return set_error("stream-K attention scratch: %s", hipGetErrorString(e));
        }
    }
    *ws = w.ws; *cnt = w.cnt;
    return 0;
}
int sk_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = -1;
    }
    return n;
}
}  // namespace

#endif   // OMX_EXPERIMENTS

int launch_attn_prefill(bf16_t* out, const bf16_t* q, const bf16_t* k, const bf16_t* v, int B, int H, int Hkv, int Tq,
                        int Tk, int D, int64_t kv_batch_stride, int64_t kv_head_stride, float scale, int mask_mode,
                        const void* mask, hipStream_t s, bool out_token_major, const AttnLayout* layout, bool f16) {
    OMX_REQUIRE(D == 64 || D == 128, "sdpa prefill: head_dim %d unsupported (64 or 128)", D);
    // float16 operands (a float16 checkpoint's prompt pass): the single-phase kernel's float16 instantiations
    OMX_REQUIRE(!f16 || mask_mode == OMX_MASK_NONE || mask_mode == OMX_MASK_CAUSAL, "sdpa prefill in float16: no mask or causal");
    PrefillArgs a = {q, k, v, out, B, H, Hkv, Tq, Tk, kv_batch_stride, kv_head_stride, scale, mask_mode, mask,
                     (int64_t)H * Tq * D, (int64_t)Tq * D, D, (int64_t)H * Tq * D, (int64_t)Tq * D, D, D};
    if (out_token_major) {   // out[b][t][h][d]: what o_proj consumes after the reference's transpose+reshape (model.rs:211-213)
        a.o_hs = D;
        a.o_ts = (int64_t)H * D;
    }
    if (layout) {            // fully strided operands (DiT blocks read q/k/v straight out of a fused projection)
        a.q_bs = layout->q_bs; a.q_hs = layout->q_hs; a.q_ts = layout->q_ts;
        a.o_bs = layout->o_bs; a.o_hs = layout->o_hs; a.o_ts = layout->o_ts;
        a.kv_ts = layout->kv_ts;
    }
    // long sequences: 128 query rows per block (each LDS fragment feeds two MFMAs); short ones keep 64-row
    // blocks so that the grid still covers the chip
    // long sequences: the two-phase kernel, 256 query rows per block, one block per CU (OMX_ATTN_PP=0: the kernels below).  Under a
    // causal mask its blocks differ in length up to the whole sequence and nothing pairs a long one with a short one on a CU (the
    // single-phase kernel does, two blocks per CU): 2048 x 32 heads ran 82 us against 65, so causal shapes take it only when the grid is
    // many rounds deep
    // (round 5) the 4-wave persistent kernel for long unmasked sequences (OMX_ATTN_W4=0: the kernels below)
    {
        const char* w4 = getenv("OMX_ATTN_W4");
        // an explicit request for one of the older kernels (their A/B switches, the two-phase trace) outranks the default choice of this
        // one: OMX_ATTN_PP=1 used to measure the four-wave kernel silently unless OMX_ATTN_W4=0 was set as well
        const bool older_asked = getenv("OMX_ATTN_PP") || getenv("OMX_ATTN_PP_TRACE") || getenv("OMX_ATTN_PP32");
        const bool want = w4 ? atoi(w4) != 0 : !older_asked;
        if (want && Tq >= 1024 && attn_flash4_supported(B, H, Hkv, Tq, Tk, D, mask_mode, f16))
            return launch_attn_flash4(out, q, k, v, B, H, Hkv, Tq, Tk, kv_batch_stride, kv_head_stride, scale, s, out_token_major, layout);
    }
    const char* ppenv = getenv("OMX_ATTN_PP");
    const long pp_blocks = (long)((Tq + 255) / 256) * H * B;
    if (!f16 && (ppenv ? atoi(ppenv) != 0 : (Tq >= 1024 && (mask_mode != OMX_MASK_CAUSAL || pp_blocks >= 2048)))) {
        const dim3 grid((Tq + 255) / 256, H, B), block(512);
#ifdef OMX_EXPERIMENTS
        // stream-K form (opt-in, OMX_ATTN_STREAMK=1): no mask, more units than CUs (every share then spans at least one whole unit: a
        // unit is cut at most once); one block per CU (a piece never waits for another: no co-residency needed).  Measured SLOWER where
        // it should pay (432 units: 384 us against 329; 280 units: 173 against 125; 512 units, nothing cut: 330 against 349): the blocks
        // of a head no longer walk its key tiles together, and the head's K / V stop being one L2-resident stream (EXPERIMENTS.md R3-4)
        const char* ske = getenv("OMX_ATTN_STREAMK");
        const int cus = sk_cus();
        const bool sk_fit = mask_mode == OMX_MASK_NONE && cus >= 8 && cus <= kSkBlocks && pp_blocks > cus && !getenv("OMX_ATTN_PP_TRACE");
        if (sk_fit && ske && atoi(ske) != 0) {
            if (sk_workspace(s, &a.sk_ws, &a.sk_cnt)) return 1;
            const int G = cus & ~7;
            if (D == 128) attn_prefill_sk_kernel<128><<<G, block, 0, s>>>(a);
            else attn_prefill_sk_kernel<64><<<G, block, 0, s>>>(a);
            OMX_LAUNCH_CHECK();
            return 0;
        }
#endif
        if (const char* te = getenv("OMX_ATTN_PP_TRACE")) {   // timeline build (head_dim 128, no mask): address of a device buffer
            a.trace = reinterpret_cast<unsigned long long*>(strtoull(te, nullptr, 0));
            OMX_REQUIRE(D == 128 && mask_mode == OMX_MASK_NONE, "attention timeline build: head_dim 128 without a mask only");
            attn_prefill_pp_kernel<128, OMX_MASK_NONE, true><<<grid, block, 0, s>>>(a);
            OMX_LAUNCH_CHECK();
            return 0;
        }
#ifdef OMX_EXPERIMENTS
        if (const char* p32 = getenv("OMX_ATTN_PP32"); p32 && atoi(p32) != 0 && D == 128) {   // the 32x32x16 form (round 4, opt-in)
#define OMX_PP32_CASE(MM)                                                            \
    if (mask_mode == MM) {                                                           \
        attn_prefill_pp32_kernel<128, MM><<<grid, block, 0, s>>>(a);                 \
        OMX_LAUNCH_CHECK();                                                          \
        return 0;                                                                    \
    }
            OMX_PP32_CASE(OMX_MASK_NONE) OMX_PP32_CASE(OMX_MASK_CAUSAL) OMX_PP32_CASE(OMX_MASK_BOOL) OMX_PP32_CASE(OMX_MASK_ADDITIVE)
#undef OMX_PP32_CASE
        }
#endif
#define OMX_PP_CASE(DD, MM)                                                          \
    if (D == DD && mask_mode == MM) {                                                \
        attn_prefill_pp_kernel<DD, MM><<<grid, block, 0, s>>>(a);                    \
        OMX_LAUNCH_CHECK();                                                          \
        return 0;                                                                    \
    }
#define OMX_PP_MASKS(DD) OMX_PP_CASE(DD, OMX_MASK_NONE) OMX_PP_CASE(DD, OMX_MASK_CAUSAL) OMX_PP_CASE(DD, OMX_MASK_BOOL) OMX_PP_CASE(DD, OMX_MASK_ADDITIVE)
        OMX_PP_MASKS(128) OMX_PP_MASKS(64)
#undef OMX_PP_MASKS
#undef OMX_PP_CASE
    }
    const char* wenv = getenv("OMX_ATTN_WIDE");
    const bool wide = wenv ? atoi(wenv) != 0 : Tq >= 512;
    const int qblk = wide ? 128 : 64;
    const dim3 grid((Tq + qblk - 1) / qblk, H, B), block(256);
    if (f16) {
#define OMX_PF16_CASE(DD, QQ, MM)                                                    \
    if (D == DD && (wide ? 2 : 1) == QQ && mask_mode == MM) {                        \
        attn_prefill_kernel<DD, QQ, MM, true><<<grid, block, 0, s>>>(a);             \
        OMX_LAUNCH_CHECK();                                                          \
        return 0;                                                                    \
    }
#define OMX_PF16_D(DD) OMX_PF16_CASE(DD, 1, OMX_MASK_NONE) OMX_PF16_CASE(DD, 1, OMX_MASK_CAUSAL) OMX_PF16_CASE(DD, 2, OMX_MASK_NONE) OMX_PF16_CASE(DD, 2, OMX_MASK_CAUSAL)
        OMX_PF16_D(128) OMX_PF16_D(64)
#undef OMX_PF16_D
#undef OMX_PF16_CASE
    }
#define OMX_PF_CASE(DD, QQ, MM)                                                      \
    if (D == DD && (wide ? 2 : 1) == QQ && mask_mode == MM) {                        \
        attn_prefill_kernel<DD, QQ, MM><<<grid, block, 0, s>>>(a);                   \
        OMX_LAUNCH_CHECK();                                                          \
        return 0;                                                                    \
    }
#define OMX_PF_MASKS(DD, QQ) OMX_PF_CASE(DD, QQ, OMX_MASK_NONE) OMX_PF_CASE(DD, QQ, OMX_MASK_CAUSAL) \
    OMX_PF_CASE(DD, QQ, OMX_MASK_BOOL) OMX_PF_CASE(DD, QQ, OMX_MASK_ADDITIVE)
    OMX_PF_MASKS(128, 1) OMX_PF_MASKS(128, 2) OMX_PF_MASKS(64, 1) OMX_PF_MASKS(64, 2)
#undef OMX_PF_MASKS
#undef OMX_PF_CASE
    return set_error("sdpa prefill: unsupported mask mode %d", mask_mode);
}

}  // namespace omx
