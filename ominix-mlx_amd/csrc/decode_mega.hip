// One persistent kernel per decode token for gfx950: every phase of every layer of
// qwen3-mlx's decode step (qwen3-mlx/src/model.rs:168-327, 423, 480-489, 733-735) plus the greedy
// sampler, with device-wide barriers (gridsync.hpp) where the launch boundaries used to be.
//
// Why: as separate launches (engine.hip's step graph) each of the 5 dependent phases per layer pays
// launch + HBM ramp + drain; the rocprof summary (profiles/r01_b_kernel_stats_ctx2048.csv) puts the
// GEMV phases at 4.2-5.8 TB/s against the 6.3 TB/s the lm_head kernel reaches in steady state, and
// attention + combine at 15 us for 9 MB.  Here
//   * one 8-wave block per CU; a block issues the first TWO register sets per wave of the NEXT phase's
//     weight rows before it waits at the barrier, so HBM keeps streaming while the barrier resolves and
//     the activation is staged (wave 0, which polls the barrier, issues its share after it);
//   * inside a block the row sets of a phase are handed out through a queue in LDS: the waves of a CU do
//     not progress at the same rate and a static split leaves the phase waiting for the slowest;
//   * the attention phase runs on the first n_ab blocks only; all other blocks own the O-projection
//     rows and pull them into registers while attention runs, so the O phase is compute-only;
//   * the split-KV combine is done by the last split block of each KV head (no extra phase).
// The arithmetic of every row / head is the one of gemv.hip / attn_decode.hip (same per-lane order,
// same rounding points), so tokens and logits are bit-identical to the step-graph path -- which stays
// as the fallback for shapes without an instantiation, D != 128, G > 4 and tensor parallelism.
//
// Data exchanged between blocks inside the launch (h, qkv, split partials, attention output, act,
// argmax partials) moves through the coherent accessors of gridsync.hpp; weights and the KV cache of
// earlier tokens are read-only in a launch and use plain / non-temporal loads.
#include "decode_mega.hpp"

#include <stdlib.h>

#include "gridsync.hpp"

namespace omx {

namespace {

constexpr int kBlock = 512;      // 8 waves: one block per CU, two waves per SIMD
constexpr int kWaves = 8;
constexpr int kGroupWaves = 4;   // attention: a block runs two 4-wave virtual blocks side by side
constexpr int kGroups = kWaves / kGroupWaves;
constexpr int kGroupThreads = kGroupWaves * 64;
constexpr int kSet = 16;         // 16-byte vectors per lane per register set (two sets in flight)
constexpr int kD = 128;          // head_dim
constexpr int kGT = 4;           // query heads per KV head held in registers
constexpr int kLPR = kD / 8;     // lanes per K/V row
constexpr int kTPW = 64 / kLPR;  // tokens per wave-instruction
constexpr int kUnroll = 4;
constexpr int kStep = kTPW * kUnroll;   // tokens per wave per step
constexpr int kMaxSplit = 512;
constexpr int kCombChunk = 40;   // split partials a combine thread keeps in flight per head
constexpr int kPartRows = 64;    // down projection, K split over four waves: rows per block

__device__ __forceinline__ u32x4 ld_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }

struct WSrc {
    const bf16_t *w0, *w1, *w2;   // row-stacked matrices sharing K (q/k/v) -- or gate (w0) / up (w1)
    int n0, n1;
};

__device__ __forceinline__ const bf16_t* row_ptr(const WSrc& s, int row, int K) {
    if (row < s.n0) return s.w0 + (size_t)row * K;
    row -= s.n0;
    if (row < s.n1) return s.w1 + (size_t)row * K;
    row -= s.n1;
    return s.w2 + (size_t)row * K;
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

__device__ __forceinline__ void unpack8(const u32x4 r, float (&x)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x[2 * e] = bf16lo(r[e]);
        x[2 * e + 1] = bf16hi(r[e]);
    }
}

// ---- weight rows: HBM -> VGPR, one wave per row (or per K quarter of a row) ----
template <int NVW, int RB, bool PAIR>
__device__ __forceinline__ void issue_set(u32x4 (&w)[kSet], const WSrc& s, int K, int r0, int r1, int koff, int lane) {
    static_assert(NVW * RB * (PAIR ? 2 : 1) <= kSet, "register set too small");
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = r0 + r;
        if (row < r1) {   // wave-uniform
            if (PAIR) {
                const u32x4* g = reinterpret_cast<const u32x4*>(s.w0 + (size_t)row * K) + koff + lane;
                const u32x4* u = reinterpret_cast<const u32x4*>(s.w1 + (size_t)row * K) + koff + lane;
#pragma unroll
                for (int j = 0; j < NVW; ++j) {
                    w[(2 * r) * NVW + j] = ld_nt(g + j * 64);
                    w[(2 * r + 1) * NVW + j] = ld_nt(u + j * 64);
                }
            } else {
                const u32x4* p = reinterpret_cast<const u32x4*>(row_ptr(s, row, K)) + koff + lane;
#pragma unroll
                for (int j = 0; j < NVW; ++j) w[r * NVW + j] = ld_nt(p + j * 64);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NVW * (PAIR ? 2 : 1); ++j) w[r * NVW * (PAIR ? 2 : 1) + j] = u32x4{0, 0, 0, 0};
        }
    }
}

template <int NVW, int RB, bool PAIR, class Epi>
__device__ __forceinline__ void compute_set(const u32x4 (&w)[kSet], const u32x4* xs, int koff, int r0, int r1, int lane,
                                            Epi&& epi) {
    constexpr int LR = PAIR ? 2 : 1;
    constexpr int NR = RB * LR;
    float acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < NVW; ++j) {
        const u32x4 xp = xs[koff + j * 64 + lane];
        float xf[8];
        unpack8(xp, xf);
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r] = dot8(w[r * NVW + j], xf, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (r0 + r < r1) epi(r0 + r, acc[LR * r], acc[LR * r + (LR - 1)]);
    }
}

// The block's rows [r0, r1) are cut into sets of RB rows.  NQ = waves sharing a row (K quarters);
// the kWaves/NQ waves of a quarter take sets from a queue in LDS: the first two per wave are fixed
// (they are issued around the barrier), the rest go to whichever wave frees a register set first.
template <int NVW, int RB, bool PAIR, int NQ>
__device__ __forceinline__ void prefetch_sets(u32x4 (&wA)[kSet], u32x4 (&wB)[kSet], const WSrc& s, int K, int r0, int r1,
                                              int koff, int lane, int j) {
    constexpr int WPQ = kWaves / NQ;
    issue_set<NVW, RB, PAIR>(wA, s, K, r0 + j * RB, r1, koff, lane);
    issue_set<NVW, RB, PAIR>(wB, s, K, r0 + (j + WPQ) * RB, r1, koff, lane);
}

template <int NVW, int RB, bool PAIR, int NQ, class Epi>
__device__ __forceinline__ void stream_sets(u32x4 (&wA)[kSet], u32x4 (&wB)[kSet], const WSrc& s, int K, int r0, int r1,
                                            int koff, const u32x4* xs, int* queue, int lane, int j, Epi&& epi) {
    constexpr int WPQ = kWaves / NQ;
    const int nsets = (r1 - r0 + RB - 1) / RB;
    int sA = j, sB = j + WPQ;
    auto grab = [&]() {
        int v = 0;
        if (lane == 0) v = atomicAdd(queue, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    while (sA < nsets || sB < nsets) {
        if (sA < nsets) {
            compute_set<NVW, RB, PAIR>(wA, xs, koff, r0 + sA * RB, r1, lane, epi);
            sA = grab();
            if (sA < nsets) issue_set<NVW, RB, PAIR>(wA, s, K, r0 + sA * RB, r1, koff, lane);
        }
        if (sB < nsets) {
            compute_set<NVW, RB, PAIR>(wB, xs, koff, r0 + sB * RB, r1, lane, epi);
            sB = grab();
            if (sB < nsets) issue_set<NVW, RB, PAIR>(wB, s, K, r0 + sB * RB, r1, koff, lane);
        }
    }
}

// ---- activation [K] -> LDS as bf16, optionally RMS-normalised.  The arithmetic is gemv.hip's 256-thread
// prologue (threads 0..255, sum over four waves); the upper half of the block stages the residual rows the
// epilogue will add, and threads 0..3 reset the phase's row-set queues.
template <int NV, bool NORM>
__device__ __forceinline__ void stage_x(u32x4* xs, float* red, int* queues, int queue_init, bf16_t* resid_lds,
                                        const bf16_t* resid_g, int n_resid, const bf16_t* xg, const bf16_t* norm_w,
                                        float eps, int K) {
    constexpr int PV = (NV * 64 + 255) / 256;
    const int t = threadIdx.x;
    if (t < 4) queues[t] = queue_init;
    u32x4 xv[PV];
    float ss = 0.f;
    if (t < 256) {
#pragma unroll
        for (int i = 0; i < PV; ++i) {
            const int v = t + i * 256;
            if (v < NV * 64) {
                const u32x4 raw = ld_coh128(reinterpret_cast<const u32x4*>(xg) + v);
                xv[i] = raw;
                if (NORM) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = bf16lo(raw[q]), hi = bf16hi(raw[q]);
                        ss = fmaf(lo, lo, ss);
                        ss = fmaf(hi, hi, ss);
                    }
                }
            }
        }
    } else if (resid_g) {
        for (int v = t - 256; v < n_resid / 8; v += 256)
            reinterpret_cast<u32x4*>(resid_lds)[v] = ld_coh128(reinterpret_cast<const u32x4*>(resid_g) + v);
    }
    if (NORM) {
        ss = wave_sum(ss);
        __syncthreads();
        if ((t & 63) == 0 && t < 256) red[t >> 6] = ss;
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += red[i];
        const float rstd = 1.0f / sqrtf(s / (float)K + eps);
        if (t < 256) {
#pragma unroll
            for (int i = 0; i < PV; ++i) {
                const int v = t + i * 256;
                if (v < NV * 64) {
                    const u32x4 nw = *(reinterpret_cast<const u32x4*>(norm_w) + v);
                    u32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[q] = pack_bf16(bf16lo(xv[i][q]) * rstd * bf16lo(nw[q]), bf16hi(xv[i][q]) * rstd * bf16hi(nw[q]));
                    xs[v] = o;
                }
            }
        }
    } else if (t < 256) {
#pragma unroll
        for (int i = 0; i < PV; ++i) {
            const int v = t + i * 256;
            if (v < NV * 64) xs[v] = xv[i];
        }
    }
    __syncthreads();
}

// value held by lane (l ^ 8) of the aligned 16-lane group (RoPE partner i <-> i + D/2)
__device__ __forceinline__ float swap_halves16(float v) { return dpp_f<0x128>(v); }   // row_ror:8

constexpr int cmax3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

// LDS map (bytes).  GEMV view and attention view overlay each other: a block is in one phase at a time.
template <int HNV, int ONV, int DNV>
struct Lds {
    static constexpr int XV = cmax3(HNV, ONV, DNV) * 64;              // staged activation, u32x4 slots
    static constexpr size_t xs = 0;
    static constexpr size_t resid = xs + (size_t)XV * 16;             // residual rows, bf16 [HNV*512]
    static constexpr size_t red = resid + (size_t)HNV * 512 * 2;      // [8] floats
    static constexpr size_t part = red + 32;                          // [kPartRows][4] floats
    static constexpr size_t queues = part + (size_t)kPartRows * 4 * 4;   // [4] ints
    static constexpr size_t gemv_end = queues + 16;
    // attention, per 4-wave group
    static constexpr size_t g_o = 0;                                               // [4][kTPW][kGT][kD] floats
    static constexpr size_t g_m = g_o + (size_t)kGroupWaves * kTPW * kGT * kD * 4; // [4][kGT]
    static constexpr size_t g_l = g_m + (size_t)kGroupWaves * kGT * 4;
    static constexpr size_t g_f = g_l + (size_t)kGroupWaves * kGT * 4;             // [kGT][kMaxSplit]
    static constexpr size_t g_L = g_f + (size_t)kGT * kMaxSplit * 4;               // [kGT]
    static constexpr size_t g_size = g_L + (size_t)kGT * 4;
    static constexpr size_t flags = (size_t)kGroups * g_size;                      // [kGroups] ints
    static constexpr size_t attn_end = flags + 16;
    static constexpr size_t total = gemv_end > attn_end ? gemv_end : attn_end;
};

template <int HNV, int ONV, int DNV>
__global__ __launch_bounds__(kBlock) void decode_mega_kernel(const MegaArgs a) {
    constexpr int DKS = DNV > 8 ? 4 : 1;          // down projection: waves sharing a row
    constexpr int DNVW = DNV / DKS;
    static_assert(DNV % DKS == 0 && DNVW <= 8 && HNV <= 8 && ONV <= 8, "unsupported shape");
    constexpr int RB_H = kSet / HNV;              // rows per register set, K = hidden
    constexpr int RB_GU = kSet / (2 * HNV);       // gate/up row pairs per set
    constexpr int RB_O = kSet / ONV;
    constexpr int RB_D = kSet / DNVW;
    using M = Lds<HNV, ONV, DNV>;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem + M::xs);
    bf16_t* resid = reinterpret_cast<bf16_t*>(smem + M::resid);
    float* red = reinterpret_cast<float*>(smem + M::red);
    float* part = reinterpret_cast<float*>(smem + M::part);
    int* queues = reinterpret_cast<int*>(smem + M::queues);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nblk = gridDim.x;
    const int hidden = a.hidden, H = a.H, Hkv = a.Hkv, G = H / Hkv;
    const int pos = a.st->pos;                    // written by an earlier launch
    const uint32_t tok = a.st->cur_token;
    GridSync gs{a.sync_words, a.epoch0, (unsigned)nblk, false};

    // ---- work split ----
    const int Tk = pos + 1;
    const int per = (Tk + a.nsplit - 1) / a.nsplit;
    const int chunk = ((per + kStep * kGroupWaves - 1) / (kStep * kGroupWaves)) * (kStep * kGroupWaves);
    const int n_active = (Tk + chunk - 1) / chunk;      // non-empty splits
    const int n_vb = Hkv * n_active;                     // virtual attention blocks (4 waves each)
    const int n_pairs = (n_vb + kGroups - 1) / kGroups;
    const int n_ab = min(n_pairs, a.attn_blocks);        // real blocks that take them, two at a time
    const bool is_attn = (int)blockIdx.x < n_ab;

    auto span = [](int n, int parts, int idx, int& r0, int& r1) {
        const int per_part = (n + parts - 1) / parts;
        r0 = min(n, idx * per_part);
        r1 = min(n, r0 + per_part);
    };
    const int Nqkv = (H + 2 * Hkv) * kD;
    int qkv_r0, qkv_r1, o_r0 = 0, o_r1 = 0, gu_r0, gu_r1, d_r0, d_r1, v_r0, v_r1;
    span(Nqkv, nblk, blockIdx.x, qkv_r0, qkv_r1);
    if (!is_attn) span(hidden, nblk - n_ab, blockIdx.x - n_ab, o_r0, o_r1);
    span(a.I, nblk, blockIdx.x, gu_r0, gu_r1);
    span(hidden, nblk, blockIdx.x, d_r0, d_r1);
    span(a.V, nblk, blockIdx.x, v_r0, v_r1);
    // down projection: quarter of K this wave streams, its index among the waves of that quarter
    const int d_q = (DKS == 1) ? 0 : (wave & 3);
    const int d_j = (DKS == 1) ? wave : (wave >> 2);
    const int d_koff = d_q * DNVW * 64;

    u32x4 wA[kSet], wB[kSet];

    // Loads for the next phase are issued just before the block waits at a barrier.  (Measured variant: the
    // polling wave issuing its share AFTER the barrier, because a wave's loads return in order and its poll
    // sits behind its own prefetch -- the release is seen ~4 us earlier by that wave but the late issue costs
    // as much)
    auto around_barrier = [&](auto&& issue, auto&& barrier) {
        issue();
        barrier();
    };

    // attention roles: 4-wave group, wave inside the group, chunk ownership inside a K/V row
    const int grp = wave / kGroupWaves, gwv = wave % kGroupWaves;
    const int ltid = threadIdx.x % kGroupThreads;
    const int c = lane % kLPR, sg = lane / kLPR;
    const bool first_half = c < kLPR / 2;
    auto vb_geom = [&](int pair, bool& on, int& kvh, int& split, int& t_begin, int& t_end) {
        const int vb = pair * kGroups + grp;
        on = vb < n_vb;
        kvh = on ? vb % Hkv : 0;
        split = on ? vb / Hkv : 0;
        t_begin = on ? split * chunk : 0;
        t_end = on ? min(Tk, t_begin + chunk) : 0;
    };

    const bf16_t* embed_row = a.embed + (size_t)tok * hidden;
    auto stamp = [&](int l, int ev) {
        if (a.trace && threadIdx.x == 0) a.trace[((size_t)l * kTraceEvents + ev) * nblk + blockIdx.x] = wall_clock64();
    };

    // layer 0's QKV rows go out before anything else
    {
        const MegaLayer& L = a.layers[0];
        const WSrc s{L.q, L.k, L.v, H * kD, Hkv * kD};
        prefetch_sets<HNV, RB_H, false, 1>(wA, wB, s, hidden, qkv_r0, qkv_r1, 0, lane, wave);
    }

    for (int l = 0; l < a.n_layers; ++l) {
        const MegaLayer& L = a.layers[l];
        const bf16_t* h_in = (l == 0) ? embed_row : a.h0;   // residual stream entering the layer

        // ===== phase 1: RMSNorm + QKV projection (model.rs:168-170, 324) =====
        stamp(l, 0);
        stage_x<HNV, true>(xs, red, queues, 2 * kWaves, nullptr, nullptr, 0, h_in, L.in_ln, a.eps, hidden);
        {
            const WSrc s{L.q, L.k, L.v, H * kD, Hkv * kD};
            stream_sets<HNV, RB_H, false, 1>(wA, wB, s, hidden, qkv_r0, qkv_r1, 0, xs, queues, lane, wave,
                                             [&](int row, float v0, float) { st_coh_bf16(a.qkv + row, f32_to_bf16(v0)); });
        }
        stamp(l, 1);
        grid_arrive(gs);

        // ===== phase 2: q/k RMSNorm + RoPE + cache append + split-KV attention + combine (model.rs:172-210) =====
        // (one branch per block role, so the O rows the other blocks hold are never live in this code)
        if (is_attn) {
            float* sm_o = reinterpret_cast<float*>(smem + grp * M::g_size + M::g_o);
            float* sm_m = reinterpret_cast<float*>(smem + grp * M::g_size + M::g_m);
            float* sm_l = reinterpret_cast<float*>(smem + grp * M::g_size + M::g_l);
            float* sm_f = reinterpret_cast<float*>(smem + grp * M::g_size + M::g_f);
            float* sm_L = reinterpret_cast<float*>(smem + grp * M::g_size + M::g_L);
            int* sm_flag = reinterpret_cast<int*>(smem + M::flags);

            u32x4 kr[kUnroll], vr[kUnroll];
            auto issue_kv = [&](const bf16_t* Kb, const bf16_t* Vb, int tbase, int t_end) {
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const int tc = max(min(tbase + u * kTPW + sg, t_end - 1), 0);
                    kr[u] = *reinterpret_cast<const u32x4*>(Kb + (size_t)tc * kD + c * 8);
                    vr[u] = *reinterpret_cast<const u32x4*>(Vb + (size_t)tc * kD + c * 8);
                }
            };
            // everything that does not depend on this layer's projections goes out before the barrier:
            // the RoPE row of this position, the q/k norm weights, the first K/V step
            float cs[8], sn[8], wq[8], wk[8];
            {
                const int i0 = (c % (kLPR / 2)) * 8;
                const f32x4* cp = reinterpret_cast<const f32x4*>(a.rope_cos + (size_t)pos * (kD / 2) + i0);
                const f32x4* sp = reinterpret_cast<const f32x4*>(a.rope_sin + (size_t)pos * (kD / 2) + i0);
                const f32x4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cs[e] = c0[e]; cs[4 + e] = c1[e];
                    sn[e] = s0[e]; sn[4 + e] = s1[e];
                }
                unpack8(*reinterpret_cast<const u32x4*>(L.q_norm + c * 8), wq);
                unpack8(*reinterpret_cast<const u32x4*>(L.k_norm + c * 8), wk);
            }
            around_barrier(
                [&]() {
                    bool on; int kvh, split, t_begin, t_end;
                    vb_geom(blockIdx.x, on, kvh, split, t_begin, t_end);
                    issue_kv(L.kc + (size_t)kvh * a.cap * kD, L.vc + (size_t)kvh * a.cap * kD, t_begin + gwv * kStep, t_end);
                },
                [&]() { grid_wait(gs); });
            stamp(l, 2);
            for (int pair = blockIdx.x; pair < n_pairs; pair += n_ab) {
                bool on; int kvh, split, t_begin, t_end;
                vb_geom(pair, on, kvh, split, t_begin, t_end);
                bf16_t* Kb = L.kc + (size_t)kvh * a.cap * kD;
                bf16_t* Vb = L.vc + (size_t)kvh * a.cap * kD;
                int t0 = t_begin + gwv * kStep;
                if (pair != (int)blockIdx.x) {
                    __syncthreads();   // the previous pair's LDS merge / combine is done
                    issue_kv(Kb, Vb, t0, t_end);
                }
                auto norm_rope = [&](const bf16_t* src, const float (&w)[8], float (&out)[8]) {
                    float x[8];
                    unpack8(ld_coh128(src + c * 8), x);
                    float ss = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss = fmaf(x[e], x[e], ss);
                    ss = group_sum<kLPR>(ss);
                    const float rstd = 1.0f / sqrtf(ss / (float)kD + a.eps);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xn = round_bf16(x[e] * rstd * w[e]);
                        const float other = swap_halves16(xn);
                        const float y = first_half ? xn * cs[e] - other * sn[e] : other * sn[e] + xn * cs[e];
                        out[e] = round_bf16(y);
                    }
                };
                float q[kGT][8];
#pragma unroll
                for (int g = 0; g < kGT; ++g) {
                    const int h = kvh * G + min(g, G - 1);
                    norm_rope(a.qkv + (size_t)h * kD, wq, q[g]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) q[g][e] *= a.scale;
                }
                stamp(l, 10);
                float m[kGT], lsum[kGT], o[kGT][8];
#pragma unroll
                for (int g = 0; g < kGT; ++g) {
                    m[g] = -INFINITY;
                    lsum[g] = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[g][e] = 0.f;
                }
                for (; t0 < t_end; t0 += kStep * kGroupWaves) {
                    float sc[kUnroll][kGT];
                    float vf[kUnroll][8];
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) {
                        const int tk = t0 + u * kTPW + sg;
                        float kf[8];
                        if (tk == pos) {
                            // this lane group owns the NEW token: build its K/V row, use it, append it (cache.rs:183-188)
                            const bf16_t* kraw = a.qkv + (size_t)H * kD + (size_t)kvh * kD;
                            const bf16_t* vraw = kraw + (size_t)Hkv * kD;
                            norm_rope(kraw, wk, kf);
                            u32x4 kp;
#pragma unroll
                            for (int e = 0; e < 4; ++e) kp[e] = pack_bf16(kf[2 * e], kf[2 * e + 1]);
                            const u32x4 vp = ld_coh128(vraw + c * 8);
                            *reinterpret_cast<u32x4*>(Kb + (size_t)pos * kD + c * 8) = kp;
                            *reinterpret_cast<u32x4*>(Vb + (size_t)pos * kD + c * 8) = vp;
                            unpack8(vp, vf[u]);
                        } else {
                            unpack8(kr[u], kf);
                            unpack8(vr[u], vf[u]);
                        }
                        if (tk >= t_end) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) vf[u][e] = 0.f;
                        }
#pragma unroll
                        for (int g = 0; g < kGT; ++g) {
                            float d = 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) d = fmaf(q[g][e], kf[e], d);
                            d = group_sum<kLPR>(d);
                            sc[u][g] = tk < t_end ? d : -INFINITY;
                        }
                    }
                    if (t0 + kStep * kGroupWaves < t_end) issue_kv(Kb, Vb, t0 + kStep * kGroupWaves, t_end);
#pragma unroll
                    for (int g = 0; g < kGT; ++g) {
                        float mx = sc[0][g];
#pragma unroll
                        for (int u = 1; u < kUnroll; ++u) mx = fmaxf(mx, sc[u][g]);
                        float wmx = readlane_f(mx, 0);
#pragma unroll
                        for (int r = 1; r < kTPW; ++r) wmx = fmaxf(wmx, readlane_f(mx, r * kLPR));
                        const float mn = fmaxf(m[g], wmx);
                        const float alpha = (mn == -INFINITY) ? 1.f : __expf(m[g] - mn);
                        m[g] = mn;
                        lsum[g] *= alpha;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[g][e] *= alpha;
#pragma unroll
                        for (int u = 0; u < kUnroll; ++u) {
                            const float p = (mn == -INFINITY) ? 0.f : __expf(sc[u][g] - mn);
                            lsum[g] += p;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[g][e] = fmaf(p, vf[u][e], o[g][e]);
                        }
                    }
                }
                stamp(l, 11);
                // every token sub-group parks its partial in LDS (same m inside a wave: plain sums)
#pragma unroll
                for (int g = 0; g < kGT; ++g) {
                    float* dst = sm_o + (((size_t)(gwv * kTPW + sg) * kGT + g) * kD + c * 8);
                    *reinterpret_cast<f32x4*>(dst) = f32x4{o[g][0], o[g][1], o[g][2], o[g][3]};
                    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[g][4], o[g][5], o[g][6], o[g][7]};
                    float lw = readlane_f(lsum[g], 0);
#pragma unroll
                    for (int r = 1; r < kTPW; ++r) lw += readlane_f(lsum[g], r * kLPR);
                    if (lane == 0) {
                        sm_m[gwv * kGT + g] = m[g];
                        sm_l[gwv * kGT + g] = lw;
                    }
                }
                __syncthreads();
                // merge the 4 waves x kTPW sub-groups, publish the split's partial
                if (on) {
                    for (int idx = ltid; idx < G * kD; idx += kGroupThreads) {
                        const int g = idx / kD, d = idx % kD;
                        float Mx = sm_m[g];
#pragma unroll
                        for (int w = 1; w < kGroupWaves; ++w) Mx = fmaxf(Mx, sm_m[w * kGT + g]);
                        float Ls = 0.f, O = 0.f;
#pragma unroll
                        for (int w = 0; w < kGroupWaves; ++w) {
                            const float mw = sm_m[w * kGT + g];
                            const float f = (mw == -INFINITY) ? 0.f : __expf(mw - Mx);
                            float ow = 0.f;
#pragma unroll
                            for (int r = 0; r < kTPW; ++r) ow += sm_o[((size_t)(w * kTPW + r) * kGT + g) * kD + d];
                            Ls = fmaf(f, sm_l[w * kGT + g], Ls);
                            O = fmaf(f, ow, O);
                        }
                        const size_t head = (size_t)kvh * G + g;
                        st_coh_f32(a.ws_o + (head * a.nsplit + split) * kD + d, O);
                        if (d == 0) {
                            st_coh_f32(a.ws_ml + (head * a.nsplit + split) * 2, Mx);
                            st_coh_f32(a.ws_ml + (head * a.nsplit + split) * 2 + 1, Ls);
                        }
                    }
                }
                // the last split block of a KV head merges its splits (attn_combine_kernel's arithmetic)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                stamp(l, 12);
                if (ltid == 0) {
                    int last = 0;
                    if (on) {
                        unsigned* cnt = a.kv_count + kvh * 16;
                        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        last = (old + 1u == (unsigned)n_active);
                        if (last) st_coh32(cnt, 0u);
                    }
                    sm_flag[grp] = last;
                }
                __syncthreads();
                stamp(l, 13);
                if (sm_flag[0] | sm_flag[1]) {              // block-uniform
                    const bool mine = sm_flag[grp] != 0;
                    const int d = ltid & (kD - 1);
                    const int g_a = ltid >> 7, g_b = g_a + 2;     // this thread's two heads
                    const bool on_a = mine && g_a < G, on_b = mine && g_b < G;
                    const size_t head0 = (size_t)kvh * G;
                    const int n4 = a.nsplit & ~3;
                    const float* src_a = a.ws_o + (head0 + (on_a ? g_a : 0)) * a.nsplit * kD + d;
                    const float* src_b = a.ws_o + (head0 + (on_b ? g_b : 0)) * a.nsplit * kD + d;
                    float va[kCombChunk], vb[kCombChunk];
                    auto load_chunk = [&](int base) {
#pragma unroll
                        for (int i = 0; i < kCombChunk; ++i) {
                            const int s = base + i;
                            va[i] = (on_a && s < n_active) ? ld_coh_f32(src_a + (size_t)s * kD) : 0.f;
                            vb[i] = (on_b && s < n_active) ? ld_coh_f32(src_b + (size_t)s * kD) : 0.f;
                        }
                    };
                    load_chunk(0);   // in flight together with the split scalars below
                    // split scalars: wave gwv -> head gwv, all four heads at once (attn_combine_kernel's phase 1)
                    if (mine && gwv < G) {
                        const float* ml = a.ws_ml + (head0 + gwv) * a.nsplit * 2;
                        float mi[kMaxSplit / 64], li[kMaxSplit / 64];
#pragma unroll
                        for (int k = 0; k < kMaxSplit / 64; ++k) {
                            const int s = lane + 64 * k;
                            const bool ok = s < n_active;
                            mi[k] = ok ? ld_coh_f32(ml + 2 * s) : -INFINITY;
                            li[k] = ok ? ld_coh_f32(ml + 2 * s + 1) : 0.f;
                        }
                        float mloc = -INFINITY;
#pragma unroll
                        for (int k = 0; k < kMaxSplit / 64; ++k)
                            if (lane + 64 * k < a.nsplit) mloc = fmaxf(mloc, mi[k]);
                        const float Mx = wave_max(mloc);
                        float lloc = 0.f;
#pragma unroll
                        for (int k = 0; k < kMaxSplit / 64; ++k) {
                            const int s = lane + 64 * k;
                            if (s < a.nsplit) {
                                const float f = (mi[k] == -INFINITY) ? 0.f : __expf(mi[k] - Mx);
                                sm_f[gwv * kMaxSplit + s] = f;
                                lloc = fmaf(f, li[k], lloc);
                            }
                        }
                        const float Lt = wave_sum(lloc);
                        if (lane == 0) sm_L[gwv] = Lt;
                    }
                    __syncthreads();
                    // the FMAs keep attn_combine_kernel's order: even/odd accumulators, the tail beyond the last
                    // multiple of 4 on acc0 (empty splits contribute exactly 0 there and are skipped here)
                    float acc0a = 0.f, acc1a = 0.f, acc0b = 0.f, acc1b = 0.f;
                    for (int base = 0; base < n_active; base += kCombChunk) {
                        if (base > 0) load_chunk(base);
#pragma unroll
                        for (int i = 0; i < kCombChunk; ++i) {
                            const int s = base + i;
                            if (s < n_active) {
                                const float fa = on_a ? sm_f[g_a * kMaxSplit + s] : 0.f;
                                const float fb = on_b ? sm_f[g_b * kMaxSplit + s] : 0.f;
                                if (s < n4 && (s & 1)) {
                                    acc1a = fmaf(fa, va[i], acc1a);
                                    acc1b = fmaf(fb, vb[i], acc1b);
                                } else {
                                    acc0a = fmaf(fa, va[i], acc0a);
                                    acc0b = fmaf(fb, vb[i], acc0b);
                                }
                            }
                        }
                    }
                    if (on_a) st_coh_bf16(a.attn_out + (head0 + g_a) * kD + d, f32_to_bf16((acc0a + acc1a) / sm_L[g_a]));
                    if (on_b) st_coh_bf16(a.attn_out + (head0 + g_b) * kD + d, f32_to_bf16((acc0b + acc1b) / sm_L[g_b]));
                    stamp(l, 14);
                }
            }
            stamp(l, 3);
            grid_arrive(gs);
            // nothing to do in the O phase: pull this wave's gate/up rows instead
            around_barrier(
                [&]() {
                    const WSrc s{L.gate, L.up, nullptr, a.I, 0};
                    prefetch_sets<HNV, RB_GU, true, 1>(wA, wB, s, hidden, gu_r0, gu_r1, 0, lane, wave);
                },
                [&]() {
                    grid_wait(gs);
                    stamp(l, 4);
                    stamp(l, 5);
                    grid_arrive(gs, false);
                    grid_wait(gs);
                });
        } else {
            around_barrier(
                [&]() {
                    const WSrc s{L.o, nullptr, nullptr, hidden, 0};
                    prefetch_sets<ONV, RB_O, false, 1>(wA, wB, s, H * kD, o_r0, o_r1, 0, lane, wave);
                },
                [&]() {
                    grid_wait(gs);
                    stamp(l, 2);
                    stamp(l, 3);
                    grid_arrive(gs, false);
                    grid_wait(gs);
                });
            stamp(l, 4);
            // ===== phase 3: O projection + residual (model.rs:214, 325) -- rows already in registers =====
            stage_x<ONV, false>(xs, red, queues, 2 * kWaves, resid, h_in, hidden, a.attn_out, nullptr, 0.f, H * kD);
            {
                const WSrc s{L.o, nullptr, nullptr, hidden, 0};
                stream_sets<ONV, RB_O, false, 1>(wA, wB, s, H * kD, o_r0, o_r1, 0, xs, queues, lane, wave, [&](int row, float v0, float) {
                    st_coh_bf16(a.h1 + row, f32_to_bf16(bf16_to_f32(resid[row]) + round_bf16(v0)));
                });
            }
            stamp(l, 5);
            grid_arrive(gs);
            around_barrier(
                [&]() {
                    const WSrc s{L.gate, L.up, nullptr, a.I, 0};
                    prefetch_sets<HNV, RB_GU, true, 1>(wA, wB, s, hidden, gu_r0, gu_r1, 0, lane, wave);
                },
                [&]() { grid_wait(gs); });
        }
        stamp(l, 6);

        // ===== phase 4: RMSNorm + gate/up + SwiGLU (model.rs:263-265, 326) =====
        stage_x<HNV, true>(xs, red, queues, 2 * kWaves, nullptr, nullptr, 0, a.h1, L.post_ln, a.eps, hidden);
        {
            const WSrc s{L.gate, L.up, nullptr, a.I, 0};
            stream_sets<HNV, RB_GU, true, 1>(wA, wB, s, hidden, gu_r0, gu_r1, 0, xs, queues, lane, wave, [&](int row, float v0, float v1) {
                // nn::silu(gate) * up, every primitive's result held in bf16 (activation.rs:876-880)
                const float g = round_bf16(v0);
                const float u = round_bf16(v1);
                const float sgm = round_bf16(1.0f / (1.0f + expf(-g)));
                st_coh_bf16(a.act + row, f32_to_bf16(round_bf16(g * sgm) * u));
            });
        }
        stamp(l, 7);
        grid_arrive(gs);
        around_barrier(
            [&]() {
                const WSrc s{L.down, nullptr, nullptr, hidden, 0};
                prefetch_sets<DNVW, RB_D, false, DKS>(wA, wB, s, a.I, d_r0, d_r1, d_koff, lane, d_j);
            },
            [&]() { grid_wait(gs); });
        stamp(l, 8);

        // ===== phase 5: down projection + residual (model.rs:266, 327) =====
        stage_x<DNV, false>(xs, red, queues, 2 * (kWaves / DKS), resid, a.h1, hidden, a.act, nullptr, 0.f, a.I);
        {
            const WSrc s{L.down, nullptr, nullptr, hidden, 0};
            if (DKS == 1) {
                stream_sets<DNVW, RB_D, false, 1>(wA, wB, s, a.I, d_r0, d_r1, 0, xs, queues, lane, wave, [&](int row, float v0, float) {
                    st_coh_bf16(a.h0 + row, f32_to_bf16(bf16_to_f32(resid[row]) + round_bf16(v0)));
                });
            } else {
                stream_sets<DNVW, RB_D, false, DKS>(wA, wB, s, a.I, d_r0, d_r1, d_koff, xs, queues + d_q, lane, d_j,
                                                    [&](int row, float v0, float) { part[(row - d_r0) * DKS + d_q] = v0; });
                __syncthreads();
                const int lr = threadIdx.x;
                if (lr < d_r1 - d_r0) {
                    float v0 = 0.f;
#pragma unroll
                    for (int w = 0; w < DKS; ++w) v0 += part[lr * DKS + w];
                    const int row = d_r0 + lr;
                    st_coh_bf16(a.h0 + row, f32_to_bf16(bf16_to_f32(resid[row]) + round_bf16(v0)));
                }
            }
        }
        stamp(l, 9);
        grid_arrive(gs);
        // next layer's QKV rows (or the vocabulary rows) around the layer's last barrier
        around_barrier(
            [&]() {
                if (l + 1 < a.n_layers) {
                    const MegaLayer& Ln = a.layers[l + 1];
                    const WSrc s{Ln.q, Ln.k, Ln.v, H * kD, Hkv * kD};
                    prefetch_sets<HNV, RB_H, false, 1>(wA, wB, s, hidden, qkv_r0, qkv_r1, 0, lane, wave);
                } else {
                    const WSrc s{a.lm_head, nullptr, nullptr, a.V, 0};
                    prefetch_sets<HNV, RB_H, false, 1>(wA, wB, s, hidden, v_r0, a.with_head ? v_r1 : v_r0, 0, lane, wave);
                }
            },
            [&]() { grid_wait(gs); });
    }

    if (a.with_head) {
        // ===== final RMSNorm + lm_head + greedy argmax (model.rs:423, 480-489, 733-735; sampler.rs:9-18) =====
        stage_x<HNV, true>(xs, red, queues, 2 * kWaves, nullptr, nullptr, 0, a.h0, a.final_norm, a.eps, hidden);
        uint64_t best = 0;
        {
            const WSrc s{a.lm_head, nullptr, nullptr, a.V, 0};
            stream_sets<HNV, RB_H, false, 1>(wA, wB, s, hidden, v_r0, v_r1, 0, xs, queues, lane, wave, [&](int row, float v0, float) {
                const bf16_t lb = f32_to_bf16(v0);
                a.logits[row] = lb;
                const uint64_t key = argmax_key(bf16_to_f32(lb), (uint32_t)row);
                best = key > best ? key : best;
            });
        }
        uint64_t* bred = reinterpret_cast<uint64_t*>(part);   // [kWaves] u64
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
            st_coh64(a.argmax_partials + blockIdx.x, b);
        }
        grid_sync(gs);
        if (blockIdx.x == 0) {
            uint64_t b2 = 0;
            for (int i = threadIdx.x; i < nblk; i += kBlock) {
                const uint64_t p = ld_coh64(a.argmax_partials + i);
                b2 = p > b2 ? p : b2;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint64_t other = __shfl_xor(b2, o, 64);
                b2 = other > b2 ? other : b2;
            }
            __syncthreads();
            if (lane == 0) bred[wave] = b2;
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int w = 1; w < kWaves; ++w) b2 = bred[w] > b2 ? bred[w] : b2;
                const uint32_t t = ~(uint32_t)(b2 & 0xFFFFFFFFull);
                a.out_ring[a.st->out_count % a.ring_cap] = t;
                a.st->out_count += 1;
                a.st->cur_token = t;
                a.st->pos = pos + 1;
            }
        }
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        // prompt token: advance and feed the next one (every block read pos / cur_token before its first barrier)
        a.st->pos = pos + 1;
        a.st->prompt_idx += 1;
        a.st->cur_token = a.prompt[a.st->prompt_idx];
    }
}

struct Variant {
    int hnv, onv, dnv;
    const void* fn;
    size_t smem;
};
#define OMX_MEGA_VARIANT(A, B, C) \
    Variant { A, B, C, (const void*)decode_mega_kernel<A, B, C>, Lds<A, B, C>::total }
const Variant kVariants[] = {
    OMX_MEGA_VARIANT(8, 8, 24),    // Qwen3-8B: hidden 4096, H*D 4096, I 12288
    OMX_MEGA_VARIANT(2, 4, 6),     // Qwen3-0.6B: 1024, 2048, 3072
    OMX_MEGA_VARIANT(4, 4, 12),    // Qwen3-1.7B: 2048, 2048, 6144
    OMX_MEGA_VARIANT(2, 2, 6),     // test shapes
    OMX_MEGA_VARIANT(2, 2, 12),
};
#undef OMX_MEGA_VARIANT

const Variant* find_variant(int hidden, int attn_width, int inter) {
    if (hidden % 512 || attn_width % 512 || inter % 512) return nullptr;
    for (const Variant& v : kVariants)
        if (v.hnv == hidden / 512 && v.onv == attn_width / 512 && v.dnv == inter / 512) return &v;
    return nullptr;
}

}  // namespace

bool mega_supported(int hidden, int attn_width, int inter, int head_dim, int group) {
    return head_dim == kD && group >= 1 && group <= kGT && find_variant(hidden, attn_width, inter) != nullptr;
}

int mega_capacity(int hidden, int attn_width, int inter, int* blocks) {
    const Variant* v = find_variant(hidden, attn_width, inter);
    OMX_REQUIRE(v != nullptr, "decode megakernel: no instantiation for hidden %d / attention width %d / intermediate %d", hidden, attn_width, inter);
    OMX_HIP_CHECK(hipFuncSetAttribute(v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v->smem));
    int per_cu = 0, dev = 0, cus = 0;
    OMX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v->fn, kBlock, v->smem));
    OMX_HIP_CHECK(hipGetDevice(&dev));
    OMX_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    *blocks = per_cu * cus;
    return 0;
}

int launch_decode_mega(const MegaArgs& a, int nblocks, hipStream_t s) {
    const Variant* v = find_variant(a.hidden, a.H * kD, a.I);
    OMX_REQUIRE(v != nullptr, "decode megakernel: unsupported shape");
    OMX_REQUIRE(a.n_layers >= 1, "decode megakernel: no layers");
    OMX_REQUIRE(a.nsplit >= 1 && a.nsplit <= kMaxSplit, "decode megakernel: nsplit %d out of range", a.nsplit);
    OMX_REQUIRE(a.attn_blocks >= 1 && a.attn_blocks < nblocks, "decode megakernel: attention blocks %d of %d", a.attn_blocks, nblocks);
    OMX_REQUIRE((a.hidden + nblocks - 1) / nblocks <= kPartRows, "decode megakernel: %d blocks too few for hidden %d", nblocks, a.hidden);
    void* params[] = {const_cast<MegaArgs*>(&a)};
    OMX_HIP_CHECK(hipLaunchKernel(v->fn, dim3(nblocks), dim3(kBlock), params, v->smem, s));
    return 0;
}

}  // namespace omx
