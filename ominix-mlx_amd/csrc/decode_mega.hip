// One persistent kernel per decode token for gfx950: every phase of every layer of
// qwen3-mlx's decode step (qwen3-mlx/src/model.rs:168-327, 423, 480-489, 733-735) plus the greedy
// sampler, with device-wide barriers (gridsync.hpp) where the launch boundaries used to be.
//
// Why: as separate launches (engine.hip's step graph) each of the 5 dependent phases per layer pays
// launch + HBM ramp + drain; the rocprof summary (profiles/r01_b_kernel_stats_ctx2048.csv) puts the
// GEMV phases at 4.2-5.8 TB/s against the 6.3 TB/s the lm_head kernel reaches in steady state, and
// attention + combine at 15 us for 9 MB.  Here
//   * a block issues the first TWO register sets of the NEXT phase's weight rows before it waits at
//     the barrier, so HBM keeps streaming while the barrier resolves and the activation is staged;
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

constexpr int kBlock = 256;
constexpr int kWaves = 4;
constexpr int kSet = 16;         // 16-byte vectors per lane per register set (two sets in flight)
constexpr int kD = 128;          // head_dim
constexpr int kGT = 4;           // query heads per KV head held in registers
constexpr int kLPR = kD / 8;     // lanes per K/V row
constexpr int kTPW = 64 / kLPR;  // tokens per wave-instruction
constexpr int kUnroll = 4;
constexpr int kStep = kTPW * kUnroll;   // tokens per wave per step
constexpr int kMaxSplit = 512;
constexpr int kPartRows = 64;    // down projection, K split over the block's waves: rows per block

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

// both register sets of [r0, r0 + 2*RB) are already in flight (issued before the barrier)
template <int NVW, int RB, bool PAIR, class Epi>
__device__ __forceinline__ void stream_rows(u32x4 (&wA)[kSet], u32x4 (&wB)[kSet], const WSrc& s, int K, int r0, int r1,
                                            int koff, const u32x4* xs, int lane, Epi&& epi) {
    for (int r = r0; r < r1; r += 2 * RB) {
        compute_set<NVW, RB, PAIR>(wA, xs, koff, r, r1, lane, epi);
        if (r + 2 * RB < r1) issue_set<NVW, RB, PAIR>(wA, s, K, r + 2 * RB, r1, koff, lane);
        if (r + RB < r1) {
            compute_set<NVW, RB, PAIR>(wB, xs, koff, r + RB, r1, lane, epi);
            if (r + 3 * RB < r1) issue_set<NVW, RB, PAIR>(wB, s, K, r + 3 * RB, r1, koff, lane);
        }
    }
}

template <int NVW, int RB, bool PAIR>
__device__ __forceinline__ void prefetch_rows(u32x4 (&wA)[kSet], u32x4 (&wB)[kSet], const WSrc& s, int K, int r0, int r1,
                                              int koff, int lane) {
    issue_set<NVW, RB, PAIR>(wA, s, K, r0, r1, koff, lane);
    issue_set<NVW, RB, PAIR>(wB, s, K, r0 + RB, r1, koff, lane);
}

// ---- activation [K] -> LDS as bf16, optionally RMS-normalised (same arithmetic as gemv.hip's prologue) ----
template <int NV, bool NORM>
__device__ __forceinline__ void stage_x(u32x4* xs, float* red, const bf16_t* xg, const bf16_t* norm_w, float eps, int K) {
    constexpr int PV = (NV * 64 + kBlock - 1) / kBlock;
    u32x4 xv[PV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < PV; ++i) {
        const int v = threadIdx.x + i * kBlock;
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
    if (NORM) {
        ss = block_sum<kWaves>(ss, red);
        const float rstd = 1.0f / sqrtf(ss / (float)K + eps);
#pragma unroll
        for (int i = 0; i < PV; ++i) {
            const int v = threadIdx.x + i * kBlock;
            if (v < NV * 64) {
                const u32x4 nw = *(reinterpret_cast<const u32x4*>(norm_w) + v);
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q] = pack_bf16(bf16lo(xv[i][q]) * rstd * bf16lo(nw[q]), bf16hi(xv[i][q]) * rstd * bf16hi(nw[q]));
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

// value held by lane (l ^ 8) of the aligned 16-lane group (RoPE partner i <-> i + D/2)
__device__ __forceinline__ float swap_halves16(float v) { return dpp_f<0x128>(v); }   // row_ror:8

template <int HNV, int ONV, int DNV>
__global__ __launch_bounds__(kBlock, 2) void decode_mega_kernel(const MegaArgs a) {
    constexpr int DKS = DNV > 8 ? 4 : 1;          // down projection: waves sharing a row
    constexpr int DNVW = DNV / DKS;
    static_assert(DNV % DKS == 0 && DNVW <= 8 && HNV <= 8 && ONV <= 8, "unsupported shape");
    constexpr int RB_H = kSet / HNV;              // rows per register set, K = hidden
    constexpr int RB_GU = kSet / (2 * HNV);       // gate/up row pairs per set
    constexpr int RB_O = kSet / ONV;
    constexpr int RB_D = kSet / DNVW;
    constexpr int XV = (HNV > ONV ? (HNV > DNV ? HNV : DNV) : (ONV > DNV ? ONV : DNV)) * 64;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // GEMV view
    u32x4* xs = reinterpret_cast<u32x4*>(smem);                              // [XV] staged activation
    float* red = reinterpret_cast<float*>(smem + (size_t)XV * 16);           // [8] block-reduce scratch
    float* part = red + 8;                                                   // [kPartRows][DKS]
    // attention view (a block is in one phase at a time)
    float* sm_o = reinterpret_cast<float*>(smem);                            // [kWaves][kTPW][kGT][kD]
    float* sm_m = sm_o + kWaves * kTPW * kGT * kD;                           // [kWaves][kGT]
    float* sm_l = sm_m + kWaves * kGT;                                       // [kWaves][kGT]
    float* sm_f = sm_l + kWaves * kGT;                                       // [2][kMaxSplit]
    float* sm_L = sm_f + 2 * kMaxSplit;                                      // [2]
    int* sm_flag = reinterpret_cast<int*>(sm_L + 2);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nblk = gridDim.x;
    const int nwaves = nblk * kWaves;
    const int gw = blockIdx.x * kWaves + wave;
    const int hidden = a.hidden, H = a.H, Hkv = a.Hkv, G = H / Hkv;
    const int pos = a.st->pos;                    // written by an earlier launch
    const uint32_t tok = a.st->cur_token;
    GridSync gs{a.sync_words, a.epoch0, (unsigned)nblk, false};

    // ---- static work split ----
    const int Tk = pos + 1;
    const int per = (Tk + a.nsplit - 1) / a.nsplit;
    const int chunk = ((per + kStep * kWaves - 1) / (kStep * kWaves)) * (kStep * kWaves);
    const int n_active = (Tk + chunk - 1) / chunk;      // non-empty splits
    const int n_vb = Hkv * n_active;                     // virtual attention blocks
    const int n_ab = min(n_vb, a.attn_blocks);           // real blocks that take them
    const bool is_attn = (int)blockIdx.x < n_ab;

    auto span = [](int n, int parts, int idx, int& r0, int& r1) {
        const int per_part = (n + parts - 1) / parts;
        r0 = min(n, idx * per_part);
        r1 = min(n, r0 + per_part);
    };
    const int Nqkv = (H + 2 * Hkv) * kD;
    int qkv_r0, qkv_r1, o_r0 = 0, o_r1 = 0, gu_r0, gu_r1, d_r0, d_r1, v_r0, v_r1;
    span(Nqkv, nwaves, gw, qkv_r0, qkv_r1);
    if (!is_attn) span(hidden, (nblk - n_ab) * kWaves, gw - n_ab * kWaves, o_r0, o_r1);
    span(a.I, nwaves, gw, gu_r0, gu_r1);
    if (DKS == 1) span(hidden, nwaves, gw, d_r0, d_r1);
    else span(hidden, nblk, blockIdx.x, d_r0, d_r1);
    span(a.V, nwaves, gw, v_r0, v_r1);
    const int d_koff = (DKS == 1) ? 0 : wave * DNVW * 64;

    u32x4 wA[kSet], wB[kSet];
    u32x4 kr[kUnroll], vr[kUnroll];

    // chunk ownership inside a K/V row
    const int c = lane % kLPR, sg = lane / kLPR;
    const bool first_half = c < kLPR / 2;
    auto issue_kv = [&](const bf16_t* Kb, const bf16_t* Vb, int tbase, int t_end) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int tc = max(min(tbase + u * kTPW + sg, t_end - 1), 0);
            kr[u] = *reinterpret_cast<const u32x4*>(Kb + (size_t)tc * kD + c * 8);
            vr[u] = *reinterpret_cast<const u32x4*>(Vb + (size_t)tc * kD + c * 8);
        }
    };
    auto vb_geom = [&](int vb, int& kvh, int& split, int& t_begin, int& t_end) {
        kvh = vb % Hkv;
        split = vb / Hkv;
        t_begin = split * chunk;
        t_end = min(Tk, t_begin + chunk);
    };

    const bf16_t* embed_row = a.embed + (size_t)tok * hidden;
    auto stamp = [&](int l, int ev) {
        if (a.trace && threadIdx.x == 0) a.trace[((size_t)l * kTraceEvents + ev) * nblk + blockIdx.x] = wall_clock64();
    };

    // layer 0's QKV rows go out before anything else
    {
        const MegaLayer& L = a.layers[0];
        const WSrc s{L.q, L.k, L.v, H * kD, Hkv * kD};
        prefetch_rows<HNV, RB_H, false>(wA, wB, s, hidden, qkv_r0, qkv_r1, 0, lane);
    }

    for (int l = 0; l < a.n_layers; ++l) {
        const MegaLayer& L = a.layers[l];
        const bf16_t* h_in = (l == 0) ? embed_row : a.h0;   // residual stream entering the layer

        // ===== phase 1: RMSNorm + QKV projection (model.rs:168-170, 324) =====
        if (l > 0) grid_wait(gs);                            // h0 of the previous layer is complete
        stamp(l, 0);
        stage_x<HNV, true>(xs, red, h_in, L.in_ln, a.eps, hidden);
        {
            const WSrc s{L.q, L.k, L.v, H * kD, Hkv * kD};
            stream_rows<HNV, RB_H, false>(wA, wB, s, hidden, qkv_r0, qkv_r1, 0, xs, lane,
                                          [&](int row, float v0, float) { st_coh_bf16(a.qkv + row, f32_to_bf16(v0)); });
        }
        stamp(l, 1);
        grid_arrive(gs);

        // ===== phase 2: q/k RMSNorm + RoPE + cache append + split-KV attention + combine (model.rs:172-210) =====
        // (one branch per block role, so the O rows the other blocks hold are never live in this code)
        if (is_attn) {
            {
                int kvh, split, t_begin, t_end;
                vb_geom(blockIdx.x, kvh, split, t_begin, t_end);
                const int t0 = t_begin + wave * kStep;
                if (t0 < t_end) issue_kv(L.kc + (size_t)kvh * a.cap * kD, L.vc + (size_t)kvh * a.cap * kD, t0, t_end);
            }
            grid_wait(gs);
            stamp(l, 2);
            float cs[8], sn[8];   // RoPE row of this position
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
            }
            for (int vb = blockIdx.x; vb < n_vb; vb += n_ab) {
                int kvh, split, t_begin, t_end;
                vb_geom(vb, kvh, split, t_begin, t_end);
                bf16_t* Kb = L.kc + (size_t)kvh * a.cap * kD;
                bf16_t* Vb = L.vc + (size_t)kvh * a.cap * kD;
                int t0 = t_begin + wave * kStep;
                if (vb != (int)blockIdx.x) {
                    __syncthreads();   // the previous virtual block's LDS merge is done
                    if (t0 < t_end) issue_kv(Kb, Vb, t0, t_end);
                }
                float wk[8];
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
                {
                    float wq[8];
                    unpack8(*reinterpret_cast<const u32x4*>(L.q_norm + c * 8), wq);
                    unpack8(*reinterpret_cast<const u32x4*>(L.k_norm + c * 8), wk);
#pragma unroll
                    for (int g = 0; g < kGT; ++g) {
                        const int h = kvh * G + min(g, G - 1);
                        norm_rope(a.qkv + (size_t)h * kD, wq, q[g]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) q[g][e] *= a.scale;
                    }
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
                for (; t0 < t_end; t0 += kStep * kWaves) {
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
                    if (t0 + kStep * kWaves < t_end) issue_kv(Kb, Vb, t0 + kStep * kWaves, t_end);
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
                    float* dst = sm_o + (((size_t)(wave * kTPW + sg) * kGT + g) * kD + c * 8);
                    *reinterpret_cast<f32x4*>(dst) = f32x4{o[g][0], o[g][1], o[g][2], o[g][3]};
                    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[g][4], o[g][5], o[g][6], o[g][7]};
                    float lw = readlane_f(lsum[g], 0);
#pragma unroll
                    for (int r = 1; r < kTPW; ++r) lw += readlane_f(lsum[g], r * kLPR);
                    if (lane == 0) {
                        sm_m[wave * kGT + g] = m[g];
                        sm_l[wave * kGT + g] = lw;
                    }
                }
                __syncthreads();
                // merge the 4 waves x kTPW sub-groups, publish the split's partial
                for (int idx = threadIdx.x; idx < G * kD; idx += kBlock) {
                    const int g = idx / kD, d = idx % kD;
                    float M = sm_m[g];
#pragma unroll
                    for (int w = 1; w < kWaves; ++w) M = fmaxf(M, sm_m[w * kGT + g]);
                    float Ls = 0.f, O = 0.f;
#pragma unroll
                    for (int w = 0; w < kWaves; ++w) {
                        const float mw = sm_m[w * kGT + g];
                        const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
                        float ow = 0.f;
#pragma unroll
                        for (int r = 0; r < kTPW; ++r) ow += sm_o[((size_t)(w * kTPW + r) * kGT + g) * kD + d];
                        Ls = fmaf(f, sm_l[w * kGT + g], Ls);
                        O = fmaf(f, ow, O);
                    }
                    const size_t head = (size_t)kvh * G + g;
                    st_coh_f32(a.ws_o + (head * a.nsplit + split) * kD + d, O);
                    if (d == 0) {
                        st_coh_f32(a.ws_ml + (head * a.nsplit + split) * 2, M);
                        st_coh_f32(a.ws_ml + (head * a.nsplit + split) * 2 + 1, Ls);
                    }
                }
                // the last split block of this KV head merges the splits (attn_combine_kernel's arithmetic)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                stamp(l, 12);
                if (threadIdx.x == 0) {
                    unsigned* cnt = a.kv_count + kvh * 16;
                    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int last = (old + 1u == (unsigned)n_active);
                    if (last) st_coh32(cnt, 0u);
                    *sm_flag = last;
                }
                __syncthreads();
                stamp(l, 13);
                if (*sm_flag) {
                    const int half = threadIdx.x >> 7;          // two heads at a time, kD threads each
                    const int d = threadIdx.x & (kD - 1);
                    for (int g0 = 0; g0 < G; g0 += 2) {
                        const int g = g0 + half;
                        const bool on = g < G;
                        const size_t head = (size_t)kvh * G + (on ? g : 0);
                        const float* ml = a.ws_ml + head * a.nsplit * 2;
                        float* f_of = sm_f + half * kMaxSplit;
                        if (on && (wave & 1) == 0) {             // first wave of the half: split-parallel scalars
                            float mloc = -INFINITY;
                            for (int i = lane; i < a.nsplit; i += 64)
                                mloc = fmaxf(mloc, i < n_active ? ld_coh_f32(ml + 2 * i) : -INFINITY);
                            const float M = wave_max(mloc);
                            float lloc = 0.f;
                            for (int i = lane; i < a.nsplit; i += 64) {
                                const float mi = i < n_active ? ld_coh_f32(ml + 2 * i) : -INFINITY;
                                const float f = (mi == -INFINITY) ? 0.f : __expf(mi - M);
                                f_of[i] = f;
                                lloc = fmaf(f, i < n_active ? ld_coh_f32(ml + 2 * i + 1) : 0.f, lloc);
                            }
                            const float Lt = wave_sum(lloc);
                            if (lane == 0) sm_L[half] = Lt;
                        }
                        __syncthreads();
                        if (on) {
                            const float* src = a.ws_o + head * a.nsplit * kD + d;
                            auto ldv = [&](int i) { return i < n_active ? ld_coh_f32(src + (size_t)i * kD) : 0.f; };
                            float acc0 = 0.f, acc1 = 0.f;
                            int i = 0;
                            for (; i + 16 <= a.nsplit; i += 16) {
                                float v[16];
#pragma unroll
                                for (int j = 0; j < 16; ++j) v[j] = ldv(i + j);
#pragma unroll
                                for (int j = 0; j < 16; j += 2) {
                                    acc0 = fmaf(f_of[i + j], v[j], acc0);
                                    acc1 = fmaf(f_of[i + j + 1], v[j + 1], acc1);
                                }
                            }
                            for (; i + 4 <= a.nsplit; i += 4) {
                                float v[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = ldv(i + j);
                                acc0 = fmaf(f_of[i], v[0], acc0);
                                acc1 = fmaf(f_of[i + 1], v[1], acc1);
                                acc0 = fmaf(f_of[i + 2], v[2], acc0);
                                acc1 = fmaf(f_of[i + 3], v[3], acc1);
                            }
                            for (; i < a.nsplit; ++i) acc0 = fmaf(f_of[i], ldv(i), acc0);
                            st_coh_bf16(a.attn_out + head * kD + d, f32_to_bf16((acc0 + acc1) / sm_L[half]));
                        }
                        __syncthreads();
                    }
                    stamp(l, 14);
                }
            }
            stamp(l, 3);
            grid_arrive(gs);
            // nothing to do in the O phase: pull this wave's gate/up rows instead
            const WSrc s{L.gate, L.up, nullptr, a.I, 0};
            prefetch_rows<HNV, RB_GU, true>(wA, wB, s, hidden, gu_r0, gu_r1, 0, lane);
            grid_wait(gs);
            stamp(l, 4);
            stamp(l, 5);
            grid_arrive(gs, false);
        } else {
            {
                const WSrc s{L.o, nullptr, nullptr, hidden, 0};
                prefetch_rows<ONV, RB_O, false>(wA, wB, s, H * kD, o_r0, o_r1, 0, lane);
            }
            grid_wait(gs);
            stamp(l, 2);
            stamp(l, 3);
            grid_arrive(gs, false);
            grid_wait(gs);
            stamp(l, 4);
            // ===== phase 3: O projection + residual (model.rs:214, 325) -- rows already in registers =====
            stage_x<ONV, false>(xs, red, a.attn_out, nullptr, 0.f, H * kD);
            const WSrc s{L.o, nullptr, nullptr, hidden, 0};
            stream_rows<ONV, RB_O, false>(wA, wB, s, H * kD, o_r0, o_r1, 0, xs, lane, [&](int row, float v0, float) {
                st_coh_bf16(a.h1 + row, f32_to_bf16(ld_coh_bf16(h_in + row) + round_bf16(v0)));
            });
            stamp(l, 5);
            grid_arrive(gs);
            const WSrc sg2{L.gate, L.up, nullptr, a.I, 0};
            prefetch_rows<HNV, RB_GU, true>(wA, wB, sg2, hidden, gu_r0, gu_r1, 0, lane);
        }
        grid_wait(gs);
        stamp(l, 6);

        // ===== phase 4: RMSNorm + gate/up + SwiGLU (model.rs:263-265, 326) =====
        stage_x<HNV, true>(xs, red, a.h1, L.post_ln, a.eps, hidden);
        {
            const WSrc s{L.gate, L.up, nullptr, a.I, 0};
            stream_rows<HNV, RB_GU, true>(wA, wB, s, hidden, gu_r0, gu_r1, 0, xs, lane, [&](int row, float v0, float v1) {
                // nn::silu(gate) * up, every primitive's result held in bf16 (activation.rs:876-880)
                const float g = round_bf16(v0);
                const float u = round_bf16(v1);
                const float sgm = round_bf16(1.0f / (1.0f + expf(-g)));
                st_coh_bf16(a.act + row, f32_to_bf16(round_bf16(g * sgm) * u));
            });
        }
        stamp(l, 7);
        grid_arrive(gs);
        {
            const WSrc s{L.down, nullptr, nullptr, hidden, 0};
            prefetch_rows<DNVW, RB_D, false>(wA, wB, s, a.I, d_r0, d_r1, d_koff, lane);
        }
        grid_wait(gs);
        stamp(l, 8);

        // ===== phase 5: down projection + residual (model.rs:266, 327) =====
        stage_x<DNV, false>(xs, red, a.act, nullptr, 0.f, a.I);
        {
            const WSrc s{L.down, nullptr, nullptr, hidden, 0};
            if (DKS == 1) {
                stream_rows<DNVW, RB_D, false>(wA, wB, s, a.I, d_r0, d_r1, 0, xs, lane, [&](int row, float v0, float) {
                    st_coh_bf16(a.h0 + row, f32_to_bf16(ld_coh_bf16(a.h1 + row) + round_bf16(v0)));
                });
            } else {
                stream_rows<DNVW, RB_D, false>(wA, wB, s, a.I, d_r0, d_r1, d_koff, xs, lane,
                                               [&](int row, float v0, float) { part[(row - d_r0) * DKS + wave] = v0; });
                __syncthreads();
                const int lr = threadIdx.x;
                if (lr < d_r1 - d_r0) {
                    float v0 = 0.f;
#pragma unroll
                    for (int w = 0; w < DKS; ++w) v0 += part[lr * DKS + w];
                    const int row = d_r0 + lr;
                    st_coh_bf16(a.h0 + row, f32_to_bf16(ld_coh_bf16(a.h1 + row) + round_bf16(v0)));
                }
            }
        }
        stamp(l, 9);
        grid_arrive(gs);
        if (l + 1 < a.n_layers) {
            const MegaLayer& Ln = a.layers[l + 1];
            const WSrc s{Ln.q, Ln.k, Ln.v, H * kD, Hkv * kD};
            prefetch_rows<HNV, RB_H, false>(wA, wB, s, hidden, qkv_r0, qkv_r1, 0, lane);
        } else if (a.with_head) {
            const WSrc s{a.lm_head, nullptr, nullptr, a.V, 0};
            prefetch_rows<HNV, RB_H, false>(wA, wB, s, hidden, v_r0, v_r1, 0, lane);
        }
        // the matching grid_wait is at the top of the next layer / before the head
    }
    grid_wait(gs);

    if (a.with_head) {
        // ===== final RMSNorm + lm_head + greedy argmax (model.rs:423, 480-489, 733-735; sampler.rs:9-18) =====
        const bf16_t* h_fin = (a.n_layers == 0) ? embed_row : a.h0;
        stage_x<HNV, true>(xs, red, h_fin, a.final_norm, a.eps, hidden);
        uint64_t best = 0;
        {
            const WSrc s{a.lm_head, nullptr, nullptr, a.V, 0};
            stream_rows<HNV, RB_H, false>(wA, wB, s, hidden, v_r0, v_r1, 0, xs, lane, [&](int row, float v0, float) {
                const bf16_t lb = f32_to_bf16(v0);
                a.logits[row] = lb;
                const uint64_t key = argmax_key(bf16_to_f32(lb), (uint32_t)row);
                best = key > best ? key : best;
            });
        }
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

constexpr size_t smem_bytes(int xv_vectors) {
    const size_t gemv = (size_t)xv_vectors * 64 * 16 + 32 + (size_t)kPartRows * 4 * 4;
    const size_t attn = ((size_t)kWaves * kTPW * kGT * kD + 2 * kWaves * kGT + 2 * kMaxSplit + 2 + 2) * 4;
    return gemv > attn ? gemv : attn;
}

struct Variant {
    int hnv, onv, dnv;
    const void* fn;
    size_t smem;
};
#define OMX_MEGA_VARIANT(A, B, C) \
    Variant { A, B, C, (const void*)decode_mega_kernel<A, B, C>, smem_bytes((A > B ? (A > C ? A : C) : (B > C ? B : C))) }
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
    OMX_REQUIRE(a.nsplit >= 1 && a.nsplit <= kMaxSplit, "decode megakernel: nsplit %d out of range", a.nsplit);
    OMX_REQUIRE(a.attn_blocks >= 1 && a.attn_blocks < nblocks, "decode megakernel: attention blocks %d of %d", a.attn_blocks, nblocks);
    OMX_REQUIRE((a.hidden + nblocks - 1) / nblocks <= kPartRows, "decode megakernel: %d blocks too few for hidden %d", nblocks, a.hidden);
    void* params[] = {const_cast<MegaArgs*>(&a)};
    OMX_HIP_CHECK(hipLaunchKernel(v->fn, dim3(nblocks), dim3(kBlock), params, v->smem, s));
    return 0;
}

}  // namespace omx
