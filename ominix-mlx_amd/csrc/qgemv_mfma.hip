// MLX 4-bit (group 64) Linear at batch 1 with the multiply on the MATRIX cores (round 6): nn::QuantizedLinear::forward,
// mlx-rs/src/nn/quantized.rs:361-385, for the dense decode step's q/k/v, gate/up, down and lm_head launches.
//
// Why: quant.hip's qgemv_kernel turns every packed word into four v_perm + four v_dot2c (+ 3 mask / shift operations): 1.5 VALU
// operations per weight, the dot products among the slow ones -- it streams at 4.6 TB/s-equivalent where the bf16 GEMV streams at
// 6.7 (EXPERIMENTS R5-5).  Here the VALU only assembles bf16 values (0x4300 | q = 128 + q: 7 cheap operations per word of 8 weights)
// and v_mfma_f32_16x16x32_bf16 multiplies: per instruction 16 output rows x 32 weights against the activation.
//
// Mapping.  A wave owns 16 output rows x one 1 024-element "superchunk" of K (16 quantisation groups = 512 B per row); the KS waves of
// a block own the KS superchunks of the same 16 rows (K == KS * 1024).  Lane (n = lane & 15, b = lane >> 4) loads, for t = 0 .. 7, the
// 16 bytes [64 t + 16 b, +16) of row n's superchunk: eight 1 KB wave loads, each row's 64 bytes contiguous.  Word j of load t is the
// B operand (k block b, output row n) of MFMA (t, j).  The A operand is NOT the activation broadcast over its 16 rows: row m of A is
// "slot" m = quantisation group m of the superchunk, holding x only where the k block it meets belongs to that group and zero
// elsewhere -- A_j [m, b] = x[128 (m >> 1) + 32 b + 8 j, +8) if (b >> 1) == (m & 1).  These four A fragments do not depend on t, so
// they are loaded once per superchunk; MFMA (t, j) then leaves the group sums of groups 2 t and 2 t + 1 in rows 2 t, 2 t + 1 of its
// result and garbage (other loads' activations against this load's weights) in the rows nobody reads.  Eight accumulators, one per t;
// lane (n, q) picks its four valid group sums D_{2q + (i >> 1)}[i] (groups 4 q + i of row n), multiplies by the group's scale, adds
// (bias - 128 scale) * sum(x over the group), and the 16 groups x KS superchunks are summed over lanes and waves in a fixed order.
#include <algorithm>

#include "act16.hpp"
#include "launch_timing.hpp"
#include "quant.hpp"

namespace omx {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ uint64_t qm_argmax_key(float v, uint32_t idx) {   // the key of quant.hip / gemv.hip: larger value, then lower index
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    if (v != v) u = 0;
    return ((uint64_t)u << 32) | (uint32_t)(~idx);
}

struct QmUnit {            // 16 rows x one superchunk: 8 KB of packed words + 1 KB of scale / bias pairs per wave
    u32x4 wd[8];
    u32x4 sb;
};
constexpr int kTileWords = 2304;   // 9 KB: [8 loads][64 lanes][4 words] + [64 lanes][4 (scale | bias << 16) words]

// checkpoint layout -> tiles (built once by the engine, launch_qgemv4m_repack): tile (rb, w) = rows [16 rb, +16) x superchunk w in the
// order the kernel's lanes consume it, so that every wave load is 1 KB of consecutive bytes and a wave's units are consecutive tiles
__global__ __launch_bounds__(256) void qgemv4m_repack_kernel(uint32_t* __restrict__ tiles, const uint32_t* __restrict__ wq,
                                                             const bf16_t* __restrict__ scales, const bf16_t* __restrict__ biases, int n,
                                                             int KS) {
    const int rb = blockIdx.x / KS, w = blockIdx.x % KS;
    const int WPR = KS * 128, GPR = KS * 16;
    uint32_t* tile = tiles + (size_t)blockIdx.x * kTileWords;
    for (int i = threadIdx.x; i < kTileWords; i += 256) {
        uint32_t v = 0;
        if (i < 2048) {
            const int t = i >> 8, lane = (i >> 2) & 63, j = i & 3;
            const int row = rb * 16 + (lane & 15);
            if (row < n) v = wq[(size_t)row * WPR + w * 128 + t * 16 + (lane >> 4) * 4 + j];
        } else {
            const int lane = ((i - 2048) >> 2) & 63, j = i & 3;
            const int row = rb * 16 + (lane & 15);
            if (row < n) {
                const size_t g = (size_t)row * GPR + w * 16 + (lane >> 4) * 4 + j;
                v = (uint32_t)scales[g] | ((uint32_t)(biases ? biases[g] : (bf16_t)0) << 16);
            }
        }
        tile[i] = v;
    }
}

// KS waves per block (K == KS * 1024), wave w owns superchunk w; a task = NU row blocks (EPI_SWIGLU: gate block + up block) whose KS
// partial sums meet in LDS; NBUF task buffers per wave: 1 = every task's loads issued when the task starts (grids of one task per block:
// the launch's whole matrix is in flight at once), 3 = the loads of tasks i + 1 and i + 2 are in flight while task i is multiplied (blocks
// that stream many tasks: the vocabulary matrix)
template <int KS, int NU, int PRO, int EPI, int NBUF>
#ifndef OMX_QM_MINW
#define OMX_QM_MINW 3          // (A/B: make VARIANT=x VARIANT_FLAGS=-DOMX_QM_MINW=2)
#endif
#ifndef OMX_QM_NT
#define OMX_QM_NT 1            // weight tiles with non-temporal loads
#endif
__global__ __launch_bounds__(KS * 64, (NU * NBUF <= 2 ? OMX_QM_MINW : 2)) void qgemv4m_kernel(const QGemvArgs a) {
    typedef Act16<false> A16;
    constexpr int NT = KS * 64, K = KS * 1024, GPR = K / 64;
    constexpr bool SWIGLU = EPI == EPI_SWIGLU;
    static_assert(!SWIGLU || NU == 2, "gate / up pairs are two row blocks per task");
    static_assert(NBUF == 1 || NBUF == 3, "one task buffer or a rolling window of three");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                        // [K], octets stored (x0,x2)(x4,x6)(x1,x3)(x5,x7)
    float* xsum = reinterpret_cast<float*>(smem + (size_t)K * 2);        // [GPR] sum of x over each quantisation group
    float* red = xsum + GPR;                                             // [2][NU][KS][16] per-wave row sums, double-buffered by task parity
    float* scr = red + 2 * NU * KS * 16;                                 // [KS] block-reduce scratch; [32] u64 argmax keys
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int n_rb = (a.N + 15) / 16;
    const int ntask = SWIGLU ? n_rb : (n_rb + NU - 1) / NU;
    const int rb0 = a.m[0].n / 16, rb1 = rb0 + a.m[1].n / 16;   // row blocks of the stacked members (q | k | v)

    auto issue = [&](QmUnit& U, int task, int u) {
        int rb = min(SWIGLU ? task : task * NU + u, n_rb - 1);
        int mi = SWIGLU ? u : 0;
        if (!SWIGLU) {
            if (a.m[1].tiles && rb >= rb0) { mi = 1; if (a.m[2].tiles && rb >= rb1) mi = 2; }
            rb -= mi == 0 ? 0 : mi == 1 ? rb0 : rb1;
        }
        const uint32_t* p = a.m[mi].tiles + ((size_t)rb * KS + w) * kTileWords + lane * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) U.wd[t] = OMX_QM_NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 256 * t)) : *reinterpret_cast<const u32x4*>(p + 256 * t);
        U.sb = OMX_QM_NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 2048)) : *reinterpret_cast<const u32x4*>(p + 2048);
    };

    QmUnit U0[NU], U1[NBUF == 3 ? NU : 1], U2[NBUF == 3 ? NU : 1];
    int task = blockIdx.x;
    if (task < ntask) {
#pragma unroll
        for (int u = 0; u < NU; ++u) issue(U0[u], task, u);
    }
    if (NBUF == 3 && task + (int)gridDim.x < ntask) {
#pragma unroll
        for (int u = 0; u < NU; ++u) issue(U1[NBUF == 3 ? u : 0], task + gridDim.x, u);
    }

    // ---- prologue: x -> LDS (RMS-normalised on the way in), octets in the order the nibble unpack produces; group sums beside it.
    //      K == NT * 16: two 16-byte vectors per thread, held in registers between the two passes of the norm ----
    {
        const bf16_t* xg = a.x;
        u32x4 raw[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) raw[it] = *reinterpret_cast<const u32x4*>(xg + threadIdx.x * 8 + it * NT * 8);
        if (PRO == PRO_RMSNORM) {
            u32x4 nwv[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) nwv[it] = *reinterpret_cast<const u32x4*>(a.norm_w + threadIdx.x * 8 + it * NT * 8);
            float ss = 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ss = fmaf(A16::lo(raw[it][c]), A16::lo(raw[it][c]), ss);
                    ss = fmaf(A16::hi(raw[it][c]), A16::hi(raw[it][c]), ss);
                }
            ss = block_sum<KS>(ss, scr);
            const float rstd = 1.0f / sqrtf(ss / (float)K + a.eps);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    raw[it][c] = A16::pack(A16::lo(raw[it][c]) * rstd * A16::lo(nwv[it][c]), A16::hi(raw[it][c]) * rstd * A16::hi(nwv[it][c]));
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = threadIdx.x * 8 + it * NT * 8;
            const u32x4 o = raw[it];
            u32x4 t;
            t[0] = __builtin_amdgcn_perm(o[1], o[0], 0x05040100u); t[1] = __builtin_amdgcn_perm(o[3], o[2], 0x05040100u);
            t[2] = __builtin_amdgcn_perm(o[1], o[0], 0x07060302u); t[3] = __builtin_amdgcn_perm(o[3], o[2], 0x07060302u);
            *reinterpret_cast<u32x4*>(xs + i) = t;
            float sv = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) sv += A16::lo(o[c]) + A16::hi(o[c]);
            sv += dpp_f<kDppXor1>(sv);
            sv += dpp_f<kDppXor2>(sv);
            sv += dpp_f<kDppHalfMirror>(sv);
            if ((threadIdx.x & 7) == 0) xsum[i >> 6] = sv;
        }
    }
    __syncthreads();

    // the four A fragments of this wave's superchunk (slot m = lane & 15 meets k block b = lane >> 4) and its lanes' four group sums
    bf16x8 xa[4];
    {
        const bool active = (q >> 1) == (n & 1);
        const bf16_t* xp = xs + w * 1024 + (n >> 1) * 128 + q * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (active) v = *reinterpret_cast<const u32x4*>(xp + j * 8);
            xa[j] = __builtin_bit_cast(bf16x8, v);
        }
    }
    const f32x4 xs4 = *reinterpret_cast<const f32x4*>(xsum + w * 16 + q * 4);

    auto multiply = [&](const QmUnit& U) -> float {
        f32x4 D[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) D[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint32_t c43 = A16::kMagicBytes;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t wdw = U.wd[t][j];
                const uint32_t lo = wdw & 0x0F0F0F0Fu, hi = (wdw >> 4) & 0x0F0F0F0Fu;
                u32x4 f;
                f[0] = __builtin_amdgcn_perm(c43, lo, 0x04010400u); f[1] = __builtin_amdgcn_perm(c43, lo, 0x04030402u);
                f[2] = __builtin_amdgcn_perm(c43, hi, 0x04010400u); f[3] = __builtin_amdgcn_perm(c43, hi, 0x04030402u);
                D[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[j], __builtin_bit_cast(bf16x8, f), D[t], 0, 0, 0);
            }
        }
        // lane (n, q): groups 4 q + i of row n sit in D[2 q + (i >> 1)][i]
        const f32x4 e = q == 0 ? D[0] : q == 1 ? D[2] : q == 2 ? D[4] : D[6];
        const f32x4 o = q == 0 ? D[1] : q == 1 ? D[3] : q == 2 ? D[5] : D[7];
        const float v[4] = {e[0], e[1], o[2], o[3]};
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float scl = bf16lo(U.sb[i]);
            const float bia = fmaf(-A16::kMagic, scl, bf16hi(U.sb[i]));
            r = fmaf(scl, v[i], r);
            r = fmaf(bia, xs4[i], r);
        }
        // the four k-block quarters of the superchunk: lanes n, n + 16, n + 32, n + 48
        r += __shfl_xor(r, 16, 64);
        r += __shfl_xor(r, 32, 64);
        return r;
    };

    uint64_t best = 0;
    int par = 0;
    auto finish = [&](const QmUnit* U, int tk) {
        float* rp = red + par * NU * KS * 16;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const float r = multiply(U[u]);
            if (lane < 16) rp[(u * KS + w) * 16 + lane] = r;
        }
        __syncthreads();
        if (threadIdx.x < (SWIGLU ? 16 : 16 * NU)) {
            const int u = threadIdx.x >> 4, nn = threadIdx.x & 15;
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < KS; ++ww) v0 += rp[(u * KS + ww) * 16 + nn];
            if (SWIGLU) {
#pragma unroll
                for (int ww = 0; ww < KS; ++ww) v1 += rp[(KS + ww) * 16 + nn];
            }
            const int row = (SWIGLU ? tk : tk * NU + u) * 16 + nn;
            if (row < a.N) {
                if (EPI == EPI_STORE) {
                    a.out[row] = A16::bits(v0);
                } else if (EPI == EPI_F32) {
                    a.out_f32[row] = v0;
                } else if (EPI == EPI_RESIDUAL) {
                    a.out[row] = A16::bits(A16::val(a.resid[row]) + A16::rnd(v0));
                } else if (EPI == EPI_SWIGLU) {
                    const float g = A16::rnd(v0), uu = A16::rnd(v1);
                    if (a.swiglu_single_round) {
                        a.out[row] = A16::bits(g / (1.0f + expf(-g)) * uu);
                    } else {
                        const float sg = A16::rnd(1.0f / (1.0f + expf(-g)));
                        a.out[row] = A16::bits(A16::rnd(g * sg) * uu);
                    }
                } else if (EPI == EPI_ARGMAX) {
                    const bf16_t lb = A16::bits(v0);
                    a.out[row] = lb;
                    const uint64_t key = qm_argmax_key(A16::val(lb), (uint32_t)(row + a.row_offset));
                    best = key > best ? key : best;
                }
            }
        }
        par ^= 1;
    };

    const int stride = gridDim.x;
    if (NBUF == 1) {
        while (task < ntask) {
            finish(U0, task);
            task += stride;
            if (task < ntask) {
#pragma unroll
                for (int u = 0; u < NU; ++u) issue(U0[u], task, u);
            }
        }
    } else {
        // task i lives in buffer i % 3; before it is multiplied the loads of task i + 2 go out into the buffer task i - 1 left
        while (task < ntask) {
            if (task + 2 * stride < ntask) {
#pragma unroll
                for (int u = 0; u < NU; ++u) issue(U2[NBUF == 3 ? u : 0], task + 2 * stride, u);
            }
            finish(U0, task);
            task += stride;
            if (task >= ntask) break;
            if (task + 2 * stride < ntask) {
#pragma unroll
                for (int u = 0; u < NU; ++u) issue(U0[u], task + 2 * stride, u);
            }
            finish(U1, task);
            task += stride;
            if (task >= ntask) break;
            if (task + 2 * stride < ntask) {
#pragma unroll
                for (int u = 0; u < NU; ++u) issue(U1[NBUF == 3 ? u : 0], task + 2 * stride, u);
            }
            finish(U2, task);
            task += stride;
        }
    }
    if (EPI == EPI_ARGMAX) {
        uint64_t* keys = reinterpret_cast<uint64_t*>(scr);
        __syncthreads();
        if (threadIdx.x < 32) keys[threadIdx.x] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t b = keys[0];
            for (int i = 1; i < 32; ++i) b = keys[i] > b ? keys[i] : b;
            a.argmax_slot[blockIdx.x] = b;
            // the engine reduces qgemv_grid(N) partials: the slots beyond this launch's grid hold "no candidate"
            for (int i = blockIdx.x + gridDim.x; i < a.rolled_stage; i += gridDim.x) a.argmax_slot[i] = 0;
        }
    }
}

template <int KS, int NU, int PRO, int EPI, int NBUF>
int launch_one(const QGemvArgs& a, int grid, hipStream_t s) {
    const size_t shmem = (size_t)KS * 1024 * 2 + (size_t)KS * 16 * 4 + (size_t)2 * NU * KS * 16 * 4 + 32 * 8 + 64;
    OMX_LAUNCH_TIMED((qgemv4m_kernel<KS, NU, PRO, EPI, NBUF>), dim3(grid), dim3(KS * 64), shmem, s, a);
    OMX_LAUNCH_CHECK();
    return 0;
}

int stream_grid() {   // blocks of a streaming launch (the vocabulary matrix): two resident blocks per CU
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    if (const char* e = getenv("OMX_QGEMV_MFMA_BPC")) return cus * std::max(1, atoi(e));
    return cus * 2;
}

template <int KS>
int launch_ks(const QGemvArgs& a_in, int pro, int epi, hipStream_t s) {
    QGemvArgs a = a_in;
    const int n_rb = (a.N + 15) / 16;
    if (epi == EPI_SWIGLU) {
        if (pro == PRO_RMSNORM) return launch_one<KS, 2, PRO_RMSNORM, EPI_SWIGLU, 1>(a, n_rb, s);
        return launch_one<KS, 2, PRO_NONE, EPI_SWIGLU, 1>(a, n_rb, s);
    }
    if (epi == EPI_ARGMAX && pro == PRO_RMSNORM) {
        const int slots = qgemv_grid(a.N), grid = std::min(std::min(n_rb, slots), stream_grid());
        a.rolled_stage = slots;      // (a field the VALU kernel's A/B switch owns: here the number of argmax slots to leave defined)
        if (grid * 3 <= n_rb) return launch_one<KS, 1, PRO_RMSNORM, EPI_ARGMAX, 3>(a, grid, s);
        return launch_one<KS, 1, PRO_RMSNORM, EPI_ARGMAX, 1>(a, std::min(n_rb, slots), s);
    }
    if (epi == EPI_STORE && pro == PRO_RMSNORM) return launch_one<KS, 1, PRO_RMSNORM, EPI_STORE, 1>(a, n_rb, s);
    if (epi == EPI_STORE && pro == PRO_NONE) return launch_one<KS, 1, PRO_NONE, EPI_STORE, 1>(a, n_rb, s);
    if (epi == EPI_RESIDUAL && pro == PRO_NONE) return launch_one<KS, 1, PRO_NONE, EPI_RESIDUAL, 1>(a, n_rb, s);
    if (epi == EPI_F32 && pro == PRO_NONE) return launch_one<KS, 1, PRO_NONE, EPI_F32, 1>(a, n_rb, s);
    return -1;
}

}  // namespace

size_t qgemv4m_tile_words(int n, int K) { return (size_t)((n + 15) / 16) * (K / 1024) * kTileWords; }
// K / 1024 waves per block: the widths of the Qwen3 family and its tensor-parallel K slices (1024 .. 12288)
static bool ks_built(int ks) { return ks == 1 || ks == 2 || ks == 3 || ks == 4 || ks == 6 || ks == 8 || ks == 12; }
bool qgemv4m_shape_ok(int K, int group, int bits) { return bits == 4 && group == 64 && K > 0 && K % 1024 == 0 && ks_built(K / 1024); }
int launch_qgemv4m_repack(uint32_t* tiles, const uint32_t* wq, const bf16_t* scales, const bf16_t* biases, int n, int K, hipStream_t s) {
    OMX_REQUIRE(tiles && wq && scales && n > 0 && K % 1024 == 0, "qgemv tile repack: bad arguments");
    qgemv4m_repack_kernel<<<(unsigned)(((n + 15) / 16) * (K / 1024)), 256, 0, s>>>(tiles, wq, scales, biases, n, K / 1024);
    OMX_LAUNCH_CHECK();
    return 0;
}

int g_qgemv_mfma_mode = 1;    // 0: never (tests flip it through omx_debug_qgemv_mfma); otherwise a matrix with tiles takes this kernel --
                              // OMX_QGEMV_MFMA=0 makes the engine (and the bench hook) build no tiles

// 0: launched; -1: not a shape / form of this kernel (the caller takes quant.hip's VALU kernel); 1: error
int launch_qgemv4m(const QGemvArgs& a, int pro, int epi, hipStream_t s) {
    if (g_qgemv_mfma_mode == 0 || a.group != 64 || a.scales_f16 || a.n_batch > 1 || a.w_sel || a.w_sel_n > 0 || a.N < 16) return -1;
    if (!qgemv4m_shape_ok(a.K, a.group, 4)) return -1;
    for (int i = 0; i < 3; ++i)
        if (a.m[i].w && !a.m[i].tiles) return -1;          // the engine (or the test hook) built no tiles for a member
    if (epi == EPI_SWIGLU) {
        if (a.N % 16 != 0) return -1;
    } else {
        // the members of a row-stacked weight (q | k | v) must not share a 16-row block
        if (a.m[1].w && a.m[0].n % 16 != 0) return -1;
        if (a.m[2].w && a.m[1].n % 16 != 0) return -1;
    }
    switch (a.K / 1024) {
        case 1: return launch_ks<1>(a, pro, epi, s);
        case 2: return launch_ks<2>(a, pro, epi, s);
        case 3: return launch_ks<3>(a, pro, epi, s);
        case 4: return launch_ks<4>(a, pro, epi, s);
        case 6: return launch_ks<6>(a, pro, epi, s);
        case 8: return launch_ks<8>(a, pro, epi, s);
        default: return launch_ks<12>(a, pro, epi, s);
    }
}

}  // namespace omx

/* test / A-B hook: 1 (or -1) = the matrix-core kernel wherever a matrix has tiles (default), 0 = quant.hip's VALU kernel everywhere */
extern "C" void omx_debug_qgemv_mfma(int on) { omx::g_qgemv_mfma_mode = on; }
