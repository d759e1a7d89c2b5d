// Flash attention forward, round-5 form: ONE persistent workgroup of 4 waves per CU, one wave per SIMD, 64 query rows per wave
// (two 32-row blocks on v_mfma_f32_32x32x16_bf16), the whole 512-entry register file per lane, head_dim 128, no mask.
//   reference: mlx_fast_scaled_dot_product_attention (mlx-c fast.h:189-198; mlx-rs/src/fast.rs:121-151) as called with mask = none by
//   FLUX joint attention (flux-klein-mlx/src/klein_model.rs:474-483) and by mlx_rs_core::scaled_dot_product_attention (utils.rs:191-209).
//
// Why another kernel (EXPERIMENTS.md R3-4, R4-14, R5-1): the 8-wave two-phase kernel pairs an MFMA wave with a softmax wave on each SIMD
// and the two ADD UP (issue-bound); here one wave per SIMD software-pipelines the two itself:
//   phase A(i): S_{i+1} = K_{i+1} Q^T (32 MFMAs)   beside   p = 2^(t_i) -> P_i, row sums, the first V_i transpose reads
//   phase B(i): O += V_i^T P_i      (32 MFMAs)   beside   row max of S_{i+1}, the (deferred) rescale decision, t = s*c - m,
//                                                          the K_{i+2} fragment reads, the other V_i reads, the LDS-DMA of K_{i+5} / V_{i+3}
// with ONE s_barrier per key tile and the filler instructions placed per MFMA gap.  The per-unit body (256 query rows of one head against
// all its key tiles) is ONE generated inline-asm statement with hand-allocated registers -- tools/gen_flash4_asm.py has the register map,
// the schedule and the hazard rules; written in HIP the same structure spilled ~100 registers and reloaded DMA offsets behind vmcnt(0).
// This file keeps what hipcc does well: the unit walk, the addresses, the start of the LDS-DMA stream.
// K / V tiles (64 keys) stream by LDS-DMA through 4-slot rings (128 KB of LDS) as ONE continuous stream over all the units a workgroup
// walks: the next unit's first tiles are in flight while the current unit finishes and stores.
// Lane layouts (32x32x16): S^T = K Q^T and O^T = V^T P^T ("swapped" products): a lane's column is its query row (lane & 31), its 16
// accumulator registers are rows 8 (r >> 2) + 4 (lane >> 5) + (r & 3); the k-slot <-> key permutation 8 (lane >> 5) + e  <->
// (e & 3) + 8 (e >> 2) + 4 (lane >> 5) is applied on BOTH operands of the second product, so P's B fragment is eight consecutive
// accumulator registers and V^T comes through ds_read_b64_tr_b16 (attn_prefill.hip pp32_unit has the same mapping).
// Rescale: deferred (cdna_hip_programming.md T13): the running maximum of a row moves only when the tile's maximum exceeds it by more
// than THR = 8 (base-2 exponent units); P then reaches 2^8 instead of 1, rounded to bf16 with the same RELATIVE error.  THR = 0
// (OMX_ATTN_W4_THR=0) is the textbook online softmax on the same code.  The decision for tile i+1 is taken while O += V_i^T P_i is in
// flight and applied to O and l after that product completes and before P_{i+1} is summed: everything at the old scale is scaled once.
#include <stdlib.h>

#include "gemm.hpp"

#include "attn_flash4_clobbers.inc"   // F4_CLOBBERS: the registers the generated body owns

namespace omx {
namespace {

constexpr int F4_KB = 64;                  // keys per tile
constexpr int F4_D = 128;
constexpr int F4_RING = 4;                 // LDS slots per operand
constexpr int F4_TILE_B = F4_KB * F4_D * 2;   // bytes per tile (16 KiB)

struct Flash4Args {
    const bf16_t *q, *k, *v;
    bf16_t* out;
    int B, H, Hkv, Tq, Tk;
    int64_t kv_batch_stride, kv_head_stride, kv_ts;
    int64_t q_bs, q_hs, q_ts, o_bs, o_hs, o_ts;
    float scale;
    int nq;          // 256-row query blocks per head
    int units;       // B * H * nq
    int xcd_map;     // units of head x on XCD x % 8 (B * H and the grid multiples of 8)
};

// VAR 0: deferred rescale (threshold 8); 1: threshold 0; 2..5 (-DOMX_F4_DIAG): timing-only builds without DMA / softmax VALU / LDS reads / all three
template <int VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_flash4_kernel(const Flash4Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[F4_RING * F4_TILE_B];
    __shared__ __attribute__((aligned(16))) unsigned char sV[F4_RING * F4_TILE_B];
    constexpr int D = F4_D;

    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l32 = lane & 31, hi = lane >> 5;
    const int nt = a.Tk / F4_KB;                         // key tiles per unit (the launcher: Tk % 256 == 0, nt >= 8)

    // ---- the units this workgroup walks: ordinal n -> (batch*head, query block) ----
    const int G = (int)gridDim.x, wg = (int)blockIdx.x;
    const int xcd = wg & 7, lw = wg >> 3, nxw = G >> 3;
    const int units_x = a.xcd_map ? (a.B * a.H / 8) * a.nq : a.units;
    auto unit_of = [&](int n, int& bh, int& qt) {
        const int u = a.xcd_map ? lw + n * nxw : wg + n * G;
        const int hq = u / a.nq;
        bh = a.xcd_map ? hq * 8 + xcd : hq;
        qt = u - hq * a.nq;
    };
    int cnt = 0;
    {
        const int first = a.xcd_map ? lw : wg, step = a.xcd_map ? nxw : G;
        if (first < units_x) cnt = (units_x - 1 - first) / step + 1;
    }
    if (cnt == 0) return;
    auto kv_base = [&](int n, const unsigned char*& kp, const unsigned char*& vp) {
        int bh = 0, qt = 0;
        unit_of(n, bh, qt);
        const int b = bh / a.H, h = bh - b * a.H;
        const int kvh = h / (a.H / a.Hkv);
        const size_t e = (size_t)b * a.kv_batch_stride + (size_t)kvh * a.kv_head_stride;
        kp = reinterpret_cast<const unsigned char*>(a.k + e);
        vp = reinterpret_cast<const unsigned char*>(a.v + e);
    };

    // ---- LDS-DMA stream: tile g of this workgroup's stream = (unit g / nt, key tile g % nt), ring slot g % 4 ----
    // per-thread source offsets (bytes from the tile's first key row): chunk ci = tid + it * 256, row = ci / 16, 16-B chunk index swizzled on
    // the SOURCE side (the DMA image is lane-linear).  K: chunk ^= row & 15 (ds_read_b128 of 32 key rows x 2 chunks); V: 32-B block ^= row & 7
    uint32_t koff[4], voff[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int ci = (int)threadIdx.x + it * 256;
        const int row = ci >> 4, c = ci & 15;
        koff[it] = (uint32_t)((row * a.kv_ts + (c ^ (row & 15)) * 8) * 2);
        voff[it] = (uint32_t)((row * a.kv_ts + (c ^ ((row & 7) << 1)) * 8) * 2);
    }
    const unsigned sK_base = (unsigned)(uintptr_t)sK, sV_base = (unsigned)(uintptr_t)sV;
    const unsigned kdst = sK_base + wave * 1024u, vdst = sV_base + wave * 1024u;
    auto dma16 = [&](const unsigned char* sbase, uint32_t voff_b, unsigned lds_dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff_b), "s"(sbase), "s"(lds_dst) : "memory");
    };
    // ---- fragment read addresses (LDS byte addresses inside slot 0) ----
    // K (A operand of S^T): lane = key row kb * 32 + l32, head-dim step i: 16-B chunk 2 i + hi, swizzled by row & 15
    uint32_t kro[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kro[i] = sK_base + (uint32_t)((l32 * 16 + ((2 * i + hi) ^ (l32 & 15))) * 16);
    // V^T (A operand of O^T) through the transposing read: the 16-lane group g4 supplies key rows 4 (g4 >> 1) + (l16 >> 2) (+ 8 for the second
    // read) of the 16-key step at dims 32 db + 16 (g4 & 1) + 4 (l16 & 3) ..+3 and receives keys 4 (g4 >> 1) + 0..3 at dim 32 db + 16 (g4 & 1) + l16
    uint32_t vro[4];
    {
        const int l16 = lane & 15, g4 = lane >> 4;
        const int v_key = 4 * (g4 >> 1) + (l16 >> 2);
#pragma unroll
        for (int db = 0; db < 4; ++db)
            vro[db] = sV_base + (uint32_t)((v_key * D + (l16 & 3) * 4 + (((2 * db + (g4 & 1)) ^ (v_key & 7)) * 16)) * 2);
    }
    const float c2 = a.scale * 1.44269504088896340736f;
    const uint32_t stride_b = (uint32_t)(F4_KB * a.kv_ts * 2);

    // ---- stream start: K_0..K_3, V_0..V_2 of the first unit (nt >= 8: all inside it) ----
    const unsigned char *kcur, *vcur;
    kv_base(0, kcur, vcur);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int it = 0; it < 4; ++it) dma16(kcur + (size_t)t * stride_b, koff[it], kdst + t * F4_TILE_B + it * 4096u);
        if (t < 3) {
#pragma unroll
            for (int it = 0; it < 4; ++it) dma16(vcur + (size_t)t * stride_b, voff[it], vdst + t * F4_TILE_B + it * 4096u);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    for (int n = 0; n < cnt; ++n) {
        int bh = 0, qt = 0;
        unit_of(n, bh, qt);
        const int b = bh / a.H, h = bh - b * a.H;
        const unsigned char *knext = kcur, *vnext = vcur;      // (past the end of the stream the DMA re-reads this unit's first tiles)
        if (n + 1 < cnt) kv_base(n + 1, knext, vnext);
        // query rows of this wave: qb * 32 + l32; byte offsets of the lane's fragments (dims 16 i + 8 hi ..+7 of its row) and output pieces
        uint32_t qoff[2], ooff[2];
        unsigned long long rowmask[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qrow = qt * 256 + (int)wave * 64 + qb * 32 + l32;
            const int rc = min(qrow, a.Tq - 1);
            qoff[qb] = (uint32_t)(((int64_t)rc * a.q_ts + hi * 8) * 2);
            ooff[qb] = (uint32_t)(((int64_t)rc * a.o_ts + hi * 4) * 2);
            rowmask[qb] = __builtin_amdgcn_ballot_w64(qrow < a.Tq);
        }
        const bf16_t* qbase = a.q + (size_t)b * a.q_bs + (size_t)h * a.q_hs;
        bf16_t* obase = a.out + (size_t)b * a.o_bs + (size_t)h * a.o_hs;
#define F4_OPERANDS                                                                                                                 \
        :                                                                                                                           \
        : "v"(koff[0]), "v"(koff[1]), "v"(koff[2]), "v"(koff[3]), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]),           \
          "v"(kro[0]), "v"(kro[1]), "v"(kro[2]), "v"(kro[3]), "v"(kro[4]), "v"(kro[5]), "v"(kro[6]), "v"(kro[7]),                   \
          "v"(vro[0]), "v"(vro[1]), "v"(vro[2]), "v"(vro[3]), "v"(qoff[0]), "v"(qoff[1]), "v"(ooff[0]), "v"(ooff[1]),               \
          "s"(qbase), "s"(obase), "s"(kcur), "s"(vcur), "s"(knext), "s"(vnext),                                                     \
          "s"(stride_b), "s"(nt), "s"(c2), "s"(kdst), "s"(vdst), "s"(rowmask[0]), "s"(rowmask[1])                                   \
        : F4_CLOBBERS
        if (VAR == 0) {
            asm volatile(
#include "attn_flash4_body.inc"
                F4_OPERANDS);
        } else if (VAR == 1) {
            asm volatile(
#include "attn_flash4_body_thr0.inc"
                F4_OPERANDS);
        }
#ifdef OMX_F4_DIAG
        else if (VAR == 2) {
            asm volatile(
#include "attn_flash4_body_d2.inc"
                F4_OPERANDS);
        } else if (VAR == 3) {
            asm volatile(
#include "attn_flash4_body_d3.inc"
                F4_OPERANDS);
        } else if (VAR == 4) {
            asm volatile(
#include "attn_flash4_body_d4.inc"
                F4_OPERANDS);
        } else if (VAR == 5) {
            asm volatile(
#include "attn_flash4_body_d5.inc"
                F4_OPERANDS);
        } else if (VAR == 6) {
            asm volatile(
#include "attn_flash4_body_d6.inc"
                F4_OPERANDS);
        } else if (VAR == 7) {
            asm volatile(
#include "attn_flash4_body_d7.inc"
                F4_OPERANDS);
        } else if (VAR == 8) {
            asm volatile(
#include "attn_flash4_body_d8.inc"
                F4_OPERANDS);
        }
#endif
#undef F4_OPERANDS
        kcur = knext;
        vcur = vnext;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the tail of the DMA stream must not outlive the workgroup's LDS)
}

int flash4_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = -1;
    }
    return n;
}

}  // namespace

// Shapes the 4-wave kernel takes (the launcher's caller falls back to the 8-wave kernels otherwise)
bool attn_flash4_supported(int B, int H, int Hkv, int Tq, int Tk, int D, int mask_mode, bool f16) {
    (void)B;
    // (the ring slot of a key tile is static in the generated body: units of whole groups of four tiles, at least two groups)
    return !f16 && D == 128 && mask_mode == OMX_MASK_NONE && Tk % (4 * F4_KB) == 0 && Tk >= 8 * F4_KB && Tq >= 1 && H % Hkv == 0;
}

int launch_attn_flash4(bf16_t* out, const bf16_t* q, const bf16_t* k, const bf16_t* v, int B, int H, int Hkv, int Tq, int Tk,
                       int64_t kv_batch_stride, int64_t kv_head_stride, float scale, hipStream_t s, bool out_token_major,
                       const AttnLayout* layout) {
    constexpr int D = F4_D;
    Flash4Args a = {};
    a.q = q; a.k = k; a.v = v; a.out = out;
    a.B = B; a.H = H; a.Hkv = Hkv; a.Tq = Tq; a.Tk = Tk;
    a.kv_batch_stride = kv_batch_stride; a.kv_head_stride = kv_head_stride; a.kv_ts = D;
    a.q_bs = (int64_t)H * Tq * D; a.q_hs = (int64_t)Tq * D; a.q_ts = D;
    a.o_bs = a.q_bs; a.o_hs = a.q_hs; a.o_ts = D;
    if (out_token_major) { a.o_hs = D; a.o_ts = (int64_t)H * D; }
    if (layout) {
        a.q_bs = layout->q_bs; a.q_hs = layout->q_hs; a.q_ts = layout->q_ts;
        a.o_bs = layout->o_bs; a.o_hs = layout->o_hs; a.o_ts = layout->o_ts;
        a.kv_ts = layout->kv_ts;
    }
    a.scale = scale;
    a.nq = (Tq + 255) / 256;
    a.units = B * H * a.nq;
    const int cus = flash4_cus();
    OMX_REQUIRE(cus > 0, "flash attention: no device");
    int G = a.units < cus ? a.units : cus;
    a.xcd_map = ((B * H) % 8 == 0 && cus % 8 == 0 && a.units >= cus) ? 1 : 0;
    if (a.xcd_map) G = cus;
    // the 32-bit DMA offsets: 64 key rows of the tile
    OMX_REQUIRE((int64_t)F4_KB * a.kv_ts * 2 < (int64_t)1 << 31, "flash attention: key row stride %lld too large", (long long)a.kv_ts);
    // 32-bit byte offsets of a lane's query row / output row from its head's base
    OMX_REQUIRE(((int64_t)Tq * a.q_ts + 8) * 2 < (int64_t)1 << 32 && ((int64_t)Tq * a.o_ts + 8) * 2 < (int64_t)1 << 32,
                "flash attention: query / output row stride too large for %d rows", Tq);
    const char* te = getenv("OMX_ATTN_W4_THR");
    int var = (te && atoi(te) == 0) ? 1 : 0;
#ifdef OMX_F4_DIAG
    if (const char* ve = getenv("OMX_ATTN_W4_VAR")) var = atoi(ve);
    if (var == 2) attn_flash4_kernel<2><<<G, 256, 0, s>>>(a);
    else if (var == 3) attn_flash4_kernel<3><<<G, 256, 0, s>>>(a);
    else if (var == 4) attn_flash4_kernel<4><<<G, 256, 0, s>>>(a);
    else if (var == 5) attn_flash4_kernel<5><<<G, 256, 0, s>>>(a);
    else if (var == 6) attn_flash4_kernel<6><<<G, 256, 0, s>>>(a);
    else if (var == 7) attn_flash4_kernel<7><<<G, 256, 0, s>>>(a);
    else if (var == 8) attn_flash4_kernel<8><<<G, 256, 0, s>>>(a);
    else
#endif
    if (var == 1) attn_flash4_kernel<1><<<G, 256, 0, s>>>(a);
    else attn_flash4_kernel<0><<<G, 256, 0, s>>>(a);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // namespace omx
