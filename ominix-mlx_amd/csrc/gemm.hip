// bf16 GEMM on the gfx950 matrix cores for the compute-bound side of the path:
//   out[M,N] = x[M,K] . W[N,K]^T (+ bias[N]) (+ residual[M,N])      -- nn::Linear at M > 4
//   (mlx-rs/src/nn/linear.rs:87-92 -> mlx_matmul / mlx_addmm, mlx-c ops.h:598-602, 36-43):
//   prefill projections (M = 2048), FLUX.2-klein DiT GEMMs (M = 4608), Paraformer encoder (M = 501);
//   and the GROUPED form used by the sparse-MoE block (gather_mm / gather_qmm semantics,
//   mlx-rs/src/ops/quantization.rs:169-279; mixtral-mlx/src/model.rs:194-275): rows sorted by expert,
//   one weight matrix per expert segment, activation rows gathered through an index array.
// Both operands are K-contiguous ("NT"), which is exactly the MFMA A/B fragment shape: a lane
// reads 8 consecutive k of one row (16 B) for either operand.
//
// Fast path (K % 64 == 0, 16-B aligned):
//   * 128x128x64 tile, 256 threads = 4 waves as 2(M) x 2(N), each wave 64x64 = 2x2 MFMA 32x32x16 tiles
//     (v_mfma_f32_32x32x16_bf16, fp32 accumulate: 64 accumulator registers per lane);
//   * HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS buffers: tile t+1 is in
//     flight while tile t feeds the MFMAs (cdna_hip_programming.md T3+T4 "minimum 2-phase" form);
//   * LDS image is lane-linear per wave-instruction, so the bank-conflict swizzle (16-B chunk index
//     ^= row & 7) is applied to the per-lane SOURCE address and again on the ds_read (rule 21);
//   * XCD-aware block order (T1): consecutive blocks of one XCD share an x row panel in its L2.
// Fallback (any K, any alignment): register-staged 64x64x32 tiles with zero fill.
#include <type_traits>
#include "gemm.hpp"
#include "act16.hpp"
#include "gridsync.hpp"   // coherent accessors for the split-K hand-over
#include <map>
#include <mutex>
#include <thread>
#include <utility>

#include <stdlib.h>

namespace omx {
namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per operand tile

struct GemmArgs {
    const bf16_t* x;      // [M, K]
    const bf16_t* w;      // [N, K]   (grouped: [E, N, K])
    const bf16_t* bias;   // [N] or null
    const bf16_t* resid;  // [M, N] or null : out = bf16(resid + bf16(acc (+bias)))
    const bf16_t* gate;   // [N] or null  : out = bf16(resid + acc * gate[col])  (DiT gated residual)
    bf16_t* out;          // [M, N]
    int M, N, K;
    int grid_m, grid_n;
    GroupedDesc g;        // grouped mode when g.tile_expert != nullptr
    int relu;             // out = max(0, acc + bias)  (Paraformer FFN, paraformer.rs:565-569)
    GemmSegs sg;          // segmented mode (256^2 kernel, SW instantiation), see gemm.hpp
    float* split_ws;      // ring kernel, gridDim.y > 1: f32 partial tiles [tile][split][64 x 64]; 256^2 kernel with ksplit > 1: [tile][split][256 x 256]
    unsigned* split_cnt;  // [tiles] arrival counters (zero between launches)
    int ksplit;           // 256^2 kernel (plain form): K halves of a tile go to different blocks (a grid of <= 128 tiles covers the chip)
    int stagger;          // four-wave kernel: sleeps of ~512 clocks per XCD slot before the first round's tiles start (OMX_GEMM_STAGGER; 0 = none)
    int no_park;          // four-wave kernel: store from the accumulator layout even where whole rows could be stored (OMX_GEMM_W4_PARK=0: the A/B)
};

// one 16-B chunk per lane per wave-instruction, 4 instructions per operand tile: row pointers of the 4
// rows this thread stages are fixed for the whole K loop
struct StageRows {
    const bf16_t* p[4];   // row base + logical k-chunk offset (the source-side swizzle)
};

__device__ __forceinline__ void stage_tile(const StageRows& r, int k0, unsigned char* lds_tile) {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int p = 0; p < 4; ++p)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(r.p[p] + k0), (lds_ptr_t)(lds_tile + (p * 256 + wave * 64) * 16), 16, 0, 0);
}

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* lds_tile, int row, int kc) {
    return *reinterpret_cast<const bf16x8*>(lds_tile + ((row << 3) + (kc ^ (row & 7))) * 16);
}

template <bool GROUPED>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_nt_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 buffers][A tile | B tile]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int m0, n0, rows_valid, out_row0;
    const bf16_t* w = a.w;
    if (GROUPED) {
        // tile table built on the device by moe_plan_kernel: tile -> (expert, first row inside its segment)
        const int tile = blockIdx.x / a.grid_n;
        if (tile >= *a.g.n_tiles) return;
        const int e = a.g.tile_expert[tile];
        const int seg = a.g.seg_start[e];
        m0 = a.g.tile_m0[tile];
        rows_valid = a.g.seg_start[e + 1] - seg;   // rows of this expert
        out_row0 = seg;                            // rows are produced in expert-sorted order
        n0 = (blockIdx.x % a.grid_n) * BN;
        w = a.w + (size_t)e * a.g.w_estride;
    } else {
        // XCD-aware remap: block b runs on XCD b % 8; give each XCD a contiguous run of tiles
        const int nblk = a.grid_m * a.grid_n;
        int bid = blockIdx.x;
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        m0 = (bid / a.grid_n) * BM;
        n0 = (bid % a.grid_n) * BN;
        rows_valid = a.M;
        out_row0 = 0;
    }
    const int wm = wave >> 1, wn = wave & 1;
    const int nt = a.K / BK;

    StageRows ra, rb;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int chunk = p * 256 + wave * 64 + lane;   // LDS chunk index (lane-linear inside the wave)
        const int row = chunk >> 3;
        const int kc = (chunk & 7) ^ (row & 7);          // logical k-chunk stored at this LDS slot
        int arow = min(m0 + row, rows_valid - 1);
        if (GROUPED) {
            arow += out_row0;                            // position in expert-sorted order
            if (a.g.row_src) arow = (int)a.g.row_src[arow];   // gather: sorted position -> source activation row
        }
        ra.p[p] = a.x + (size_t)arow * a.K + kc * 8;
        rb.p[p] = w + (size_t)min(n0 + row, a.N - 1) * a.K + kc * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage_tile(ra, 0, smem);
    stage_tile(rb, 0, smem + TILE_BYTES);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        unsigned char* bufA = smem + cur * 2 * TILE_BYTES;
        unsigned char* bufB = bufA + TILE_BYTES;
        if (t + 1 < nt) {
            unsigned char* nA = smem + (cur ^ 1) * 2 * TILE_BYTES;
            stage_tile(ra, (t + 1) * BK, nA);
            stage_tile(rb, (t + 1) * BK, nA + TILE_BYTES);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int kc = ks * 2 + (lane >> 5);
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = lds_frag(bufA, wm * 64 + i * 32 + (lane & 31), kc);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = lds_frag(bufB, wn * 64 + j * 32 + (lane & 31), kc);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);   // next tile landed (vmcnt) and our ds_reads retired (lgkmcnt)
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= a.N) continue;
            const float bv = a.bias ? bf16_to_f32(a.bias[col]) : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < rows_valid) {
                    const size_t o = (size_t)(out_row0 + row) * a.N + col;
                    float v = acc[i][j][r] + bv;
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (a.gate) v = bf16_to_f32(a.resid[o]) + v * bf16_to_f32(a.gate[col]);
                    else if (a.resid) v = bf16_to_f32(a.resid[o]) + round_bf16(v);
                    a.out[o] = f32_to_bf16(v);
                }
            }
        }
}


// ---- 256 x 256 x 64 tile, 8 waves, one block per CU: the deep-pipelined form for large GEMMs ----
// (cdna_hip_programming.md "256^2 8-phase": the 128^2 / one-barrier-per-K-step structure above tops out near
// 0.9 PF because every wave pays ~8 LDS-DMA issues per 16 MFMAs; here a wave owns 128 x 64 of the output, so
// the same 8 issues buy 32 MFMAs' worth of work, and the K step is cut into 4 phases -- one 64 x 32 quadrant of
// the wave's tile each -- with the staging spread over them, two LDS-DMA instructions per wave and phase.)
//
//   wave (wr, wc) = (wave >> 2, wave & 3) owns rows wr*128 + [0,128), cols wc*64 + [0,64) of the block tile.
//   LDS (128 KiB): 2 buffers x 4 pieces of [128 rows][64 k] (16 KiB, source-side XOR swizzle as above), cut by
//   WHEN they are read, not by where they sit in the tile:
//     X0 / X1 = A rows  {wr*128 + s*64 + [0,64)}  for both wr      (s = 0: read in phase 1, s = 1: phase 3)
//     Y0 / Y1 = B cols  {wc*64  + s*32 + [0,32)}  for all four wc  (s = 0: read in phase 1, s = 1: phase 2)
//   phase 1: read X0, Y0 -> MFMA quadrant (0,0)      stage Y1 of tile t+1
//   phase 2: read Y1     -> MFMA quadrant (0,1)      stage X1 of tile t+1
//   phase 3: read X1     -> MFMA quadrant (1,1)      stage X0 of tile t+2
//   phase 4:                MFMA quadrant (1,0)      stage Y0 of tile t+2
//   so every piece is issued 5-6 phases before its first read and FOUR newer pieces stay in flight across each
//   counted wait (vmcnt(8) in phases 4, 1, 2 -- never 0 in the steady state).
//   Each phase is {reads + staging [+ wait]; barrier; MFMAs at raised priority; barrier}; the wr = 1 waves run
//   one barrier behind the wr = 0 waves, so on every SIMD one wave issues loads while the other feeds the
//   matrix core.  Hazards: a piece is re-staged two phases after the phase that reads it (at least two barriers
//   after the reads of BOTH wave groups retired), and it is read one phase after the wait that retires it.
namespace big {
constexpr int TN = 256, TK = 64, NT = 512;
constexpr int HALF = 128 * TK * 2;           // bytes per half-tile
constexpr int BUF = 4 * HALF;                // A0 | A1 | B0 | B1
constexpr int SMEM = 2 * BUF;
constexpr int smem_bytes(int tile_rows) { return 2 * (2 * (tile_rows / 2) * TK * 2 + 2 * HALF); }   // X pieces of tile_rows / 2 rows
}  // namespace big

#define OMX_BAR() asm volatile("s_barrier" ::: "memory")

// MF = 16: v_mfma_f32_16x16x32_bf16, eight independent accumulators per k step inside a phase (a dependent MFMA is
// eight issues away); MF = 32: v_mfma_f32_32x32x16_bf16, two accumulators per phase (bit-identical to the 128^2 kernel)
// IMPL (with SW): implicit 3x3 convolution.  The A operand is a zero-bordered NHWC activation [(H+2), (W+2), C]; output
// pixel (y, x) starts at padded pixel (y, x) and K tile t (64 wide, K = 9 * C ordered (tap, channel), C = 64 * 2^sh) reads
// tap = t >> sh at channel (t & (2^sh - 1)) * 64 -- a wave-uniform offset per K tile instead of a materialised im2col matrix.
// TMR = 128: a 128 x 256 tile with the same eight waves (2 x 4, each 64 x 64), phases and stagger -- for grids whose 256^2 tiles cover
// at most half of the chip (the O / down projections of a 2 048-token prompt: 8 x 16 tiles); every output element is the same
// MFMA chain as in the 256-row form, so the two are bit-identical.  X pieces are 64 rows (ONE 16-byte chunk per thread and stage).
// F16: float16 operands and results (a float16 checkpoint's prompt pass): v_mfma_f32_16x16x32_f16 on the same fragments, every rounding
// point of the epilogues in float16 (act16.hpp)
template <int MF, bool SW = false, bool IMPL = false, int TMR = 256, bool F16 = false>
__global__ __launch_bounds__(big::NT) void gemm_bf16_nt_256_kernel(const GemmArgs a) {
    static_assert(!IMPL || SW, "the implicit-convolution staging lives in the segmented variant");
    static_assert(!SW || MF == 16, "the SwiGLU epilogue is written for the 16x16 accumulator layout");
    static_assert(TMR == 256 || (TMR == 128 && !SW && MF == 16), "the 128-row tile exists in the plain form");
    static_assert(!F16 || (MF == 16 && !IMPL), "float16: the 16x16x32 forms without the implicit convolution");
    typedef Act16<F16> A16;
    using namespace big;
    constexpr int WM = TMR / 2, XH = TMR / 4;      // rows per wave group, rows per wave group and half-piece
    constexpr int NX = TMR / 128;                  // 16-byte chunks per thread of an X piece
    constexpr int HALFX = (TMR / 2) * TK * 2;      // bytes per X piece
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    // XCD-aware remap (bijective form)
    const int nblk = a.grid_m * a.grid_n;
    int bid = blockIdx.x;
    // split K (plain form only): blocks [split * nblk, (split + 1) * nblk) own the split-th part of every tile's K range
    int ksplit = 1, split = 0;
    if constexpr (!SW) {
        if (a.ksplit > 1) {
            ksplit = a.ksplit;
            split = bid / nblk;
            bid -= split * nblk;
        }
    }
    const int tile_id = bid;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // grouped order inside the XCD's run: the 32 tiles its CUs hold at one time form an 8 x 4 patch (12 operand
    // panels through the XCD's L2 per K step) instead of a 1 x 32 strip (33 panels)
    int tm, tn;
    {
        constexpr int GM = 8;
        const int per_group = GM * a.grid_n;
        const int group = bid / per_group, in_group = bid % per_group;
        const int first_m = group * GM;
        const int gm = min(a.grid_m - first_m, GM);
        tm = first_m + in_group % gm;
        tn = in_group / gm;
    }
    // segmented mode: the column tile belongs to one projection (own weight, bias, output, width); n0 is local to it
    const bf16_t* seg_w = a.w;
    const bf16_t* seg_bias = nullptr;
    bf16_t* seg_out = a.out;
    int seg_cols = a.N, seg_ld = a.N;
    bool seg_act = false;
    if constexpr (SW) {
        const GemmSegs& g = a.sg;
        if (tn >= g.act_tile0) {
            seg_act = true;
            tn -= g.act_tile0;
            seg_cols = g.half;
        } else {
            const int sidx = (g.n_plain > 1 && tn >= g.plain[1].tile0) + (g.n_plain > 2 && tn >= g.plain[2].tile0);
            const GemmSeg& sgm = g.plain[sidx];
            seg_w = sgm.w; seg_bias = sgm.bias; seg_out = sgm.out; seg_cols = sgm.cols; seg_ld = sgm.ld;
            tn -= sgm.tile0;
        }
    }
    // grouped rows (MoE prefill): row tile tm is an entry of the device-built tile table -> (expert, first row inside its
    // segment of the expert-sorted order); rows are gathered through row_src, weights are the expert's slice of the stacks
    int m0 = tm * TMR, rows_valid = a.M, row_base = 0;
    size_t w_off = 0;
    const uint32_t* row_src = nullptr;
    if constexpr (SW) {
        if (a.g.tile_expert) {
            if (tm >= *a.g.n_tiles) return;
            const int e = a.g.tile_expert[tm];
            row_base = a.g.seg_start[e];
            rows_valid = a.g.seg_start[e + 1] - row_base;
            m0 = a.g.tile_m0[tm];
            row_src = a.g.row_src;
            w_off = (size_t)e * a.g.w_estride;
        }
    }
    const int n0 = tn * TN;
    const int nt_all = a.K / TK, t_first = split * (nt_all / ksplit);
    const int nt = split == ksplit - 1 ? nt_all - t_first : nt_all / ksplit;

    // staging sources: thread handles chunks c = i * 512 + tid (i = 0, 1) of every piece; piece row r of X_s is
    // tile row (r >> 6) * 128 + s * 64 + (r & 63), piece row r of Y_s is tile col (r >> 5) * 64 + s * 32 + (r & 31)
    const bf16_t* srcX[2][2];
    const bf16_t* srcY[2][2];
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * NT + threadIdx.x;
            const int row = c >> 3;
            const int kc = (c & 7) ^ (row & 7);
            const int trow = (row / XH) * WM + sidx * XH + (row % XH);   // (X pieces: rows [0, TMR / 2), i < NX only)
            const int tcol = (row >> 5) * 64 + sidx * 32 + (row & 31);
            if constexpr (IMPL) {
                const int p = min(m0 + trow, a.M - 1), py = p / a.sg.im_W, px = p - py * a.sg.im_W;
                srcX[sidx][i] = a.x + ((size_t)py * (a.sg.im_W + 2) + px) * a.sg.im_C + kc * 8;
            } else if constexpr (SW) {
                int xr = min(m0 + trow, rows_valid - 1) + row_base;
                if (row_src) xr = (int)row_src[xr];
                srcX[sidx][i] = a.x + (size_t)xr * a.K + kc * 8;
            }
            if constexpr (SW) {
                if (seg_act) {
                    // Y0 (tile cols wc * 64 + [0, 32)) <- gate rows, Y1 (wc * 64 + [32, 64)) <- up rows of the same 32 outputs
                    const int oc = n0 / 2 + (row >> 5) * 32 + (row & 31);
                    srcY[sidx][i] = (sidx ? a.sg.w_up : a.sg.w_gate) + w_off + (size_t)min(oc, seg_cols - 1) * a.K + kc * 8;
                } else {
                    srcY[sidx][i] = seg_w + w_off + (size_t)min(n0 + tcol, seg_cols - 1) * a.K + kc * 8;
                }
            } else {
                srcX[sidx][i] = a.x + (size_t)min(m0 + trow, a.M - 1) * a.K + kc * 8 + t_first * TK;
                srcY[sidx][i] = a.w + (size_t)min(n0 + tcol, a.N - 1) * a.K + kc * 8 + t_first * TK;
            }
        }
    auto stage = [&](const bf16_t* const (&src)[2], int k0, unsigned char* piece) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[i] + k0), (lds_ptr_t)(piece + (i * NT + wave * 64) * 16), 16, 0, 0);
    };
    auto stage_x = [&](const bf16_t* const (&src)[2], int k0, unsigned char* piece) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[i] + k0), (lds_ptr_t)(piece + (i * NT + wave * 64) * 16), 16, 0, 0);
    };
    // counted waits: every wait leaves the four most recent stages in flight (two X pieces of NX loads, two Y pieces of 2)
#define OMX_WAIT_RING() do { if constexpr (NX == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } while (0)
    // element offset of K tile t in the A operand
    auto kx = [&](int t) -> int {
        if constexpr (IMPL) {
            const int tap = t >> a.sg.im_sh, c0 = (t & ((1 << a.sg.im_sh) - 1)) * TK;
            const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;   // tap / 3, tap % 3 for tap < 9
            return (dy * (a.sg.im_W + 2) + dx) * a.sg.im_C + c0;
        } else {
            return t * TK;
        }
    };
    // buffer layout: X0 | X1 | Y0 | Y1
    constexpr int PX0 = 0, PX1 = HALFX, PY0 = 2 * HALFX, PY1 = 2 * HALFX + HALF;
    constexpr int BUFB = 2 * HALFX + 2 * HALF;    // bytes per ring buffer (= BUF at 256 rows)

    // accumulators: MF = 32 -> [4][2] tiles of 32x32 (16 regs each); MF = 16 -> [8][4] tiles of 16x16 (4 regs each)
    constexpr int RT = WM / MF, CT = 64 / MF, AR = MF * MF / 64;   // row tiles, col tiles, registers per tile
    constexpr int KS = TK / (MF == 32 ? 16 : 32);                     // MFMA k steps per K tile
    constexpr int KCH = MF == 32 ? 2 : 4;                            // 16-B chunks per k step
    constexpr int LR = MF == 32 ? 31 : 15;                           // lane -> row mask
    constexpr int LS = MF == 32 ? 5 : 4;                             // lane -> k chunk shift
    typedef float accv __attribute__((ext_vector_type(AR)));
    accv acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < AR; ++r) acc[i][j][r] = 0.f;

    // prologue: what the steady state would have issued before tile 0's first phase, in its order
    stage_x(srcX[0], kx(0), smem + PX0);
    stage(srcY[0], 0, smem + PY0);
    stage(srcY[1], 0, smem + PY1);
    stage_x(srcX[1], kx(0), smem + PX1);
    if (nt > 1) {
        stage_x(srcX[0], kx(1), smem + BUFB + PX0);
        stage(srcY[0], TK, smem + BUFB + PY0);
        OMX_WAIT_RING();   // X0, Y0 of tile 0 have landed
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    OMX_BAR();
    if (wr == 1) OMX_BAR();   // the two wave groups run one barrier apart from here on

    constexpr int RQ = RT / 2, CQ = CT / 2;   // tiles per quadrant
    bf16x8 fa[RQ][KS], fb[2][CQ][KS];
    const int arow = lane & LR, kh = lane >> LS;
    auto mfma = [&](const bf16x8& x, const bf16x8& y, accv& c) {
        // operands swapped (W fragment first): the accumulator holds the tile transposed, see the epilogue
        if constexpr (MF == 32) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, c, 0, 0, 0);
        else if constexpr (F16) {
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, y), __builtin_bit_cast(h8, x), c, 0, 0, 0);
        } else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, c, 0, 0, 0);
    };
    auto quadrant = [&](int qm, int qn) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < RQ; ++i)
#pragma unroll
                for (int j = 0; j < CQ; ++j) mfma(fa[i][ks], fb[qn][j][ks], acc[qm * RQ + i][qn * CQ + j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < nt; ++t) {
        unsigned char* cur = smem + (t & 1) * BUFB;
        unsigned char* nxt = smem + ((t + 1) & 1) * BUFB;
        const bool has1 = t + 1 < nt, has2 = t + 2 < nt;
        const int k1 = (t + 1) * TK, k2 = (t + 2) * TK;
        auto read_a = [&](int sub) {
            const unsigned char* px = cur + (sub ? PX1 : PX0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < RQ; ++i) fa[i][ks] = lds_frag(px, wr * XH + i * MF + arow, ks * KCH + kh);
        };
        auto read_b = [&](int sub) {
            const unsigned char* py = cur + (sub ? PY1 : PY0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < CQ; ++j) fb[sub][j][ks] = lds_frag(py, wc * 32 + j * MF + arow, ks * KCH + kh);
        };

        // ---- phase 1 ----
        read_b(0);
        read_a(0);
        if (has1) {
            stage(srcY[1], k1, nxt + PY1);
            OMX_WAIT_RING();   // Y1 of this tile (read in phase 2) has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        OMX_BAR();
        quadrant(0, 0);
        OMX_BAR();
        // ---- phase 2 ----
        read_b(1);
        if (has1) {
            stage_x(srcX[1], kx(t + 1), nxt + PX1);
            OMX_WAIT_RING();   // X1 of this tile (read in phase 3) has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        OMX_BAR();
        quadrant(0, 1);
        OMX_BAR();
        // ---- phase 3 ----
        read_a(1);
        if (has2) stage_x(srcX[0], kx(t + 2), cur + PX0);
        OMX_BAR();
        quadrant(1, 1);
        OMX_BAR();
        // ---- phase 4 ----
        if (has2) {
            stage(srcY[0], k2, cur + PY0);
            OMX_WAIT_RING();   // X0, Y0 of tile t+1 have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        OMX_BAR();
        quadrant(1, 0);
        OMX_BAR();
    }
    if (wr == 0) OMX_BAR();   // pairs with the extra barrier of the wr = 1 waves

    if constexpr (!SW) {
        if (ksplit > 1) {
            // every split parks its f32 partial tile; the LAST block of the tile to arrive re-reads all of them in split order (its own
            // included: one fixed summation order whoever is last) and runs the epilogue -- the ring kernel's hand-over
            __shared__ unsigned s_last;
            constexpr int PER_WAVE = RT * CT * 64 * AR;
            float* mine = a.split_ws + (((size_t)tile_id * ksplit + split) * 8 + wave) * PER_WAVE;
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int g = 0; g < AR / 4; ++g) {
                        float* p = mine + (((i * CT + j) * (AR / 4) + g) * 64 + lane) * 4;
                        st_coh64(p, (uint64_t)__float_as_uint(acc[i][j][4 * g]) | ((uint64_t)__float_as_uint(acc[i][j][4 * g + 1]) << 32));
                        st_coh64(p + 2, (uint64_t)__float_as_uint(acc[i][j][4 * g + 2]) | ((uint64_t)__float_as_uint(acc[i][j][4 * g + 3]) << 32));
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                s_last = __hip_atomic_fetch_add(a.split_cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)ksplit - 1 ? 1u : 0u;
            __syncthreads();
            if (!s_last) return;
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int r = 0; r < AR; ++r) acc[i][j][r] = 0.f;
            for (int sp = 0; sp < ksplit; ++sp) {
                const float* theirs = a.split_ws + (((size_t)tile_id * ksplit + sp) * 8 + wave) * PER_WAVE;
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
#pragma unroll
                        for (int g = 0; g < AR / 4; ++g) {
                            const u32x4 v = ld_coh128(theirs + (((i * CT + j) * (AR / 4) + g) * 64 + lane) * 4);
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[i][j][4 * g + r] += __uint_as_float(v[r]);
                        }
            }
            if (threadIdx.x == 0) __hip_atomic_store(a.split_cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // epilogue.  The products were issued as W-tile x X-tile (operands swapped), so a lane holds runs of FOUR
    // consecutive output columns of one row: 16x16 -> row = lane & 15, cols 4 * (lane >> 4) + [0,4);
    // 32x32 -> row = lane & 31, cols 8 * g + 4 * (lane >> 5) + [0,4) for g = 0..3.  One 8-byte store per run.
    if constexpr (SW) {
        if (seg_act) {
            const int c0 = n0 / 2 + wc * 32 + 4 * (lane >> 4);
            const bool per_op = a.sg.act_mode == 1;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int lrow = m0 + wr * WM + i * MF + (lane & LR);
                if (lrow >= rows_valid) continue;
                const int row = row_base + lrow;
#pragma unroll
                for (int j = 0; j < CT / 2; ++j) {
                    const int col = c0 + j * MF;
                    if (col >= seg_cols) continue;   // half is a multiple of 4: a run is inside or outside
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gt = A16::rnd(acc[i][j][e]), up = A16::rnd(acc[i][j + CT / 2][e]);
                        if (per_op) {   // nn::silu(gate) * up, each primitive rounded to bf16 (silu_mul_kernel, prefill.hip)
                            const float sg = A16::rnd(1.0f / (1.0f + expf(-gt)));
                            v[e] = A16::rnd(gt * sg) * up;
                        } else {        // fused_swiglu: one rounding (swiglu_strided_kernel, dit.hip)
                            v[e] = gt / (1.0f + expf(-gt)) * up;
                        }
                    }
                    *reinterpret_cast<u32x2*>(a.sg.out_act + (size_t)row * a.sg.ld_act + col) =
                        u32x2{A16::pack(v[0], v[1]), A16::pack(v[2], v[3])};
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int lrow = m0 + wr * WM + i * MF + (lane & LR);
                if (lrow >= rows_valid) continue;
                const int row = row_base + lrow;
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const int col = n0 + wc * 64 + j * MF + 4 * (lane >> 4);
                    if (col >= seg_cols) continue;   // widths are multiples of 4
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
                    if constexpr (IMPL) {   // convolution epilogue: bias, optional shortcut, any width (conv_out has 3 columns)
                        const bf16_t* rs = a.sg.im_resid;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (col + e >= seg_cols) break;
                            const size_t o = (size_t)row * seg_ld + col + e;
                            float x = v[e] + (seg_bias ? A16::val(seg_bias[col + e]) : 0.f);
                            if (rs) x = A16::val(rs[o]) + A16::rnd(x);
                            seg_out[o] = A16::bits(x);
                        }
                        continue;
                    }
                    if (seg_bias) {
                        const u32x2 b = *reinterpret_cast<const u32x2*>(seg_bias + col);
                        v[0] += A16::lo(b[0]); v[1] += A16::hi(b[0]); v[2] += A16::lo(b[1]); v[3] += A16::hi(b[1]);
                    }
                    *reinterpret_cast<u32x2*>(seg_out + (size_t)row * seg_ld + col) = u32x2{A16::pack(v[0], v[1]), A16::pack(v[2], v[3])};
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int row = m0 + wr * WM + i * MF + (lane & LR);
        if (row >= a.M) continue;
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int g = 0; g < AR / 4; ++g) {
                const int col = n0 + wc * 64 + j * MF + (MF == 32 ? 8 * g + 4 * (lane >> 5) : 4 * (lane >> 4));
                if (col >= a.N) continue;
                const size_t o = (size_t)row * a.N + col;
                const bool full = col + 3 < a.N && (a.N & 3) == 0;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                if (full) {
                    if (a.bias) {
                        const u32x2 b = *reinterpret_cast<const u32x2*>(a.bias + col);
                        v[0] += A16::lo(b[0]); v[1] += A16::hi(b[0]); v[2] += A16::lo(b[1]); v[3] += A16::hi(b[1]);
                    }
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (a.gate) {
                        const u32x2 r = *reinterpret_cast<const u32x2*>(a.resid + o);
                        const u32x2 gt = *reinterpret_cast<const u32x2*>(a.gate + col);
                        v[0] = A16::lo(r[0]) + v[0] * A16::lo(gt[0]); v[1] = A16::hi(r[0]) + v[1] * A16::hi(gt[0]);
                        v[2] = A16::lo(r[1]) + v[2] * A16::lo(gt[1]); v[3] = A16::hi(r[1]) + v[3] * A16::hi(gt[1]);
                    } else if (a.resid) {
                        const u32x2 r = *reinterpret_cast<const u32x2*>(a.resid + o);
                        v[0] = A16::lo(r[0]) + A16::rnd(v[0]); v[1] = A16::hi(r[0]) + A16::rnd(v[1]);
                        v[2] = A16::lo(r[1]) + A16::rnd(v[2]); v[3] = A16::hi(r[1]) + A16::rnd(v[3]);
                    }
                    *reinterpret_cast<u32x2*>(a.out + o) = u32x2{A16::pack(v[0], v[1]), A16::pack(v[2], v[3])};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (col + e >= a.N) break;
                        float x = v[e] + (a.bias ? A16::val(a.bias[col + e]) : 0.f);
                        if (a.relu) x = fmaxf(x, 0.f);
                        if (a.gate) x = A16::val(a.resid[o + e]) + x * A16::val(a.gate[col + e]);
                        else if (a.resid) x = A16::val(a.resid[o + e]) + A16::rnd(x);
                        a.out[o + e] = A16::bits(x);
                    }
                }
            }
    }
}
#undef OMX_WAIT_RING
#undef OMX_BAR

// ---- 256 x 256 tile, FOUR waves, the K loop as generated assembly (round 5; tools/gen_gemm5_asm.py -> gemm5_body.inc has the register map,
//      the buffer protocol and the schedule): one wave per SIMD owning the whole register file, 128 x 128 of the tile per wave = 8 x 8 accumulators
//      of 16x16x32 (256 AGPRs), both operands by LDS-DMA into two 64-k buffers each, three barriers per 64 k -- the geometry of the vendor
//      library's best kernel on this chip (tools/blaslt_probe.py; EXPERIMENTS.md R5-4 has the measurements and why this one does not ship as the
//      default).  This file keeps what hipcc does well: tile order, segments / expert rows, the per-thread source rows (clamped at the edges,
//      gathered for the MoE form), the epilogues.  LDS: [X buffer 0 | X 1 | W 0 | W 1] of 256 rows x 64 k (128 KiB) + 20 parameter dwords
//      per thread.  K % 128 == 0, no implicit convolution, no K split; SW: the segmented projection (plain segments + SwiGLU pair tiles:
//      inside a wave's 128 columns the first 64 are gate rows, the last 64 the up rows of the SAME outputs, so a lane holds both). ----
namespace w5 {
constexpr int tiles_bytes(int tmr) { return 2 * tmr * 128 + 2 * 32768; }     // two X buffers of tmr rows, two W buffers of 256 rows, 64 k each
constexpr int smem_bytes(int tmr) { return tiles_bytes(tmr) + 256 * (tmr / 32 + 12) * 4; }
constexpr int SMEM = smem_bytes(256);
}
#include "gemm5_body.inc"

// TMR = 128 (plain form): a 128 x 256 tile, a wave owns 64 x 128 -- for grids of exactly one such tile per CU (the O / down projections of a
// 2 048-token prompt), where the 256^2 tile would leave half of the chip idle; 4 X pieces per wave, 128 accumulator AGPRs.
template <bool SW, bool F16, int VAR = 0, int TMR = 256>
__global__ __launch_bounds__(256) void gemm_nt_w4_kernel(const GemmArgs a) {
    static_assert(TMR == 256 || (TMR == 128 && !SW && VAR == 0), "the 128-row tile exists in the plain form");
    typedef Act16<F16> A16;
    constexpr int NB = 8;                          // 16-column blocks per wave
    constexpr int NBI = TMR / 32;                  // 16-row blocks per wave
    constexpr int WROWS = TMR / 2;                 // rows per wave
    constexpr int XP = TMR / 32;                   // X pieces (8 rows each) a wave stages per K step
    constexpr int NPRM = XP + 12;
    constexpr int WCOLS = 16 * NB;                 // columns per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    if (a.stagger > 0 && blockIdx.x < 256) {      // (experiment: the first round's tiles start spread out, so that later rounds do not all write at once)
        const int n = __builtin_amdgcn_readfirstlane((int)((blockIdx.x >> 3) & 31u) * a.stagger);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1, l16 = lane & 15, kg = lane >> 4;
    // tile order: the 256^2 kernel's (XCD-aware remap, 8 x 4 patches per XCD)
    const int nblk = a.grid_m * a.grid_n;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm, tn;
    {
        constexpr int GM = 8;
        const int per_group = GM * a.grid_n;
        const int group = bid / per_group, in_group = bid % per_group;
        const int first_m = group * GM;
        const int gm = min(a.grid_m - first_m, GM);
        tm = first_m + in_group % gm;
        tn = in_group / gm;
    }
    const bf16_t* seg_w = a.w;
    const bf16_t* seg_bias = nullptr;
    bf16_t* seg_out = a.out;
    int seg_cols = a.N, seg_ld = a.N;
    bool seg_act = false;
    if constexpr (SW) {
        const GemmSegs& g = a.sg;
        if (tn >= g.act_tile0) {
            seg_act = true;
            tn -= g.act_tile0;
            seg_cols = g.half;
        } else {
            const int sidx = (g.n_plain > 1 && tn >= g.plain[1].tile0) + (g.n_plain > 2 && tn >= g.plain[2].tile0);
            const GemmSeg& sgm = g.plain[sidx];
            seg_w = sgm.w; seg_bias = sgm.bias; seg_out = sgm.out; seg_cols = sgm.cols; seg_ld = sgm.ld;
            tn -= sgm.tile0;
        }
    }
    int m0 = tm * TMR, rows_valid = a.M, row_base = 0;
    size_t w_off = 0;
    const uint32_t* row_src = nullptr;
    if constexpr (SW) {
        if (a.g.tile_expert) {
            if (tm >= *a.g.n_tiles) return;
            const int e = a.g.tile_expert[tm];
            row_base = a.g.seg_start[e];
            rows_valid = a.g.seg_start[e + 1] - row_base;
            m0 = a.g.tile_m0[tm];
            row_src = a.g.row_src;
            w_off = (size_t)e * a.g.w_estride;
        }
    }
    const int n0 = tn * 256;

    // ---- per-thread parameters of the K loop (gen_gemm5_asm.py): DMA source offsets of this thread's 8 X and 8 W pieces (piece wave * 8 + it =
    //      tile rows (wave * 8 + it) * 8 + (lane >> 3), 16-B chunk (lane & 7) ^ ((row >> 1) & 7) of the row's 128 B), fragment read addresses
    //      per k half kh (row = l16 of the block, chunk (4 kh + kg) ^ ((row >> 1) & 7)).  W tile row R sits in wave column R / 128; in a
    //      SwiGLU tile its first 64 rows are gate rows, the other 64 the up rows of the same outputs (one tile = 128 outputs): a wave
    //      stages 64 tile rows -- gate OR up rows. ----
    uint32_t prm[NPRM];
    {
        const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int it = 0; it < XP; ++it) {
            const int R = (wave * XP + it) * 8 + r8;
            const int chunk = slot ^ ((R >> 1) & 7);
            int xr = min(m0 + R, rows_valid - 1) + row_base;
            if constexpr (SW) {
                if (row_src) xr = (int)row_src[xr];
            }
            prm[it] = (uint32_t)(((int64_t)xr * a.K + chunk * 8) * 2);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int R = (wave * 8 + it) * 8 + r8;
            const int chunk = slot ^ ((R >> 1) & 7);
            int wrow;
            if (SW && seg_act) wrow = min(n0 / 2 + (R / WCOLS) * (WCOLS / 2) + (R & (WCOLS / 2 - 1)), seg_cols - 1);
            else wrow = min(n0 + R, seg_cols - 1);
            prm[XP + it] = (uint32_t)(((int64_t)wrow * a.K + chunk * 8) * 2);
        }
        const unsigned tiles = (unsigned)(uintptr_t)smem;
        const int swf = (l16 >> 1) & 7;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            prm[XP + 8 + kh] = tiles + (unsigned)((wr * WROWS + l16) * 128 + ((4 * kh + kg) ^ swf) * 16);
            prm[XP + 10 + kh] = tiles + (unsigned)(2 * TMR * 128) + (unsigned)((wc * WCOLS + l16) * 128 + ((4 * kh + kg) ^ swf) * 16);
        }
    }
    u32x4* pblock = reinterpret_cast<u32x4*>(smem + w5::tiles_bytes(TMR)) + threadIdx.x * (NPRM / 4);
#pragma unroll
    for (int k = 0; k < NPRM / 4; ++k) pblock[k] = u32x4{prm[4 * k], prm[4 * k + 1], prm[4 * k + 2], prm[4 * k + 3]};
    const unsigned param_addr = (unsigned)(uintptr_t)pblock;
    const bf16_t* xbase = a.x;
    const bool up_rows = (wave & 1) != 0;          // (a wave's W pieces are tile rows [wave * 64, + 64): the second half of a wave column)
    const bf16_t* wbase = (SW && seg_act ? (up_rows ? a.sg.w_up : a.sg.w_gate) : seg_w) + w_off;
    const int ntrips = __builtin_amdgcn_readfirstlane((a.K / 64 - 2) / 2);
    const unsigned ldsx = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + (unsigned)wave * (unsigned)(XP * 1024));                   // this wave's X pieces
    const unsigned ldsww = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + (unsigned)(2 * TMR * 128) + (unsigned)wave * 8192u);     // ... and W pieces
    {   // (the asm takes these in SGPRs; values hipcc cannot prove uniform would be handed over in VGPRs)
        auto uni = [](const bf16_t* p) {
            const uint64_t v = (uint64_t)(uintptr_t)p;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi2 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
            return reinterpret_cast<const bf16_t*>((uintptr_t)(((uint64_t)hi2 << 32) | lo));
        };
        xbase = uni(xbase);
        wbase = uni(wbase);
    }

    const int wave_odd = __builtin_amdgcn_readfirstlane(wave & 1);     // (the generator's two-schedule form branches on it; unused otherwise)
    f32x16 acc[NBI * 2];      // acc[o] = a[16 o : 16 o + 15]: tile (i, j) of the wave's NBI x 8 is acc[i * 2 + (j >> 2)][(j & 3) * 4 + e]
#pragma unroll
    for (int t = 0; t < NBI * 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#define G5_OPERANDS                                                                                                                                  \
    : "+{a[0:15]}"(acc[0]), "+{a[16:31]}"(acc[1]), "+{a[32:47]}"(acc[2]), "+{a[48:63]}"(acc[3]), "+{a[64:79]}"(acc[4]), "+{a[80:95]}"(acc[5]),        \
      "+{a[96:111]}"(acc[6]), "+{a[112:127]}"(acc[7]), "+{a[128:143]}"(acc[8]), "+{a[144:159]}"(acc[9]), "+{a[160:175]}"(acc[10]),                    \
      "+{a[176:191]}"(acc[11]), "+{a[192:207]}"(acc[12]), "+{a[208:223]}"(acc[13]), "+{a[224:239]}"(acc[14]), "+{a[240:255]}"(acc[15])               \
    : "v"(param_addr), "s"(xbase), "s"(wbase), "s"(ntrips), "s"(ldsx), "s"(ldsww), "s"(wave_odd)                                                    \
    : G5_CLOBBERS
#define G5H_OPERANDS                                                                                                                                 \
    : "+{a[0:15]}"(acc[0]), "+{a[16:31]}"(acc[1]), "+{a[32:47]}"(acc[2]), "+{a[48:63]}"(acc[3]), "+{a[64:79]}"(acc[4]), "+{a[80:95]}"(acc[5]),        \
      "+{a[96:111]}"(acc[6]), "+{a[112:127]}"(acc[7])                                                                                                \
    : "v"(param_addr), "s"(xbase), "s"(wbase), "s"(ntrips), "s"(ldsx), "s"(ldsww), "s"(wave_odd)                                                    \
    : G5_CLOBBERS
    if constexpr (TMR == 128) {
        if constexpr (F16) asm volatile(G5H_BODY_F16 G5H_OPERANDS);
        else asm volatile(G5H_BODY G5H_OPERANDS);
    } else if (VAR == 0) {
        if constexpr (F16) asm volatile(G5_BODY_F16 G5_OPERANDS);
        else asm volatile(G5_BODY G5_OPERANDS);
    }
#ifdef OMX_G5_DIAG   // timing-only builds (tools/gen_gemm5_asm.py --diag, tools/gemm5_diag.py; OMX_GEMM_W4_VAR=1..7: the K loop without its DMA, its
                     // fragment reads, its barriers / landing waits, all three, and with only the reads, only the DMA, only the barriers; garbage results)
    else if (VAR == 1) asm volatile(G5_BODY_D1 G5_OPERANDS);
    else if (VAR == 2) asm volatile(G5_BODY_D2 G5_OPERANDS);
    else if (VAR == 3) asm volatile(G5_BODY_D3 G5_OPERANDS);
    else if (VAR == 4) asm volatile(G5_BODY_D4 G5_OPERANDS);
    else if (VAR == 5) asm volatile(G5_BODY_D5 G5_OPERANDS);
    else if (VAR == 6) asm volatile(G5_BODY_D6 G5_OPERANDS);
    else if (VAR == 7) asm volatile(G5_BODY_D7 G5_OPERANDS);
    else if (VAR == 8) { asm volatile(G5_BODY G5_OPERANDS); if (a.M > 0) return; }      // the whole K loop, no epilogue (what the epilogue costs per tile)
#endif
#undef G5_OPERANDS
#undef G5H_OPERANDS

    // ---- epilogue.  Tile (i, j), element e = row (wr * WROWS + i * 16 + l16), column (wc * 128 + j * 16 + 4 kg + e) of the tile: a lane holds
    //      runs of four consecutive columns, 16 x NBI of them.  Interior tiles (every row and column exists) take straight-line code per
    //      epilogue kind, chosen by ONE uniform branch (the general form -- every run and element checked, every option carried -- is 20 000
    //      instructions a wave would fetch once per tile), and they do not store from the accumulator layout: an 8-byte store per run
    //      writes 16 rows x 32 bytes per instruction, 4 096 write requests per tile that took ~12 us of every tile (tools/gemm_tile_overhead.py:
    //      the K loop without its epilogue has NO per-tile cost worth the name).  The wave parks its finished 16-bit tile in LDS (its own
    //      WROWS x 272-byte region of the tile buffers, dead by now) and stores whole rows: 16 bytes per lane, 4 rows x 256 bytes per instruction. ----
#define ACC(i, j, e) acc[(i) * 2 + ((j) >> 2)][((j) & 3) * 4 + (e)]
    auto runs = [&](auto&& fn) {   // fn(i, j): row block, column block
#pragma unroll
        for (int i = 0; i < NBI; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) fn(i, j);
    };
    const bool interior = m0 + TMR <= rows_valid && n0 + 256 <= (SW && seg_act ? 2 * seg_cols : seg_cols) && (seg_ld & 3) == 0;
    const int lrow0 = m0 + wr * WROWS + l16;                // + i * 16
    constexpr int PARK_LD = 272;                            // bytes per parked row (256 + 16: the 8-byte writes and 16-byte reads 2-way at worst)
    unsigned char* park = smem + wave * (WROWS * PARK_LD);
    // a run's two packed words: to the parking row (staged) or straight to memory
    auto put = [&](auto staged_c, bf16_t* gptr, int i, int j, uint32_t p0, uint32_t p1) {
        if constexpr (decltype(staged_c)::value) *reinterpret_cast<u32x2*>(park + (i * 16 + l16) * PARK_LD + j * 32 + kg * 8) = u32x2{p0, p1};
        else *reinterpret_cast<u32x2*>(gptr) = u32x2{p0, p1};
    };
    // the parked tile -> memory, `cols` (128 or 64) 16-bit columns per row starting at base (row 0 of the wave's rows)
    auto flush = [&](bf16_t* base, size_t ld, int cols) {
        __syncthreads();
        if (cols == 128) {
#pragma unroll 8
            for (int it = 0; it < WROWS / 4; ++it) {
                const int row = 4 * it + (lane >> 4), c = lane & 15;
                *reinterpret_cast<u32x4*>(base + (size_t)row * ld + c * 8) = *reinterpret_cast<const u32x4*>(park + row * PARK_LD + c * 16);
            }
        } else {
#pragma unroll 8
            for (int it = 0; it < WROWS / 8; ++it) {
                const int row = 8 * it + (lane >> 3), c = lane & 7;
                *reinterpret_cast<u32x4*>(base + (size_t)row * ld + c * 8) = *reinterpret_cast<const u32x4*>(park + row * PARK_LD + c * 16);
            }
        }
    };
    auto aligned = [&](const void* p, size_t ld) { return !a.no_park && (reinterpret_cast<uintptr_t>(p) & 15u) == 0 && (ld & 7) == 0; };
    if constexpr (SW) {
        if (seg_act) {
            const int c0 = n0 / 2 + wc * (WCOLS / 2) + 4 * kg;      // + j * 16, j < 4 (gate); the up value of the same output is tile j + 4
            const bool staged = interior && aligned(a.sg.out_act, (size_t)a.sg.ld_act);
            if (staged) __syncthreads();      // every wave is done with the tile buffers (and the parameter block)
            auto act = [&](auto per_op_c, auto checked_c, auto staged_c) {
                constexpr bool per_op = decltype(per_op_c)::value, checked = decltype(checked_c)::value;
                runs([&](int i, int j) {
                    if (j >= NB / 2) return;
                    const int lrow = lrow0 + i * 16, col = c0 + j * 16;
                    if (checked && (lrow >= rows_valid || col >= seg_cols)) return;   // half is a multiple of 4: a run is inside or outside
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gt = A16::rnd(ACC(i, j, e)), up = A16::rnd(ACC(i, j + NB / 2, e));
                        if (per_op) {   // nn::silu(gate) * up, each primitive rounded (silu_mul_kernel, prefill.hip)
                            const float sg = A16::rnd(1.0f / (1.0f + expf(-gt)));
                            v[e] = A16::rnd(gt * sg) * up;
                        } else {        // fused_swiglu: one rounding (swiglu_strided_kernel, dit.hip)
                            v[e] = gt / (1.0f + expf(-gt)) * up;
                        }
                    }
                    put(staged_c, a.sg.out_act + (size_t)(row_base + lrow) * a.sg.ld_act + col, i, j, A16::pack(v[0], v[1]), A16::pack(v[2], v[3]));
                });
            };
            if (staged) {
                if (a.sg.act_mode == 1) act(std::true_type{}, std::false_type{}, std::true_type{});
                else act(std::false_type{}, std::false_type{}, std::true_type{});
                flush(a.sg.out_act + (size_t)(row_base + m0 + wr * WROWS) * a.sg.ld_act + n0 / 2 + wc * (WCOLS / 2), (size_t)a.sg.ld_act, 64);
            } else if (interior) {
                if (a.sg.act_mode == 1) act(std::true_type{}, std::false_type{}, std::false_type{});
                else act(std::false_type{}, std::false_type{}, std::false_type{});
            } else {
                if (a.sg.act_mode == 1) act(std::true_type{}, std::true_type{}, std::false_type{});
                else act(std::false_type{}, std::true_type{}, std::false_type{});
            }
        } else {
            const int c0 = n0 + wc * WCOLS + 4 * kg;
            const bool staged = interior && aligned(seg_out, (size_t)seg_ld);
            if (staged) __syncthreads();
            auto plain = [&](auto bias_c, auto checked_c, auto staged_c) {
                constexpr bool BIAS = decltype(bias_c)::value, checked = decltype(checked_c)::value;
                runs([&](int i, int j) {
                    const int lrow = lrow0 + i * 16, col = c0 + j * 16;
                    if (checked && (lrow >= rows_valid || col >= seg_cols)) return;   // widths are multiples of 4
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ACC(i, j, e);
                    if (BIAS) {
                        const u32x2 bb = *reinterpret_cast<const u32x2*>(seg_bias + col);
                        v[0] += A16::lo(bb[0]); v[1] += A16::hi(bb[0]); v[2] += A16::lo(bb[1]); v[3] += A16::hi(bb[1]);
                    }
                    put(staged_c, seg_out + (size_t)(row_base + lrow) * seg_ld + col, i, j, A16::pack(v[0], v[1]), A16::pack(v[2], v[3]));
                });
            };
            if (staged) {
                if (seg_bias) plain(std::true_type{}, std::false_type{}, std::true_type{});
                else plain(std::false_type{}, std::false_type{}, std::true_type{});
                flush(seg_out + (size_t)(row_base + m0 + wr * WROWS) * seg_ld + n0 + wc * WCOLS, (size_t)seg_ld, 128);
            } else if (interior) {
                if (seg_bias) plain(std::true_type{}, std::false_type{}, std::false_type{});
                else plain(std::false_type{}, std::false_type{}, std::false_type{});
            } else {
                if (seg_bias) plain(std::true_type{}, std::true_type{}, std::false_type{});
                else plain(std::false_type{}, std::true_type{}, std::false_type{});
            }
        }
        return;
    }
    const int c0 = n0 + wc * WCOLS + 4 * kg;
    if (interior && !a.relu) {
        const bool staged = aligned(a.out, (size_t)a.N);
        if (staged) __syncthreads();
        auto plain = [&](auto bias_c, auto mode_c, auto staged_c) {
            constexpr bool BIAS = decltype(bias_c)::value;
            constexpr int MODE = decltype(mode_c)::value;      // 0 store, 1 residual add (two roundings), 2 gated residual
            runs([&](int i, int j) {
                const int col = c0 + j * 16;
                const size_t o = (size_t)(lrow0 + i * 16) * a.N + col;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ACC(i, j, e);
                if (BIAS) {
                    const u32x2 bb = *reinterpret_cast<const u32x2*>(a.bias + col);
                    v[0] += A16::lo(bb[0]); v[1] += A16::hi(bb[0]); v[2] += A16::lo(bb[1]); v[3] += A16::hi(bb[1]);
                }
                if (MODE == 2) {
                    const u32x2 r = *reinterpret_cast<const u32x2*>(a.resid + o);
                    const u32x2 gt = *reinterpret_cast<const u32x2*>(a.gate + col);
                    v[0] = A16::lo(r[0]) + v[0] * A16::lo(gt[0]); v[1] = A16::hi(r[0]) + v[1] * A16::hi(gt[0]);
                    v[2] = A16::lo(r[1]) + v[2] * A16::lo(gt[1]); v[3] = A16::hi(r[1]) + v[3] * A16::hi(gt[1]);
                } else if (MODE == 1) {
                    const u32x2 r = *reinterpret_cast<const u32x2*>(a.resid + o);
                    v[0] = A16::lo(r[0]) + A16::rnd(v[0]); v[1] = A16::hi(r[0]) + A16::rnd(v[1]);
                    v[2] = A16::lo(r[1]) + A16::rnd(v[2]); v[3] = A16::hi(r[1]) + A16::rnd(v[3]);
                }
                put(staged_c, a.out + o, i, j, A16::pack(v[0], v[1]), A16::pack(v[2], v[3]));
            });
        };
        const int mode = a.gate ? 2 : a.resid ? 1 : 0;
        auto by_mode = [&](auto bias_c, auto staged_c) {
            if (mode == 0) plain(bias_c, std::integral_constant<int, 0>{}, staged_c);
            else if (mode == 1) plain(bias_c, std::integral_constant<int, 1>{}, staged_c);
            else plain(bias_c, std::integral_constant<int, 2>{}, staged_c);
        };
        if (staged) {
            if (a.bias) by_mode(std::true_type{}, std::true_type{});
            else by_mode(std::false_type{}, std::true_type{});
            flush(a.out + (size_t)(m0 + wr * WROWS) * a.N + n0 + wc * WCOLS, (size_t)a.N, 128);
        } else {
            if (a.bias) by_mode(std::true_type{}, std::false_type{});
            else by_mode(std::false_type{}, std::false_type{});
        }
        return;
    }
    // the general form: any edge, any option, element by element where a run is cut
    runs([&](int i, int j) {
        const int row = lrow0 + i * 16, col = c0 + j * 16;
        if (row >= a.M || col >= a.N) return;
        const size_t o = (size_t)row * a.N + col;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (col + e >= a.N) break;
            float x = ACC(i, j, e) + (a.bias ? A16::val(a.bias[col + e]) : 0.f);
            if (a.relu) x = fmaxf(x, 0.f);
            if (a.gate) x = A16::val(a.resid[o + e]) + x * A16::val(a.gate[col + e]);
            else if (a.resid) x = A16::val(a.resid[o + e]) + A16::rnd(x);
            a.out[o + e] = A16::bits(x);
        }
    });
#undef ACC
}

// does the four-wave kernel take this launch?  It wins where the chip is full and the power limit sets the pace (>= 256 tiles: +2..6 % over the
// eight-wave kernel, bit-identical results); with fewer tiles than CUs the eight waves hide more latency (-6..-11 %).  OMX_GEMM_W4=0 / 1: never /
// whenever the shape allows.
static bool w5_takes(int64_t x_rows, int K, int64_t w_rows_max, int n_tiles) {
    const char* e = getenv("OMX_GEMM_W4");
    const int mode = e ? atoi(e) : -1;
    if (mode == 0 || (mode < 0 && n_tiles < 256)) return false;
    return K % 128 == 0 && (x_rows * K + 64) * 2 < ((int64_t)1 << 32) && (w_rows_max * K + 64) * 2 < ((int64_t)1 << 32);
}

#ifdef OMX_EXPERIMENTS   // measured negative (EXPERIMENTS.md R5-3: 0.86-0.94 of the eight-phase kernel above): `make EXPERIMENTS=1`, OMX_GEMM_ASM=8
// ---- 256 x 256 tile with the K loop as generated assembly (round 5; tools/gen_gemm4_asm.py -> gemm4_body.inc has the register map, the
//      ring protocol and the hazard rules): eight waves, two per SIMD, 128 x 64 of the tile per wave = 8 accumulators of 32x32x16
//      (128 AGPRs), free-running between ONE barrier per 32 k.  This file keeps what hipcc does well: tile order, segments / expert rows,
//      the per-thread source rows (clamped at the edges, gathered for the MoE form), the epilogues.
//      LDS: [X buffer 0 | X 1 | W 0 | W 1] of 256 rows x 64 k (128 KiB) + 16 parameter dwords per thread (32 KiB).
//      K % 128 == 0, bf16, no implicit convolution, no K split; SW: the segmented projection (plain segments + SwiGLU pair tiles: inside
//      a wave's 64 columns the first 32 are gate rows, the last 32 the up rows of the SAME outputs, so a lane holds both).
namespace w4 {
constexpr int TILES_B = 4 * 32768, PARAM_B = 512 * 64, SMEM = TILES_B + PARAM_B;
}
#include "gemm4_body.inc"

template <bool SW, int VAR = 0>
__global__ __launch_bounds__(512) void gemm_bf16_nt_asm_kernel(const GemmArgs a) {
    typedef Act16<false> A16;
    constexpr int CJ = 2;                          // 32-column blocks per wave
    constexpr int WCOLS = 32 * CJ;                 // columns per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3, l32 = lane & 31, hi = lane >> 5;
    // tile order: the 256^2 kernel's (XCD-aware remap, 8 x 4 patches per XCD)
    const int nblk = a.grid_m * a.grid_n;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm, tn;
    {
        constexpr int GM = 8;
        const int per_group = GM * a.grid_n;
        const int group = bid / per_group, in_group = bid % per_group;
        const int first_m = group * GM;
        const int gm = min(a.grid_m - first_m, GM);
        tm = first_m + in_group % gm;
        tn = in_group / gm;
    }
    const bf16_t* seg_w = a.w;
    const bf16_t* seg_bias = nullptr;
    bf16_t* seg_out = a.out;
    int seg_cols = a.N, seg_ld = a.N;
    bool seg_act = false;
    if constexpr (SW) {
        const GemmSegs& g = a.sg;
        if (tn >= g.act_tile0) {
            seg_act = true;
            tn -= g.act_tile0;
            seg_cols = g.half;
        } else {
            const int sidx = (g.n_plain > 1 && tn >= g.plain[1].tile0) + (g.n_plain > 2 && tn >= g.plain[2].tile0);
            const GemmSeg& sgm = g.plain[sidx];
            seg_w = sgm.w; seg_bias = sgm.bias; seg_out = sgm.out; seg_cols = sgm.cols; seg_ld = sgm.ld;
            tn -= sgm.tile0;
        }
    }
    int m0 = tm * 256, rows_valid = a.M, row_base = 0;
    size_t w_off = 0;
    const uint32_t* row_src = nullptr;
    if constexpr (SW) {
        if (a.g.tile_expert) {
            if (tm >= *a.g.n_tiles) return;
            const int e = a.g.tile_expert[tm];
            row_base = a.g.seg_start[e];
            rows_valid = a.g.seg_start[e + 1] - row_base;
            m0 = a.g.tile_m0[tm];
            row_src = a.g.row_src;
            w_off = (size_t)e * a.g.w_estride;
        }
    }
    const int n0 = tn * 256;

    // ---- per-thread parameters of the K loop (gen_gemm4_asm.py): DMA source offsets of this thread's 4 X and 4 W pieces (piece wave * 4 + it =
    //      tile rows (wave * 4 + it) * 8 + (lane >> 3), 16-B chunk (lane & 7) ^ ((row >> 1) & 7) of the row's 128 B), fragment read addresses
    //      per quarter q of the row (row = l32 of the wave's block, chunk (2 q + hi) ^ ((row >> 1) & 7)).  W tile row R sits in wave column
    //      R / 64; in a SwiGLU tile its first 32 rows are gate rows, the other 32 the up rows of the same outputs (one tile = 128
    //      outputs): a wave stages 32 tile rows -- gate OR up rows. ----
    uint32_t prm[16];
    {
        const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int R = (wave * 4 + it) * 8 + r8;
            const int chunk = slot ^ ((R >> 1) & 7);
            int xr = min(m0 + R, rows_valid - 1) + row_base;
            if constexpr (SW) {
                if (row_src) xr = (int)row_src[xr];
            }
            prm[it] = (uint32_t)(((int64_t)xr * a.K + chunk * 8) * 2);
            int wrow;
            if (SW && seg_act) wrow = min(n0 / 2 + (R / WCOLS) * (WCOLS / 2) + (R & (WCOLS / 2 - 1)), seg_cols - 1);
            else wrow = min(n0 + R, seg_cols - 1);
            prm[4 + it] = (uint32_t)(((int64_t)wrow * a.K + chunk * 8) * 2);
        }
        const unsigned tiles = (unsigned)(uintptr_t)smem;
        const int swf = (l32 >> 1) & 7;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            prm[8 + q] = tiles + (unsigned)((wr * 128 + l32) * 128 + ((2 * q + hi) ^ swf) * 16);
            prm[12 + q] = tiles + 65536u + (unsigned)((wc * WCOLS + l32) * 128 + ((2 * q + hi) ^ swf) * 16);
        }
    }
    u32x4* pblock = reinterpret_cast<u32x4*>(smem + w4::TILES_B) + threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) pblock[k] = u32x4{prm[4 * k], prm[4 * k + 1], prm[4 * k + 2], prm[4 * k + 3]};
    const unsigned param_addr = (unsigned)(uintptr_t)pblock;
    const bf16_t* xbase = a.x;
    const bool up_rows = (wave & 1) != 0;          // (a wave's W pieces are tile rows [wave * 32, + 32): the second half of a wave column)
    const bf16_t* wbase = (SW && seg_act ? (up_rows ? a.sg.w_up : a.sg.w_gate) : seg_w) + w_off;
    const int npairs = a.K / 128;
    const unsigned ldsw = (unsigned)(uintptr_t)smem + (unsigned)wave * 4096u;

    constexpr int NACC = 4 * CJ;
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#define G8_OPERANDS                                                                                                                   \
    : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7])                   \
    : "v"(param_addr), "s"(xbase), "s"(wbase), "s"(npairs), "s"(ldsw)                                                                \
    : G8_CLOBBERS
    if (VAR == 0) asm volatile(G8_BODY G8_OPERANDS);
#ifdef OMX_G4_DIAG   // timing-only builds (tools/gen_gemm4_asm.py --diag; OMX_GEMM_ASM_VAR=1..4: no DMA, no fragment reads, no barrier / waits, none of them)
    else if (VAR == 1) asm volatile(G8_BODY_D1 G8_OPERANDS);
    else if (VAR == 2) asm volatile(G8_BODY_D2 G8_OPERANDS);
    else if (VAR == 3) asm volatile(G8_BODY_D3 G8_OPERANDS);
    else if (VAR == 4) asm volatile(G8_BODY_D4 G8_OPERANDS);
#endif
#undef G8_OPERANDS

    // ---- epilogue: acc[i * CJ + j][4 g + e] = row (wr * 128 + i * 32 + l32), column (wc * WCOLS + j * 32 + 8 g + 4 hi + e) of the tile ----
    if constexpr (SW) {
        if (seg_act) {
            const bool per_op = a.sg.act_mode == 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lrow = m0 + wr * 128 + i * 32 + l32;
                if (lrow >= rows_valid) continue;
                const int row = row_base + lrow;
#pragma unroll
                for (int j = 0; j < CJ / 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = n0 / 2 + wc * (WCOLS / 2) + j * 32 + 8 * g + 4 * hi;
                        if (col >= seg_cols) continue;   // half is a multiple of 4: a run is inside or outside
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float gt = A16::rnd(acc[i * CJ + j][4 * g + e]), up = A16::rnd(acc[i * CJ + j + CJ / 2][4 * g + e]);
                            if (per_op) {   // nn::silu(gate) * up, each primitive rounded to bf16 (silu_mul_kernel, prefill.hip)
                                const float sg = A16::rnd(1.0f / (1.0f + expf(-gt)));
                                v[e] = A16::rnd(gt * sg) * up;
                            } else {        // fused_swiglu: one rounding (swiglu_strided_kernel, dit.hip)
                                v[e] = gt / (1.0f + expf(-gt)) * up;
                            }
                        }
                        *reinterpret_cast<u32x2*>(a.sg.out_act + (size_t)row * a.sg.ld_act + col) =
                            u32x2{A16::pack(v[0], v[1]), A16::pack(v[2], v[3])};
                    }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lrow = m0 + wr * 128 + i * 32 + l32;
                if (lrow >= rows_valid) continue;
                const int row = row_base + lrow;
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = n0 + wc * WCOLS + j * 32 + 8 * g + 4 * hi;
                        if (col >= seg_cols) continue;   // widths are multiples of 4
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i * CJ + j][4 * g + e];
                        if (seg_bias) {
                            const u32x2 b = *reinterpret_cast<const u32x2*>(seg_bias + col);
                            v[0] += A16::lo(b[0]); v[1] += A16::hi(b[0]); v[2] += A16::lo(b[1]); v[3] += A16::hi(b[1]);
                        }
                        *reinterpret_cast<u32x2*>(seg_out + (size_t)row * seg_ld + col) = u32x2{A16::pack(v[0], v[1]), A16::pack(v[2], v[3])};
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + wr * 128 + i * 32 + l32;
        if (row >= a.M) continue;
#pragma unroll
        for (int j = 0; j < CJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wc * WCOLS + j * 32 + 8 * g + 4 * hi;
                if (col >= a.N) continue;
                const size_t o = (size_t)row * a.N + col;
                const bool full = col + 3 < a.N && (a.N & 3) == 0;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i * CJ + j][4 * g + e];
                if (full) {
                    if (a.bias) {
                        const u32x2 b = *reinterpret_cast<const u32x2*>(a.bias + col);
                        v[0] += A16::lo(b[0]); v[1] += A16::hi(b[0]); v[2] += A16::lo(b[1]); v[3] += A16::hi(b[1]);
                    }
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (a.gate) {
                        const u32x2 r = *reinterpret_cast<const u32x2*>(a.resid + o);
                        const u32x2 gt = *reinterpret_cast<const u32x2*>(a.gate + col);
                        v[0] = A16::lo(r[0]) + v[0] * A16::lo(gt[0]); v[1] = A16::hi(r[0]) + v[1] * A16::hi(gt[0]);
                        v[2] = A16::lo(r[1]) + v[2] * A16::lo(gt[1]); v[3] = A16::hi(r[1]) + v[3] * A16::hi(gt[1]);
                    } else if (a.resid) {
                        const u32x2 r = *reinterpret_cast<const u32x2*>(a.resid + o);
                        v[0] = A16::lo(r[0]) + A16::rnd(v[0]); v[1] = A16::hi(r[0]) + A16::rnd(v[1]);
                        v[2] = A16::lo(r[1]) + A16::rnd(v[2]); v[3] = A16::hi(r[1]) + A16::rnd(v[3]);
                    }
                    *reinterpret_cast<u32x2*>(a.out + o) = u32x2{A16::pack(v[0], v[1]), A16::pack(v[2], v[3])};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (col + e >= a.N) break;
                        float x = v[e] + (a.bias ? A16::val(a.bias[col + e]) : 0.f);
                        if (a.relu) x = fmaxf(x, 0.f);
                        if (a.gate) x = A16::val(a.resid[o + e]) + x * A16::val(a.gate[col + e]);
                        else if (a.resid) x = A16::val(a.resid[o + e]) + A16::rnd(x);
                        a.out[o + e] = A16::bits(x);
                    }
                }
            }
    }
}

// does the asm-scheduled kernel take this shape?  (OMX_GEMM_ASM=0: the hipcc-scheduled eight-wave kernel)
static int asm_waves() {
    const char* e = getenv("OMX_GEMM_ASM");
    return e ? atoi(e) : 0;
}
static bool w4_takes(int64_t x_rows, int K, int64_t w_rows_max) {
    if (asm_waves() == 0) return false;
    return K % 128 == 0 && (x_rows * K + 64) * 2 < ((int64_t)1 << 32) && (w_rows_max * K + 64) * 2 < ((int64_t)1 << 32);
}

#else
static bool w4_takes(int64_t, int, int64_t) { return false; }
#endif   // OMX_EXPERIMENTS

// ---- 64 x 64 x 64 tile, deep LDS-DMA ring: GEMMs whose 128^2 grid cannot fill the chip ----
// (Paraformer layers: 501 x 512 x 512 is 16 tiles of 128^2; the text encoder and the DiT txt stream at 512 rows; short prompts.)
// With one or two blocks per CU and ONE tile of prefetch the 128^2 kernel pays a full memory round trip per K step
// (8 K steps of a 512-wide contraction took 15 us).  Here a block owns a 64 x 64 output (4x the blocks), and the K loop runs
// over a ring of NS stages of [A 64x64 | B 64x64] (16 KiB each) filled by global_load_lds: NS - 1 tiles are in flight
// while one is consumed, the wait is a counted vmcnt, one barrier per K step (the stage read in step t-1 is refilled right
// after the barrier of step t).  NS = 8 (128 KiB, one block per CU) when the grid has at most one block per CU -- a
// 512-wide contraction is then issued in full before the first wait -- NS = 4 (64 KiB, two blocks per CU) otherwise.
// Wave (wr, wc) owns 32 x 32 as 2 x 2 tiles of v_mfma_f32_16x16x32_bf16, operands swapped as in the 256^2 kernel so a
// lane holds four consecutive output columns (8-byte stores); same epilogue options as the other kernels.
namespace skinny {
constexpr int TM = 64, TN = 64, TK = 64, NT = 256;
constexpr int HALF = TM * TK * 2;       // 8 KiB: one operand tile
constexpr int STAGE = 2 * HALF;
}

template <int LOADS>
__device__ __forceinline__ void wait_vm_tiles(int tiles) {   // wait until at most `tiles` staged tiles (LOADS loads each) are pending
    switch (tiles) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LOADS) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LOADS) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * LOADS) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * LOADS) : "memory"); break;
    }
}

// SEG: the column tiles are the concatenation of up to three projections of the same input (GemmSegs::plain, tile0 in units of
// 64 columns): own weight, bias, output and width per segment -- q/k/v of a short prompt as ONE grid instead of three, two of
// which would be 32 blocks.
template <int NS, bool SEG = false>
__global__ __launch_bounds__(skinny::NT) void gemm_bf16_nt_skinny_kernel(const GemmArgs a) {
    using namespace skinny;
    static_assert(NS >= 2 && NS <= 8, "ring depth: the counted waits cover up to 6 pending tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware remap: each XCD gets a contiguous run of tiles (row-major: neighbours share the activation panel)
    int bid = blockIdx.x;
    {
        const int nblk = a.grid_m * a.grid_n;
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // an XCD's run of tiles walks the smaller operand's panels fastest, so its L2 pulls the larger operand only once
    // (weights wider than the activation block: column runs; every XCD then reads 1/8 of W instead of all of it)
    const bool col_runs = a.N > a.M;
    const int m0 = (col_runs ? bid % a.grid_m : bid / a.grid_n) * TM;
    int tn = col_runs ? bid / a.grid_m : bid % a.grid_n;
    const bf16_t* seg_w = a.w;
    const bf16_t* seg_bias = a.bias;
    bf16_t* seg_out = a.out;
    int seg_cols = a.N, seg_ld = a.N;
    bool seg_act = false;   // SwiGLU segment: a 64-column tile = 32 gate + 32 up columns -> 32 activation columns
    if constexpr (SEG) {
        const GemmSegs& g = a.sg;
        if (tn >= g.act_tile0) {
            seg_act = true;
            tn -= g.act_tile0;
            seg_cols = g.half;
            seg_bias = nullptr;
        } else {
            const int sidx = (g.n_plain > 1 && tn >= g.plain[1].tile0) + (g.n_plain > 2 && tn >= g.plain[2].tile0);
            const GemmSeg& sgm = g.plain[sidx];
            seg_w = sgm.w; seg_bias = sgm.bias; seg_out = sgm.out; seg_cols = sgm.cols; seg_ld = sgm.ld;
            tn -= sgm.tile0;
        }
    }
    const int n0 = tn * TN;
    // split K (gridDim.y > 1): this block owns K steps [t0, t0 + nt) of the tile; the partial tiles meet in the epilogue
    const int nt_all = a.K / TK, nsplit = gridDim.y, split = blockIdx.y;
    const int t0 = (int)((long long)nt_all * split / nsplit);
    const int nt = (int)((long long)nt_all * (split + 1) / nsplit) - t0;

    const bf16_t* srcA[2];
    const bf16_t* srcB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = i * NT + threadIdx.x;          // LDS chunk (16 B) of the operand tile, lane-linear inside a wave
        const int row = c >> 3;
        const int kc = (c & 7) ^ (row & 7);           // logical k-chunk kept at this slot (source-side swizzle)
        srcA[i] = a.x + (size_t)min(m0 + row, a.M - 1) * a.K + kc * 8 + (size_t)t0 * TK;
        if (SEG && seg_act) {
            // wave column wc = row >> 5 reads its first 16 B-rows from the gate weight, the next 16 from the up weight: gate and up of
            // one output column land in accumulator tiles j = 0 and j = 1 of the same lane
            const int oc = n0 / 2 + (row >> 5) * 16 + (row & 15);
            srcB[i] = (((row >> 4) & 1) ? a.sg.w_up : a.sg.w_gate) + (size_t)min(oc, seg_cols - 1) * a.K + kc * 8 + (size_t)t0 * TK;
        } else {
            srcB[i] = seg_w + (size_t)min(n0 + row, seg_cols - 1) * a.K + kc * 8 + (size_t)t0 * TK;
        }
    }
    auto stage = [&](int t) {
        unsigned char* st = smem + (t % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(srcA[i] + t * TK), (lds_ptr_t)(st + (i * NT + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(srcB[i] + t * TK), (lds_ptr_t)(st + HALF + (i * NT + wave * 64) * 16), 16, 0, 0);
    };

    typedef float accv __attribute__((ext_vector_type(4)));
    accv acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    for (int t = 0; t < NS - 1 && t < nt; ++t) stage(t);
    const int arow = lane & 15, kh = lane >> 4;
    for (int t = 0; t < nt; ++t) {
        // tiles issued so far: 0 .. min(t + NS - 2, nt - 1); everything younger than tile t may stay in flight
        wait_vm_tiles<4>(min(NS - 2, nt - 1 - t));
        __syncthreads();                              // tile t is in LDS for every wave; everyone is done reading tile t - 1
        if (t + NS - 1 < nt) stage(t + NS - 1);       // refills the stage tile t - 1 was read from
        const unsigned char* pa = smem + (t % NS) * STAGE;
        const unsigned char* pb = pa + HALF;
#pragma unroll
        for (int ks = 0; ks < TK / 32; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = lds_frag(pa, wr * 32 + i * 16 + arow, ks * 4 + kh);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = lds_frag(pb, wc * 32 + j * 16 + arow, ks * 4 + kh);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    }

    if (nsplit > 1) {
        // Every split stores its f32 partial tile write-through; the LAST one to arrive (one relaxed atomic per block) re-reads
        // all of them in split order -- its own included, so the sum has one fixed order whoever arrives last -- and runs the
        // epilogue.  (Same hand-over as the split-KV combine of attn_decode.hip.)
        __shared__ unsigned s_last;
        const size_t tile = blockIdx.x;
        float* mine = a.split_ws + ((tile * nsplit + split) * 4 + wave) * (4 * 64 * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* p = mine + ((i * 2 + j) * 64 + lane) * 4;
                st_coh64(p, (uint64_t)__float_as_uint(acc[i][j][0]) | ((uint64_t)__float_as_uint(acc[i][j][1]) << 32));
                st_coh64(p + 2, (uint64_t)__float_as_uint(acc[i][j][2]) | ((uint64_t)__float_as_uint(acc[i][j][3]) << 32));
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            s_last = __hip_atomic_fetch_add(a.split_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nsplit - 1 ? 1u : 0u;
        __syncthreads();
        if (!s_last) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) {
            const float* theirs = a.split_ws + ((tile * nsplit + sp) * 4 + wave) * (4 * 64 * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x4 v = ld_coh128(theirs + ((i * 2 + j) * 64 + lane) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += __uint_as_float(v[r]);
                }
        }
        if (threadIdx.x == 0) __hip_atomic_store(a.split_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    if constexpr (SEG) {
        if (seg_act) {
            const bool per_op = a.sg.act_mode == 1;
            const int col = n0 / 2 + wc * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = m0 + wr * 32 + i * 16 + (lane & 15);
                if (row >= a.M || col >= seg_cols) continue;   // half is a multiple of 4
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gt = round_bf16(acc[i][0][e]), up = round_bf16(acc[i][1][e]);
                    if (per_op) {
                        const float sg = round_bf16(1.0f / (1.0f + expf(-gt)));
                        v[e] = round_bf16(gt * sg) * up;
                    } else {
                        v[e] = gt / (1.0f + expf(-gt)) * up;
                    }
                }
                *reinterpret_cast<u32x2*>(a.sg.out_act + (size_t)row * a.sg.ld_act + col) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            }
            return;
        }
    }
    // epilogue: W-tile x X-tile products, so a lane holds output row (lane & 15), columns 4 * (lane >> 4) + [0, 4) of each tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + wr * 32 + i * 16 + (lane & 15);
        if (row >= a.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 32 + j * 16 + 4 * (lane >> 4);
            if (col >= seg_cols) continue;
            const size_t o = (size_t)row * seg_ld + col;
            const bool full = col + 3 < seg_cols && (seg_cols & 3) == 0 && (seg_ld & 3) == 0;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
            if (full) {
                if (seg_bias) {
                    const u32x2 b = *reinterpret_cast<const u32x2*>(seg_bias + col);
                    v[0] += bf16lo(b[0]); v[1] += bf16hi(b[0]); v[2] += bf16lo(b[1]); v[3] += bf16hi(b[1]);
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (a.gate) {
                    const u32x2 r = *reinterpret_cast<const u32x2*>(a.resid + o);
                    const u32x2 gt = *reinterpret_cast<const u32x2*>(a.gate + col);
                    v[0] = bf16lo(r[0]) + v[0] * bf16lo(gt[0]); v[1] = bf16hi(r[0]) + v[1] * bf16hi(gt[0]);
                    v[2] = bf16lo(r[1]) + v[2] * bf16lo(gt[1]); v[3] = bf16hi(r[1]) + v[3] * bf16hi(gt[1]);
                } else if (a.resid) {
                    const u32x2 r = *reinterpret_cast<const u32x2*>(a.resid + o);
                    v[0] = bf16lo(r[0]) + round_bf16(v[0]); v[1] = bf16hi(r[0]) + round_bf16(v[1]);
                    v[2] = bf16lo(r[1]) + round_bf16(v[2]); v[3] = bf16hi(r[1]) + round_bf16(v[3]);
                }
                *reinterpret_cast<u32x2*>(seg_out + o) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (col + e >= seg_cols) break;
                    float x = v[e] + (seg_bias ? bf16_to_f32(seg_bias[col + e]) : 0.f);
                    if (a.relu) x = fmaxf(x, 0.f);
                    if (a.gate) x = bf16_to_f32(a.resid[o + e]) + x * bf16_to_f32(a.gate[col + e]);
                    else if (a.resid) x = bf16_to_f32(a.resid[o + e]) + round_bf16(x);
                    seg_out[o + e] = f32_to_bf16(x);
                }
            }
        }
    }
}

// ---- fallback: any K / alignment.  64x64 tile, BK = 32, register staging with zero fill ----
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_nt_generic_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[64][40];   // +8 pad: conflict-free 16-B fragment reads
    __shared__ __attribute__((aligned(16))) bf16_t sB[64][40];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;   // each wave: 32 x 32 output = one MFMA tile
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < a.K; k0 += 32) {
        // 64 rows x 32 k per operand = 2048 elements; 256 threads x 8
        {
            const int row = threadIdx.x >> 2, kk = (threadIdx.x & 3) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + kk + e;
                const int gm = m0 + row, gn = n0 + row;
                sA[row][kk + e] = (gm < a.M && k < a.K) ? a.x[(size_t)gm * a.K + k] : (bf16_t)0;
                sB[row][kk + e] = (gn < a.N && k < a.K) ? a.w[(size_t)gn * a.K + k] : (bf16_t)0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 fa = *reinterpret_cast<const bf16x8*>(&sA[wm * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8]);
            const bf16x8 fb = *reinterpret_cast<const bf16x8*>(&sB[wn * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= a.N) return;
    const float bv = a.bias ? bf16_to_f32(a.bias[col]) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < a.M) {
            float v = acc[r] + bv;
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.gate) v = bf16_to_f32(a.resid[(size_t)row * a.N + col]) + v * bf16_to_f32(a.gate[col]);
            else if (a.resid) v = bf16_to_f32(a.resid[(size_t)row * a.N + col]) + round_bf16(v);
            a.out[(size_t)row * a.N + col] = f32_to_bf16(v);
        }
    }
}

int ensure_attr() {
    static bool attr_set = false;
    if (!attr_set) {
        const int shmem = 4 * TILE_BYTES;
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_256_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, big::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_256_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, big::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_256_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_256_kernel<16, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_256_kernel<16, false, false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, big::smem_bytes(128)));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_256_kernel<16, false, false, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big::smem_bytes(128)));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_256_kernel<16, false, false, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_256_kernel<16, true, false, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, false, 0, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::smem_bytes(128)));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, true, 0, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::smem_bytes(128)));
#ifdef OMX_G5_DIAG
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, false, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, false, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, w5::SMEM));
#endif
#ifdef OMX_EXPERIMENTS
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_asm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, w4::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_asm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, w4::SMEM));
#endif
#ifdef OMX_G4_DIAG
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_asm_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, w4::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_asm_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, w4::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_asm_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, w4::SMEM));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_asm_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, w4::SMEM));
#endif
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_skinny_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * skinny::STAGE));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_skinny_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * skinny::STAGE));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_skinny_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * skinny::STAGE));
        OMX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf16_nt_skinny_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * skinny::STAGE));
        attr_set = true;
    }
    return 0;
}

}  // namespace

// split-K scratch of the ring kernel: ONE fixed-size buffer per stream (two streams may run such GEMMs at the same time; host
// threads sharing the null stream get one each), sized for the largest launch ring_splits() can produce (blocks * splits <= 512
// tiles of 64 x 64 partials, <= 160 arrival counters).  The size never depends on the call, so the split decision is a function
// of the shape alone and a captured launch takes the same path -- and sums in the same order -- as an eager one.  A first use
// inside a stream capture allocates under a relaxed capture mode (hipMalloc is not a stream operation).
namespace {
constexpr size_t kSplitWsFloats = (size_t)512 * 64 * 64, kSplitCnt = 256;
constexpr size_t kSplitBigFloats = (size_t)128 * 2 * 256 * 256;   // 256^2 kernel: up to 128 tiles in two K halves (64 MiB, on first use)
struct SplitWs { float* ws = nullptr; unsigned* cnt = nullptr; float* big = nullptr; };
std::mutex g_split_mu;
std::map<std::pair<hipStream_t, std::thread::id>, SplitWs> g_split_ws;
std::pair<hipStream_t, std::thread::id> split_key(hipStream_t s) { return {s, s ? std::thread::id() : std::this_thread::get_id()}; }
int split_workspace(hipStream_t s, size_t floats, size_t tiles, float** ws, unsigned** cnt) {
    OMX_REQUIRE(floats <= kSplitWsFloats && tiles <= kSplitCnt, "ring GEMM split-K: %zu partial floats / %zu tiles exceed the fixed scratch", floats, tiles);
    std::lock_guard<std::mutex> lk(g_split_mu);
    SplitWs& w = g_split_ws[split_key(s)];
    if (!w.ws) {
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        OMX_HIP_CHECK(hipThreadExchangeStreamCaptureMode(&mode));
        hipError_t e = hipMalloc((void**)&w.ws, kSplitWsFloats * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&w.cnt, kSplitCnt * 4);
        if (e == hipSuccess) {
            // the counters must be zero before the first launch on `s` reads them.  `s` may be a non-blocking stream (not ordered
            // against the null stream): zero them ON `s` -- unless `s` is being captured, where the memset would become a graph
            // node; then zero through the null stream and wait for it here (the graph only runs after the capture ends)
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            const bool capturing = s != nullptr && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
            if (capturing) {
                e = hipMemset(w.cnt, 0, kSplitCnt * 4);
                if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            } else {
                e = hipMemsetAsync(w.cnt, 0, kSplitCnt * 4, s);
            }
        }
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (e != hipSuccess) {
            if (w.ws) (void)hipFree(w.ws);
            if (w.cnt) (void)hipFree(w.cnt);
            g_split_ws.erase(split_key(s));
            return set_error("ring GEMM split-K scratch: %s", hipGetErrorString(e));
        }
    }
    *ws = w.ws; *cnt = w.cnt;
    return 0;
}
// the 256^2 kernel's partial tiles: the same counters (every launch leaves them zero), a second, larger buffer
int split_workspace_big(hipStream_t s, size_t floats, size_t tiles, float** ws, unsigned** cnt) {
    OMX_REQUIRE(floats <= kSplitBigFloats && tiles <= kSplitCnt, "GEMM split-K: %zu partial floats / %zu tiles exceed the fixed scratch", floats, tiles);
    float* small = nullptr;
    if (split_workspace(s, 0, tiles, &small, cnt)) return 1;
    std::lock_guard<std::mutex> lk(g_split_mu);
    SplitWs& w = g_split_ws[split_key(s)];
    if (!w.big) {
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        OMX_HIP_CHECK(hipThreadExchangeStreamCaptureMode(&mode));
        const hipError_t e = hipMalloc((void**)&w.big, kSplitBigFloats * 4);
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (e != hipSuccess) { w.big = nullptr; return set_error("GEMM split-K scratch: %s", hipGetErrorString(e)); }
    }
    *ws = w.big;
    return 0;
}
}  // namespace

// the owner of `s` is about to destroy it: give its split-K scratch back (engine / model destructors)
void gemm_release_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_split_mu);
    auto it = g_split_ws.find(split_key(s));
    if (it == g_split_ws.end()) return;
    (void)hipStreamSynchronize(s);
    if (it->second.ws) (void)hipFree(it->second.ws);
    if (it->second.big) (void)hipFree(it->second.big);
    if (it->second.cnt) (void)hipFree(it->second.cnt);
    g_split_ws.erase(it);
}

namespace {
// K splits of a ring-kernel launch: only when the tile grid leaves most CUs without a block AND the K loop is long -- the
// serial chain of K steps, each a memory round trip, is then what the launch takes (down projection of a 128-token prompt:
// 128 blocks x 192 steps).  At least 8 steps stay in each split (Paraformer FFN down: 64 tiles x 32 steps -> 4 splits, 17 -> 12 us).
int ring_splits(int blocks, int nt) {
    const char* env = getenv("OMX_GEMM_SPLITK");
    if (env && env[0] == '0') return 1;
    if (blocks > 160 || nt < 32) return 1;
    return std::max(1, std::min(std::min(8, 512 / blocks), nt / 8));
}
}  // namespace

// a caller's tile preference for its next plain GEMMs on this thread (0: none).  128: the 128 x 256 tile even where the shape alone
// would pick many small tiles -- a GEMM that runs BESIDE another one on a second stream (the DiT's 512-row txt chain next to the img
// chain: 48 such blocks fit the CUs the img grid leaves idle, 384 tiles of 64^2 queue behind its workgroups)
static thread_local int g_tile_hint = 0;
void gemm_tile_hint(int rows) { g_tile_hint = rows; }
// float16 operands / results for this thread's next GEMMs (a float16 checkpoint's prompt pass, engine.hip): the eight-wave kernel's
// float16 instantiations serve every shape (plain: 256- or 128-row tiles; segmented: 256-row tiles)
static thread_local bool g_gemm_f16 = false;
bool gemm_set_f16(bool on) { const bool was = g_gemm_f16; g_gemm_f16 = on; return was; }   // returns the previous state (nested scopes)

// the two experiment-only switches of the four-wave kernel (R5-8: staggered tile starts, un-parked epilogue): read ONCE -- every GEMM launch
// used to call getenv for them on the host's hot path, in five copies of this block
static void apply_experiment_switches(GemmArgs& a) {
    static const int stagger = [] { const char* e = getenv("OMX_GEMM_STAGGER"); return e ? atoi(e) : 0; }();
    static const bool no_park = [] { const char* e = getenv("OMX_GEMM_W4_PARK"); return e && e[0] == '0'; }();
    a.stagger = stagger;
    a.no_park = no_park;
}

static int launch_gemm_impl(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, const bf16_t* resid,
                            const bf16_t* gate, int M, int N, int K, hipStream_t s, int relu = 0) {
    OMX_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
    GemmArgs a = {x, w, bias, resid, gate, out, M, N, K, (M + BM - 1) / BM, (N + BN - 1) / BN, {}, relu};
    apply_experiment_switches(a);
    const bool fast = (K % BK == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15u) == 0;
    {   // a handful of rows: stream the weights once (gemv_rows.hip) instead of a matrix-core tile grid that is mostly padding;
        // OMX_GEMV_ROWS=0 keeps the GEMM kernels
        const char* re = getenv("OMX_GEMV_ROWS");
        const bool rows_off = re && re[0] == '0';
        if (!g_gemm_f16 && !rows_off && M <= 8 && (int64_t)N * K >= (1 << 20) && gemv_rows_supported(M, N, K, x, w))
            return launch_gemv_rows(out, x, w, bias, resid, gate, M, N, K, relu, s);
    }
    if (g_gemm_f16) {
        OMX_REQUIRE(fast, "float16 gemm: K %% 64 == 0 and 16-byte aligned operands expected (M=%d K=%d)", M, K);
        if (ensure_attr()) return 1;
        const int t256 = ((M + 255) / 256) * ((N + 255) / 256);
        const char* te = getenv("OMX_GEMM_TILE");
        if (t256 <= 128 && !(te && atoi(te) == 256)) {   // at most half of the chip in 256^2 tiles: 128 x 256
            a.grid_m = (M + 127) / 128;
            a.grid_n = (N + 255) / 256;
            if (w5_takes(M, K, N, a.grid_m * a.grid_n)) gemm_nt_w4_kernel<false, true, 0, 128><<<a.grid_m * a.grid_n, 256, w5::smem_bytes(128), s>>>(a);
            else gemm_bf16_nt_256_kernel<16, false, false, 128, true><<<a.grid_m * a.grid_n, big::NT, big::smem_bytes(128), s>>>(a);
        } else {
            a.grid_m = (M + 255) / 256;
            a.grid_n = (N + 255) / 256;
            if (w5_takes(M, K, N, a.grid_m * a.grid_n)) gemm_nt_w4_kernel<false, true><<<a.grid_m * a.grid_n, 256, w5::SMEM, s>>>(a);
            else gemm_bf16_nt_256_kernel<16, false, false, 256, true><<<a.grid_m * a.grid_n, big::NT, big::SMEM, s>>>(a);
        }
        OMX_LAUNCH_CHECK();
        return 0;
    }
    if (fast) {
        if (ensure_attr()) return 1;
        // 256^2 tiles when they still cover the chip (one block per CU); OMX_GEMM_TILE=128|256 forces a kernel
        const int tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
        const char* tile_env = getenv("OMX_GEMM_TILE");
        const int forced = tile_env ? atoi(tile_env) : 0;
        // 64 .. 128 tiles of 256^2 leave half of the CUs idle (prefill of 2048 tokens: O and down projection, 8 x 16 tiles).  Opt-in
        // (OMX_GEMM_KSPLIT=1): each tile's K range in two halves -> up to 256 blocks, f32 partials summed by the tile's last block.
        // Measured (EXPERIMENTS.md R3-5): back to back on one matrix (weights resident in the Infinity Cache) K = 12288 runs 246 -> 202 us,
        // K = 4096 83 -> 124 us (the partials are 64 MiB written and read back); inside the real 2048-token prefill, where every
        // layer's weights come from HBM and the partials compete with them, 29.21 -> 29.21 ms: no gain, so it is not the default.
        const char* ks_env = getenv("OMX_GEMM_KSPLIT");
        const int ksplit = (ks_env && ks_env[0] == '1' && forced == 0 && tiles256 >= 64 && tiles256 <= 128 && K >= 4096 && K % 128 == 0) ? 2 : 1;
        const bool use256 = forced == 256 || ksplit > 1 || (forced != 128 && (tiles256 >= 160 || (tiles256 >= 100 && K >= 8192)));
        // 80 .. 128 tiles of 256^2 (O / down projection of a 2 048-token prompt: 8 x 16): 128 x 256 tiles of the same kernel put one
        // block on 160 .. 256 CUs.  OMX_GEMM_ROWS128=0 keeps the old choice, =1 takes the 128-row tile for every shape (tests, A/B).
        const char* r128_env = getenv("OMX_GEMM_ROWS128");
        const int r128 = r128_env ? atoi(r128_env) : -1;
        if (ksplit == 1 && (r128 == 1 || (r128 != 0 && forced == 0 && ((tiles256 >= 80 && tiles256 <= 128) || (g_tile_hint == 128 && M >= 128 && N >= 256))))) {
            a.grid_m = (M + 127) / 128;
            a.grid_n = (N + 255) / 256;
            if (w5_takes(M, K, N, a.grid_m * a.grid_n)) gemm_nt_w4_kernel<false, false, 0, 128><<<a.grid_m * a.grid_n, 256, w5::smem_bytes(128), s>>>(a);
            else gemm_bf16_nt_256_kernel<16, false, false, 128><<<a.grid_m * a.grid_n, big::NT, big::smem_bytes(128), s>>>(a);
        } else if (use256) {
            a.grid_m = (M + 255) / 256;
            a.grid_n = (N + 255) / 256;
            a.ksplit = ksplit;
            if (ksplit > 1 && split_workspace_big(s, (size_t)tiles256 * ksplit * 256 * 256, (size_t)tiles256, &a.split_ws, &a.split_cnt)) return 1;
            const char* mf_env = getenv("OMX_GEMM_MFMA");
            if (ksplit == 1 && !mf_env && w5_takes(M, K, N, a.grid_m * a.grid_n)) {
                const int blocks = a.grid_m * a.grid_n;
#ifdef OMX_G5_DIAG
                const char* ve = getenv("OMX_GEMM_W4_VAR");
                const int var = ve ? atoi(ve) : 0;
                if (var == 1) gemm_nt_w4_kernel<false, false, 1><<<blocks, 256, w5::SMEM, s>>>(a);
                else if (var == 2) gemm_nt_w4_kernel<false, false, 2><<<blocks, 256, w5::SMEM, s>>>(a);
                else if (var == 3) gemm_nt_w4_kernel<false, false, 3><<<blocks, 256, w5::SMEM, s>>>(a);
                else if (var == 4) gemm_nt_w4_kernel<false, false, 4><<<blocks, 256, w5::SMEM, s>>>(a);
                else if (var == 5) gemm_nt_w4_kernel<false, false, 5><<<blocks, 256, w5::SMEM, s>>>(a);
                else if (var == 6) gemm_nt_w4_kernel<false, false, 6><<<blocks, 256, w5::SMEM, s>>>(a);
                else if (var == 7) gemm_nt_w4_kernel<false, false, 7><<<blocks, 256, w5::SMEM, s>>>(a);
                else if (var == 8) gemm_nt_w4_kernel<false, false, 8><<<blocks, 256, w5::SMEM, s>>>(a);
                else
#endif
                gemm_nt_w4_kernel<false, false><<<blocks, 256, w5::SMEM, s>>>(a);
            } else
#ifdef OMX_EXPERIMENTS
            if (ksplit == 1 && !mf_env && w4_takes(M, K, N)) {
                const int blocks = a.grid_m * a.grid_n;
#ifdef OMX_G4_DIAG
                const char* ve = getenv("OMX_GEMM_ASM_VAR");
                const int var = ve ? atoi(ve) : 0;
                if (var == 1) gemm_bf16_nt_asm_kernel<false, 1><<<blocks, 512, w4::SMEM, s>>>(a);
                else if (var == 2) gemm_bf16_nt_asm_kernel<false, 2><<<blocks, 512, w4::SMEM, s>>>(a);
                else if (var == 3) gemm_bf16_nt_asm_kernel<false, 3><<<blocks, 512, w4::SMEM, s>>>(a);
                else if (var == 4) gemm_bf16_nt_asm_kernel<false, 4><<<blocks, 512, w4::SMEM, s>>>(a);
                else
#endif
                gemm_bf16_nt_asm_kernel<false><<<blocks, 512, w4::SMEM, s>>>(a);
            } else
#endif
            if (mf_env && atoi(mf_env) == 32) gemm_bf16_nt_256_kernel<32><<<a.grid_m * a.grid_n * ksplit, big::NT, big::SMEM, s>>>(a);
            else gemm_bf16_nt_256_kernel<16><<<a.grid_m * a.grid_n * ksplit, big::NT, big::SMEM, s>>>(a);
        } else if (forced == 64 || (forced == 0 && a.grid_m * a.grid_n <= 128)) {
            // the 128^2 grid leaves CUs idle: 64^2 tiles with a deep prefetch ring (one block per CU: 8 stages; two: 4)
            a.grid_m = (M + 63) / 64;
            a.grid_n = (N + 63) / 64;
            const int blocks = a.grid_m * a.grid_n, ns = ring_splits(blocks, K / 64);
            if (ns > 1) {
                if (split_workspace(s, (size_t)blocks * ns * 64 * 64, (size_t)blocks, &a.split_ws, &a.split_cnt)) return 1;
                if (blocks * ns <= 256) gemm_bf16_nt_skinny_kernel<8><<<dim3(blocks, ns), skinny::NT, 8 * skinny::STAGE, s>>>(a);
                else gemm_bf16_nt_skinny_kernel<4><<<dim3(blocks, ns), skinny::NT, 4 * skinny::STAGE, s>>>(a);
            } else if (blocks <= 256) {
                gemm_bf16_nt_skinny_kernel<8><<<blocks, skinny::NT, 8 * skinny::STAGE, s>>>(a);
            } else {
                gemm_bf16_nt_skinny_kernel<4><<<blocks, skinny::NT, 4 * skinny::STAGE, s>>>(a);
            }
        } else {
            gemm_bf16_nt_kernel<false><<<a.grid_m * a.grid_n, NTHREADS, 4 * TILE_BYTES, s>>>(a);
        }
    } else {
        gemm_bf16_nt_generic_kernel<<<dim3((N + 63) / 64, (M + 63) / 64), NTHREADS, 0, s>>>(a);
    }
    OMX_LAUNCH_CHECK();
    return 0;
}

static int seg_tiles(const GemmSegs& g) {
    int t = 0;
    for (int i = 0; i < g.n_plain; ++i) t += (g.plain[i].cols + 255) / 256;
    return t + (g.half > 0 ? (g.half + 127) / 128 : 0);
}

bool gemm_segmented_supported(int M, int K, const GemmSegs& g) {
    if (M <= 0 || K <= 0 || K % BK != 0 || g.n_plain < 0 || g.n_plain > 3 || (g.n_plain == 0 && g.half <= 0)) return false;
    for (int i = 0; i < g.n_plain; ++i)
        if (g.plain[i].cols <= 0 || g.plain[i].cols % 4 != 0 || g.plain[i].ld % 4 != 0) return false;
    if (g.half < 0 || g.half % 4 != 0 || (g.half > 0 && g.ld_act % 4 != 0)) return false;
    return true;   // >= 160 tiles of 256^2: the deep-pipelined kernel; smaller problems: one grid of the 64^2 ring kernel
}

// Policy next to capability: is ONE segmented launch the faster schedule?  Measured on MI355X: yes for the 256^2 kernel and for
// plain segments at any size; a SwiGLU pair on the ring kernel only up to 256 rows (at 512 rows two 128^2 launches + the
// elementwise kernel are ~30 us faster per layer of a 4B-parameter encoder).
static bool rows_route(int M, int K, const GemmSegs& g) {
    const char* re = getenv("OMX_GEMV_ROWS");
    if ((re && re[0] == '0') || M > 8 || !gemv_rows_segmented_supported(M, K, g)) return false;
    int64_t rows = 2 * (int64_t)g.half;
    for (int i = 0; i < g.n_plain; ++i) rows += g.plain[i].cols;
    return rows * K >= (1 << 20);
}

bool gemv_rows_takes_norm(int M, int K, const GemmSegs& g) {
    const char* e = getenv("OMX_ROWS_NORM");   // 0: keep the separate RMSNorm launch (A/B, tests)
    // every block redoes the norm of all M staged rows: measured on the Qwen3-8B verify pass, 1 / 2 / 3 / 4 rows -9 / -7 / -5 / -7 %,
    // 5 rows flat, 8 rows +6 % -- taken up to 4 rows
    return !(e && e[0] == '0') && M <= 4 && K <= 4096 && rows_route(M, K, g);
}

bool gemm_segmented_preferred(int M, int K, const GemmSegs& g) {
    if (!gemm_segmented_supported(M, K, g)) return false;
    if (rows_route(M, K, g)) return true;   // a handful of rows: one weight-streaming launch (gemv_rows.hip)
    if (((M + 255) / 256) * seg_tiles(g) >= 160 || g.half == 0) return true;
    return M <= 256;
}

int launch_gemm_bf16_segmented(const bf16_t* x, int M, int K, const GemmSegs& segs, hipStream_t s) {
    OMX_REQUIRE(gemm_segmented_supported(M, K, segs), "segmented gemm: unsupported shape (M=%d K=%d, %d plain segments, half=%d)", M, K,
                segs.n_plain, segs.half);
    uintptr_t align = reinterpret_cast<uintptr_t>(x);
    for (int i = 0; i < segs.n_plain; ++i) {
        OMX_REQUIRE(segs.plain[i].w && segs.plain[i].out, "segmented gemm: null weight / output in segment %d", i);
        align |= reinterpret_cast<uintptr_t>(segs.plain[i].w);
    }
    if (segs.half > 0) {
        OMX_REQUIRE(segs.w_gate && segs.w_up && segs.out_act, "segmented gemm: null gate / up / activation pointer");
        align |= reinterpret_cast<uintptr_t>(segs.w_gate) | reinterpret_cast<uintptr_t>(segs.w_up);
    }
    OMX_REQUIRE((align & 15u) == 0, "segmented gemm: operands must be 16-byte aligned");
    if (!g_gemm_f16 && rows_route(M, K, segs)) return launch_gemv_rows_segmented(x, M, K, segs, s);
    OMX_REQUIRE(!segs.pre_norm_w, "segmented gemm: an in-launch RMSNorm exists on the weight-streaming route only (gemv_rows_takes_norm)");
    if (ensure_attr()) return 1;
    GemmArgs a = {};
    apply_experiment_switches(a);
    a.x = x; a.M = M; a.K = K;
    a.sg = segs;
    if (!g_gemm_f16 && ((M + 255) / 256) * seg_tiles(segs) < 160) {   // small problem, plain segments: one grid of 64^2 ring-kernel tiles
        int t64 = 0, n64 = 0;
        for (int i = 0; i < segs.n_plain; ++i) {
            a.sg.plain[i].tile0 = t64;
            t64 += (segs.plain[i].cols + 63) / 64;
            n64 += segs.plain[i].cols;
        }
        a.sg.act_tile0 = segs.half > 0 ? t64 : 0x7FFFFFFF;
        if (segs.half > 0) t64 += (segs.half + 31) / 32;
        n64 += 2 * segs.half;
        a.N = n64;
        a.grid_m = (M + 63) / 64;
        a.grid_n = t64;
        if (a.grid_m * a.grid_n <= 256) gemm_bf16_nt_skinny_kernel<8, true><<<a.grid_m * a.grid_n, skinny::NT, 8 * skinny::STAGE, s>>>(a);
        else gemm_bf16_nt_skinny_kernel<4, true><<<a.grid_m * a.grid_n, skinny::NT, 4 * skinny::STAGE, s>>>(a);
        OMX_LAUNCH_CHECK();
        return 0;
    }
    int t = 0, n = 0;
    for (int i = 0; i < segs.n_plain; ++i) {
        a.sg.plain[i].tile0 = t;
        t += (segs.plain[i].cols + 255) / 256;
        n += segs.plain[i].cols;
    }
    a.sg.act_tile0 = segs.half > 0 ? t : 0x7FFFFFFF;
    a.N = n + 2 * segs.half;
    a.grid_m = (M + 255) / 256;
    a.grid_n = seg_tiles(segs);
    int64_t w_rows_max = segs.half;
    for (int i = 0; i < segs.n_plain; ++i) w_rows_max = std::max<int64_t>(w_rows_max, segs.plain[i].cols);
    if (w5_takes(M, K, w_rows_max, a.grid_m * a.grid_n)) {
        if (g_gemm_f16) gemm_nt_w4_kernel<true, true><<<a.grid_m * a.grid_n, 256, w5::SMEM, s>>>(a);
        else gemm_nt_w4_kernel<true, false><<<a.grid_m * a.grid_n, 256, w5::SMEM, s>>>(a);
    } else
    if (g_gemm_f16) gemm_bf16_nt_256_kernel<16, true, false, 256, true><<<a.grid_m * a.grid_n, big::NT, big::SMEM, s>>>(a);
#ifdef OMX_EXPERIMENTS
    else if (w4_takes(M, K, w_rows_max)) gemm_bf16_nt_asm_kernel<true><<<a.grid_m * a.grid_n, 512, w4::SMEM, s>>>(a);
#endif
    else
    gemm_bf16_nt_256_kernel<16, true><<<a.grid_m * a.grid_n, big::NT, big::SMEM, s>>>(a);
    OMX_LAUNCH_CHECK();
    return 0;
}

int launch_gemm_bf16_segmented_grouped(const bf16_t* x, int max_rows, int K, const GemmSegs& segs, const GroupedDesc& g, int max_tiles,
                                       hipStream_t s) {
    OMX_REQUIRE(max_rows > 0 && K > 0 && K % BK == 0 && max_tiles > 0 && g.tile_expert && g.tile_m0 && g.seg_start && g.n_tiles,
                "grouped segmented gemm: bad arguments");
    OMX_REQUIRE(segs.n_plain >= 0 && segs.n_plain <= 3 && (segs.n_plain > 0 || segs.half > 0), "grouped segmented gemm: no segment");
    uintptr_t align = reinterpret_cast<uintptr_t>(x);
    for (int i = 0; i < segs.n_plain; ++i) {
        OMX_REQUIRE(segs.plain[i].w && segs.plain[i].out && segs.plain[i].cols > 0 && segs.plain[i].cols % 4 == 0 && segs.plain[i].ld % 4 == 0,
                    "grouped segmented gemm: bad plain segment %d", i);
        align |= reinterpret_cast<uintptr_t>(segs.plain[i].w);
    }
    if (segs.half > 0) {
        OMX_REQUIRE(segs.w_gate && segs.w_up && segs.out_act && segs.half % 4 == 0 && segs.ld_act % 4 == 0, "grouped segmented gemm: bad SwiGLU segment");
        align |= reinterpret_cast<uintptr_t>(segs.w_gate) | reinterpret_cast<uintptr_t>(segs.w_up);
    }
    OMX_REQUIRE((align & 15u) == 0 && (g.w_estride * 2) % 16 == 0, "grouped segmented gemm: operands must be 16-byte aligned");
    if (ensure_attr()) return 1;
    GemmArgs a = {};
    apply_experiment_switches(a);
    a.x = x; a.M = max_rows; a.K = K;
    a.g = g;
    a.sg = segs;
    int t = 0, n = 0;
    for (int i = 0; i < segs.n_plain; ++i) {
        a.sg.plain[i].tile0 = t;
        t += (segs.plain[i].cols + 255) / 256;
        n += segs.plain[i].cols;
    }
    a.sg.act_tile0 = segs.half > 0 ? t : 0x7FFFFFFF;
    a.N = n + 2 * segs.half;
    a.grid_m = max_tiles;
    a.grid_n = seg_tiles(segs);
    int64_t w_rows_g = segs.half;
    for (int i = 0; i < segs.n_plain; ++i) w_rows_g = std::max<int64_t>(w_rows_g, segs.plain[i].cols);
    if (w5_takes(max_rows, K, w_rows_g, a.grid_m * a.grid_n)) {   // (32-bit source offsets: inside the activations / inside ONE expert's matrices)
        if (g_gemm_f16) gemm_nt_w4_kernel<true, true><<<a.grid_m * a.grid_n, 256, w5::SMEM, s>>>(a);
        else gemm_nt_w4_kernel<true, false><<<a.grid_m * a.grid_n, 256, w5::SMEM, s>>>(a);
    } else
    if (g_gemm_f16) gemm_bf16_nt_256_kernel<16, true, false, 256, true><<<a.grid_m * a.grid_n, big::NT, big::SMEM, s>>>(a);   // (a float16 model's prompt)
    else
    gemm_bf16_nt_256_kernel<16, true><<<a.grid_m * a.grid_n, big::NT, big::SMEM, s>>>(a);
    OMX_LAUNCH_CHECK();
    return 0;
}

bool conv3x3_implicit_supported(int H, int W, int C, int Cout) {
    if (H <= 0 || W <= 0 || C < 64 || C % 64 != 0 || ((C / 64) & (C / 64 - 1)) != 0 || Cout <= 0) return false;
    const long long M = (long long)H * W;
    return M <= 0x7FFFFFFF && ((M + 255) / 256) * ((Cout + 255) / 256) >= 160 && (long long)(H + 2) * (W + 2) * C <= 0x7FFFFFFFLL;
}

int launch_conv3x3_implicit(bf16_t* out, const bf16_t* padded, const bf16_t* w, const bf16_t* bias, const bf16_t* resid, int H, int W,
                            int C, int Cout, hipStream_t s) {
    OMX_REQUIRE(conv3x3_implicit_supported(H, W, C, Cout), "implicit conv: unsupported shape %dx%d, %d -> %d channels", H, W, C, Cout);
    OMX_REQUIRE(out && padded && w && ((reinterpret_cast<uintptr_t>(padded) | reinterpret_cast<uintptr_t>(w)) & 15u) == 0,
                "implicit conv: null or misaligned operand");
    if (ensure_attr()) return 1;
    GemmArgs a = {};
    apply_experiment_switches(a);
    a.x = padded; a.M = H * W; a.K = 9 * C; a.N = Cout;
    a.sg.n_plain = 1;
    a.sg.plain[0] = {w, bias, out, Cout, Cout, 0};
    a.sg.act_tile0 = 0x7FFFFFFF;
    a.sg.im_C = C; a.sg.im_W = W; a.sg.im_resid = resid;
    int sh = 0;
    while ((64 << sh) < C) ++sh;
    a.sg.im_sh = sh;
    a.grid_m = (a.M + 255) / 256;
    a.grid_n = (Cout + 255) / 256;
    gemm_bf16_nt_256_kernel<16, true, true><<<a.grid_m * a.grid_n, big::NT, big::SMEM, s>>>(a);
    OMX_LAUNCH_CHECK();
    return 0;
}

static GemmSegs swiglu_segs(bf16_t* out_plain, int ld_plain, bf16_t* out_act, int ld_act, const bf16_t* w, int n_plain, int half, int K) {
    GemmSegs g = {};
    if (n_plain > 0) {
        g.n_plain = 1;
        g.plain[0] = {w, nullptr, out_plain, n_plain, ld_plain, 0};
    }
    g.w_gate = w + (size_t)n_plain * K;
    g.w_up = g.w_gate + (size_t)half * K;
    g.out_act = out_act; g.half = half; g.ld_act = ld_act; g.act_mode = 0;
    return g;
}

bool gemm_swiglu_supported(int M, int n_plain, int half, int K) {
    if (n_plain < 0 || half <= 0 || n_plain % 4 != 0) return false;
    return gemm_segmented_supported(M, K, swiglu_segs(nullptr, 4, nullptr, 4, nullptr, n_plain, half, K));
}

bool gemm_swiglu_preferred(int M, int n_plain, int half, int K) {
    if (n_plain < 0 || half <= 0 || n_plain % 4 != 0) return false;
    return gemm_segmented_preferred(M, K, swiglu_segs(nullptr, 4, nullptr, 4, nullptr, n_plain, half, K));
}

int launch_gemm_bf16_swiglu(bf16_t* out_plain, int ld_plain, bf16_t* out_act, int ld_act, const bf16_t* x, const bf16_t* w, int M,
                            int n_plain, int half, int K, hipStream_t s) {
    return launch_gemm_bf16_segmented(x, M, K, swiglu_segs(out_plain, n_plain > 0 ? ld_plain : 4, out_act, ld_act, w, n_plain, half, K), s);
}

int launch_gemm_bf16_ex(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, const bf16_t* resid, int M,
                        int N, int K, hipStream_t s) {
    return launch_gemm_impl(out, x, w, bias, resid, nullptr, M, N, K, s);
}

int launch_gemm_bf16_bias_relu(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, int M, int N, int K,
                               hipStream_t s) {
    return launch_gemm_impl(out, x, w, bias, nullptr, nullptr, M, N, K, s, 1);
}

int launch_gemm_bf16_gated(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* resid, const bf16_t* gate, int M,
                           int N, int K, hipStream_t s) {
    OMX_REQUIRE(resid && gate, "gated gemm: residual and gate are required");
    return launch_gemm_impl(out, x, w, nullptr, resid, gate, M, N, K, s);
}

int launch_gemm_bf16(bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* bias, int M, int N, int K,
                     hipStream_t s) {
    return launch_gemm_bf16_ex(out, x, w, bias, nullptr, M, N, K, s);
}

int launch_gemm_bf16_grouped(bf16_t* out, const bf16_t* x, const bf16_t* w, int max_rows, int N, int K,
                             const GroupedDesc& g, int max_tiles, hipStream_t s) {
    OMX_REQUIRE(max_rows > 0 && N > 0 && K > 0 && max_tiles > 0, "grouped gemm: bad shape");
    OMX_REQUIRE(K % BK == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15u) == 0,
                "grouped gemm: K=%d must be a multiple of %d and operands 16-byte aligned", K, BK);
    GemmArgs a = {x, w, nullptr, nullptr, nullptr, out, max_rows, N, K, max_tiles, (N + BN - 1) / BN, g, 0};
    apply_experiment_switches(a);
    if (ensure_attr()) return 1;
    gemm_bf16_nt_kernel<true><<<max_tiles * a.grid_n, NTHREADS, 4 * TILE_BYTES, s>>>(a);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // namespace omx
