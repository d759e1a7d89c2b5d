// float32 GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact f32 products and accumulation, 157 TF
// chip peak = 1/16 of the bf16 rate).  The one model on the path that the reference runs in float32 is Paraformer
// (funasr-mlx/src/paraformer.rs:496-532, 560-570, 618-634, 981-1053: f32 weights converted from PyTorch, f32 activations):
// 0.17 TFLOP for 30 s of audio, GEMMs of 200-500 rows -- so these kernels are built for SMALL problems (64 x 64 tiles so that a
// [501, 512] output still gives 64 blocks, split-K when even that leaves the chip idle), not for a roofline number.
//   out[b][m, n] = alpha * sum_k A[b][m, k] * B[b](k, n) (+ bias[n]) (relu) (+ resid[m, n])
//   B(k, n) = B[n * ldb + k]   ("NT": an nn::Linear weight [N, K], mlx-rs/src/nn/linear.rs:87-92)    or
//           = B[k * ldb + n]   ("NN": P . V of the explicit attention, paraformer.rs:513-515)
// Three kernels, newest last:
//   gemm_f32_kernel        64 x 64 x 64 per block, 4 waves, one 32 x 32 accumulator tile per wave; operands global -> registers -> LDS rows
//                          padded to 65 floats (4-byte fragment reads and staging writes conflict-free).  NN and NT, any alignment.
//   gemm_f32_small_kernel  32 x 32 tiles of 16 x 16 x 4 MFMAs (round 5) for grids that do not fill the chip.  NN and NT, any alignment.
//   gemm_f32_pipe_kernel   (round 6) what every aligned NT product runs on: the same 64 x 64 tile software-pipelined -- see its own comment.
#include "gemm.hpp"
#include "workspace.hpp"

namespace omx {

namespace {

constexpr int TM = 64, TN = 64, TK = 64, LDS_LD = TK + 1;
using f32x16 = __attribute__((ext_vector_type(16))) float;

struct F32Args {
    const float* a; const float* b; const float* bias; const float* resid; float* out; float* partial;
    int M, N, K;
    int64_t lda, ldb, ldc, ldr, sa, sb, sc;   // leading dimensions and per-batch strides (elements)
    int batch, splits, relu, b_nn;
    float alpha;
};

// stage one [64 rows x 64 k] operand tile held k-contiguous in memory (A, or B in NT form) into registers: thread t covers
// row t / 16 (+ 16 i), k quad t % 16
__device__ __forceinline__ void load_rows(f32x4 (&r)[4], const float* base, int64_t ld, int row0, int rows, int k0, int kend) {
    const int rq = threadIdx.x >> 4, kq = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + rq + 16 * i, k = k0 + kq;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < rows) {
            const float* p = base + (int64_t)row * ld + k;
            if (k + 3 < kend && ((reinterpret_cast<uintptr_t>(p) & 15u) == 0)) v = *reinterpret_cast<const f32x4*>(p);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (k + e < kend) ? p[e] : 0.f;
            }
        }
        r[i] = v;
    }
}
__device__ __forceinline__ void store_rows(float* lds, const f32x4 (&r)[4]) {
    const int rq = threadIdx.x >> 4, kq = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[(rq + 16 * i) * LDS_LD + kq + e] = r[i][e];
}
// B in NN form: the tile is [64 k x 64 n] with n contiguous in memory; staged TRANSPOSED into the same [n][k] LDS layout
__device__ __forceinline__ void load_cols(f32x4 (&r)[4], const float* base, int64_t ld, int n0, int ncols, int k0, int kend) {
    const int kr = threadIdx.x >> 4, nq = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + kr + 16 * i, n = n0 + nq;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < kend) {
            const float* p = base + (int64_t)k * ld + n;
            if (n + 3 < ncols && ((reinterpret_cast<uintptr_t>(p) & 15u) == 0)) v = *reinterpret_cast<const f32x4*>(p);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (n + e < ncols) ? p[e] : 0.f;
            }
        }
        r[i] = v;
    }
}
__device__ __forceinline__ void store_cols(float* lds, const f32x4 (&r)[4]) {
    const int kr = threadIdx.x >> 4, nq = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[(nq + e) * LDS_LD + kr + 16 * i] = r[i][e];
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const F32Args g) {
    __shared__ float As[TM * LDS_LD];
    __shared__ float Bs[TN * LDS_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;                    // the wave's 32 x 32 quadrant
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int bz = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
    const float* A = g.a + (int64_t)bz * g.sa;
    const float* B = g.b + (int64_t)bz * g.sb;
    // this split's K range, in whole K tiles
    const int ktiles = (g.K + TK - 1) / TK, per = (ktiles + g.splits - 1) / g.splits;
    const int kbeg = split * per * TK, kend = min(g.K, (split + 1) * per * TK);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // (measured: a second K tile in flight in registers -- two alternating register sets, 127 VGPRs -- made the 30 s pass 11.0 ms
    //  instead of 10.1 ms: the kernel is not waiting for memory, its tiles are chains of dependent 64-cycle MFMAs)
    f32x4 ra[4], rb[4];
    if (kbeg < kend) {
        load_rows(ra, A, g.lda, m0, g.M, kbeg, kend);
        if (g.b_nn) load_cols(rb, B, g.ldb, n0, g.N, kbeg, kend);
        else load_rows(rb, B, g.ldb, n0, g.N, kbeg, kend);
    }
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        __syncthreads();                                        // the previous tile's fragment reads are done
        store_rows(As, ra);
        if (g.b_nn) store_cols(Bs, rb); else store_rows(Bs, rb);
        __syncthreads();
        if (k0 + TK < kend) {                                   // next tile in flight under this tile's matrix work
            load_rows(ra, A, g.lda, m0, g.M, k0 + TK, kend);
            if (g.b_nn) load_cols(rb, B, g.ldb, n0, g.N, k0 + TK, kend);
            else load_rows(rb, B, g.ldb, n0, g.N, k0 + TK, kend);
        }
        const float* ap = As + (wm * 32 + (lane & 31)) * LDS_LD + (lane >> 5);
        const float* bp = Bs + (wn * 32 + (lane & 31)) * LDS_LD + (lane >> 5);
        // all 64 fragment reads of the tile go out as one burst ahead of the 32 dependent MFMAs (64 cycles each): with an
        // 8-step unroll every group of 8 paid the LDS latency again
        float fa[TK / 2], fb[TK / 2];
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) { fa[kk / 2] = ap[kk]; fb[kk / 2] = bp[kk]; }
#pragma unroll
        for (int kk = 0; kk < TK / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb[kk], acc, 0, 0, 0);
    }
    // C/D map of the 32 x 32 forms: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= g.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= g.M) continue;
        if (g.splits > 1) {
            g.partial[(((int64_t)bz * g.splits + split) * g.M + row) * g.N + col] = acc[r];
        } else {
            float v = acc[r] * g.alpha;
            if (g.bias) v += g.bias[col];
            if (g.relu) v = fmaxf(v, 0.f);
            if (g.resid) v += g.resid[(int64_t)row * g.ldr + col];
            g.out[(int64_t)bz * g.sc + (int64_t)row * g.ldc + col] = v;
        }
    }
}

// ---- 32 x 32 tiles, four waves of 16 x 16 on v_mfma_f32_16x16x4_f32 (round 5) -- for the problems above whose 64 x 64 tile grid does not
//      fill the chip several times over: four times the blocks without a K split (no partial tiles written and re-read, no reduce launch),
//      16 KiB of LDS and few registers, so several blocks share a CU and one block's staging hides behind another's MFMA chain.
//      Same operands, same epilogue, K ascending in fours per output (the 32x32x2 form sums in twos: results differ in the last f32 bits).
constexpr int SM = 32, SN = 32, S_LD = TK + 4;      // rows of 68 floats: 16-byte aligned for the float4 staging stores, fragment reads 2-way at worst

__device__ __forceinline__ void small_load_rows(f32x4 (&r)[2], const float* base, int64_t ld, int row0, int rows, int k0, int kend) {
    const int rq = threadIdx.x >> 4, kq = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = row0 + rq + 16 * i, k = k0 + kq;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < rows) {
            const float* p = base + (int64_t)row * ld + k;
            if (k + 3 < kend && ((reinterpret_cast<uintptr_t>(p) & 15u) == 0)) v = *reinterpret_cast<const f32x4*>(p);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (k + e < kend) ? p[e] : 0.f;
            }
        }
        r[i] = v;
    }
}
__device__ __forceinline__ void small_store_rows(float* lds, const f32x4 (&r)[2]) {
    const int rq = threadIdx.x >> 4, kq = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(lds + (rq + 16 * i) * S_LD + kq) = r[i];
}
// B in NN form: [64 k x 32 n], n contiguous in memory; thread t covers k = t / 8 (+ 32 i), n quad t % 8; staged transposed into [n][k]
__device__ __forceinline__ void small_load_cols(f32x4 (&r)[2], const float* base, int64_t ld, int n0, int ncols, int k0, int kend) {
    const int kr = threadIdx.x >> 3, nq = (threadIdx.x & 7) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = k0 + kr + 32 * i, n = n0 + nq;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < kend) {
            const float* p = base + (int64_t)k * ld + n;
            if (n + 3 < ncols && ((reinterpret_cast<uintptr_t>(p) & 15u) == 0)) v = *reinterpret_cast<const f32x4*>(p);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (n + e < ncols) ? p[e] : 0.f;
            }
        }
        r[i] = v;
    }
}
__device__ __forceinline__ void small_store_cols(float* lds, const f32x4 (&r)[2]) {
    const int kr = threadIdx.x >> 3, nq = (threadIdx.x & 7) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[(nq + e) * S_LD + kr + 32 * i] = r[i][e];
}

__global__ __launch_bounds__(256) void gemm_f32_small_kernel(const F32Args g) {
    __shared__ __attribute__((aligned(16))) float As[SM * S_LD];
    __shared__ __attribute__((aligned(16))) float Bs[SN * S_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;                    // the wave's 16 x 16 quadrant
    const int m0 = blockIdx.y * SM, n0 = blockIdx.x * SN;
    const int bz = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
    const float* A = g.a + (int64_t)bz * g.sa;
    const float* B = g.b + (int64_t)bz * g.sb;
    const int ktiles = (g.K + TK - 1) / TK, per = (ktiles + g.splits - 1) / g.splits;
    const int kbeg = split * per * TK, kend = min(g.K, (split + 1) * per * TK);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 ra[2], rb[2];
    if (kbeg < kend) {
        small_load_rows(ra, A, g.lda, m0, g.M, kbeg, kend);
        if (g.b_nn) small_load_cols(rb, B, g.ldb, n0, g.N, kbeg, kend);
        else small_load_rows(rb, B, g.ldb, n0, g.N, kbeg, kend);
    }
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        __syncthreads();
        small_store_rows(As, ra);
        if (g.b_nn) small_store_cols(Bs, rb); else small_store_rows(Bs, rb);
        __syncthreads();
        if (k0 + TK < kend) {
            small_load_rows(ra, A, g.lda, m0, g.M, k0 + TK, kend);
            if (g.b_nn) small_load_cols(rb, B, g.ldb, n0, g.N, k0 + TK, kend);
            else small_load_rows(rb, B, g.ldb, n0, g.N, k0 + TK, kend);
        }
        // A fragment: row lane & 15, k = lane >> 4 of each group of four; B likewise (column lane & 15)
        const float* ap = As + (wm * 16 + (lane & 15)) * S_LD + (lane >> 4);
        const float* bp = Bs + (wn * 16 + (lane & 15)) * S_LD + (lane >> 4);
        float fa[TK / 4], fb[TK / 4];
#pragma unroll
        for (int kk = 0; kk < TK / 4; ++kk) { fa[kk] = ap[4 * kk]; fb[kk] = bp[4 * kk]; }
#pragma unroll
        for (int kk = 0; kk < TK / 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[kk], fb[kk], acc, 0, 0, 0);
    }
    // C/D map of the 16 x 16 form: col = lane & 15, row = 4 (lane >> 4) + reg
    const int col = n0 + wn * 16 + (lane & 15);
    if (col >= g.N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 16 + 4 * (lane >> 4) + r;
        if (row >= g.M) continue;
        if (g.splits > 1) {
            g.partial[(((int64_t)bz * g.splits + split) * g.M + row) * g.N + col] = acc[r];
        } else {
            float v = acc[r] * g.alpha;
            if (g.bias) v += g.bias[col];
            if (g.relu) v = fmaxf(v, 0.f);
            if (g.resid) v += g.resid[(int64_t)row * g.ldr + col];
            g.out[(int64_t)bz * g.sc + (int64_t)row * g.ldc + col] = v;
        }
    }
}

// ---- 64 x 64 tiles, one wave per SIMD, software-pipelined (round 6).  What the two kernels above lose is not matrix time but everything
//      around it: a load under a condition compiles to a branch with a full vmcnt(0) wait behind it (four serial memory round trips per K
//      tile), the LDS traffic is 32 (16) four-byte writes and 64 (32) four-byte reads per thread per tile, and the matrix pipe idles
//      while the block stages and reads.  Here
//        * the terms of a dot product are taken in a permuted order -- lane (n, q) of v_mfma_f32_16x16x4_f32 step j, MFMA e multiplies
//          k = 16 j + 4 q + e -- so a fragment is ONE 16-byte LDS read per four MFMAs, and the staging writes are 16-byte too;
//        * staging is branch-free (rows / k quads past the end re-read the last one; a select zeroes the k quads at the LDS write) and runs
//          PR tiles ahead in a register ring, every request unconditional so that the compiler's vmcnt bookkeeping stays exact;
//        * a wave owns a 32 x 32 quadrant as 2 x 2 fragments (four independent accumulator chains, 64 MFMAs per K tile) and hides its own
//          memory work in the issue slots between its own MFMAs: while tile t is multiplied out of registers, the fragments of tile t + 1
//          are read from one LDS buffer and tile t + 2 is written into the other -- one barrier per tile.
//      Measured dead ends of the same round (EXPERIMENTS R6-6): fragments straight from global memory with the same permutation (16 rows
//      per 16-lane group is the worst case for the texture addresser); two waves per SIMD, in phase or half a tile out of phase ("ping-
//      pong"): a wave streaming MFMAs holds its SIMD's issue stage, the other wave's LDS instructions wait for the whole stream
//      (cycle stamps: 1180 cycles to get 8 ds_read_b128 through beside an MFMA stream, 124 without) -- only a wave's OWN instructions
//      slot in between its MFMAs.
#ifndef OMX_PIPE_MFMA32
#define OMX_PIPE_MFMA32 1          // the 32 x 32 x 2 MFMA (64 cycles each: a memory instruction fits whole into the gap behind one); 0: 2 x 2 of 16 x 16 x 4
#endif
constexpr int PK = 64, P_LD = PK + 4, PR = 3;                   // rows of 68 floats: 16-byte aligned
constexpr size_t P_LDS_BYTES = (size_t)2 * 2 * 64 * P_LD * sizeof(float);

template <bool FULLK>
__global__ __launch_bounds__(256) void gemm_f32_pipe_kernel(const F32Args g) {
    extern __shared__ __attribute__((aligned(16))) float pipe_lds[];
    float* As = pipe_lds;                                       // [2][64][P_LD]
    float* Bs = pipe_lds + 2 * 64 * P_LD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, n = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int bz = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
    const int ktiles = (g.K + PK - 1) / PK, per = (ktiles + g.splits - 1) / g.splits;
    const int kbeg = split * per * PK, kend = min(g.K, (split + 1) * per * PK);
    const int nt = kbeg < kend ? (kend - kbeg + PK - 1) / PK : 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#if OMX_PIPE_MFMA32
    (void)n; (void)q; (void)zero4;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#else
    f32x4 acc00 = zero4, acc01 = zero4, acc10 = zero4, acc11 = zero4;
#endif
    // staging: thread (rq, kq) covers rows rq + 16 i of both operands' tiles, k quad kq.  Buffer loads: the per-thread part of the address is
    // eight 32-bit offsets computed once, the tile's k offset is a scalar, and a row past the end of the matrix is out of the descriptor's
    // range and reads as zero -- no address arithmetic and no clamping per tile.  FULLK (K a multiple of the tile): no k tail either; otherwise
    // a k quad past the end re-reads the last quad and a select zeroes it at the LDS write.
    const int rq = threadIdx.x >> 4, kq = (threadIdx.x & 15) * 4;
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a + (int64_t)bz * g.sa), 0,
                                                                            (int)((((int64_t)g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.b + (int64_t)bz * g.sb), 0,
                                                                            (int)((((int64_t)g.N - 1) * g.ldb + g.K) * 4), 0x00020000);
    int ao[4], bo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ao[i] = (int)(((int64_t)(m0 + rq + 16 * i) * g.lda + kq) * 4);
        bo[i] = (int)(((int64_t)(n0 + rq + 16 * i) * g.ldb + kq) * 4);
    }
    u32x4 ra[PR][4], rb[PR][4];
#if OMX_PIPE_MFMA32
    f32x4 fa[2][8], fb[2][8];                                   // [register set][k octet]
#else
    f32x4 fa[2][4][2], fb[2][4][2];                             // [register set][k step][fragment row block]
#endif
#define OMX_PIPE_REQUEST(slot, tile)                                                                   \
    {                                                                                                  \
        const int k0_ = kbeg + min((tile), nt - 1) * PK;                                               \
        const int back_ = FULLK ? 0 : 4 * max(k0_ + kq + 4 - kend, 0);      /* bytes to step back onto the last whole quad */ \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            ra[slot][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, ao[i] - back_, 4 * k0_, 0)); \
            rb[slot][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, bo[i] - back_, 4 * k0_, 0)); \
        }                                                                                              \
    }
// (the zeroing select sits at the first USE of the registers, so that the requests stay in flight across the tiles in between)
#define OMX_PIPE_STAGE(slot, tile, buf)                                                                \
    {                                                                                                  \
        const bool in_ = FULLK || kbeg + min((tile), nt - 1) * PK + kq < kend;                         \
        const u32x4 zero_ = {0u, 0u, 0u, 0u};                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            *reinterpret_cast<u32x4*>(As + (buf) * 64 * P_LD + (rq + 16 * i) * P_LD + kq) = in_ ? ra[slot][i] : zero_; \
            *reinterpret_cast<u32x4*>(Bs + (buf) * 64 * P_LD + (rq + 16 * i) * P_LD + kq) = in_ ? rb[slot][i] : zero_; \
        }                                                                                              \
    }
#if OMX_PIPE_MFMA32
// lane (r, hh) of v_mfma_f32_32x32x2_f32 reads row r, k = 8 j + 4 hh .. + 3: MFMA e of fragment j multiplies the k pair {8 j + e, 8 j + 4 + e}
#define OMX_PIPE_FRAGMENTS(set, buf)                                                                   \
    {                                                                                                  \
        const float* at_ = As + (buf) * 64 * P_LD + (wm * 32 + (lane & 31)) * P_LD + 4 * (lane >> 5);  \
        const float* bt_ = Bs + (buf) * 64 * P_LD + (wn * 32 + (lane & 31)) * P_LD + 4 * (lane >> 5);  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                \
            fa[set][j] = *reinterpret_cast<const f32x4*>(at_ + 8 * j);                                 \
            fb[set][j] = *reinterpret_cast<const f32x4*>(bt_ + 8 * j);                                 \
        }                                                                                              \
    }
#define OMX_PIPE_MULTIPLY(set)                                                                         \
    {                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                  \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                              \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][j][e], fb[set][j][e], acc, 0, 0, 0); \
    }
// the iteration's instruction mix, spelled out for the scheduler: 16 fragment reads, then 8 x (staging write + request), each behind one MFMA
#define OMX_PIPE_INTERLEAVE()                                                                          \
    {                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
        }                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                         \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                         \
        }                                                                                              \
    }
#else
#define OMX_PIPE_FRAGMENTS(set, buf)                                                                   \
    {                                                                                                  \
        const float* at_ = As + (buf) * 64 * P_LD + (wm * 32 + n) * P_LD + 4 * q;                      \
        const float* bt_ = Bs + (buf) * 64 * P_LD + (wn * 32 + n) * P_LD + 4 * q;                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                            \
                fa[set][j][i] = *reinterpret_cast<const f32x4*>(at_ + 16 * i * P_LD + 16 * j);         \
                fb[set][j][i] = *reinterpret_cast<const f32x4*>(bt_ + 16 * i * P_LD + 16 * j);         \
            }                                                                                          \
    }
#define OMX_PIPE_MULTIPLY(set)                                                                         \
    {                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
                acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][j][0][e], fb[set][j][0][e], acc00, 0, 0, 0); \
                acc01 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][j][0][e], fb[set][j][1][e], acc01, 0, 0, 0); \
                acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][j][1][e], fb[set][j][0][e], acc10, 0, 0, 0); \
                acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][j][1][e], fb[set][j][1][e], acc11, 0, 0, 0); \
            }                                                                                          \
    }
// the iteration's instruction mix, spelled out for the scheduler: 16 fragment reads, 8 staging writes and 8 requests, each behind one MFMA
#define OMX_PIPE_INTERLEAVE()                                                                          \
    {                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
        }                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                         \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                         \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                         \
        }                                                                                              \
    }
#endif
#define OMX_PIPE_BOUNDARY()                                                                            \
    {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        __syncthreads();                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
    if (nt > 0) {
        // ring: tile x waits in slot x % PR from its request until it is staged.  Every request is unconditional -- one past the last tile
        // re-reads the last tile (cache hits, staged into a buffer nobody reads again).
#pragma unroll
        for (int x = 0; x < PR; ++x) OMX_PIPE_REQUEST(x, x)
        OMX_PIPE_STAGE(0, 0, 0)
        OMX_PIPE_REQUEST(0, PR)
        OMX_PIPE_BOUNDARY()
        OMX_PIPE_FRAGMENTS(0, 0)
        OMX_PIPE_STAGE(1 % PR, 1, 1)
        OMX_PIPE_REQUEST(1 % PR, 1 + PR)
        OMX_PIPE_BOUNDARY()
        // iteration t: the MFMAs of tile t run on fragments read an iteration ago; between them, the fragments of tile t + 1 come out of
        // buffer (t + 1) & 1 and tile t + 2 goes into buffer t & 1 (whose fragments everyone read before the last boundary).
        // Unrolled by 2 PR so that the slot (mod PR) and the buffer / register set (mod 2) are both compile-time.
        int t = 0;
        for (; t + 2 * PR <= nt; t += 2 * PR) {
#pragma unroll
            for (int u = 0; u < 2 * PR; ++u) {
                OMX_PIPE_FRAGMENTS((u + 1) & 1, (u + 1) & 1)
                OMX_PIPE_STAGE((u + 2) % PR, t + u + 2, u & 1)
                OMX_PIPE_REQUEST((u + 2) % PR, t + u + 2 + PR)
                OMX_PIPE_MULTIPLY(u & 1)
                OMX_PIPE_INTERLEAVE()
                OMX_PIPE_BOUNDARY()
            }
        }
#pragma unroll
        for (int u = 0; u < 2 * PR - 1; ++u)
            if (t + u < nt) {
                OMX_PIPE_FRAGMENTS((u + 1) & 1, (u + 1) & 1)
                OMX_PIPE_STAGE((u + 2) % PR, t + u + 2, u & 1)
                OMX_PIPE_REQUEST((u + 2) % PR, t + u + 2 + PR)
                OMX_PIPE_MULTIPLY(u & 1)
                OMX_PIPE_INTERLEAVE()
                OMX_PIPE_BOUNDARY()
            }
    }
#undef OMX_PIPE_REQUEST
#undef OMX_PIPE_STAGE
#undef OMX_PIPE_FRAGMENTS
#undef OMX_PIPE_MULTIPLY
#undef OMX_PIPE_INTERLEAVE
#undef OMX_PIPE_BOUNDARY
    // The tile goes out through LDS (over the first staging buffer: every fragment read is behind the last boundary): a lane of the MFMA result
    // holds one column of four rows, so storing from the accumulators writes 64-byte pieces of sixteen different rows per instruction; from
    // LDS a wave stores four whole 256-byte rows per instruction, and bias / residual are read 16 bytes at a time.
    // D of the 16 x 16 form: col = lane & 15, row = 4 (lane >> 4) + reg
    float* C = pipe_lds;                                        // [64][P_LD]
#if OMX_PIPE_MFMA32
    // D of the 32 x 32 form: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) C[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * P_LD + wn * 32 + (lane & 31)] = acc[r];
#else
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const f32x4 acc = f == 0 ? acc00 : f == 1 ? acc01 : f == 2 ? acc10 : acc11;
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(wm * 32 + 16 * (f >> 1) + 4 * q + r) * P_LD + wn * 32 + 16 * (f & 1) + n] = acc[r];
    }
#endif
    __syncthreads();
    const bool vec = (g.N & 3) == 0 && (g.splits > 1 ? (reinterpret_cast<uintptr_t>(g.partial) & 15u) == 0
                                                     : (g.ldc & 3) == 0 && (g.sc & 3) == 0 && (!g.resid || (g.ldr & 3) == 0) &&
                                                           ((reinterpret_cast<uintptr_t>(g.bias) | reinterpret_cast<uintptr_t>(g.resid)) & 15u) == 0);
    const int col = n0 + kq;                                    // this thread's four columns, rows rq + 16 i
    if (col >= g.N) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + rq + 16 * i;
        if (row >= g.M) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(C + (rq + 16 * i) * P_LD + kq);
        if (g.splits > 1) {
            float* dst = g.partial + (((int64_t)bz * g.splits + split) * g.M + row) * g.N + col;
            if (vec) *reinterpret_cast<f32x4*>(dst) = v;
            else
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < g.N) dst[e] = v[e];
            continue;
        }
        float* dst = g.out + (int64_t)bz * g.sc + (int64_t)row * g.ldc + col;
        if (vec && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
            v *= g.alpha;
            if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col);
            if (g.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (g.resid) v += *reinterpret_cast<const f32x4*>(g.resid + (int64_t)row * g.ldr + col);
            *reinterpret_cast<f32x4*>(dst) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (col + e >= g.N) break;
                float x = v[e] * g.alpha;
                if (g.bias) x += g.bias[col + e];
                if (g.relu) x = fmaxf(x, 0.f);
                if (g.resid) x += g.resid[(int64_t)row * g.ldr + col + e];
                dst[e] = x;
            }
        }
    }
}

// sums the split-K partials in split order (deterministic) and applies the epilogue.  (Measured alternative: the last block to
// arrive at a per-tile counter reduces in the same launch -- write-through 4-byte partial stores and the serial re-read made
// the 30 s pass 14.4 ms instead of 10.1 ms; a second launch of 256 K elements is cheaper.)
__global__ __launch_bounds__(256) void gemm_f32_reduce_kernel(const F32Args g) {
    const int64_t total = (int64_t)g.batch * g.M * g.N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % g.N), row = (int)((i / g.N) % g.M), bz = (int)(i / ((int64_t)g.M * g.N));
        float v = 0.f;
        for (int s = 0; s < g.splits; ++s) v += g.partial[(((int64_t)bz * g.splits + s) * g.M + row) * g.N + col];
        v *= g.alpha;
        if (g.bias) v += g.bias[col];
        if (g.relu) v = fmaxf(v, 0.f);
        if (g.resid) v += g.resid[(int64_t)row * g.ldr + col];
        g.out[(int64_t)bz * g.sc + (int64_t)row * g.ldc + col] = v;
    }
}

}  // namespace

int launch_gemm_f32(const GemmF32& p, hipStream_t s) {
    OMX_REQUIRE(p.a && p.b && p.out && p.M > 0 && p.N > 0 && p.K > 0 && p.batch >= 1, "gemm_f32: bad arguments (M=%d N=%d K=%d)", p.M, p.N, p.K);
    F32Args g = {p.a, p.b, p.bias, p.resid, p.out, nullptr, p.M, p.N, p.K, p.lda, p.ldb, p.ldc, p.ldr ? p.ldr : p.ldc, p.sa, p.sb, p.sc,
                 p.batch, 1, p.relu, p.b_nn, p.alpha};
    const bool defer = p.defer_partial && p.defer_splits;      // the caller applies the epilogue (and whatever follows) in its own launch
    if (defer) {
        OMX_REQUIRE(p.batch == 1 && p.ldc == p.N, "gemm_f32: a deferred epilogue takes one dense [M, N] product");
        g.bias = nullptr; g.resid = nullptr; g.relu = 0; g.alpha = 1.0f;
    }
    const int gx = (p.N + TN - 1) / TN, gy = (p.M + TM - 1) / TM, tiles = gx * gy * p.batch, ktiles = (p.K + TK - 1) / TK;
    // split K while the tile grid leaves most of the chip idle and every split keeps >= 2 K tiles: a wave's tile is a chain of
    // K / 2 dependent 64-cycle MFMAs (7.8 us at K = 512) whatever the tile shape, so for 64-tile outputs (out_proj, ffn_down,
    // P . V) the only way to use the other 192 CUs is to cut K.  (Swept on the 30 s pass: a floor of 4 / 8 K tiles per split 10.8 /
    // 11.9 ms, no splitting 15.4 ms, a block budget of 512 / 1024 instead of 256: 10.3 / 10.5 ms -- this rule: 10.2 ms.)
    // (round 5) a 64 x 64 grid that does not cover the chip at least `kSmallBelow / 256` times over goes to the 32 x 32 kernel: four times the
    // blocks, several of them per CU.  OMX_F32_SMALL=0 keeps the old choice, =1 takes the small tiles for every shape (tests, A/B).
    // (round 6) the one-barrier kernel takes every NT product; OMX_F32_PIPE=0 keeps the older staged kernels (A/B, tests)
    static const int pipe_mode = [] { const char* e = getenv("OMX_F32_PIPE"); return e ? atoi(e) : 1; }();
    auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15u) == 0; };
    const bool small_enough = (int64_t)p.M * p.lda < (1ll << 28) && (int64_t)p.N * p.ldb < (1ll << 28);     // 32-bit byte offsets in the buffer loads
    if (pipe_mode && small_enough && !p.b_nn && p.K % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0 && p.sa % 4 == 0 && p.sb % 4 == 0 && al16(p.a) && al16(p.b)) {
        static const int pbudget = [] { const char* e = getenv("OMX_F32_PIPE_BUDGET"); return e ? atoi(e) : 256; }();
        static const bool lds_ok = [] {
            return hipFuncSetAttribute((const void*)gemm_f32_pipe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_LDS_BYTES) == hipSuccess &&
                   hipFuncSetAttribute((const void*)gemm_f32_pipe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_LDS_BYTES) == hipSuccess;
        }();
        OMX_REQUIRE(lds_ok, "gemm_f32: the device refused %zu bytes of LDS per block", P_LDS_BYTES);
        const int px = (p.N + 63) / 64, py = (p.M + 63) / 64, ptiles = px * py * p.batch, pk = (p.K + PK - 1) / PK;
        int splits = 1;
        while (ptiles * splits * 2 <= pbudget && pk / (splits * 2) >= 2) splits *= 2;
        g.splits = splits;
        if (splits > 1) {
            void* ws = nullptr;
            if (get_workspace_aux(&ws, (size_t)p.batch * splits * p.M * p.N * sizeof(float), s)) return 1;
            g.partial = (float*)ws;
        }
        if (p.K % PK == 0) gemm_f32_pipe_kernel<true><<<dim3(px, py, p.batch * splits), 256, P_LDS_BYTES, s>>>(g);
        else gemm_f32_pipe_kernel<false><<<dim3(px, py, p.batch * splits), 256, P_LDS_BYTES, s>>>(g);
        OMX_LAUNCH_CHECK();
        if (defer) {
            *p.defer_partial = splits > 1 ? g.partial : p.out;
            *p.defer_splits = splits;
            return 0;
        }
        if (splits > 1) {
            const int64_t total = (int64_t)p.batch * p.M * p.N;
            gemm_f32_reduce_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 4096), 256, 0, s>>>(g);
            OMX_LAUNCH_CHECK();
        }
        return 0;
    }
    const char* se = getenv("OMX_F32_SMALL");
    const int small_mode = se ? atoi(se) : -1;
    constexpr int kSmallBelow = 1024;
    const bool small = small_mode == 1 || (small_mode != 0 && tiles < kSmallBelow);
    int splits = 1;
    if (small) {
        const int sx = (p.N + SN - 1) / SN, sy = (p.M + SM - 1) / SM, stiles = sx * sy * p.batch;
        const char* be = getenv("OMX_F32_SMALL_BUDGET");
        // (round 6) a deferred epilogue folds the split sum into the caller's tail launch, so a second doubling costs no extra launch:
        // 1024 there (encoder 5.73 -> 5.48 ms), 512 where the reduce is its own launch (decoder 2.12 vs 2.19 ms at 1024).
        const int budget = be ? atoi(be) : (defer ? 1024 : 512);
        while (stiles * splits * 2 <= budget && ktiles / (splits * 2) >= 2) splits *= 2;
        g.splits = splits;
        if (splits > 1) {
            void* ws = nullptr;
            if (get_workspace_aux(&ws, (size_t)p.batch * splits * p.M * p.N * sizeof(float), s)) return 1;
            g.partial = (float*)ws;
        }
        gemm_f32_small_kernel<<<dim3(sx, sy, p.batch * splits), 256, 0, s>>>(g);
        OMX_LAUNCH_CHECK();
        if (defer) {
            *p.defer_partial = splits > 1 ? g.partial : p.out;
            *p.defer_splits = splits;
            return 0;
        }
        if (splits > 1) {
            const int64_t total = (int64_t)p.batch * p.M * p.N;
            gemm_f32_reduce_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 4096), 256, 0, s>>>(g);
            OMX_LAUNCH_CHECK();
        }
        return 0;
    }
    while (tiles * splits * 2 <= 256 && ktiles / (splits * 2) >= 2) splits *= 2;
    g.splits = splits;
    if (splits > 1) {
        void* ws = nullptr;
        if (get_workspace_aux(&ws, (size_t)p.batch * splits * p.M * p.N * sizeof(float), s)) return 1;
        g.partial = (float*)ws;
    }
    gemm_f32_kernel<<<dim3(gx, gy, p.batch * splits), 256, 0, s>>>(g);
    OMX_LAUNCH_CHECK();
    if (defer) {
        *p.defer_partial = splits > 1 ? g.partial : p.out;
        *p.defer_splits = splits;
        return 0;
    }
    if (splits > 1) {
        const int64_t total = (int64_t)p.batch * p.M * p.N;
        gemm_f32_reduce_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 4096), 256, 0, s>>>(g);
        OMX_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace omx
