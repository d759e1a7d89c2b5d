// float32 attention in one launch for the model the reference runs in f32 (Paraformer: funasr-mlx/src/paraformer.rs:509-516 encoder
// self-attention, :1090-1102 decoder cross-attention):  out = softmax(q k^T * scale) v  per head, head width 128, no mask, the explicit
// form's arithmetic -- every score of a query row is formed first, then max / exp(x - max) / sum / divide, then the product with v, all f32.
// It replaces four launches of the building-block path (q k^T GEMM, row softmax, P v split-K GEMM, split sum) whose 2 x [heads, Tq, Tk]
// score round trip through HBM and launch gaps were 35 us of a 110 us encoder layer.
//
// One block = 16 query rows of one head, four waves.  Keys are dealt to the waves in quarters (NT tiles of 16 keys each, Tk <= 64 NT).
//   scores:  v_mfma_f32_16x16x4_f32, A = q rows, B = k rows.  Both operands are k-contiguous in memory, so each lane loads float4s and
//            the four lane groups of one MFMA take the k indices {16 j + 4 g + e}: the order of a dot product's terms is free as long as
//            A and B agree, and this one needs no transposition and no LDS staging (a lane permutation after the load, see below).  Four key tiles (4 x 8 float4 per lane) are requested before
//            the first MFMA and the requests stay that far ahead, so the MFMA chain only ever waits for the first tile.
//   softmax: the 16 x Tk scores sit in LDS (33 KB); one wave per row, same three passes as the oracle.
//   P . v:   A = probabilities from LDS (one ds_read_b128 per 16 keys), B = v rows loaded as float4 along the head width: tile e of a
//            load holds columns {4 n + e}, so eight accumulators cover the 128 columns from two loads per key group.  The v loads are issued
//            BEFORE the softmax (they do not depend on it) and land under it.
//   sum:     the four waves' partial [16, 128] outputs meet in LDS (over the score buffer) and are added in wave order -- deterministic.
//   split:   when the tile grid leaves CUs idle the keys are dealt to 2 or 4 blocks per tile; each leaves its normalised product with its rows'
//            (max, sum) and attn_f32_combine_kernel weighs them together in share order (also deterministic).
#include "gemm.hpp"
#include "workspace.hpp"

namespace omx {

namespace {

constexpr int AF_ROWS = 16, AF_HD = 128, AF_SLOT = AF_ROWS * AF_HD + 2 * AF_ROWS;

template <int NT>
__global__ __launch_bounds__(256) void attn_f32_kernel(float* __restrict__ out, const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, int64_t ldq, int64_t ldkv, int64_t ldo, int Tq, int Tk_all,
                                                       float scale, float* __restrict__ part) {
    constexpr int KW = NT * 16;            // keys per wave
    constexpr int KP = 4 * KW;             // padded key count
    constexpr int S_LD = KP + 4;
    constexpr int S_FLOATS = AF_ROWS * S_LD > 4 * AF_ROWS * AF_HD ? AF_ROWS * S_LD : 4 * AF_ROWS * AF_HD;
    __shared__ __attribute__((aligned(16))) float S[S_FLOATS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * AF_ROWS, head = blockIdx.y;
    const int key0 = wave * KW;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // key split (gridDim.z > 1): this block sees only keys [KP z, KP (z + 1)) -- k / v rebased, Tk = its share.  Everything down to the
    // normalised product is the unsplit kernel on that share; the shares meet at the very end.
    const int nsplit = gridDim.z, kbase = blockIdx.z * KP;
    const int Tk = min(Tk_all - kbase, KP);
    k += (int64_t)kbase * ldkv;
    v += (int64_t)kbase * ldkv;
    __shared__ float row_max[AF_ROWS], row_sum[AF_ROWS];

    // ---- q and k fragments.  The MFMA wants lane (n, g) to hold row n, k group g -- sixteen different rows across sixteen neighbouring lanes,
    // the worst case for the texture addresser (one 16-byte request per lane: the first version of this kernel spent 7 us of its 20 there).
    // So the LOAD is done in the addresser's favourite shape -- lane s takes row s >> 2, group s & 3: four neighbouring lanes share 64
    // contiguous bytes -- and a ds_bpermute per dword moves each value to the lane the MFMA reads it from (source lane 4 n + g).
    const int srow = lane >> 2, sg = lane & 3;
    const int from = 4 * (4 * n + g);
    // (the loads are buffer loads: a plain 16-byte load whose elements are only ever used one by one gets split into four 4-byte loads by the
    // optimiser -- four times the requests, the opposite of the point; an intrinsic's result is left whole)
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(k), 0, 0x7fffffff, 0x00020000);
    auto to_fragment = [&](const u32x4& v) {
        f32x4 r;
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, (int)v[c]));
        return r;
    };
    f32x4 qf[8];
    u32x4 qraw[8];
    {
        // rows / keys past the end read the last one instead of branching around the load: a surplus query row is never stored, a
        // surplus key's score is never read and its probability is written as 0 below
        const int qo = (int)(((int64_t)min(m0 + srow, Tq - 1) * ldq + head * AF_HD + 4 * sg) * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) qraw[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(q_rsrc, qo + 64 * j, 0, 0));
    }
    // Software pipeline by hand: the compiler's scheduler sinks every load next to its first use (one exposed L2 round trip per tile -- the
    // first version of this kernel took 31 us that way), so the request blocks are fenced with sched_barrier: four key tiles are in flight
    // before the first MFMA, each pair of tiles computed is replaced by the next pair's requests, and the pair after the current one is
    // moved into fragment order under the current pair's MFMAs.
    constexpr int LEAD = NT < 4 ? NT : 4;
    u32x4 kraw[NT][8];
    f32x4 kf[NT][8];
    auto request_keys = [&](int t) {
        const int ko = (int)(((int64_t)min(key0 + 16 * t + srow, Tk - 1) * ldkv + head * AF_HD + 4 * sg) * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) kraw[t][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, ko + 64 * j, 0, 0));
    };
    auto permute_keys = [&](int t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[t][j] = to_fragment(kraw[t][j]);
    };
#pragma unroll
    for (int t = 0; t < LEAD; ++t) request_keys(t);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = to_fragment(qraw[j]);
    permute_keys(0);
    if (NT > 1) permute_keys(1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- scores, two key tiles at a time (two independent accumulator chains keep the matrix pipe busy)
#pragma unroll
    for (int t = 0; t < NT; t += 2) {
        if (t + LEAD < NT) request_keys(t + LEAD);
        if (t + LEAD + 1 < NT) request_keys(t + LEAD + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < NT) permute_keys(t + 2);
        if (t + 3 < NT) permute_keys(t + 3);
        f32x4 a0 = zero4, a1 = zero4;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[j][e], kf[t][j][e], a0, 0, 0, 0);
                if (t + 1 < NT) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[j][e], kf[t + 1][j][e], a1, 0, 0, 0);
            }
        // D: column (key) = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            S[(4 * g + r) * S_LD + key0 + 16 * t + n] = a0[r] * scale;
            if (t + 1 < NT) S[(4 * g + r) * S_LD + key0 + 16 * (t + 1) + n] = a1[r] * scale;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- v requests for the product below: key group j of this wave, lane group g, term e -> key key0 + 16 j + 4 g + e; halves of the width.
    //      The first LEAD groups go out here, ahead of the softmax they do not depend on; the rest replace the groups consumed.
    f32x4 vf[NT][4][2];
    auto request_values = [&](int j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = min(key0 + 16 * j + 4 * g + e, Tk - 1);
            const float* vp = v + (int64_t)key * ldkv + head * AF_HD + 4 * n;
            vf[j][e][0] = *reinterpret_cast<const f32x4*>(vp);
            vf[j][e][1] = *reinterpret_cast<const f32x4*>(vp + 64);
        }
    };
#pragma unroll
    for (int j = 0; j < LEAD; ++j) request_values(j);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // ---- softmax over the keys, rows 4 wave .. 4 wave + 3 (paraformer.rs:514: softmax(scores, axis = -1) in f32)
    // a row is NT values per lane, held in registers between the passes
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float* p = S + (4 * wave + r) * S_LD;
        float x[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NT; ++i) { x[i] = lane + 64 * i < Tk ? p[lane + 64 * i] : -INFINITY; mx = fmaxf(mx, x[i]); }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i) { x[i] = lane + 64 * i < Tk ? expf(x[i] - mx) : 0.f; sum += x[i]; }
        sum = wave_sum(sum);
#pragma unroll
        for (int i = 0; i < NT; ++i) p[lane + 64 * i] = lane + 64 * i < Tk ? x[i] / sum : 0.f;
        if (lane == 0) { row_max[4 * wave + r] = mx; row_sum[4 * wave + r] = sum; }
    }
    __syncthreads();
    // ---- partial out[16, 128] over this wave's keys: accumulator c = 4 half + e' holds columns 64 half + 4 n + e'
    f32x4 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = zero4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (j + LEAD < NT) request_values(j + LEAD);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 pa = *reinterpret_cast<const f32x4*>(S + n * S_LD + key0 + 16 * j + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[e], vf[j][e][c >> 2][c & 3], acc[c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                   // every wave is done reading the probabilities
    float* R = S + wave * AF_ROWS * AF_HD;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) R[(4 * g + r) * AF_HD + 64 * (c >> 2) + 4 * n + (c & 3)] = acc[c][r];
    __syncthreads();
    // ---- the four partials in wave order; thread t: row t / 16, columns 8 (t % 16) .. + 7
    const int row = threadIdx.x >> 4, col = (threadIdx.x & 15) * 8;
    f32x4 o[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        o[h] = *reinterpret_cast<const f32x4*>(S + row * AF_HD + col + 4 * h);
#pragma unroll
        for (int w = 1; w < 4; ++w) o[h] += *reinterpret_cast<const f32x4*>(S + (w * AF_ROWS + row) * AF_HD + col + 4 * h);
    }
    if (nsplit == 1) {
        if (m0 + row < Tq) {
#pragma unroll
            for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(out + (int64_t)(m0 + row) * ldo + head * AF_HD + col + 4 * h) = o[h];
        }
        return;
    }
    // ---- key split: share z leaves its normalised product O_z and its rows' (max m_z, sum l_z) for attn_f32_combine_kernel
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    float* mine = part + ((int64_t)tile * nsplit + blockIdx.z) * AF_SLOT;
#pragma unroll
    for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(mine + row * AF_HD + col + 4 * h) = o[h];
    if (threadIdx.x < AF_ROWS) {
        mine[AF_ROWS * AF_HD + threadIdx.x] = row_max[threadIdx.x];
        mine[AF_ROWS * AF_HD + AF_ROWS + threadIdx.x] = row_sum[threadIdx.x];
    }
}

// the shares of a tile combined in share order:  softmax over all keys . v  =  sum_z w_z O_z / sum_z w_z,   w_z = l_z exp(m_z - max_z m_z)
// -- the explicit form's value with one extra rounding per factor.  (Measured alternative: the last share to arrive at a per-tile counter
// combines inside the attention launch -- the device-scope fences around the counter made that launch 46 us instead of 17.)
__global__ __launch_bounds__(256) void attn_f32_combine_kernel(float* __restrict__ out, const float* __restrict__ part, int nsplit, int64_t ldo, int Tq,
                                                               int row_tiles) {
    const int tile = blockIdx.x, m0 = (tile % row_tiles) * AF_ROWS, head = tile / row_tiles;
    const int row = threadIdx.x >> 4, col = (threadIdx.x & 15) * 8;
    if (m0 + row >= Tq) return;
    const float* base = part + (int64_t)tile * nsplit * AF_SLOT;
    float m = -INFINITY;
    for (int z = 0; z < nsplit; ++z) m = fmaxf(m, base[z * AF_SLOT + AF_ROWS * AF_HD + row]);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 num[2] = {zero4, zero4};
    float den = 0.f;
    for (int z = 0; z < nsplit; ++z) {
        const float* pz = base + z * AF_SLOT;
        const float w = pz[AF_ROWS * AF_HD + AF_ROWS + row] * expf(pz[AF_ROWS * AF_HD + row] - m);
        den += w;
#pragma unroll
        for (int h = 0; h < 2; ++h) num[h] += w * *reinterpret_cast<const f32x4*>(pz + row * AF_HD + col + 4 * h);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(out + (int64_t)(m0 + row) * ldo + head * AF_HD + col + 4 * h) = num[h] / den;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// -1: the shape is not this kernel's (the caller keeps its building-block path); 0 launched; 1 error
int launch_attn_f32(float* out, const float* q, const float* k, const float* v, int64_t ldq, int64_t ldkv, int64_t ldo, int Tq, int Tk, int heads,
                    float scale, hipStream_t s) {
    static const int mode = [] { const char* e = getenv("OMX_ATTN_F32"); return e ? atoi(e) : 1; }();
    if (!mode || Tq < 1 || Tk < 1 || Tk > 512 || heads < 1) return -1;
    if ((int64_t)Tq * ldq >= (1ll << 28) || (int64_t)Tk * ldkv >= (1ll << 28)) return -1;      // 32-bit byte offsets in the buffer loads
    if ((ldq | ldkv | ldo) & 3 || !aligned16(out) || !aligned16(q) || !aligned16(k) || !aligned16(v)) return -1;
    const int tiles = ((Tq + AF_ROWS - 1) / AF_ROWS) * heads;
    // one block streams ALL of its head's k and v (2 x Tk x 512 bytes) through one CU at ~11 bytes a clock -- with 128 tiles or fewer half the
    // chip sits idle while the other half waits for memory.  So the keys are dealt to 2 or 4 blocks per tile while that keeps ~100 keys per
    // block and <= 256 blocks, and a second, small launch combines the shares.  Measured on the 30 s pass: 56 tiles x 4 shares 8.0 + 5.0 us
    // against 17.3 in one launch; 128 tiles x 2 shares 12.5 + 4.8 against 17.4 (encoder 4.46 against 4.55 ms).
    // OMX_ATTN_F32_SPLIT=1 keeps one block per tile.
    static const int max_split = [] { const char* e = getenv("OMX_ATTN_F32_SPLIT"); return e ? atoi(e) : 4; }();
    int split = 1;
    while (split * 2 <= max_split && tiles * split * 2 <= 256 && (Tk + split * 2 - 1) / (split * 2) >= 96) split *= 2;
    const int per = (Tk + split - 1) / split;                  // keys per block; the kernel width is the smallest that holds them
    const int nt = per <= 128 ? 2 : per <= 256 ? 4 : 8;
    while (split > 1 && (split - 1) * 64 * nt >= Tk) split /= 2;   // (never a share without keys)
    float* part = nullptr;
    if (split > 1) {
        void* ws = nullptr;
        if (get_workspace_aux(&ws, (size_t)tiles * split * AF_SLOT * sizeof(float), s)) return 1;
        part = (float*)ws;
    }
    const int row_tiles = (Tq + AF_ROWS - 1) / AF_ROWS;
    const dim3 grid(row_tiles, heads, split);
    const int width = split > 1 ? nt : (Tk <= 128 ? 2 : Tk <= 256 ? 4 : 8);
    if (width == 2) attn_f32_kernel<2><<<grid, 256, 0, s>>>(out, q, k, v, ldq, ldkv, ldo, Tq, Tk, scale, part);
    else if (width == 4) attn_f32_kernel<4><<<grid, 256, 0, s>>>(out, q, k, v, ldq, ldkv, ldo, Tq, Tk, scale, part);
    else attn_f32_kernel<8><<<grid, 256, 0, s>>>(out, q, k, v, ldq, ldkv, ldo, Tq, Tk, scale, part);
    OMX_LAUNCH_CHECK();
    if (split > 1) attn_f32_combine_kernel<<<tiles, 256, 0, s>>>(out, part, split, ldo, Tq, row_tiles);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // namespace omx
