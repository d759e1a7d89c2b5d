// float32 attention in one launch for the model the reference runs in f32 (Paraformer: funasr-mlx/src/paraformer.rs:509-516 encoder
// self-attention, :1090-1102 decoder cross-attention):  out = softmax(q k^T * scale) v  per head, head width 128, no mask, the explicit
// form's arithmetic -- every score of a query row is formed first, then max / exp(x - max) / sum / divide, then the product with v, all f32.
// It replaces four launches of the building-block path (q k^T GEMM, row softmax, P v split-K GEMM, split sum) whose 2 x [heads, Tq, Tk]
// score round trip through HBM and launch gaps were 35 us of a 110 us encoder layer.
//
// One block = 16 query rows of one head, four waves.  Keys are dealt to the waves in quarters (NT tiles of 16 keys each, Tk <= 64 NT).
//   scores:  v_mfma_f32_16x16x4_f32, A = q rows, B = k rows.  Both operands are k-contiguous in memory, so each lane loads float4s and
//            the four lane groups of one MFMA take the k indices {16 j + 4 g + e}: the order of a dot product's terms is free as long as
//            A and B agree, and this one needs no transposition and no LDS staging.  Four key tiles (4 x 8 float4 per lane) are requested before
//            the first MFMA and the requests stay that far ahead, so the MFMA chain only ever waits for the first tile.
//   softmax: the 16 x Tk scores sit in LDS (33 KB); one wave per row, same three passes as the oracle.
//   P . v:   A = probabilities from LDS (one ds_read_b128 per 16 keys), B = v rows loaded as float4 along the head width: tile e of a
//            load holds columns {4 n + e}, so eight accumulators cover the 128 columns from two loads per key group.  The v loads are issued
//            BEFORE the softmax (they do not depend on it) and land under it.
//   sum:     the four waves' partial [16, 128] outputs meet in LDS (over the score buffer) and are added in wave order -- deterministic.
#include "gemm.hpp"

namespace omx {

namespace {

constexpr int AF_ROWS = 16, AF_HD = 128;

template <int NT>
__global__ __launch_bounds__(256) void attn_f32_kernel(float* __restrict__ out, const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, int64_t ldq, int64_t ldkv, int64_t ldo, int Tq, int Tk,
                                                       float scale) {
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

    // ---- requests: q fragment (row m0 + n), then this wave's keys
    f32x4 qf[8];
    {
        // rows / keys past the end read the last one instead of branching around the load: a surplus query row is never stored, a
        // surplus key's score is never read and its probability is written as 0 below
        const int row = min(m0 + n, Tq - 1);
        const float* qp = q + (int64_t)row * ldq + head * AF_HD + 4 * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[j] = *reinterpret_cast<const f32x4*>(qp + 16 * j);
    }
    // Software pipeline by hand: the compiler's scheduler sinks every load next to its first use (one exposed L2 round trip per tile -- the
    // first version of this kernel took 31 us that way), so the request blocks are fenced with sched_barrier: four key tiles are in flight
    // before the first MFMA and each pair of tiles computed is replaced by the next pair's requests.
    constexpr int LEAD = NT < 4 ? NT : 4;
    f32x4 kf[NT][8];
    auto request_keys = [&](int t) {
        const int key = min(key0 + 16 * t + n, Tk - 1);
        const float* kp = k + (int64_t)key * ldkv + head * AF_HD + 4 * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[t][j] = *reinterpret_cast<const f32x4*>(kp + 16 * j);
    };
#pragma unroll
    for (int t = 0; t < LEAD; ++t) request_keys(t);
    __builtin_amdgcn_sched_barrier(0);
    // ---- scores, two key tiles at a time (two independent accumulator chains keep the matrix pipe busy)
#pragma unroll
    for (int t = 0; t < NT; t += 2) {
        if (t + LEAD < NT) request_keys(t + LEAD);
        if (t + LEAD + 1 < NT) request_keys(t + LEAD + 1);
        __builtin_amdgcn_sched_barrier(0);
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
    {
        const int row = threadIdx.x >> 4, col = (threadIdx.x & 15) * 8;
        if (m0 + row < Tq) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 s = *reinterpret_cast<const f32x4*>(S + row * AF_HD + col + 4 * h);
#pragma unroll
                for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(S + (w * AF_ROWS + row) * AF_HD + col + 4 * h);
                *reinterpret_cast<f32x4*>(out + (int64_t)(m0 + row) * ldo + head * AF_HD + col + 4 * h) = s;
            }
        }
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// -1: the shape is not this kernel's (the caller keeps its building-block path); 0 launched; 1 error
int launch_attn_f32(float* out, const float* q, const float* k, const float* v, int64_t ldq, int64_t ldkv, int64_t ldo, int Tq, int Tk, int heads,
                    float scale, hipStream_t s) {
    static const int mode = [] { const char* e = getenv("OMX_ATTN_F32"); return e ? atoi(e) : 1; }();
    if (!mode || Tq < 1 || Tk < 1 || Tk > 512 || heads < 1) return -1;
    if ((ldq | ldkv | ldo) & 3 || !aligned16(out) || !aligned16(q) || !aligned16(k) || !aligned16(v)) return -1;
    const dim3 grid((Tq + AF_ROWS - 1) / AF_ROWS, heads);
    if (Tk <= 128) attn_f32_kernel<2><<<grid, 256, 0, s>>>(out, q, k, v, ldq, ldkv, ldo, Tq, Tk, scale);
    else if (Tk <= 256) attn_f32_kernel<4><<<grid, 256, 0, s>>>(out, q, k, v, ldq, ldkv, ldo, Tq, Tk, scale);
    else attn_f32_kernel<8><<<grid, 256, 0, s>>>(out, q, k, v, ldq, ldkv, ldo, Tq, Tk, scale);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // namespace omx
