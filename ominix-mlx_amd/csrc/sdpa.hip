// omx_sdpa / omx_linear: per-op ABI entry points that pick the decode (HBM-streaming) or the
// prefill (MFMA) kernel family by shape.
#include <map>
#include <mutex>

#include "attn.hpp"
#include "gemv.hpp"
#include "gemm.hpp"
#include "workspace.hpp"

namespace omx {
namespace {
void* g_ws = nullptr;
size_t g_ws_bytes = 0;
bool g_ws_owned = false;
std::mutex g_ws_mu;
}  // namespace

int get_workspace(void** ptr, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (bytes > g_ws_bytes) {
        OMX_REQUIRE(g_ws_owned || g_ws == nullptr,
                    "workspace of %zu bytes set by omx_set_workspace is too small (%zu needed)", g_ws_bytes, bytes);
        if (g_ws) OMX_HIP_CHECK(hipFree(g_ws));
        size_t want = bytes < (size_t)(8u << 20) ? (size_t)(8u << 20) : bytes;
        OMX_HIP_CHECK(hipMalloc(&g_ws, want));
        g_ws_bytes = want;
        g_ws_owned = true;
    }
    *ptr = g_ws;
    return 0;
}

// a second scratch family, ONE BUFFER PER STREAM: for launchers that are called while their caller holds pointers into the main
// workspace (the f32 GEMM's split-K partials inside the Paraformer layers, the split-KV partials of the short batched prefill / verify
// attention).  Two engines or models on their own streams must not share it -- one would overwrite the other's partials, and growing
// it would free the buffer under the other user -- so it is keyed by the stream, like the ring GEMM's split-K scratch (gemm.hip).
namespace {
struct AuxWs { void* p = nullptr; size_t bytes = 0; bool pinned = false; };
std::map<uintptr_t, AuxWs> g_aux;
}  // namespace
int get_workspace_aux(void** ptr, size_t bytes, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    AuxWs& w = g_aux[reinterpret_cast<uintptr_t>(s)];
    if (bytes > w.bytes) {
        // a captured graph holds pointers into this buffer (workspace_aux_pin): moving it would leave the graph writing freed memory
        OMX_REQUIRE(!w.pinned, "the stream's scratch (%zu bytes) is held by a captured decode step and cannot grow to %zu bytes", w.bytes, bytes);
        if (w.p) {
            OMX_HIP_CHECK(hipStreamSynchronize(s));      // nothing of THIS stream still reads the old buffer
            OMX_HIP_CHECK(hipFree(w.p));
            w = AuxWs{};
        }
        const size_t want = bytes < (size_t)(8u << 20) ? (size_t)(8u << 20) : bytes;
        OMX_HIP_CHECK(hipMalloc(&w.p, want));
        w.bytes = want;
    }
    *ptr = w.p;
    return 0;
}
void workspace_aux_pin(hipStream_t s, bool pinned) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_aux.find(reinterpret_cast<uintptr_t>(s));
    if (it != g_aux.end()) it->second.pinned = pinned;
}
// the owner of `s` is about to destroy it
void workspace_release_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_aux.find(reinterpret_cast<uintptr_t>(s));
    if (it == g_aux.end()) return;
    (void)hipStreamSynchronize(s);
    if (it->second.p) (void)hipFree(it->second.p);
    g_aux.erase(it);
}

// ---- float32 / float16 SDPA: the explicit form on the exact-f32 matrix cores (gemm_f32.hip) -- scores = scale * q k^T, masked row
//      softmax, scores . v, everything in f32 (mlx-rs/src/fast.rs:303-331 runs the op in f32 and f16; MLX accumulates in f32 for both).
//      One wave per score row. ----
namespace {
template <class MT>
__global__ __launch_bounds__(256) void masked_softmax_rows_kernel(float* __restrict__ s, int64_t rows, int Tq, int Tk, int mask_mode,
                                                                  const MT* __restrict__ mask) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int qi = (int)(row % Tq);
    float* r = s + row * Tk;
    const int shift = Tk - Tq;
    auto val = [&](int j) -> float {
        float v = r[j];
        if (mask_mode == OMX_MASK_CAUSAL && j > qi + shift) v = -INFINITY;
        if (mask_mode == OMX_MASK_BOOL && !reinterpret_cast<const uint8_t*>(mask)[(size_t)qi * Tk + j]) v = -INFINITY;
        if (mask_mode == OMX_MASK_ADDITIVE) v += (float)mask[(size_t)qi * Tk + j];
        return v;
    };
    float m = -INFINITY;
    for (int j = lane; j < Tk; j += 64) m = fmaxf(m, val(j));
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < Tk; j += 64) {
        const float e = (m == -INFINITY) ? 0.f : expf(val(j) - m);
        r[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    for (int j = lane; j < Tk; j += 64) r[j] *= inv;
}
__global__ void f16_to_f32_kernel(float* dst, const f16_t* src, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
__global__ void f32_to_f16_kernel(f16_t* dst, const float* src, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (f16_t)src[i];
}
// strided rows -> dense f32 (K / V arrive as views of a larger cache buffer): [B, Hkv, T, D] with batch / head strides
__global__ void f16_rows_to_f32_kernel(float* dst, const f16_t* src, int B, int Hkv, int T, int D, int64_t bs, int64_t hs) {
    const int64_t n = (int64_t)B * Hkv * T * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = i % D, t = (i / D) % T, h = (i / ((int64_t)D * T)) % Hkv, b = i / ((int64_t)D * T * Hkv);
        dst[i] = (float)src[b * bs + h * hs + t * D + d];
    }
}

// q [B,H,Tq,D], k / v [B,Hkv,Tk,D] (strides), out [B,H,Tq,D], all f32; additive mask f32
int sdpa_f32(float* out, const float* q, const float* k, const float* v, int B, int H, int Hkv, int Tq, int Tk, int D,
             int64_t kv_bs, int64_t kv_hs, float scale, int mask_mode, const float* mask, hipStream_t s) {
    const int G = H / Hkv;
    // (the scores get their own stream-ordered allocation: the GEMMs below may take the stream's aux scratch for split-K partials)
    float* scores = nullptr;
    OMX_HIP_CHECK(hipMallocAsync((void**)&scores, (size_t)B * H * Tq * Tk * sizeof(float) + 256, s));
    struct Free { float* p; hipStream_t s; ~Free() { (void)hipFreeAsync(p, s); } } free_scores{scores, s};
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < Hkv; ++h) {   // the G query heads of a KV head: one batched launch, K / V with batch stride 0 (no tiling, fast.rs:118)
            GemmF32 g = {};
            g.a = q + ((size_t)b * H + (size_t)h * G) * Tq * D; g.b = k + b * kv_bs + h * kv_hs; g.out = scores + ((size_t)b * H + (size_t)h * G) * Tq * Tk;
            g.M = Tq; g.N = Tk; g.K = D; g.lda = D; g.ldb = D; g.ldc = Tk;
            g.sa = (int64_t)Tq * D; g.sb = 0; g.sc = (int64_t)Tq * Tk; g.batch = G; g.alpha = scale;
            if (launch_gemm_f32(g, s)) return 1;
        }
    const int64_t rows = (int64_t)B * H * Tq;
    masked_softmax_rows_kernel<float><<<(unsigned)((rows + 3) / 4), 256, 0, s>>>(scores, rows, Tq, Tk, mask_mode, mask);
    OMX_LAUNCH_CHECK();
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < Hkv; ++h) {
            GemmF32 g = {};
            g.a = scores + ((size_t)b * H + (size_t)h * G) * Tq * Tk; g.b = v + b * kv_bs + h * kv_hs; g.out = out + ((size_t)b * H + (size_t)h * G) * Tq * D;
            g.M = Tq; g.N = D; g.K = Tk; g.lda = Tk; g.ldb = D; g.ldc = D; g.b_nn = 1;
            g.sa = (int64_t)Tq * Tk; g.sb = 0; g.sc = (int64_t)Tq * D; g.batch = G; g.alpha = 1.0f;
            if (launch_gemm_f32(g, s)) return 1;
        }
    return 0;
}
}  // namespace

int decode_nsplit(int Tk, int BHkv) {
    // one 64-token step per block until the grid reaches ~2 blocks per CU
    int n = (Tk + 63) / 64;
    const int cap = (512 + BHkv - 1) / BHkv;
    if (n > cap) n = cap;
    if (n < 1) n = 1;
    return n;
}
}  // namespace omx

extern "C" {

int omx_set_workspace(void* ws, size_t bytes) {
    std::lock_guard<std::mutex> lk(omx::g_ws_mu);
    if (omx::g_ws_owned && omx::g_ws) (void)hipFree(omx::g_ws);
    omx::g_ws = ws;
    omx::g_ws_bytes = ws ? bytes : 0;
    omx::g_ws_owned = false;
    return 0;
}

size_t omx_sdpa_workspace_bytes(int B, int H, int Tq, int D) {
    if (Tq != 1) return 0;
    return omx::attn_decode_ws_bytes(B * H, 512, D);
}

int omx_sdpa(void* out, const void* q, const void* k, const void* v, int B, int H, int Hkv, int Tq, int Tk, int D,
             int64_t kv_batch_stride, int64_t kv_head_stride, float scale, int mask_mode, const void* mask,
             omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out && q && k && v, "omx_sdpa: null tensor");
    OMX_REQUIRE(B > 0 && H > 0 && Hkv > 0 && Tq > 0 && Tk > 0 && D > 0, "omx_sdpa: non-positive shape");
    OMX_REQUIRE(H % Hkv == 0, "omx_sdpa: n_q_heads=%d must be a multiple of n_kv_heads=%d", H, Hkv);
    OMX_REQUIRE(mask_mode >= OMX_MASK_NONE && mask_mode <= OMX_MASK_ADDITIVE, "omx_sdpa: invalid mask mode %d", mask_mode);
    OMX_REQUIRE(mask_mode < OMX_MASK_BOOL || mask != nullptr, "omx_sdpa: mask mode %d needs a mask array", mask_mode);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == OMX_FLOAT32) {
        const int mm = (Tq == 1 && mask_mode == OMX_MASK_CAUSAL) ? OMX_MASK_NONE : mask_mode;
        return omx::sdpa_f32((float*)out, (const float*)q, (const float*)k, (const float*)v, B, H, Hkv, Tq, Tk, D, kv_batch_stride, kv_head_stride,
                             scale, mm, (const float*)mask, s);
    }
    if (dtype == OMX_FLOAT16 && Tq > 1 && (D == 64 || D == 128) && (mask_mode == OMX_MASK_NONE || mask_mode == OMX_MASK_CAUSAL) &&
        !getenv("OMX_SDPA_F16_EXPLICIT")) {
        // (round 4) the flash kernel's float16 instantiation: f32 scores / softmax / accumulators, P rounded to float16 for the second
        // product, one rounding of the output -- no [Tq, Tk] scores in memory (OMX_SDPA_F16_EXPLICIT=1: the form below)
        return omx::launch_attn_prefill((omx::bf16_t*)out, (const omx::bf16_t*)q, (const omx::bf16_t*)k, (const omx::bf16_t*)v, B, H, Hkv, Tq, Tk, D,
                                        kv_batch_stride, kv_head_stride, scale, mask_mode, nullptr, s, false, nullptr, /*f16=*/true);
    }
    if (dtype == OMX_FLOAT16) {   // f32 arithmetic on widened copies, one rounding of the output to f16
        const size_t nq = (size_t)B * H * Tq * D, nkv = (size_t)B * Hkv * Tk * D, nm = mask_mode == OMX_MASK_ADDITIVE ? (size_t)Tq * Tk : 0;
        float* buf = nullptr;
        OMX_HIP_CHECK(hipMallocAsync((void**)&buf, (2 * nq + 2 * nkv + nm) * sizeof(float), s));
        float *qf = buf, *of = qf + nq, *kf = of + nq, *vf = kf + nkv, *mf = vf + nkv;
        omx::f16_to_f32_kernel<<<256, 256, 0, s>>>(qf, (const omx::f16_t*)q, (int64_t)nq);
        omx::f16_rows_to_f32_kernel<<<512, 256, 0, s>>>(kf, (const omx::f16_t*)k, B, Hkv, Tk, D, kv_batch_stride, kv_head_stride);
        omx::f16_rows_to_f32_kernel<<<512, 256, 0, s>>>(vf, (const omx::f16_t*)v, B, Hkv, Tk, D, kv_batch_stride, kv_head_stride);
        if (nm) omx::f16_to_f32_kernel<<<256, 256, 0, s>>>(mf, (const omx::f16_t*)mask, (int64_t)nm);
        const int mm = (Tq == 1 && mask_mode == OMX_MASK_CAUSAL) ? OMX_MASK_NONE : mask_mode;
        int rc = omx::sdpa_f32(of, qf, kf, vf, B, H, Hkv, Tq, Tk, D, (int64_t)Hkv * Tk * D, (int64_t)Tk * D, scale, mm,
                               mm == OMX_MASK_ADDITIVE ? mf : (const float*)mask, s);
        if (!rc) omx::f32_to_f16_kernel<<<256, 256, 0, s>>>((omx::f16_t*)out, of, (int64_t)nq);
        (void)hipFreeAsync(buf, s);
        if (rc) return 1;
        OMX_LAUNCH_CHECK();
        return 0;
    }
    OMX_REQUIRE(dtype == OMX_BFLOAT16, "omx_sdpa: bfloat16, float16 and float32 are implemented (got dtype %d)", (int)dtype);
    if (Tq == 1) {
        omx::AttnDecodeArgs a = {};
        a.q = (const omx::bf16_t*)q;
        a.k = (const omx::bf16_t*)k;
        a.v = (const omx::bf16_t*)v;
        a.kv_batch_stride = kv_batch_stride;
        a.kv_head_stride = kv_head_stride;
        a.B = B; a.H = H; a.Hkv = Hkv; a.Tk = Tk;
        a.scale = scale;
        a.mask_mode = (mask_mode == OMX_MASK_CAUSAL) ? OMX_MASK_NONE : mask_mode;   // Tq==1: causal sees all keys
        a.mask = mask;
        a.nsplit = omx::decode_nsplit(Tk, B * Hkv);
        void* ws = nullptr;
        if (omx::get_workspace(&ws, omx::attn_decode_ws_bytes(B * H, a.nsplit, D))) return 1;
        a.ws_o = (float*)ws;
        a.ws_ml = a.ws_o + (size_t)B * H * a.nsplit * D;
        a.out = (omx::bf16_t*)out;
        return omx::launch_attn_decode(a, D, s);
    }
    return omx::launch_attn_prefill((omx::bf16_t*)out, (const omx::bf16_t*)q, (const omx::bf16_t*)k,
                                    (const omx::bf16_t*)v, B, H, Hkv, Tq, Tk, D, kv_batch_stride, kv_head_stride, scale,
                                    mask_mode, mask, s);
}

int omx_linear(void* out, const void* x, const void* w, const void* bias, int M, int N, int K, omx_dtype dtype,
               omx_stream stream) {
    OMX_REQUIRE(out && x && w, "omx_linear: null tensor");
    OMX_REQUIRE(M >= 0 && N > 0 && K > 0, "omx_linear: bad shape M=%d N=%d K=%d", M, N, K);
    OMX_REQUIRE(dtype == OMX_BFLOAT16 || dtype == OMX_FLOAT32 || dtype == OMX_FLOAT16, "omx_linear: bfloat16, float16 and float32 are implemented (got dtype %d)", (int)dtype);
    if (M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == OMX_FLOAT16) {   // the eight-wave kernel's float16 form (round 4): float16 operands, f32 accumulation, one rounding to float16
        OMX_REQUIRE(M > 8 && K % 64 == 0, "omx_linear: float16 takes more than 8 rows and K %% 64 == 0 (M=%d K=%d)", M, K);
        const bool was = omx::gemm_set_f16(true);
        const int rc = omx::launch_gemm_bf16((omx::bf16_t*)out, (const omx::bf16_t*)x, (const omx::bf16_t*)w, (const omx::bf16_t*)bias, M, N, K, s);
        omx::gemm_set_f16(was);
        return rc;
    }
    if (dtype == OMX_FLOAT32) {   // exact-f32 matrix cores (gemm_f32.hip): the Paraformer path's dtype
        omx::GemmF32 g = {(const float*)x, (const float*)w, (const float*)bias, nullptr, (float*)out, M, N, K, K, K, N, 0, 0, 0, 0, 1, 0, 0, 1.0f};
        return omx::launch_gemm_f32(g, s);
    }
    if (M <= 4 && bias == nullptr && K % 8 == 0 && K <= 65536) {   // HBM-streaming GEMV (tuned widths, generic kernel otherwise)
        for (int m = 0; m < M; ++m) {
            omx::GemvArgs a = {};
            a.w0 = (const omx::bf16_t*)w;
            a.n0 = N; a.N = N; a.K = K;
            a.x = (const omx::bf16_t*)x + (size_t)m * K;
            a.out = (omx::bf16_t*)out + (size_t)m * N;
            if (omx::launch_gemv(a, omx::PRO_NONE, omx::EPI_STORE, s)) return 1;
        }
        return 0;
    }
    return omx::launch_gemm_bf16((omx::bf16_t*)out, (const omx::bf16_t*)x, (const omx::bf16_t*)w,
                                 (const omx::bf16_t*)bias, M, N, K, s);
}

int omx_linear_swiglu(void* out_plain, void* out_act, const void* x, const void* w, int M, int n_plain, int half, int K,
                      omx_dtype dtype, omx_stream stream) {
    OMX_REQUIRE(out_act && x && w && (n_plain == 0 || out_plain), "omx_linear_swiglu: null tensor");
    OMX_REQUIRE(dtype == OMX_BFLOAT16, "omx_linear_swiglu: only bfloat16 is implemented (got dtype %d)", (int)dtype);
    OMX_REQUIRE(M >= 0 && n_plain >= 0 && half > 0 && K > 0, "omx_linear_swiglu: bad shape M=%d plain=%d half=%d K=%d", M, n_plain, half, K);
    if (M == 0) return 0;
    OMX_REQUIRE(omx::gemm_swiglu_supported(M, n_plain, half, K),
                "omx_linear_swiglu: shape M=%d plain=%d half=%d K=%d is outside the fused kernels (plain %% 4, half %% 4, K %% 64); "
                "use omx_linear + omx_fused_swiglu", M, n_plain, half, K);
    return omx::launch_gemm_bf16_swiglu((omx::bf16_t*)out_plain, n_plain, (omx::bf16_t*)out_act, half, (const omx::bf16_t*)x,
                                        (const omx::bf16_t*)w, M, n_plain, half, K, (hipStream_t)stream);
}

}  // extern "C"
