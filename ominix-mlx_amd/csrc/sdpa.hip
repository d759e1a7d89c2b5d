// omx_sdpa / omx_linear: per-op ABI entry points that pick the decode (HBM-streaming) or the
// prefill (MFMA) kernel family by shape.
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

// a second, independent buffer: for launchers that are called while their CALLER holds pointers into the main workspace
// (the f32 GEMM's split-K partials inside the Paraformer layers)
static void* g_ws2 = nullptr;
static size_t g_ws2_bytes = 0;
int get_workspace_aux(void** ptr, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (bytes > g_ws2_bytes) {
        if (g_ws2) OMX_HIP_CHECK(hipFree(g_ws2));        // (hipFree waits for the device: nothing still reads the old buffer)
        const size_t want = bytes < (size_t)(8u << 20) ? (size_t)(8u << 20) : bytes;
        OMX_HIP_CHECK(hipMalloc(&g_ws2, want));
        g_ws2_bytes = want;
    }
    *ptr = g_ws2;
    return 0;
}

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
    OMX_REQUIRE(dtype == OMX_BFLOAT16, "omx_sdpa: only bfloat16 is implemented (got dtype %d)", (int)dtype);
    hipStream_t s = (hipStream_t)stream;
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
    OMX_REQUIRE(dtype == OMX_BFLOAT16 || dtype == OMX_FLOAT32, "omx_linear: bfloat16 and float32 are implemented (got dtype %d)", (int)dtype);
    if (M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
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
