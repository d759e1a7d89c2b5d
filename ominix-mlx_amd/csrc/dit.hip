// FLUX.2-klein DiT / MMDiT forward (BASELINE config 5) on the matrix cores.
//   reference: flux-klein-mlx/src/klein_model.rs -- FluxKlein::forward_with_rope :799-854,
//   KleinDoubleBlock::forward :399-522, KleinSingleBlock::forward :603-674, SharedModulation :248-254,
//   apply_rope :124-162, modulate/gate :909-925; layers.rs:256-283 timestep_embedding.
// The reference issues every LayerNorm / modulate / Linear / reshape / RoPE / matmul / softmax / concat as a
// separate lazy op and materialises the [24, S, S] score tensor twice per double block.  Here a block is:
//   [LN + modulate] -> GEMM(s) -> [per-head RMSNorm + interleaved RoPE, in place on the projection] ->
//   ONE joint flash attention over the [txt, img] sequence (the txt/img projections are written to adjacent
//   row ranges of one buffer, so the reference's concatenations cost nothing) -> GEMM with the gated
//   residual in its epilogue -> [LN + modulate] -> GEMM -> fused_swiglu on the two halves -> GEMM + gated
//   residual.  Single blocks read q/k/v/gate/up straight out of the fused 27648-wide projection through
//   strides and write attention and MLP outputs side by side into the [S, 12288] operand of to_out.
// Activations are bf16 (fp32 accumulate / fp32 softmax and norms); see DESIGN.md for the parity tolerance
// against the reference's f32 path.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "gemm.hpp"

namespace omx {
namespace {

// per-head RMSNorm (weight [128]) then RoPE on interleaved pairs (2i, 2i+1), in place.
// x: rows of `heads` x 128 starting every `ld` elements; cos/sin [S, 128] f32 with duplicated pairs.
__global__ __launch_bounds__(256) void klein_qk_norm_rope_kernel(bf16_t* __restrict__ xq, bf16_t* __restrict__ xk, int64_t ld, int S, int heads,
                                                                 const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk,
                                                                 const float* __restrict__ cosr, const float* __restrict__ sinr,
                                                                 int rope_row0, float eps) {
    // blockIdx.y = 0: queries, 1: keys (one launch for both); 16 (token, head) rows per block, 16 lanes x 8 elements each
    constexpr int D = 128, LPR = 16;
    bf16_t* x = blockIdx.y ? xk : xq;
    const bf16_t* w = blockIdx.y ? wk : wq;
    const int lane = threadIdx.x & 63, c = lane % LPR;
    const int64_t row = (int64_t)blockIdx.x * 16 + threadIdx.x / LPR;   // (token, head)
    if (row >= (int64_t)S * heads) return;
    const int t = (int)(row / heads), h = (int)(row % heads);
    bf16_t* p = x + (size_t)t * ld + (size_t)h * D + c * 8;
    const u32x4 r = *reinterpret_cast<const u32x4*>(p);
    const u32x4 wr = *reinterpret_cast<const u32x4*>(w + c * 8);
    // cos/sin rows hold every pair's value twice ([c0,c0,c1,c1,..]): two 16-byte loads each, even entries used
    const f32x4* cp = reinterpret_cast<const f32x4*>(cosr + (size_t)(rope_row0 + t) * D + c * 8);
    const f32x4* sp = reinterpret_cast<const f32x4*>(sinr + (size_t)(rope_row0 + t) * D + c * 8);
    const f32x4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
    const float cs[4] = {c0[0], c0[2], c1[0], c1[2]}, sn[4] = {s0[0], s0[2], s1[0], s1[2]};
    float v[8], wv[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = bf16lo(r[e]); v[2 * e + 1] = bf16hi(r[e]);
        wv[2 * e] = bf16lo(wr[e]); wv[2 * e + 1] = bf16hi(wr[e]);
    }
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
    ss = group_sum<LPR>(ss);
    const float rstd = 1.0f / sqrtf(ss / (float)D + eps);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = v[2 * e] * rstd * wv[2 * e], x1 = v[2 * e + 1] * rstd * wv[2 * e + 1];
        o[e] = pack_bf16(x0 * cs[e] - x1 * sn[e], x1 * cs[e] + x0 * sn[e]);
    }
    *reinterpret_cast<u32x4*>(p) = o;
}

// mlx_rs_core::fused_swiglu(up, gate) on strided column blocks: out[s, j] = silu(g[s, j]) * u[s, j]
__global__ __launch_bounds__(256) void swiglu_strided_kernel(bf16_t* __restrict__ out, int64_t ldo, const bf16_t* __restrict__ g,
                                                             const bf16_t* __restrict__ u, int64_t ldi, int S, int n) {
    const int nv = n / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)S * nv; i += (int64_t)gridDim.x * 256) {
        const int s = (int)(i / nv), j = (int)(i % nv) * 8;
        const u32x4 gv = *reinterpret_cast<const u32x4*>(g + (size_t)s * ldi + j);
        const u32x4 uv = *reinterpret_cast<const u32x4*>(u + (size_t)s * ldi + j);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g0 = bf16lo(gv[e]), g1 = bf16hi(gv[e]);
            o[e] = pack_bf16(g0 / (1.0f + expf(-g0)) * bf16lo(uv[e]), g1 / (1.0f + expf(-g1)) * bf16hi(uv[e]));
        }
        *reinterpret_cast<u32x4*>(out + (size_t)s * ldo + j) = o;
    }
}

// elementwise helpers on small vectors: mode 0 silu(x); mode 1 (1 + scale) * x + shift broadcast over rows
__global__ __launch_bounds__(256) void small_ew_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ x,
                                                       const bf16_t* __restrict__ shift, const bf16_t* __restrict__ scale,
                                                       int64_t n, int cols, int mode) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = bf16_to_f32(x[i]);
        float r;
        if (mode == 0) r = v / (1.0f + expf(-v));
        else r = (1.0f + bf16_to_f32(scale[i % cols])) * v + bf16_to_f32(shift[i % cols]);
        out[i] = f32_to_bf16(r);
    }
}

uint32_t crc32_name(const char* s) {
    uint32_t crc = 0xFFFFFFFFu;
    for (; *s; ++s) {
        crc ^= (uint8_t)*s;
        for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    }
    return ~crc;
}

}  // namespace
// tensor-parallel tail of a row-split projection: out = bf16(resid + sum * gate[col]) on the all-reduced partial
// generate_klein.rs:441-443 (sampler.rs:174-186): latent += (t_next - t_curr) * v, product and sum rounded separately;
// the master latent stays float32, the bf16 copy is what the next forward reads
__global__ __launch_bounds__(256) void euler_step_kernel(float* __restrict__ latent, const bf16_t* __restrict__ v, float dt,
                                                         bf16_t* __restrict__ latent_bf16, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float scaled = bf16_to_f32(v[i]) * dt;
        const float z = latent[i] + scaled;
        latent[i] = z;
        if (latent_bf16) latent_bf16[i] = f32_to_bf16(z);
    }
}

__global__ __launch_bounds__(256) void gated_residual_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ resid,
                                                             const bf16_t* __restrict__ gate, const bf16_t* __restrict__ sum,
                                                             int64_t n, int h) {
    for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * blockDim.x * 8) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(resid + i);
        const u32x4 v = *reinterpret_cast<const u32x4*>(sum + i);
        const u32x4 g = *reinterpret_cast<const u32x4*>(gate + (i % h));
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            o[q] = pack_bf16(bf16lo(r[q]) + bf16lo(v[q]) * bf16lo(g[q]), bf16hi(r[q]) + bf16hi(v[q]) * bf16hi(g[q]));
        *reinterpret_cast<u32x4*>(out + i) = o;
    }
}

}  // namespace omx

using namespace omx;

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int kNcclBfloat16 = 9, kNcclSum = 0;

struct omx_klein_ {
    omx_klein_config cfg;
    std::map<std::string, const bf16_t*> w;
    std::map<std::string, size_t> wbytes;      // size of every registered tensor (omx_klein_set_weight / the synthetic generator)
    std::vector<void*> owned;
    hipStream_t stream = nullptr;      // the stream helpers launch on (switched to stream_txt for the txt half of a double block)
    hipStream_t stream_main = nullptr, stream_txt = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bf16_t *proj_txt = nullptr, *act_txt = nullptr;   // MLP scratch of the concurrently running txt stream
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    // tensor parallel: heads and the MLP width are sharded, the residual stream is replicated
    int H_l = 0, h_l = 0, mh_l = 0;
    void* comm = nullptr;
    nccl_allreduce_fn allreduce = nullptr;
    bf16_t* partial = nullptr;
    // activations, sized for (s_txt, s_img)
    int s_txt = 0, s_img = 0;
    bf16_t *x = nullptr, *x2 = nullptr, *xm = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *att = nullptr,
           *proj = nullptr, *comb = nullptr, *act = nullptr, *vec = nullptr, *svec = nullptr, *temb = nullptr,
           *tmid = nullptr, *mod_img = nullptr, *mod_txt = nullptr, *mod_single = nullptr, *ada = nullptr,
           *lat_in = nullptr;
};

namespace {

// `elems`: bf16 elements the caller is about to read behind the pointer (the registered tensor must hold at least that many)
int kget(omx_klein m, const std::string& name, const bf16_t** out, size_t elems) {
    auto it = m->w.find(name);
    if (it == m->w.end()) return set_error("WeightNotFound: %s", name.c_str());
    auto sz = m->wbytes.find(name);
    if (sz != m->wbytes.end() && sz->second < elems * 2)
        return set_error("ShapeMismatch: %s holds %zu bytes, this forward reads %zu", name.c_str(), sz->second, elems * 2);
    *out = it->second;
    return 0;
}

template <class T>
int kalloc(omx_klein m, T** p, size_t n) {
    void* q = nullptr;
    OMX_HIP_CHECK(hipMalloc(&q, n * sizeof(T) + 64));
    *p = (T*)q;
    m->owned.push_back(q);
    return 0;
}

int linear(omx_klein m, bf16_t* out, const bf16_t* x, const char* wname, int M, int N, int K) {
    const bf16_t* w = nullptr;
    if (kget(m, wname, &w, (size_t)N * K)) return 1;
    return omx_linear(out, x, w, nullptr, M, N, K, OMX_BFLOAT16, m->stream);
}

int ensure_buffers(omx_klein m, int s_txt, int s_img) {
    if (m->s_txt == s_txt && m->s_img == s_img) return 0;
    OMX_REQUIRE(m->s_txt == 0, "omx_klein: sequence lengths are fixed by the first forward (%d txt, %d img)", m->s_txt, m->s_img);
    const omx_klein_config& c = m->cfg;
    const size_t S = (size_t)s_txt + s_img, h = c.hidden_size, mh = c.mlp_hidden;
    if (kalloc(m, &m->x, S * h) || kalloc(m, &m->x2, S * h) || kalloc(m, &m->xm, S * h) || kalloc(m, &m->q, S * h) ||
        kalloc(m, &m->k, S * h) || kalloc(m, &m->v, S * h) || kalloc(m, &m->att, S * h) ||
        kalloc(m, &m->proj, S * (3 * h + 2 * mh)) || kalloc(m, &m->comb, S * (h + mh)) || kalloc(m, &m->act, S * mh) ||
        kalloc(m, &m->vec, h) || kalloc(m, &m->svec, h) || kalloc(m, &m->temb, (size_t)256) || kalloc(m, &m->tmid, h) ||
        kalloc(m, &m->mod_img, 6 * h) || kalloc(m, &m->mod_txt, 6 * h) || kalloc(m, &m->mod_single, 3 * h) ||
        kalloc(m, &m->ada, 2 * h) || kalloc(m, &m->lat_in, (size_t)s_img * c.in_channels) || kalloc(m, &m->partial, S * h) ||
        kalloc(m, &m->proj_txt, (size_t)s_txt * 2 * mh) || kalloc(m, &m->act_txt, (size_t)s_txt * mh))
        return 1;
    m->s_txt = s_txt;
    m->s_img = s_img;
    return 0;
}

int attention(omx_klein m, bf16_t* out, int64_t o_ts, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, int S) {
    const omx_klein_config& c = m->cfg;
    AttnLayout L = {0, c.head_dim, ld, ld, 0, c.head_dim, o_ts};
    return launch_attn_prefill(out, q, k, v, 1, m->H_l, m->H_l, S, S, c.head_dim, 0, c.head_dim,
                               1.0f / sqrtf((float)c.head_dim), OMX_MASK_NONE, nullptr, m->stream, false, &L);
}

// out = resid + gate * (x . W^T): one fused GEMM on a single GPU; under tensor parallelism W holds this rank's
// input columns, the bf16 partial is all-reduced (RCCL sum) and the gated residual is applied afterwards
int gated_projection(omx_klein m, bf16_t* out, const bf16_t* x, const bf16_t* w, const bf16_t* resid, const bf16_t* gate,
                     int M, int N, int K) {
    hipStream_t s = m->stream;
    if (m->cfg.tp_size == 1 && m->allreduce == nullptr) return launch_gemm_bf16_gated(out, x, w, resid, gate, M, N, K, s);
    OMX_REQUIRE(m->allreduce != nullptr, "tp_size > 1 but no communicator set (omx_klein_set_comm)");
    if (launch_gemm_bf16(m->partial, x, w, nullptr, M, N, K, s)) return 1;
    OMX_REQUIRE(m->allreduce(m->partial, m->partial, (size_t)M * N, kNcclBfloat16, kNcclSum, m->comm, s) == 0, "ncclAllReduce failed");
    gated_residual_kernel<<<1024, 256, 0, s>>>(out, resid, gate, m->partial, (int64_t)M * N, N);
    OMX_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

int omx_klein_create(omx_klein* out, const omx_klein_config* cfg) {
    OMX_REQUIRE(out && cfg, "omx_klein_create: null argument");
    OMX_REQUIRE(cfg->head_dim == 128 && cfg->hidden_size == cfg->num_heads * cfg->head_dim,
                "InvalidConfig: klein needs head_dim 128 and hidden_size = heads * head_dim");
    OMX_REQUIRE(cfg->hidden_size % 64 == 0 && cfg->mlp_hidden % 64 == 0 && cfg->in_channels % 64 == 0 && cfg->txt_embed_dim % 64 == 0,
                "InvalidConfig: klein widths must be multiples of 64");
    OMX_REQUIRE(cfg->tp_size >= 0 && cfg->tp_rank >= 0 && cfg->tp_rank < (cfg->tp_size > 0 ? cfg->tp_size : 1),
                "InvalidConfig: tp rank %d of %d", cfg->tp_rank, cfg->tp_size);
    omx_klein m = new omx_klein_();
    m->cfg = *cfg;
    if (m->cfg.tp_size == 0) m->cfg.tp_size = 1;
    const int n = m->cfg.tp_size;
    OMX_REQUIRE(cfg->num_heads % n == 0 && (cfg->mlp_hidden / n) % 64 == 0 && cfg->mlp_hidden % n == 0,
                "InvalidConfig: heads %d / mlp width %d must divide by tp_size %d (mlp shard a multiple of 64)", cfg->num_heads, cfg->mlp_hidden, n);
    m->H_l = cfg->num_heads / n;
    m->h_l = m->H_l * cfg->head_dim;
    m->mh_l = cfg->mlp_hidden / n;
    OMX_HIP_CHECK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    m->stream_main = m->stream;
    OMX_HIP_CHECK(hipStreamCreateWithFlags(&m->stream_txt, hipStreamNonBlocking));
    OMX_HIP_CHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    OMX_HIP_CHECK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    OMX_HIP_CHECK(hipEventCreate(&m->ev0));
    OMX_HIP_CHECK(hipEventCreate(&m->ev1));
    *out = m;
    return 0;
}

int omx_klein_destroy(omx_klein m) {
    if (!m) return 0;
    (void)hipStreamSynchronize(m->stream);
    for (void* p : m->owned) (void)hipFree(p);
    (void)hipEventDestroy(m->ev0);
    (void)hipEventDestroy(m->ev1);
    (void)hipStreamDestroy(m->stream_main);
    if (m->stream_txt) (void)hipStreamDestroy(m->stream_txt);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    delete m;
    return 0;
}

int omx_klein_set_weight(omx_klein m, const char* name, const void* ptr, size_t nbytes) {
    OMX_REQUIRE(m && name && ptr, "omx_klein_set_weight: null argument");
    OMX_REQUIRE(nbytes > 0 && nbytes % 2 == 0, "omx_klein_set_weight: %s: %zu bytes is not a bf16 tensor", name, nbytes);
    m->w[name] = (const bf16_t*)ptr;
    m->wbytes[name] = nbytes;
    return 0;
}

int omx_klein_set_comm(omx_klein m, void* comm, void* allreduce_fn) {
    OMX_REQUIRE(m, "omx_klein_set_comm: null model");
    m->comm = comm;
    m->allreduce = (nccl_allreduce_fn)allreduce_fn;
    return 0;
}

int omx_klein_synth_weights(omx_klein m, uint32_t base_seed) {
    OMX_REQUIRE(m, "omx_klein_synth_weights: null model");
    const omx_klein_config& c = m->cfg;
    const int h = c.hidden_size, mh = c.mlp_hidden, D = c.head_dim;
    const int r = c.tp_rank, hl = m->h_l, ml = m->mh_l;
    struct Seg { int64_t start, len; };
    const float amp_w = (float)(0.02 * sqrt(3.0)), amp_n = (float)(0.01 * sqrt(3.0));
    auto seed_of = [&](const std::string& name) { return base_seed ^ crc32_name(("klein." + name).c_str()); };
    // replicated tensor (or a single GPU): the whole logical [rows, cols]
    auto make = [&](const std::string& name, size_t rows, size_t cols, bool norm) -> int {
        bf16_t* p = nullptr;
        if (kalloc(m, &p, rows * cols)) return 1;
        if (omx_fill_uniform(p, rows * cols, seed_of(name), norm ? amp_n : amp_w, norm ? 1.0f : 0.0f, OMX_BFLOAT16, m->stream)) return 1;
        m->w[name] = p;
        m->wbytes[name] = rows * cols * 2;
        return 0;
    };
    // this rank's ROWS of a logical [*, cols] tensor: the listed row ranges stacked
    auto make_rows = [&](const std::string& name, std::initializer_list<Seg> segs, int64_t cols) -> int {
        int64_t total = 0;
        for (const Seg& g : segs) total += g.len;
        bf16_t* p = nullptr;
        if (kalloc(m, &p, (size_t)(total * cols))) return 1;
        int64_t off = 0;
        for (const Seg& g : segs) {
            if (omx_fill_uniform_2d(p + off * cols, g.len, cols, cols, g.start, 0, seed_of(name), amp_w, 0.f, OMX_BFLOAT16, m->stream)) return 1;
            off += g.len;
        }
        m->w[name] = p;
        m->wbytes[name] = (size_t)(total * cols) * 2;
        return 0;
    };
    // this rank's COLUMNS of a logical [rows, cols_full] tensor: the listed column ranges side by side
    auto make_cols = [&](const std::string& name, int64_t rows, std::initializer_list<Seg> segs, int64_t cols_full) -> int {
        int64_t total = 0, widest = 0;
        for (const Seg& g : segs) { total += g.len; widest = g.len > widest ? g.len : widest; }
        bf16_t *p = nullptr, *tmp = nullptr;
        if (kalloc(m, &p, (size_t)(rows * total))) return 1;
        OMX_HIP_CHECK(hipMalloc((void**)&tmp, (size_t)(rows * widest) * 2));
        int64_t off = 0;
        for (const Seg& g : segs) {
            if (omx_fill_uniform_2d(tmp, rows, g.len, cols_full, 0, g.start, seed_of(name), amp_w, 0.f, OMX_BFLOAT16, m->stream)) return 1;
            OMX_HIP_CHECK(hipMemcpy2DAsync(p + off, (size_t)total * 2, tmp, (size_t)g.len * 2, (size_t)g.len * 2, (size_t)rows,
                                           hipMemcpyDeviceToDevice, m->stream));
            OMX_HIP_CHECK(hipStreamSynchronize(m->stream));   // tmp is reused by the next segment
            off += g.len;
        }
        OMX_HIP_CHECK(hipFree(tmp));
        m->w[name] = p;
        m->wbytes[name] = (size_t)(rows * total) * 2;
        return 0;
    };
    if (make("x_embedder.weight", h, c.in_channels, false) || make("context_embedder.weight", h, c.txt_embed_dim, false) ||
        make("time_embed_1.weight", h, 256, false) || make("time_embed_2.weight", h, h, false) ||
        make("double_mod_img.linear.weight", 6 * (size_t)h, h, false) || make("double_mod_txt.linear.weight", 6 * (size_t)h, h, false) ||
        make("single_mod.linear.weight", 3 * (size_t)h, h, false) || make("norm_out.weight", 2 * (size_t)h, h, false) ||
        make("proj_out.weight", c.in_channels, h, false))
        return 1;
    // shard plan == klein.shard_state_dict (ominix-mlx_amd/klein.py): heads and MLP columns are contiguous per rank
    for (int i = 0; i < c.depth; ++i) {
        const std::string b = "double_blocks." + std::to_string(i) + ".";
        for (const char* st : {"img", "txt"}) {
            const std::string sname = b + st + "_";
            if (make_rows(sname + "to_q.weight", {{(int64_t)r * hl, hl}}, h) || make_rows(sname + "to_k.weight", {{(int64_t)r * hl, hl}}, h) ||
                make_rows(sname + "to_v.weight", {{(int64_t)r * hl, hl}}, h) || make_cols(sname + "to_out.weight", h, {{(int64_t)r * hl, hl}}, h) ||
                make(sname + "norm_q.weight", 1, D, true) || make(sname + "norm_k.weight", 1, D, true) ||
                make_rows(sname + "mlp_in.weight", {{(int64_t)r * ml, ml}, {(int64_t)mh + (int64_t)r * ml, ml}}, h) ||
                make_cols(sname + "mlp_out.weight", h, {{(int64_t)r * ml, ml}}, mh))
                return 1;
        }
    }
    for (int i = 0; i < c.depth_single; ++i) {
        const std::string b = "single_blocks." + std::to_string(i) + ".";
        if (make_rows(b + "to_qkv_mlp.weight",
                      {{(int64_t)r * hl, hl}, {(int64_t)h + (int64_t)r * hl, hl}, {2 * (int64_t)h + (int64_t)r * hl, hl},
                       {3 * (int64_t)h + (int64_t)r * ml, ml}, {3 * (int64_t)h + mh + (int64_t)r * ml, ml}}, h) ||
            make_cols(b + "to_out.weight", h, {{(int64_t)r * hl, hl}, {(int64_t)h + (int64_t)r * ml, ml}}, (int64_t)h + mh) ||
            make(b + "norm_q.weight", 1, D, true) || make(b + "norm_k.weight", 1, D, true))
            return 1;
    }
    OMX_HIP_CHECK(hipStreamSynchronize(m->stream));
    return 0;
}

/* FluxKlein::forward_with_rope: latent [s_img, in_channels], txt [s_txt, txt_embed_dim] (bf16, device),
 * timestep = t * 1000, rope cos/sin [s_txt + s_img, 128] f32 (device; txt rows first) -> out [s_img, in_channels] */
int omx_klein_forward_with_rope(omx_klein m, void* out, const void* latent, const void* txt_embed, int s_img, int s_txt,
                                float timestep, const float* rope_cos, const float* rope_sin) {
    OMX_REQUIRE(m && out && latent && txt_embed && rope_cos && rope_sin, "omx_klein_forward_with_rope: null argument");
    OMX_REQUIRE(s_img > 0 && s_txt > 0, "omx_klein_forward_with_rope: empty sequence");
    const omx_klein_config& c = m->cfg;
    if (ensure_buffers(m, s_txt, s_img)) return 1;
    m->stream = m->stream_main;   // an earlier forward that failed inside a txt half may have left the side stream selected
    hipStream_t s = m->stream;
    const int h = c.hidden_size, S = s_txt + s_img;
    const int H = m->H_l, hl = m->h_l, mh = m->mh_l;   // this rank's heads / attention width / MLP width
    const float rms_eps = 1e-5f;   // RmsNorm::DEFAULT_EPS
    OMX_HIP_CHECK(hipEventRecord(m->ev0, s));
    bf16_t* x = m->x;     // [S, h]: rows [0, s_txt) = txt stream, [s_txt, S) = img stream
    bf16_t* x2 = m->x2;
    // input projections (klein_model.rs:810-811)
    if (linear(m, x + (size_t)s_txt * h, (const bf16_t*)latent, "x_embedder.weight", s_img, h, c.in_channels)) return 1;
    if (linear(m, x, (const bf16_t*)txt_embed, "context_embedder.weight", s_txt, h, c.txt_embed_dim)) return 1;
    {   // timestep_embedding(t, 256) = [cos | sin], layers.rs:256-283 (256 values: host)
        std::vector<bf16_t> te(256);
        for (int i = 0; i < 128; ++i) {
            const float freq = expf(-logf(10000.0f) * (float)i / 128.0f);
            const float arg = timestep * freq;
            te[i] = f32_to_bf16((float)cos((double)arg));
            te[128 + i] = f32_to_bf16((float)sin((double)arg));
        }
        OMX_HIP_CHECK(hipMemcpyAsync(m->temb, te.data(), 512, hipMemcpyHostToDevice, s));
        OMX_HIP_CHECK(hipStreamSynchronize(s));   // te is a stack buffer
    }
    if (linear(m, m->tmid, m->temb, "time_embed_1.weight", 1, h, 256)) return 1;
    small_ew_kernel<<<8, 256, 0, s>>>(m->tmid, m->tmid, nullptr, nullptr, h, h, 0);
    if (linear(m, m->vec, m->tmid, "time_embed_2.weight", 1, h, h)) return 1;
    small_ew_kernel<<<8, 256, 0, s>>>(m->svec, m->vec, nullptr, nullptr, h, h, 0);        // silu(vec), shared by all modulations
    if (linear(m, m->mod_img, m->svec, "double_mod_img.linear.weight", 1, 6 * h, h)) return 1;
    if (linear(m, m->mod_txt, m->svec, "double_mod_txt.linear.weight", 1, 6 * h, h)) return 1;
    if (linear(m, m->mod_single, m->svec, "single_mod.linear.weight", 1, 3 * h, h)) return 1;
    OMX_LAUNCH_CHECK();

    const bf16_t* w = nullptr;
    // The txt (512 rows) and img (4096 rows) halves of a double block are independent between the joint attentions.  On one
    // GPU the txt GEMMs fill 96 of 256 CUs and the img ones 192: they run on two HIP streams, forked and joined with events
    // around each half, so the small txt chain hides under the img chain.
    const char* dual_env = getenv("OMX_KLEIN_DUAL_STREAM");
    const bool dual = !(dual_env && dual_env[0] == '0') && c.tp_size <= 1 && m->allreduce == nullptr && s_txt > 0;
    // SwiGLU in the epilogue of the producing GEMM (OMX_KLEIN_FUSE_SWIGLU=0 keeps the stored projection + separate kernel)
    const char* fuse_env = getenv("OMX_KLEIN_FUSE_SWIGLU");
    const bool fuse_act = !(fuse_env && fuse_env[0] == '0');
    hipStream_t const s_main = m->stream_main;
    auto fork = [&]() -> int {
        if (!dual) return 0;
        OMX_HIP_CHECK(hipEventRecord(m->ev_fork, s_main));
        OMX_HIP_CHECK(hipStreamWaitEvent(m->stream_txt, m->ev_fork, 0));
        return 0;
    };
    auto join = [&]() -> int {
        if (!dual) return 0;
        OMX_HIP_CHECK(hipEventRecord(m->ev_join, m->stream_txt));
        OMX_HIP_CHECK(hipStreamWaitEvent(s_main, m->ev_join, 0));
        return 0;
    };
    // the side stream's GEMMs (512 rows) as 128 x 256 tiles: 48 workgroups that fit the CUs the img grid leaves idle, instead of 384
    // tiles of 64^2 queueing behind its workgroups (OMX_KLEIN_TXT_ROWS128=0: the shape's own choice)
    // every exit path -- the many `return 1` below included -- leaves no tile preference and the model's own stream behind
    struct HintGuard {
        hipStream_t* slot; hipStream_t keep;
        ~HintGuard() { gemm_tile_hint(0); *slot = keep; }
    } hint_guard{&m->stream, s_main};
    gemm_tile_hint(0);
    const char* t128_env = getenv("OMX_KLEIN_TXT_ROWS128");
    const bool txt_rows128 = dual && !(t128_env && t128_env[0] == '0');
    auto on_stream = [&](int st) {   // txt half -> side stream, img half -> main stream
        m->stream = (dual && st == 0) ? m->stream_txt : s_main;
        s = m->stream;
        gemm_tile_hint(txt_rows128 && st == 0 ? 128 : 0);
    };
    for (int i = 0; i < c.depth; ++i) {
        const std::string b = "double_blocks." + std::to_string(i) + ".";
        // ---- attention half: per-stream LN+modulate and q/k/v projections into adjacent row ranges ----
        if (fork()) return 1;
        for (int st = 0; st < 2; ++st) {
            on_stream(st);
            const char* sn = st ? "img_" : "txt_";
            const bf16_t* mod = st ? m->mod_img : m->mod_txt;
            const int rows = st ? s_img : s_txt;
            const size_t r0 = st ? (size_t)s_txt : 0;
            if (omx_fused_modulate(m->xm + r0 * h, x + r0 * h, mod /*shift1*/, mod + h /*scale1*/, 1, rows, h, 1e-6f, OMX_BFLOAT16, s)) return 1;
            if (linear(m, m->q + r0 * hl, m->xm + r0 * h, (b + sn + "to_q.weight").c_str(), rows, hl, h)) return 1;
            if (linear(m, m->k + r0 * hl, m->xm + r0 * h, (b + sn + "to_k.weight").c_str(), rows, hl, h)) return 1;
            if (linear(m, m->v + r0 * hl, m->xm + r0 * h, (b + sn + "to_v.weight").c_str(), rows, hl, h)) return 1;
            const unsigned blocks = (unsigned)(((size_t)rows * H + 15) / 16);
            const bf16_t* wkn = nullptr;
            if (kget(m, b + sn + "norm_q.weight", &w, (size_t)c.head_dim) || kget(m, b + sn + "norm_k.weight", &wkn, (size_t)c.head_dim)) return 1;
            klein_qk_norm_rope_kernel<<<dim3(blocks, 2), 256, 0, s>>>(m->q + r0 * hl, m->k + r0 * hl, hl, rows, H, w, wkn, rope_cos, rope_sin,
                                                                      (int)r0, rms_eps);
            OMX_LAUNCH_CHECK();
        }
        on_stream(1);
        if (join()) return 1;
        // ONE joint attention over [txt, img]: img and txt queries both see all keys (klein_model.rs:461-483)
        if (attention(m, m->att, hl, m->q, m->k, m->v, hl, S)) return 1;
        if (fork()) return 1;
        for (int st = 0; st < 2; ++st) {
            on_stream(st);
            bf16_t* proj = (dual && st == 0) ? m->proj_txt : m->proj;
            bf16_t* act = (dual && st == 0) ? m->act_txt : m->act;
            const char* sn = st ? "img_" : "txt_";
            const bf16_t* mod = st ? m->mod_img : m->mod_txt;
            const int rows = st ? s_img : s_txt;
            const size_t r0 = st ? (size_t)s_txt : 0;
            if (kget(m, b + sn + "to_out.weight", &w, (size_t)h * hl)) return 1;
            if (gated_projection(m, x2 + r0 * h, m->att + r0 * hl, w, x + r0 * h, mod + 2 * h /*gate1*/, rows, h, hl)) return 1;
            // ---- MLP half ----
            if (omx_fused_modulate(m->xm + r0 * h, x2 + r0 * h, mod + 3 * h, mod + 4 * h, 1, rows, h, 1e-6f, OMX_BFLOAT16, s)) return 1;
            if (fuse_act && gemm_swiglu_preferred(rows, 0, mh, h)) {
                if (kget(m, b + sn + "mlp_in.weight", &w, (size_t)2 * mh * h)) return 1;
                if (launch_gemm_bf16_swiglu(nullptr, 0, act, mh, m->xm + r0 * h, w, rows, 0, mh, h, s)) return 1;
            } else {
                if (linear(m, proj, m->xm + r0 * h, (b + sn + "mlp_in.weight").c_str(), rows, 2 * mh, h)) return 1;
                swiglu_strided_kernel<<<2048, 256, 0, s>>>(act, mh, proj /*gate = first half*/, proj + mh /*up*/, 2 * mh, rows, mh);
                OMX_LAUNCH_CHECK();
            }
            if (kget(m, b + sn + "mlp_out.weight", &w, (size_t)h * mh)) return 1;
            if (gated_projection(m, x + r0 * h, act, w, x2 + r0 * h, mod + 5 * h /*gate2*/, rows, h, mh)) return 1;
        }
        on_stream(1);
        if (join()) return 1;
    }
    // x already holds [txt, img] (klein_model.rs:833)
    const int64_t ldp = 3 * (int64_t)hl + 2 * mh, ldc = (int64_t)hl + mh;
    for (int i = 0; i < c.depth_single; ++i) {
        const std::string b = "single_blocks." + std::to_string(i) + ".";
        const bf16_t* mod = m->mod_single;
        if (omx_fused_modulate(m->xm, x, mod, mod + h, 1, S, h, 1e-6f, OMX_BFLOAT16, s)) return 1;
        const bool fused = fuse_act && gemm_swiglu_preferred(S, 3 * hl, mh, h);
        if (fused) {   // q/k/v columns -> proj, SwiGLU of the MLP columns straight into comb[:, hl:]
            if (kget(m, b + "to_qkv_mlp.weight", &w, (size_t)ldp * h)) return 1;
            if (launch_gemm_bf16_swiglu(m->proj, (int)ldp, m->comb + hl, (int)ldc, m->xm, w, S, 3 * hl, mh, h, s)) return 1;
        } else if (linear(m, m->proj, m->xm, (b + "to_qkv_mlp.weight").c_str(), S, (int)ldp, h)) return 1;
        const unsigned blocks = (unsigned)(((size_t)S * H + 15) / 16);
        const bf16_t* wkn = nullptr;
        if (kget(m, b + "norm_q.weight", &w, (size_t)c.head_dim) || kget(m, b + "norm_k.weight", &wkn, (size_t)c.head_dim)) return 1;
        klein_qk_norm_rope_kernel<<<dim3(blocks, 2), 256, 0, s>>>(m->proj, m->proj + hl, ldp, S, H, w, wkn, rope_cos, rope_sin, 0, rms_eps);
        OMX_LAUNCH_CHECK();
        if (attention(m, m->comb, ldc, m->proj, m->proj + hl, m->proj + 2 * hl, ldp, S)) return 1;          // cols [0, hl)
        if (!fused) {
            swiglu_strided_kernel<<<2048, 256, 0, s>>>(m->comb + hl, ldc, m->proj + 3 * hl, m->proj + 3 * hl + mh, ldp, S, mh);   // cols [hl, hl+mh)
            OMX_LAUNCH_CHECK();
        }
        if (kget(m, b + "to_out.weight", &w, (size_t)h * ldc)) return 1;
        if (gated_projection(m, x2, m->comb, w, x, mod + 2 * h, S, h, (int)ldc)) return 1;
        bf16_t* t = x; x = x2; x2 = t;
    }
    // final layer: RmsNorm (weight = ones), AdaLN chunks [scale, shift], proj_out (klein_model.rs:845-853)
    if (linear(m, m->ada, m->svec, "norm_out.weight", 1, 2 * h, h)) return 1;
    bf16_t* img_out = x + (size_t)s_txt * h;
    if (omx_rms_norm(m->xm, img_out, nullptr, s_img, h, rms_eps, OMX_BFLOAT16, s)) return 1;
    small_ew_kernel<<<2048, 256, 0, s>>>(m->xm, m->xm, m->ada + h /*shift*/, m->ada /*scale*/, (int64_t)s_img * h, h, 1);
    OMX_LAUNCH_CHECK();
    if (linear(m, (bf16_t*)out, m->xm, "proj_out.weight", s_img, c.in_channels, h)) return 1;
    OMX_HIP_CHECK(hipEventRecord(m->ev1, s));
    OMX_HIP_CHECK(hipStreamSynchronize(s));
    OMX_HIP_CHECK(hipEventElapsedTime(&m->last_ms, m->ev0, m->ev1));
    return 0;
}

int omx_klein_euler_step(void* latent_f32, const void* v_bf16, float dt, void* latent_bf16, int64_t n, omx_stream stream) {
    OMX_REQUIRE(latent_f32 && v_bf16, "omx_klein_euler_step: null tensor");
    if (n == 0) return 0;
    euler_step_kernel<<<1024, 256, 0, (hipStream_t)stream>>>((float*)latent_f32, (const bf16_t*)v_bf16, dt, (bf16_t*)latent_bf16, n);
    OMX_LAUNCH_CHECK();
    return 0;
}

int omx_klein_last_ms(omx_klein m, float* ms) {
    OMX_REQUIRE(m && ms, "omx_klein_last_ms: null argument");
    *ms = m->last_ms;
    return 0;
}

int omx_klein_debug_read(omx_klein m, const char* name, void* host, size_t n_elems) {
    OMX_REQUIRE(m && name && host, "omx_klein_debug_read: null argument");
    const std::string s(name);
    const void* src = s == "x" ? m->x : s == "x2" ? m->x2 : s == "vec" ? m->vec : s == "mod_img" ? m->mod_img : nullptr;
    OMX_REQUIRE(src != nullptr, "omx_klein_debug_read: unknown buffer %s", name);
    OMX_HIP_CHECK(hipMemcpy(host, src, n_elems * 2, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
