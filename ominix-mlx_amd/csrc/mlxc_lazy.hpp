// Deferred execution behind the mlx-c boundary (round 6; included by mlxc.hip after its helpers).
//
// The ABI is lazy BY CONTRACT: an op returns an array whose value is only guaranteed after mlx_eval / mlx_async_eval / item / data
// (mlx-c transforms.h:30,42; mlx-rs/src/transforms/mod.rs:67-85), and MLX itself runs the decode step as a graph it may fuse
// (nn::silu is `compile`d, mlx-rs/src/nn/activation.rs:876-880).  Round 5 launched at every call: an unmodified qwen3-mlx made 1 384
// eager calls per token at 4.7 us of host time each and got 0.46 of the fused engine.  Here the hot ops of the four callers only
// RECORD (result buffer allocated, shape known, launch closure + typed operands kept); the list is executed
//   * at mlx_eval / mlx_async_eval / mlx_array_eval / mlx_synchronize / item / data, and
//   * before ANY other access to device data: Arr::ptr() is the one gate to a buffer's address, and it flushes first unless the caller
//     is a recorded launch itself -- an op that was never taught to defer stays correct without knowing this file exists.
// Before the launches go out a peephole pass rewrites the decode idioms onto the GEMV family the engine uses (gemv.hpp):
//     rms_norm(x) -> x @ W^T (M == 1)                          RMSNorm prologue (the norm launch disappears when every reader fused it)
//     h + (o @ W^T)                                            residual epilogue
//     ((g * sigmoid(g)) * (x @ Wu^T)), g = x @ Wg^T            gate / up pair with the SwiGLU epilogue (nn::silu roundings)
//     argmax(x @ W^T, axis = -1)                               logits + per-block argmax keys + a tiny finalise launch
//     q_norm / k_norm -> rope -> cache slice_update (k, v)     the engine's one prompt-pass launch for all of it (prefill.hpp), and the
//                                                              three projections before it as ONE row-stacked GEMV launch
// An intermediate may only vanish when NOTHING outside the pending list can read it: its buffer's reference count must equal the
// references the pending records hold (any live handle, view or vector entry makes it externally visible and it is computed as recorded).
#pragma once

#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <thread>
#include <unordered_map>

#include "gemv.hpp"
#include "prefill.hpp"
#include "quant.hpp"

namespace {

enum RecKind { RK_GENERIC = 0, RK_RMSNORM, RK_MATMUL, RK_ADD, RK_MUL, RK_SIGMOID, RK_ARGMAX, RK_ROPE, RK_SLICE_UPDATE, RK_SDPA, RK_QMM };

struct Rec {
    int kind = RK_GENERIC;
    // operands: a[0] = the result, a[1..] = inputs (views share their buffers: that is what keeps them alive and what the pass counts)
    Arr a[5];
    int na = 0;
    float f0 = 0.f, f1 = 0.f;
    int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
    bool flag = false;            // RK_MATMUL: the M == 1 bf16 NT form launch_gemv serves (a[1] = x row, a[2] = W^T view of [N, K]);
                                  // RK_QMM: the same against an MLX-packed matrix (a[2] packed words, a[3] scales, a[4] biases; i3 = group | bits << 16)
    std::vector<int> iv;
    std::function<int(Rec&)> run;
    bool dead = false;
    bool held = false;            // its operands are counted in their buffers' `inflight` (a deferred record, until it is dropped)
    Rec() = default;
    Rec(Rec&&) = default;
    Rec& operator=(Rec&&) = default;
    Rec(const Rec&) = delete;
    Rec& operator=(const Rec&) = delete;
    void hold() {
        for (int k = 0; k < na; ++k)
            if (a[k].buf) a[k].buf->inflight.fetch_add(1);
        held = true;
    }
    ~Rec() {
        if (!held) return;
        for (int k = 0; k < na; ++k)
            if (a[k].buf) a[k].buf->inflight.fetch_sub(1);     // (a moved-from record holds no buffers any more)
    }
};

std::vector<Rec> g_pending;
bool g_lazy_on = true, g_lazy_env_read = false;
int g_fuse_mode = -1;             // -1: OMX_MLX_FUSE at the first flush; 0 / 1 set through omx_mlx_lazy_mode
long g_lazy_stats[6] = {0, 0, 0, 0, 0, 0};   // recorded, launched, fused launches, flushes, ns spent in flushes (host), ns of them in the peephole pass

bool lazy_enabled() {
    if (!g_lazy_env_read) {
        const char* e = getenv("OMX_MLX_LAZY");
        g_lazy_on = !(e && e[0] == '0');
        g_lazy_env_read = true;
    }
    return g_lazy_on;
}

// execute or defer.  A recorded op that runs another lazy-aware op from inside its launch closure executes it on the spot.
int record(Rec&& r) {
    if (!lazy_enabled() || g_lazy_busy > 0) {
        ++g_lazy_busy;
        const int rc = r.run(r);
        --g_lazy_busy;
        return rc;
    }
    ++g_lazy_stats[0];
    r.hold();
    if (r.na && r.a[0].buf) r.a[0].buf->seq = g_flush_seq + 1;      // the batch this record will go out with
    g_pending.push_back(std::move(r));
    g_n_pending = g_pending.size();
    if (g_pending.size() >= 8192) return submit_pending();   // (an unbounded list is memory held, not work saved)
    return 0;
}

bool same_view(const Arr& x, const Arr& y) {
    return x.buf.get() == y.buf.get() && x.off == y.off && x.size() == y.size();
}

__global__ void lazy_argmax_finalize_kernel(uint32_t* out, const unsigned long long* keys, int n) {
    __shared__ unsigned long long sm[256];
    unsigned long long b = 0;
    for (int i = threadIdx.x; i < n; i += 256) b = keys[i] > b ? keys[i] : b;
    sm[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] = sm[threadIdx.x + o] > sm[threadIdx.x] ? sm[threadIdx.x + o] : sm[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = ~(uint32_t)(sm[0] & 0xFFFFFFFFull);
}

struct ScatterPlan {       // q_norm + k_norm + rope (q, k) + the two cache writes of one attention layer as one launch (prefill.hpp)
    int nq = -1, nk = -1, rq = -1, rk = -1, uk = -1, uv = -1;
    int T = 0, H = 0, Hkv = 0, D = 0, cap = 0, offset = 0;
    std::shared_ptr<Arr> qkv;      // set by a stacked q | k | v GEMV plan: the projections live there, not in their recorded buffers
    int nq_rows = 0, nk_rows = 0;
};
struct FusePlan {          // one launch_gemv replacing the records it absorbed; launched at the position of record `at`
    int at = -1;
    int mmk = -1, mmv = -1;        // row-stacked q | k | v launch: mm0 = q, these = k, v; the result goes to *stack_out
    std::shared_ptr<Arr> stack_out;
    bool is_scatter = false;
    ScatterPlan sc;
    bool gemm = false;             // the records are prompt-pass products (many rows): the 256^2 GEMM family's epilogues and segment stacking
    int mm0 = -1, mm1 = -1;        // the matmul record(s): mm1 = the up projection of a SwiGLU pair
    int norm = -1;                 // RMSNorm record feeding them (prologue), or -1
    int epi = omx::EPI_STORE;
    int tail = -1;                 // the add / multiply / argmax record whose result this launch writes (-1: the matmul's own)
    int resid_operand = 0;         // which input of the add is the residual
};

// cos / sin tables of fast::rope's angles (position * scale * base^(-i / half), evaluated in fp64 exactly as elementwise.hip's rope_kernel
// and the engine's table do), grown on demand
__global__ void lazy_rope_table_kernel(float* cos_t, float* sin_t, int cap, int half, double neg_log_base_over_half, double scale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cap * half) return;
    const int t = idx / half, i = idx % half;
    const double ang = ((double)t * scale) * exp((double)i * neg_log_base_over_half);
    double sn, cs;
    sincos(ang, &sn, &cs);
    cos_t[idx] = (float)cs;
    sin_t[idx] = (float)sn;
}
struct RopeTable { float base, scale; int half, cap; float *cos_t, *sin_t; };
std::vector<RopeTable> g_rope_tables;
int rope_table(float base, float scale, int half, int need, const float** cos_t, const float** sin_t) {
    for (auto& t : g_rope_tables)
        if (t.base == base && t.scale == scale && t.half == half) {
            if (t.cap < need) {
                // (the old tables may still be read by launches in flight: they are left to the stream's order and freed behind it)
                OMX_HIP_CHECK(hipStreamSynchronize(g_stream));
                (void)hipFree(t.cos_t); (void)hipFree(t.sin_t);
                t.cap = 0; t.cos_t = t.sin_t = nullptr;
            } else {
                *cos_t = t.cos_t; *sin_t = t.sin_t;
                return 0;
            }
        }
    int cap = 4096;
    while (cap < need) cap *= 2;
    RopeTable* t = nullptr;
    for (auto& e : g_rope_tables)
        if (e.base == base && e.scale == scale && e.half == half) t = &e;
    if (!t) { g_rope_tables.push_back(RopeTable{base, scale, half, 0, nullptr, nullptr}); t = &g_rope_tables.back(); }
    OMX_HIP_CHECK(hipMalloc((void**)&t->cos_t, (size_t)cap * half * 4));
    OMX_HIP_CHECK(hipMalloc((void**)&t->sin_t, (size_t)cap * half * 4));
    const int n = cap * half;
    lazy_rope_table_kernel<<<(n + 255) / 256, 256, 0, g_stream>>>(t->cos_t, t->sin_t, cap, half, -log((double)base) / (double)half, (double)scale);
    OMX_LAUNCH_CHECK();
    t->cap = cap;
    *cos_t = t->cos_t; *sin_t = t->sin_t;
    return 0;
}

int run_scatter(std::vector<Rec>& recs, const ScatterPlan& p) {
    using namespace omx;
    const Rec &nq = recs[p.nq], &nk = recs[p.nk], &rq = recs[p.rq], &uk = recs[p.uk], &uv = recs[p.uv];
    const float *cs = nullptr, *sn = nullptr;
    if (rope_table(rq.f0, rq.f1, p.D / 2, p.offset + p.T, &cs, &sn)) return 1;
    const bf16_t *ql, *kl, *vl;
    if (p.qkv) {
        ql = (const bf16_t*)p.qkv->ptr();
        kl = ql + p.nq_rows;
        vl = kl + p.nk_rows;
    } else {
        ql = (const bf16_t*)nq.a[1].ptr(); kl = (const bf16_t*)nk.a[1].ptr(); vl = (const bf16_t*)uv.a[1].ptr();
    }
    bf16_t* kc = (bf16_t*)uk.a[0].ptr() - (size_t)p.offset * p.D;
    bf16_t* vc = (bf16_t*)uv.a[0].ptr() - (size_t)p.offset * p.D;
    return launch_qk_norm_rope_scatter(ql, kl, vl, (const bf16_t*)nq.a[2].ptr(), (const bf16_t*)nk.a[2].ptr(), cs, sn, (bf16_t*)rq.a[0].ptr(), kc, vc,
                                       p.T, p.H, p.Hkv, p.D, p.cap, p.offset, nq.f0, g_stream, false);
}

// the derived forms of a packed matrix the fused kernels want, built once per weight buffer and kept WITH it (Buf::aux): the scale | bias
// words quant.hip's kernels fetch with one load, and -- 4-bit, group 64, K = 4096 / 12288 -- the matrix-core tiles of qgemv_mfma.hip
struct QAux {
    uint32_t* sb = nullptr;
    uint32_t* tiles = nullptr;
    ~QAux() { if (sb) (void)hipFree(sb); if (tiles) (void)hipFree(tiles); }
};
int qmat_of(const Rec& m, omx::QMat* out) {
    using namespace omx;
    const int N = m.i0, K = m.i1, group = m.i3 & 0xFFFF, bits = m.i3 >> 16;
    const Arr& w = m.a[2];
    *out = QMat{(const uint32_t*)w.ptr(), (const bf16_t*)m.a[3].ptr(), m.na > 4 ? (const bf16_t*)m.a[4].ptr() : nullptr, N};
    if (w.off != 0 || m.a[3].off != 0) return 0;                 // (a view into a larger buffer: no cache keyed by the buffer)
    if (!w.buf->aux) {
        auto aux = std::make_shared<QAux>();
        const size_t ng = (size_t)N * (K / group);
        if (K % 2048 == 0) {
            OMX_HIP_CHECK(hipMalloc((void**)&aux->sb, ng * 4));
            if (launch_quant_interleave(aux->sb, out->scales, out->biases, ng, g_stream)) return 1;
        }
        if (qgemv4m_shape_ok(K, group, bits) && N >= 16) {
            OMX_HIP_CHECK(hipMalloc((void**)&aux->tiles, qgemv4m_tile_words(N, K) * 4));
            if (launch_qgemv4m_repack(aux->tiles, out->w, out->scales, out->biases, N, K, g_stream)) return 1;
        }
        w.buf->aux = aux;
    }
    const QAux* a = static_cast<const QAux*>(w.buf->aux.get());
    out->sb = a->sb;
    out->tiles = a->tiles;
    return 0;
}

int run_plan_packed(std::vector<Rec>& recs, const FusePlan& p) {
    using namespace omx;
    const Rec& m0 = recs[p.mm0];
    const int K = m0.i1, N = m0.i0, group = m0.i3 & 0xFFFF, bits = m0.i3 >> 16;
    QGemvArgs g = {};
    if (qmat_of(m0, &g.m[0])) return 1;
    g.N = N; g.K = K; g.group = group;
    int pro = PRO_NONE;
    if (p.norm >= 0) {
        const Rec& nr = recs[p.norm];
        g.x = (const bf16_t*)nr.a[1].ptr();
        g.norm_w = (const bf16_t*)nr.a[2].ptr();
        g.eps = nr.f0;
        pro = PRO_RMSNORM;
    } else {
        g.x = (const bf16_t*)m0.a[1].ptr();
    }
    if (p.stack_out) {
        const Rec &mk = recs[p.mmk], &mv = recs[p.mmv];
        if (qmat_of(mk, &g.m[1]) || qmat_of(mv, &g.m[2])) return 1;
        g.N = N + mk.i0 + mv.i0;
        Arr* t = new_arr({g.N}, MLX_BFLOAT16);
        if (!t) return set_error("deferred q | k | v projection: out of device memory");
        *p.stack_out = *t;
        delete t;
        g.out = (bf16_t*)p.stack_out->ptr();
        return launch_qgemv(g, bits, pro, EPI_STORE, g_stream);
    }
    if (p.epi == EPI_SWIGLU) {
        if (qmat_of(recs[p.mm1], &g.m[1])) return 1;
        g.out = (bf16_t*)recs[p.tail].a[0].ptr();
    } else if (p.epi == EPI_RESIDUAL) {
        g.resid = (const bf16_t*)recs[p.tail].a[p.resid_operand].ptr();
        g.out = (bf16_t*)recs[p.tail].a[0].ptr();
    } else if (p.epi == EPI_ARGMAX && p.norm < 0) {
        g.out = (bf16_t*)m0.a[0].ptr();
        if (launch_qgemv(g, bits, PRO_NONE, EPI_STORE, g_stream)) return 1;
        return recs[p.tail].run(recs[p.tail]);
    } else if (p.epi == EPI_ARGMAX) {
        const int nslot = qgemv_grid(N);
        Arr* s = new_arr({nslot * 2}, MLX_UINT32);
        if (!s) return set_error("deferred lm_head: out of device memory");
        Arr slots = *s;
        delete s;
        g.argmax_slot = (unsigned long long*)slots.ptr();
        g.out = (bf16_t*)m0.a[0].ptr();
        if (launch_qgemv(g, bits, PRO_RMSNORM, EPI_ARGMAX, g_stream)) return 1;
        lazy_argmax_finalize_kernel<<<1, 256, 0, g_stream>>>((uint32_t*)recs[p.tail].a[0].ptr(), (const unsigned long long*)slots.ptr(), nslot);
        OMX_LAUNCH_CHECK();
        return 0;
    } else {
        g.out = (bf16_t*)m0.a[0].ptr();
    }
    return launch_qgemv(g, bits, pro, p.epi, g_stream);
}

int run_plan(std::vector<Rec>& recs, const FusePlan& p) {
    using namespace omx;
    if (p.is_scatter) return run_scatter(recs, p.sc);
    if (recs[p.mm0].kind == RK_QMM) return run_plan_packed(recs, p);
    if (p.gemm) {
        const Rec& m0 = recs[p.mm0];
        const int N = m0.i0, K = m0.i1, M = m0.i2;
        const bf16_t* x = (const bf16_t*)m0.a[1].ptr();
        if (p.stack_out) {                      // two or three products of one activation: one segmented launch, each into its own result
            GemmSegs sg = {};
            const int idx[3] = {p.mm0, p.mmk, p.mmv};
            for (int i = 0; i < 3; ++i)
                if (idx[i] >= 0) {
                    const Rec& m = recs[idx[i]];
                    sg.plain[sg.n_plain++] = GemmSeg{(const bf16_t*)m.a[2].ptr(), nullptr, (bf16_t*)m.a[0].ptr(), m.i0, m.i0, 0};
                }
            return launch_gemm_bf16_segmented(x, M, K, sg, g_stream);
        }
        if (p.epi == EPI_SWIGLU) {              // act = silu(x Wg^T) * (x Wu^T) with nn::silu's roundings, the gate / up pair never in HBM
            GemmSegs sg = {};
            sg.w_gate = (const bf16_t*)m0.a[2].ptr();
            sg.w_up = (const bf16_t*)recs[p.mm1].a[2].ptr();
            sg.out_act = (bf16_t*)recs[p.tail].a[0].ptr();
            sg.half = N; sg.ld_act = N; sg.act_mode = 1;
            return launch_gemm_bf16_segmented(x, M, K, sg, g_stream);
        }
        if (p.epi == EPI_RESIDUAL)
            return launch_gemm_bf16_ex((bf16_t*)recs[p.tail].a[0].ptr(), x, (const bf16_t*)m0.a[2].ptr(), nullptr,
                                       (const bf16_t*)recs[p.tail].a[p.resid_operand].ptr(), M, N, K, g_stream);
        return launch_gemm_bf16((bf16_t*)m0.a[0].ptr(), x, (const bf16_t*)m0.a[2].ptr(), nullptr, M, N, K, g_stream);
    }
    const Rec& m0 = recs[p.mm0];
    GemvArgs g = {};
    const int K = m0.i1, N = m0.i0;
    g.w0 = (const bf16_t*)m0.a[2].ptr();
    g.n0 = N; g.N = N; g.K = K;
    if (p.norm >= 0) {
        const Rec& nr = recs[p.norm];
        g.x = (const bf16_t*)nr.a[1].ptr();
        g.norm_w = (const bf16_t*)nr.a[2].ptr();
        g.eps = nr.f0;
    } else {
        g.x = (const bf16_t*)m0.a[1].ptr();
    }
    Arr slots;
    if (p.stack_out) {         // q | k | v rows in one launch, into one fresh buffer
        const Rec &mk = recs[p.mmk], &mv = recs[p.mmv];
        g.w1 = (const bf16_t*)mk.a[2].ptr(); g.n1 = mk.i0;
        g.w2 = (const bf16_t*)mv.a[2].ptr(); g.n2 = mv.i0;
        g.N = N + mk.i0 + mv.i0;
        Arr* t = new_arr({g.N}, MLX_BFLOAT16);
        if (!t) return set_error("deferred q | k | v projection: out of device memory");
        *p.stack_out = *t;
        delete t;
        g.out = p.stack_out->ptr();
        return launch_gemv(g, p.norm >= 0 ? PRO_RMSNORM : PRO_NONE, EPI_STORE, g_stream);
    }
    if (p.epi == EPI_SWIGLU) {
        g.w1 = (const bf16_t*)recs[p.mm1].a[2].ptr();
        g.n1 = N;
        g.out = recs[p.tail].a[0].ptr();
    } else if (p.epi == EPI_RESIDUAL) {
        g.resid = (const bf16_t*)recs[p.tail].a[p.resid_operand].ptr();
        g.out = recs[p.tail].a[0].ptr();
    } else if (p.epi == EPI_ARGMAX && p.norm < 0) {    // (no argmax epilogue without the norm prologue in the family: product, then the recorded argmax)
        g.out = m0.a[0].ptr();
        if (launch_gemv(g, PRO_NONE, EPI_STORE, g_stream)) return 1;
        return recs[p.tail].run(recs[p.tail]);
    } else if (p.epi == EPI_ARGMAX) {
        const int nslot = gemv_grid(N, K, EPI_ARGMAX, 0);
        Arr* s = new_arr({nslot * 2}, MLX_UINT32);
        if (!s) return set_error("deferred lm_head: out of device memory");
        slots = *s;
        delete s;
        g.argmax_slot = (unsigned long long*)slots.ptr();
        g.out = m0.a[0].ptr();                       // the logits row is written too (it may be read later)
        if (launch_gemv(g, p.norm >= 0 ? PRO_RMSNORM : PRO_NONE, EPI_ARGMAX, g_stream)) return 1;
        lazy_argmax_finalize_kernel<<<1, 256, 0, g_stream>>>((uint32_t*)recs[p.tail].a[0].ptr(), (const unsigned long long*)slots.ptr(), nslot);
        OMX_LAUNCH_CHECK();
        return 0;
    } else {
        g.out = m0.a[0].ptr();
    }
    return launch_gemv(g, p.norm >= 0 ? PRO_RMSNORM : PRO_NONE, p.epi, g_stream);
}

// the peephole pass: fills `plans` (keyed by the record index that launches them) and marks absorbed records dead
void fuse_pending(std::vector<Rec>& recs, std::unordered_map<int, FusePlan>& plans) {
    using namespace omx;
    const int n = (int)recs.size();
    // one flat table over the buffers the list touches (this pass runs once per decoded token on ~800 records: node-based maps were a third
    // of a millisecond): per buffer the references the list holds, the one record that writes it (-2: several), the last slice_update into it,
    // and its readers as a slice of one array
    struct BufInfo { const Buf* b; int nref, prod, writer, rd_begin, rd_n; };
    size_t cap = 64;
    while (cap < (size_t)n * 8) cap <<= 1;
    static thread_local std::vector<int> slot_of;        // hash slot -> index into `info` (+1; 0 = empty)
    static thread_local std::vector<BufInfo> info;
    static thread_local std::vector<int> rd, opid;
    slot_of.assign(cap, 0);
    info.clear();
    opid.assign((size_t)n * 5, -1);
    auto find = [&](const Buf* b) -> int {
        size_t h = ((uintptr_t)b >> 4) * 0x9E3779B97F4A7C15ull >> 20 & (cap - 1);
        while (slot_of[h]) {
            if (info[slot_of[h] - 1].b == b) return slot_of[h] - 1;
            h = (h + 1) & (cap - 1);
        }
        return -1;
    };
    auto intern = [&](const Buf* b) -> int {
        size_t h = ((uintptr_t)b >> 4) * 0x9E3779B97F4A7C15ull >> 20 & (cap - 1);
        while (slot_of[h]) {
            if (info[slot_of[h] - 1].b == b) return slot_of[h] - 1;
            h = (h + 1) & (cap - 1);
        }
        info.push_back(BufInfo{b, 0, -1, -1, 0, 0});
        slot_of[h] = (int)info.size();
        return (int)info.size() - 1;
    };
    bool any_mm = false;
    for (int i = 0; i < n; ++i) {
        const Rec& r = recs[i];
        for (int k = 0; k < r.na; ++k)
            if (r.a[k].buf) {
                const int id = intern(r.a[k].buf.get());
                opid[(size_t)i * 5 + k] = id;
                ++info[id].nref;
                if (k == 0) {
                    info[id].prod = info[id].prod == -1 ? i : -2;
                    if (r.kind == RK_SLICE_UPDATE) info[id].writer = i;
                } else {
                    ++info[id].rd_n;          // (an upper bound: the fill below lists a record once per buffer)
                }
            }
        any_mm = any_mm || ((r.kind == RK_MATMUL || r.kind == RK_QMM) && r.flag) || (r.kind == RK_MATMUL && r.i3 == 1);
    }
    if (!any_mm) return;
    {
        int at = 0;
        for (auto& bi : info) { bi.rd_begin = at; at += bi.rd_n; bi.rd_n = 0; }
        rd.assign((size_t)at, -1);
        for (int i = 0; i < n; ++i) {
            const Rec& r = recs[i];
            for (int k = 1; k < r.na; ++k) {
                const int id = opid[(size_t)i * 5 + k];
                if (id < 0) continue;
                BufInfo& bi = info[id];
                if (bi.rd_n && rd[bi.rd_begin + bi.rd_n - 1] == i) continue;
                rd[bi.rd_begin + bi.rd_n++] = i;
            }
        }
    }
    auto internal = [&](const Arr& x) {   // nothing outside the pending list can see this buffer
        if (!x.buf || !x.buf->owned) return false;
        const int id = find(x.buf.get());
        return id >= 0 && (long)x.buf.use_count() == (long)info[id].nref;
    };
    auto producer = [&](const Arr& x, int before) -> int {
        if (!x.buf) return -1;
        const int id = find(x.buf.get());
        if (id < 0 || info[id].prod < 0 || info[id].prod >= before) return -1;
        return same_view(recs[info[id].prod].a[0], x) ? info[id].prod : -1;
    };
    auto producer_of_buf = [&](const Arr& x) -> int {      // the record whose result this is a re-shaped / transposed view of
        if (!x.buf) return -1;
        const int id = find(x.buf.get());
        if (id < 0 || info[id].prod < 0) return -1;
        const Arr& o = recs[info[id].prod].a[0];
        return (o.off == x.off && o.size() == x.size()) ? info[id].prod : -1;
    };
    auto readers_of = [&](const Arr& x, const int** first) -> int {
        const int id = x.buf ? find(x.buf.get()) : -1;
        if (id < 0) { *first = nullptr; return 0; }
        *first = rd.data() + info[id].rd_begin;
        return info[id].rd_n;
    };
    auto only_readers = [&](const Arr& x, std::initializer_list<int> who) {
        const int* first = nullptr;
        const int cnt = readers_of(x, &first);
        for (int t = 0; t < cnt; ++t) {
            bool ok = false;
            for (int w : who) ok = ok || first[t] == w;
            if (!ok) return false;
        }
        return true;
    };
    auto writer_of = [&](const Arr& x) -> int {
        const int id = x.buf ? find(x.buf.get()) : -1;
        return id < 0 ? -1 : info[id].writer;
    };
    auto gemv_rec = [&](int i) { return i >= 0 && (recs[i].kind == RK_MATMUL || recs[i].kind == RK_QMM) && recs[i].flag && !recs[i].dead; };
    // OMX_MLX_FUSE_GEMM: bit 0 segment stacking, bit 1 SwiGLU pair, bit 2 residual epilogue of the prompt-pass products (default 7; 0: none)
    static const int gemm_mask = [] { const char* e = getenv("OMX_MLX_FUSE_GEMM"); return e ? atoi(e) : 7; }();
    auto gemm_rec = [&](int i) { return gemm_mask != 0 && i >= 0 && recs[i].kind == RK_MATMUL && recs[i].i3 == 1 && !recs[i].dead; };
    auto mm_rec = [&](int i) { return gemv_rec(i) || gemm_rec(i); };
    std::vector<int> plan_of(n, -1);       // matmul record -> the record index its plan is keyed by
    // ---- epilogues, found from the record that ends the idiom ----
    for (int j = 0; j < n; ++j) {
        Rec& r = recs[j];
        if (r.dead) continue;
        if (r.kind == RK_MUL && r.flag) {
            // act = (g * sigmoid(g)) * u, either operand order at both levels
            for (int sw = 0; sw < 2; ++sw) {
                const int m2 = producer(r.a[1 + sw], j), U = producer(r.a[2 - sw], j);
                if (m2 < 0 || !mm_rec(U) || recs[m2].kind != RK_MUL || !recs[m2].flag || recs[m2].dead) continue;
                int G = -1, S = -1;
                for (int sw2 = 0; sw2 < 2 && G < 0; ++sw2) {
                    const int g0 = producer(recs[m2].a[1 + sw2], m2), s0 = producer(recs[m2].a[2 - sw2], m2);
                    if (mm_rec(g0) && gemm_rec(g0) == gemm_rec(U) && s0 >= 0 && recs[s0].kind == RK_SIGMOID && !recs[s0].dead && same_view(recs[s0].a[1], recs[g0].a[0])) { G = g0; S = s0; }
                }
                if (G < 0 || G == U || plan_of[G] >= 0 || plan_of[U] >= 0) continue;
                if (gemm_rec(G) && !(gemm_mask & 2)) continue;
                const Rec &rg = recs[G], &ru = recs[U];
                if (rg.i0 != ru.i0 || rg.i1 != ru.i1 || rg.kind != ru.kind || rg.i3 != ru.i3 || !same_view(rg.a[1], ru.a[1])) continue;
                if (rg.i0 % 4 != 0) continue;
                if (!internal(rg.a[0]) || !internal(ru.a[0]) || !internal(recs[S].a[0]) || !internal(recs[m2].a[0])) continue;
                if (!only_readers(rg.a[0], {S, m2}) || !only_readers(ru.a[0], {j}) || !only_readers(recs[S].a[0], {m2}) || !only_readers(recs[m2].a[0], {j})) continue;
                FusePlan p;
                p.at = j; p.mm0 = G; p.mm1 = U; p.epi = EPI_SWIGLU; p.tail = j; p.gemm = gemm_rec(G);
                plans[j] = p;
                plan_of[G] = plan_of[U] = j;
                recs[G].dead = recs[U].dead = recs[S].dead = recs[m2].dead = true;
                break;
            }
        } else if (r.kind == RK_ADD && r.flag) {
            for (int sw = 0; sw < 2; ++sw) {
                const int M = producer(r.a[2 - sw], j);
                if (!mm_rec(M) || plan_of[M] >= 0) continue;
                if (gemm_rec(M) && !(gemm_mask & 4)) continue;
                if (!internal(recs[M].a[0]) || !only_readers(recs[M].a[0], {j})) continue;
                if (r.a[1 + sw].size() != (size_t)recs[M].i0 * (size_t)recs[M].i2) continue;
                FusePlan p;
                p.at = j; p.mm0 = M; p.epi = EPI_RESIDUAL; p.tail = j; p.resid_operand = 1 + sw; p.gemm = gemm_rec(M);
                plans[j] = p;
                plan_of[M] = j;
                recs[M].dead = true;
                break;
            }
        } else if (r.kind == RK_ARGMAX && r.flag) {
            const int M = producer(r.a[1], j);
            if (!gemv_rec(M) || plan_of[M] >= 0 || !only_readers(recs[M].a[0], {j})) continue;
            FusePlan p;
            p.at = j; p.mm0 = M; p.epi = EPI_ARGMAX; p.tail = j;
            plans[j] = p;
            plan_of[M] = j;
            recs[M].dead = true;
        }
    }
    // ---- the remaining M == 1 products launch as plain GEMVs at their own position (so that they can take a prologue) ----
    for (int i = 0; i < n; ++i)
        if (mm_rec(i) && plan_of[i] < 0) {
            FusePlan p;
            p.at = i; p.mm0 = i; p.epi = EPI_STORE; p.gemm = gemm_rec(i);
            plans[i] = p;
            plan_of[i] = i;
        }
    // ---- prompt pass: the plain products of ONE activation (q / k / v) as one segmented launch, each still into its own result ----
    for (int i = 0; i < n; ++i) {
        if (!(gemm_mask & 1) || !gemm_rec(i) || plan_of[i] != i || plans[i].epi != EPI_STORE || plans[i].stack_out) continue;
        int grp[3] = {i, -1, -1}, cnt = 1;
        for (int j2 = i + 1; j2 < n && j2 <= i + 24 && cnt < 3; ++j2)
            if (gemm_rec(j2) && plan_of[j2] == j2 && plans[j2].epi == EPI_STORE && !plans[j2].stack_out && same_view(recs[j2].a[1], recs[i].a[1]) &&
                recs[j2].i1 == recs[i].i1 && recs[j2].i2 == recs[i].i2)
                grp[cnt++] = j2;
        // the launch sits where the LAST product was recorded: nothing recorded before that point may read an earlier member's result
        // (gate / up with the SwiGLU rewrite switched off: sigmoid(gate) is recorded between the two products)
        while (cnt >= 2) {
            bool early_reader = false;
            for (int c = 0; c + 1 < cnt && !early_reader; ++c) {
                const int* first = nullptr;
                const int nr = readers_of(recs[grp[c]].a[0], &first);
                for (int t = 0; t < nr; ++t) early_reader = early_reader || first[t] <= grp[cnt - 1];
            }
            if (!early_reader) break;
            --cnt;                               // drop the last member and try the shorter group
        }
        if (cnt < 2) continue;
        // the activation must not be rewritten between the first and the last of them (only an in-place cache update could, and it never targets x)
        GemmSegs probe = {};
        for (int c = 0; c < cnt; ++c) probe.plain[probe.n_plain++] = GemmSeg{nullptr, nullptr, nullptr, recs[grp[c]].i0, recs[grp[c]].i0, 0};
        if (!gemm_segmented_supported(recs[i].i2, recs[i].i1, probe)) continue;
        const int last = grp[cnt - 1];
        FusePlan st = plans[i];
        for (int c = 0; c < cnt; ++c) plans.erase(grp[c]);
        st.at = last; st.mm0 = grp[0]; st.mmk = grp[1]; st.mmv = cnt > 2 ? grp[2] : -1;
        st.stack_out = std::make_shared<Arr>();           // (marks the stacked form; the results stay in the records' own buffers)
        plans[last] = st;
        for (int c = 0; c < cnt; ++c) {
            plan_of[grp[c]] = last;
            if (grp[c] != last) recs[grp[c]].dead = true;
        }
    }
    // ---- attention preparation: anchored at the SDPA record.  q <- rope <- rms_norm(q_norm) <- [1, H, T, D] view of the q projection;
    //      the k operand's buffer is written by a slice_update whose update is rope <- rms_norm(k_norm) <- view of the k projection, the
    //      v operand's by a slice_update of the v projection's view (KVCache::update_and_fetch, cache.rs:140-193) ----
    auto headed_view = [&](const Arr& v, int heads, int T, int D) {   // [1, heads, T, D] over row-major [1, T, heads * D]
        return v.shape.size() == 4 && v.shape[0] == 1 && v.shape[1] == heads && v.shape[2] == T && v.shape[3] == D && v.strides[3] == 1 &&
               v.strides[1] == (size_t)D && (T == 1 || v.strides[2] == (size_t)heads * D) && v.dt == MLX_BFLOAT16;
    };
    for (int j = 0; j < n; ++j) {
        Rec& sd = recs[j];
        if (sd.kind != RK_SDPA || sd.dead) continue;
        const Arr &q2 = sd.a[1], &kv = sd.a[2], &vv = sd.a[3];
        if (q2.shape.size() != 4 || q2.dt != MLX_BFLOAT16) continue;
        const int H = q2.shape[1], T = q2.shape[2], D = q2.shape[3], Hkv = kv.shape[1];
        if ((D != 64 && D != 128) || q2.shape[0] != 1) continue;
        ScatterPlan sp;
        sp.rq = producer(q2, j);
        sp.uk = writer_of(kv); sp.uv = writer_of(vv);
        if (sp.rq < 0 || sp.uk < 0 || sp.uv < 0) continue;
        if (sp.uk >= j || sp.uv >= j || recs[sp.uk].dead || recs[sp.uv].dead) continue;
        const Rec &rq = recs[sp.rq], &uk = recs[sp.uk], &uv = recs[sp.uv];
        if (rq.kind != RK_ROPE || !rq.flag || rq.dead || !is_contig(rq.a[0])) continue;
        sp.rk = producer(uk.a[1], sp.uk);
        if (sp.rk < 0 || recs[sp.rk].kind != RK_ROPE || !recs[sp.rk].flag || recs[sp.rk].dead) continue;
        const Rec& rk = recs[sp.rk];
        sp.nq = producer(rq.a[1], sp.rq);
        sp.nk = producer(rk.a[1], sp.rk);
        if (sp.nq < 0 || sp.nk < 0) continue;
        const Rec &nq = recs[sp.nq], &nk = recs[sp.nk];
        if (nq.kind != RK_RMSNORM || nk.kind != RK_RMSNORM || nq.dead || nk.dead || nq.na < 3 || nk.na < 3 || nq.i0 != D || nk.i0 != D || nq.f0 != nk.f0) continue;
        if (!is_contig(nq.a[2]) || !is_contig(nk.a[2]) || nq.a[2].dt != MLX_BFLOAT16 || nk.a[2].dt != MLX_BFLOAT16) continue;
        // rope: the plain rotate-half form over the whole head, both at the position the cache write starts at
        if (rq.i0 != D || rk.i0 != D || rq.i1 || rk.i1 || rq.i2 != rk.i2 || rq.f0 != rk.f0 || rq.f1 != rk.f1 || rq.f0 <= 0.f) continue;
        if (!headed_view(nq.a[1], H, T, D) || !headed_view(nk.a[1], Hkv, T, D) || !headed_view(uv.a[1], Hkv, T, D)) continue;
        // the cache regions: [1, Hkv, T, D] at token rq.i2 of a row-major [1, Hkv, cap, D]
        auto region_ok = [&](const Arr& r, int* cap) {
            if (r.shape.size() != 4 || r.shape[0] != 1 || r.shape[1] != Hkv || r.shape[2] != T || r.shape[3] != D || r.strides[3] != 1 ||
                r.strides[2] != (size_t)D || r.strides[1] % (size_t)D != 0 || r.dt != MLX_BFLOAT16) return false;
            *cap = (int)(r.strides[1] / (size_t)D);
            return r.off >= (size_t)rq.i2 * D * 2 && *cap >= rq.i2 + T;
        };
        int capk = 0, capv = 0;
        if (!region_ok(uk.a[0], &capk) || !region_ok(uv.a[0], &capv) || capk != capv) continue;
        if (kv.strides[1] != (size_t)capk * D || vv.strides[1] != (size_t)capk * D) continue;     // the operands ARE those caches
        if (kv.off + (size_t)rq.i2 * D * 2 != uk.a[0].off || vv.off + (size_t)rq.i2 * D * 2 != uv.a[0].off) continue;
        // nothing outside may see the values that stop existing: the two normalised rows and the rotated k
        if (!internal(nq.a[0]) || !internal(nk.a[0]) || !internal(rk.a[0])) continue;
        if (!only_readers(nq.a[0], {sp.rq}) || !only_readers(nk.a[0], {sp.rk}) || !only_readers(rk.a[0], {sp.uk})) continue;
        sp.T = T; sp.H = H; sp.Hkv = Hkv; sp.D = D; sp.cap = capk; sp.offset = rq.i2;
        const int at = std::max(std::max(sp.rq, sp.rk), std::max(sp.uk, sp.uv));
        {   // the one launch sits where the LAST of the six was recorded: nothing recorded before that may read the rotated q or the caches
            bool early = false;
            for (const Arr* x : {&rq.a[0], &uk.a[0], &uv.a[0]}) {
                const int* first = nullptr;
                const int nr = readers_of(*x, &first);
                for (int t = 0; t < nr; ++t) early = early || (first[t] <= at && first[t] != sp.uk && first[t] != sp.uv && first[t] != sp.rq && first[t] != sp.rk);
            }
            if (early) continue;
        }
        // the three projections as one row-stacked launch: plain GEMV plans on one activation, nobody else reading their rows
        const int Mq = producer_of_buf(nq.a[1]), Mk = producer_of_buf(nk.a[1]), Mv = producer_of_buf(uv.a[1]);
        if (Mq >= 0 && Mk >= 0 && Mv >= 0 && gemv_rec(Mq) && gemv_rec(Mk) && gemv_rec(Mv) && plan_of[Mq] == Mq && plan_of[Mk] == Mk && plan_of[Mv] == Mv &&
            plans[Mq].epi == EPI_STORE && plans[Mk].epi == EPI_STORE && plans[Mv].epi == EPI_STORE && same_view(recs[Mq].a[1], recs[Mk].a[1]) &&
            same_view(recs[Mq].a[1], recs[Mv].a[1]) && recs[Mq].i1 == recs[Mk].i1 && recs[Mq].i1 == recs[Mv].i1 && recs[Mq].i0 == H * D &&
            recs[Mq].kind == recs[Mk].kind && recs[Mq].kind == recs[Mv].kind && recs[Mq].i3 == recs[Mk].i3 && recs[Mq].i3 == recs[Mv].i3 &&
            recs[Mk].i0 == Hkv * D && recs[Mv].i0 == Hkv * D && T == 1 && internal(recs[Mq].a[0]) && internal(recs[Mk].a[0]) && internal(recs[Mv].a[0]) &&
            only_readers(recs[Mq].a[0], {sp.nq}) && only_readers(recs[Mk].a[0], {sp.nk}) && only_readers(recs[Mv].a[0], {sp.uv})) {
            const int last = std::max(Mq, std::max(Mk, Mv));
            FusePlan st = plans[Mq];
            plans.erase(Mq); plans.erase(Mk); plans.erase(Mv);
            st.at = last; st.mm0 = Mq; st.mmk = Mk; st.mmv = Mv;
            st.stack_out = std::make_shared<Arr>();
            sp.qkv = st.stack_out; sp.nq_rows = H * D; sp.nk_rows = Hkv * D;
            plans[last] = st;
            plan_of[Mq] = plan_of[Mk] = plan_of[Mv] = last;
            for (int m : {Mq, Mk, Mv})
                if (m != last) recs[m].dead = true;
        }
        FusePlan fp;
        fp.at = at; fp.is_scatter = true; fp.sc = sp;
        for (int d : {sp.nq, sp.nk, sp.rq, sp.rk, sp.uk, sp.uv})
            if (d != at) recs[d].dead = true;
        plans[at] = fp;
    }
    // ---- prologues: an RMSNorm whose result only GEMV plans read, all over the whole row, moves into them ----
    for (int i = 0; i < n; ++i) {
        Rec& r = recs[i];
        if (r.kind != RK_RMSNORM || !r.flag || r.dead || !internal(r.a[0])) continue;
        const int* first = nullptr;
        const int cnt = readers_of(r.a[0], &first);
        if (cnt == 0) continue;
        bool ok = true;
        for (int t = 0; t < cnt; ++t) {
            const int c = first[t];
            const Rec& m = recs[c];
            ok = ok && (m.kind == RK_MATMUL || m.kind == RK_QMM) && m.flag && plan_of[c] >= 0 && same_view(m.a[1], r.a[0]) && m.i1 == r.i0 && gemv_k_supported(m.i1, true) &&
                 plans[plan_of[c]].epi != EPI_RESIDUAL;   // (the family has no norm prologue + residual epilogue form)
        }
        if (!ok) continue;
        for (int t = 0; t < cnt; ++t) plans[plan_of[first[t]]].norm = i;
        r.dead = true;
    }
}

}  // namespace

// ---- the launch worker (OMX_MLX_ASYNC=1; measured NEGATIVE, off by default: EXPERIMENTS R6-4) ----
// Recording a decode step (~1 400 calls) and launching what is left of it (~330 kernels) are both host work of about a millisecond; on one
// thread they add up and a 4-bit step (1.6 ms on the device) waits for the host.  With the worker, mlx_async_eval only HANDS the recorded
// list to a thread that rewrites and launches it while the caller goes on recording the next step -- the overlap MLX gets from its own
// scheduler thread.  Everything else that needs the launches to have happened (an op that never learnt to defer, mlx_eval, a buffer growing)
// waits for the worker to run dry first; item() / data() wait only for the batch that produces their value.  On the pool's boxes the two
// threads slow each other down by more than the overlap buys (the launch path 1.4 -> 3.7 ms per token while the caller records beside it:
// 4-bit Qwen3-8B 439 -> 244 tok/s, bf16 unchanged because it is device-bound either way), so by default the caller's thread launches at
// the evaluation point.
namespace {
struct Batch { std::vector<Rec> recs; uint64_t seq; };
struct Worker {
    std::mutex mu;
    std::condition_variable work, done;
    std::deque<Batch> queue;
    bool busy = false, started = false;
    uint64_t issued = 0;          // batches launched so far (== seq of the last one)
    int failed = 0;
    std::string message;
};
Worker* g_worker = new Worker();  // (never destroyed: the thread may sit in its wait when the process ends)
int g_async_mode = -1;

// rewrite + launch one batch on the calling thread
int run_batch(std::vector<Rec>& recs, uint64_t seq) {
    ++g_lazy_busy;
    ++g_lazy_stats[3];
    const auto t_begin = std::chrono::steady_clock::now();
    std::unordered_map<int, FusePlan> plans;
    if (g_fuse_mode < 0) { const char* e = getenv("OMX_MLX_FUSE"); g_fuse_mode = (e && e[0] == '0') ? 0 : 1; }
    if (g_fuse_mode) fuse_pending(recs, plans);
    g_lazy_stats[5] += (long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
    int rc = 0;
    for (int i = 0; i < (int)recs.size() && !rc; ++i) {
        Rec& r = recs[i];
        auto it = plans.find(i);
        if (it != plans.end()) {
            rc = run_plan(recs, it->second);
            ++g_lazy_stats[2];
        } else if (!r.dead) {
            rc = r.run(r);
            ++g_lazy_stats[1];
        }
    }
    if (!rc) {
        hipEvent_t& ev = g_flush_ev[seq % kEvRing];
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
        if (ev) (void)hipEventRecord(ev, g_stream);
    }
    --g_lazy_busy;
    recs.clear();        // (drops the records' references: dead intermediates return to the pool here)
    g_lazy_stats[4] += (long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
    return rc;
}

void worker_main() {
    Worker& w = *g_worker;
    std::unique_lock<std::mutex> lk(w.mu);
    for (;;) {
        w.work.wait(lk, [&] { return !w.queue.empty(); });
        Batch b = std::move(w.queue.front());
        w.queue.pop_front();
        w.busy = true;
        lk.unlock();
        const int rc = run_batch(b.recs, b.seq);
        std::string msg;
        if (rc) { msg = omx_last_error(); omx_clear_error(); }
        lk.lock();
        if (rc && !w.failed) { w.failed = 1; w.message = msg; }
        w.issued = b.seq;
        w.busy = false;
        g_batches_in_flight.fetch_sub(1, std::memory_order_release);
        w.done.notify_all();
    }
}
bool async_enabled() {
    if (g_async_mode < 0) { const char* e = getenv("OMX_MLX_ASYNC"); g_async_mode = (e && e[0] == '1') ? 1 : 0; }
    return g_async_mode == 1;
}
// a failure the worker met, reported on the caller's thread (once)
int take_worker_error() {
    Worker& w = *g_worker;
    std::unique_lock<std::mutex> lk(w.mu);
    if (!w.failed) return 0;
    w.failed = 0;
    const std::string msg = w.message;
    lk.unlock();
    return set_error("%s", msg.empty() ? "a deferred op failed" : msg.c_str());
}
int drain_worker() {
    Worker& w = *g_worker;
    if (w.started) {
        std::unique_lock<std::mutex> lk(w.mu);
        w.done.wait(lk, [&] { return w.queue.empty() && !w.busy; });
    }
    return take_worker_error();
}
}  // namespace

int wait_issued(uint64_t seq) {
    Worker& w = *g_worker;
    if (w.started) {
        std::unique_lock<std::mutex> lk(w.mu);
        w.done.wait(lk, [&] { return w.issued >= seq || (w.queue.empty() && !w.busy); });
    }
    return take_worker_error();
}

// hand everything recorded so far to the launch worker (or launch it here when there is none); returns without waiting for the launches
int submit_pending() {
    if (g_lazy_busy > 0) return 0;
    if (g_pending.empty()) return 0;
    Batch b;
    b.recs.swap(g_pending);
    g_n_pending = 0;
    b.seq = ++g_flush_seq;
    if (!async_enabled()) return run_batch(b.recs, b.seq);
    Worker& w = *g_worker;
    {
        std::lock_guard<std::mutex> lk(w.mu);
        if (!w.started) {
            w.started = true;
            std::thread(worker_main).detach();
        }
        w.queue.push_back(std::move(b));
        g_batches_in_flight.fetch_add(1, std::memory_order_release);
    }
    w.work.notify_one();
    return 0;
}

// everything recorded so far has been launched on the handle layer's stream when this returns
int flush_pending() {
    if (g_lazy_busy > 0) return 0;
    const int rc = submit_pending();
    const int rd = drain_worker();
    return (rc || rd) ? 1 : 0;
}
