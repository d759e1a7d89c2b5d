/*
 * TEST INFRASTRUCTURE ONLY -- plain-C CPU restatement ("port") of the qwen3-mlx dense decode
 * step that drives the mlx-rs-core hot path.  Used (a) by tests/ to cross-check the numpy
 * oracle, (b) by bench.py's cpu_baseline leg as the timed CPU implementation on the GPU box's
 * host cores.  The product (ominix-mlx_amd/) never links, loads or calls this file.
 *
 * Follows, op by op with bf16 rounding of every op output (what MLX does for bf16 arrays):
 *   TransformerBlock::forward  qwen3-mlx/src/model.rs:321-332
 *   Attention::forward         qwen3-mlx/src/model.rs:161-215   (q/k RMSNorm, RoPE, KVCache, SDPA)
 *   Mlp::forward               qwen3-mlx/src/model.rs:263-267
 *   KVCache::update_and_fetch  mlx-rs-core/src/cache.rs:134-194  (append at offset)
 *   RMSNorm / RoPE / SDPA      mlx-rs/src/fast.rs:15-46,121-151,165-180 (MLX core v0.30.1
 *                              semantics; pinned by tests/test_oracle_kats.py via the numpy twin)
 * PARITY UNPINNED at model level (no golden ids/logits exist in the reference).
 *
 * Accumulations are in double (order-free estimate of MLX's fp32 accumulate), rounded once to
 * bf16 per op, exactly like oracle/ref_core.py -- so the two restatements agree bit-for-bit
 * except where libm's exp/sin/cos differ in the last ulp.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint16_t bf16;

static inline float bf2f(bf16 b) {
    union { uint32_t u; float f; } c;
    c.u = (uint32_t)b << 16;
    return c.f;
}
static inline bf16 f2bf(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (bf16)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16)(u >> 16);
}
static inline float rbf(double x) { return bf2f(f2bf((float)x)); }

/* ---- synthetic tensors: twin of oracle/synth.py and ominix-mlx_amd/csrc/runtime.hip ---- */
static inline uint32_t hash_u32(uint64_t idx, uint32_t seed) {
    uint32_t x = (uint32_t)(idx * 0x9E3779B1ull + seed);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
void oracle_fill_uniform_bf16(bf16* dst, int64_t n, uint32_t seed, float amp, float offset) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float u = (float)(hash_u32((uint64_t)i, seed) >> 8) * (1.0f / 16777216.0f);
        const float t = 2.0f * u - 1.0f;
        const float a = amp * t;          /* separate roundings, as in the device kernel */
        dst[i] = f2bf(offset + a);
    }
}

/* ---- a4 RMSNorm: y = w * x * rsqrt(mean(x^2)+eps), output bf16 (fast.rs:171-179) ---- */
void oracle_rms_norm_bf16(bf16* out, const bf16* x, const bf16* w, int rows, int dim, float eps) {
    for (int r = 0; r < rows; ++r) {
        double ss = 0;
        for (int i = 0; i < dim; ++i) { const double v = bf2f(x[(size_t)r * dim + i]); ss += v * v; }
        const double rstd = 1.0 / sqrt(ss / dim + (double)eps);
        for (int i = 0; i < dim; ++i)
            out[(size_t)r * dim + i] = f2bf((float)(bf2f(x[(size_t)r * dim + i]) * rstd * (w ? bf2f(w[i]) : 1.0)));
    }
}

/* ---- a5 Linear at M == 1: out[n] = bf16(sum_k x[k] W[n,k])  (nn/linear.rs:87-92) ---- */
void oracle_gemv_bf16(bf16* out, const bf16* w, const bf16* x, int N, int K) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const bf16* wr = w + (size_t)n * K;
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)bf2f(wr[k]) * (double)bf2f(x[k]);
        out[n] = f2bf((float)acc);
    }
}

/* ---- a3 RoPE, non-traditional half-split pairing, one position (fast.rs:15-46) ---- */
void oracle_rope_row_bf16(bf16* x, int D, float base, float scale, int pos) {
    const int half = D / 2;
    for (int i = 0; i < half; ++i) {
        const double inv = exp(-(double)i * (log((double)base) / half));
        const double ang = (double)pos * (double)scale * inv;
        const double c = cos(ang), s = sin(ang);
        const double x1 = bf2f(x[i]), x2 = bf2f(x[i + half]);
        x[i] = f2bf((float)(x1 * c - x2 * s));
        x[i + half] = f2bf((float)(x1 * s + x2 * c));
    }
}

typedef struct {
    int hidden, inter, heads, kv_heads, head_dim, cap;
    float eps, rope_theta, rope_scale;
} oracle_layer_cfg;

typedef struct {
    const bf16 *q, *k, *v, *o, *gate, *up, *down, *q_norm, *k_norm, *in_ln, *post_ln;
    bf16 *kcache, *vcache;   /* [kv_heads, cap, head_dim] */
} oracle_layer;

/* one decoder block at decode time (L == 1): h[hidden] updated in place; pos = cache offset */
void oracle_qwen3_layer_decode(const oracle_layer_cfg* c, const oracle_layer* L, bf16* h, int pos, bf16* scratch) {
    const int hd = c->hidden, D = c->head_dim, H = c->heads, Hkv = c->kv_heads, I = c->inter, G = H / Hkv;
    bf16* xn = scratch;               /* hidden */
    bf16* q = xn + hd;                /* H*D */
    bf16* k = q + (size_t)H * D;      /* Hkv*D */
    bf16* v = k + (size_t)Hkv * D;    /* Hkv*D */
    bf16* att = v + (size_t)Hkv * D;  /* H*D */
    bf16* g = att + (size_t)H * D;    /* I */
    bf16* u = g + I;                  /* I */
    bf16* tmp = u + I;                /* hidden */
    oracle_rms_norm_bf16(xn, h, L->in_ln, 1, hd, c->eps);
    oracle_gemv_bf16(q, L->q, xn, H * D, hd);
    oracle_gemv_bf16(k, L->k, xn, Hkv * D, hd);
    oracle_gemv_bf16(v, L->v, xn, Hkv * D, hd);
    oracle_rms_norm_bf16(q, q, L->q_norm, H, D, c->eps);
    oracle_rms_norm_bf16(k, k, L->k_norm, Hkv, D, c->eps);
    for (int i = 0; i < H; ++i) oracle_rope_row_bf16(q + (size_t)i * D, D, c->rope_theta, c->rope_scale, pos);
    for (int i = 0; i < Hkv; ++i) oracle_rope_row_bf16(k + (size_t)i * D, D, c->rope_theta, c->rope_scale, pos);
    for (int i = 0; i < Hkv; ++i) {   /* KVCache append (cache.rs:183-188) */
        memcpy(L->kcache + ((size_t)i * c->cap + pos) * D, k + (size_t)i * D, (size_t)D * 2);
        memcpy(L->vcache + ((size_t)i * c->cap + pos) * D, v + (size_t)i * D, (size_t)D * 2);
    }
    const int T = pos + 1;
    const double scale = (double)(1.0f / sqrtf((float)D));
#pragma omp parallel for schedule(static)
    for (int hq = 0; hq < H; ++hq) {   /* SDPA, softmax in high precision, GQA by head group */
        const int kvh = hq / G;
        const bf16* kc = L->kcache + (size_t)kvh * c->cap * D;
        const bf16* vc = L->vcache + (size_t)kvh * c->cap * D;
        double* s = (double*)malloc((size_t)T * sizeof(double));
        double mx = -INFINITY;
        for (int t = 0; t < T; ++t) {
            double d = 0;
            for (int e = 0; e < D; ++e) d += (double)bf2f(q[(size_t)hq * D + e]) * scale * (double)bf2f(kc[(size_t)t * D + e]);
            s[t] = d;
            if (d > mx) mx = d;
        }
        double den = 0;
        for (int t = 0; t < T; ++t) { s[t] = exp(s[t] - mx); den += s[t]; }
        for (int e = 0; e < D; ++e) {
            double o = 0;
            for (int t = 0; t < T; ++t) o += s[t] * (double)bf2f(vc[(size_t)t * D + e]);
            att[(size_t)hq * D + e] = f2bf((float)(o / den));
        }
        free(s);
    }
    oracle_gemv_bf16(tmp, L->o, att, hd, H * D);
    for (int i = 0; i < hd; ++i) h[i] = f2bf(bf2f(h[i]) + bf2f(tmp[i]));
    oracle_rms_norm_bf16(xn, h, L->post_ln, 1, hd, c->eps);
    oracle_gemv_bf16(g, L->gate, xn, I, hd);
    oracle_gemv_bf16(u, L->up, xn, I, hd);
    for (int i = 0; i < I; ++i) {      /* nn::silu(gate) * up, each primitive rounded to bf16 */
        const double gv = bf2f(g[i]);
        const float sg = rbf(1.0 / (1.0 + exp(-gv)));
        const float sl = rbf(gv * (double)sg);
        g[i] = f2bf((float)((double)sl * (double)bf2f(u[i])));
    }
    oracle_gemv_bf16(tmp, L->down, g, hd, I);
    for (int i = 0; i < hd; ++i) h[i] = f2bf(bf2f(h[i]) + bf2f(tmp[i]));
}

size_t oracle_qwen3_scratch_elems(const oracle_layer_cfg* c) {
    return (size_t)c->hidden * 2 + (size_t)c->heads * c->head_dim * 2 + (size_t)c->kv_heads * c->head_dim * 2 +
           (size_t)c->inter * 2 + 64;
}

/* final norm + lm_head + greedy argmax (model.rs:423,480-489,733-735); returns the token id */
uint32_t oracle_qwen3_head(const bf16* h, const bf16* norm_w, const bf16* lm_head, int hidden, int vocab, float eps,
                           bf16* logits_out, bf16* scratch) {
    oracle_rms_norm_bf16(scratch, h, norm_w, 1, hidden, eps);
    oracle_gemv_bf16(logits_out, lm_head, scratch, vocab, hidden);
    uint32_t best = 0;
    float bv = bf2f(logits_out[0]);
    for (int i = 1; i < vocab; ++i) {
        const float v = bf2f(logits_out[i]);
        if (v > bv) { bv = v; best = (uint32_t)i; }
    }
    return best;
}
