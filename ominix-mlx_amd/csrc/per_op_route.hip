// The DROP-IN route, measured (VERDICT r4 "Next" 5): what an UNMODIFIED qwen3-mlx gets from this library -- no fused engine, every mlx-rs
// call of Model::forward + Generate::next arriving as one mlx_* call on the handle ABI (include/omx_mlx_c.h).  Round 6: the handle layer
// records those calls and executes them at mlx_async_eval / item (mlxc_lazy.hpp, the ABI's own contract: /root/reference/mlx-rs/src/
// transforms/mod.rs:67-85), rewriting the decode idioms onto the engine's GEMV family; OMX_MLX_LAZY=0 is round 5's eager execution.
//   replayed, call for call:  qwen3-mlx/src/model.rs:161-215 (Attention::forward: q / k / v = nn::Linear = x.matmul(w.t()), reshape +
//   transpose_axes, q_norm / k_norm, nn::Rope, KVCache::update_and_fetch = cache.rs:140-193, fast::scaled_dot_product_attention, o_proj),
//   :263-267 (Mlp: down(silu(gate(x)) * up(x)), silu = x * sigmoid(x)), :314-340 (block: two rms_norm + two adds), :387-433 (Model::forward:
//   embedding take, mask "causal" for L > 1, final norm, lm_head), :804-843 (Generate::next: last row, argmax, item).
// The weights are the engine's own device tensors, borrowed (omx_qwen3_get_weight + omx_mlx_array_from_device), so the tokens can be compared
// with omx_qwen3_prefill / _decode on the same checkpoint.  Only public C entry points are called below -- this file could live outside
// the library; it sits in it so that bench.py reaches it through ctypes.
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/omx_mlx_c.h"
#include "common.hpp"

namespace {

struct PerOp {
    mlx_stream s;
    long calls = 0;
    std::vector<mlx_array> trash;      // intermediates of the running forward pass, freed after each token like Rust drops them
    bool failed = false;

    mlx_array keep(mlx_array a) { trash.push_back(a); return a; }
    void sweep() { for (auto& a : trash) mlx_array_free(a); trash.clear(); }
#define OP(call)                                 \
    mlx_array r = mlx_array_new();               \
    ++calls;                                     \
    if ((call) != 0) failed = true;              \
    return keep(r)
    mlx_array rms_norm(mlx_array x, mlx_array w, float eps) { OP(mlx_fast_rms_norm(&r, x, w, eps, s)); }
    mlx_array t(mlx_array a) { OP(mlx_transpose(&r, a, s)); }
    mlx_array matmul(mlx_array a, mlx_array b) { OP(mlx_matmul(&r, a, b, s)); }
    mlx_array linear(mlx_array x, mlx_array w) { return matmul(x, t(w)); }      // nn::Linear::forward (linear.rs:87-92), no bias
    // nn::QuantizedLinear::forward (quantized.rs:366-375): quantized_matmul(x, w, scales, biases, transpose = true, group_size, bits)
    int q_group = 64, q_bits = 4;
    mlx_array qmm(mlx_array x, mlx_array w, mlx_array sc, mlx_array bi) {
        OP(mlx_quantized_matmul(&r, x, w, sc, bi, true, mlx_optional_int{q_group, true}, mlx_optional_int{q_bits, true}, "affine", s));
    }
    mlx_array dequantize(mlx_array w, mlx_array sc, mlx_array bi) {
        OP(mlx_dequantize(&r, w, sc, bi, mlx_optional_int{q_group, true}, mlx_optional_int{q_bits, true}, "affine", mlx_optional_dtype{MLX_BFLOAT16, false}, s));
    }
    mlx_array reshape(mlx_array a, std::vector<int> sh) { OP(mlx_reshape(&r, a, sh.data(), sh.size(), s)); }
    mlx_array transpose_axes(mlx_array a, std::vector<int> ax) { OP(mlx_transpose_axes(&r, a, ax.data(), ax.size(), s)); }
    mlx_array rope(mlx_array x, int dims, float base, int offset) {
        OP(mlx_fast_rope(&r, x, dims, false, mlx_optional_float{base, true}, 1.0f, offset, mlx_array{nullptr}, s));
    }
    mlx_array sdpa(mlx_array q, mlx_array k, mlx_array v, float scale, const char* mode) {
        OP(mlx_fast_scaled_dot_product_attention(&r, q, k, v, scale, mode, mlx_array{nullptr}, mlx_array{nullptr}, s));
    }
    mlx_array add(mlx_array a, mlx_array b) { OP(mlx_add(&r, a, b, s)); }
    mlx_array mul(mlx_array a, mlx_array b) { OP(mlx_multiply(&r, a, b, s)); }
    mlx_array sigmoid(mlx_array a) { OP(mlx_sigmoid(&r, a, s)); }
    mlx_array take_axis(mlx_array a, mlx_array idx, int axis) { OP(mlx_take_axis(&r, a, idx, axis, s)); }
    mlx_array argmax(mlx_array a) { OP(mlx_argmax_axis(&r, a, -1, false, s)); }
    mlx_array slice(mlx_array a, std::vector<int> st, std::vector<int> sp) {
        std::vector<int> one(st.size(), 1);
        OP(mlx_slice(&r, a, st.data(), st.size(), sp.data(), sp.size(), one.data(), one.size(), s));
    }
#undef OP
};

// KVCache (mlx-rs-core/src/cache.rs:91-194): buffers grown in steps of 256, slice_update writes, [.., :offset, :] views
struct KvCache {
    mlx_array k{nullptr}, v{nullptr};
    int offset = 0, cap = 0;
    static constexpr int STEP = 256;
    void release() {
        if (k.ctx) mlx_array_free(k);
        if (v.ctx) mlx_array_free(v);
        k = v = mlx_array{nullptr};
    }
    // returns the [.., :offset, :] views (owned by the PerOp's trash)
    bool update_and_fetch(PerOp& P, mlx_array keys, mlx_array values, int Hkv, int n_new, int D, mlx_array* ko, mlx_array* vo) {
        const int prev = offset;
        if (!k.ctx || prev + n_new > cap) {                      // cache.rs:150-178
            const int n_steps = (STEP + n_new - 1) / STEP, new_size = n_steps * STEP;
            const int sh[4] = {1, Hkv, new_size, D};
            mlx_array nk = mlx_array_new(), nv = mlx_array_new();
            P.calls += 2;
            if (mlx_zeros(&nk, sh, 4, MLX_BFLOAT16, P.s) || mlx_zeros(&nv, sh, 4, MLX_BFLOAT16, P.s)) return false;
            const bool had = k.ctx != nullptr;
            if (k.ctx) {
                mlx_array ok = k, ov = v;
                if (prev % STEP != 0) {
                    ok = P.slice(k, {0, 0, 0, 0}, {1, Hkv, prev, D});
                    ov = P.slice(v, {0, 0, 0, 0}, {1, Hkv, prev, D});
                }
                mlx_array ck = mlx_array_new(), cv = mlx_array_new();
                mlx_vector_array vk = mlx_vector_array_new(), vv = mlx_vector_array_new();
                mlx_vector_array_append_value(vk, ok); mlx_vector_array_append_value(vk, nk);
                mlx_vector_array_append_value(vv, ov); mlx_vector_array_append_value(vv, nv);
                P.calls += 2;
                const bool bad = mlx_concatenate_axis(&ck, vk, 2, P.s) || mlx_concatenate_axis(&cv, vv, 2, P.s);
                mlx_vector_array_free(vk); mlx_vector_array_free(vv);
                mlx_array_free(nk); mlx_array_free(nv);
                if (bad) return false;
                release();
                k = ck; v = cv;
            } else {
                k = nk; v = nv;
            }
            cap = (had ? (prev % STEP ? prev : cap) : 0) + new_size;
        }
        offset += n_new;
        const int st[4] = {0, 0, prev, 0}, sp[4] = {1, Hkv, offset, D}, one[4] = {1, 1, 1, 1};
        // k.index_mut((Ellipsis, prev..offset, ..), &keys) -> mlx_slice_update; the old handle is dropped afterwards (donation: in place)
        mlx_array k2 = mlx_array_new(), v2 = mlx_array_new();
        P.calls += 2;
        if (mlx_slice_update(&k2, k, keys, st, 4, sp, 4, one, 4, P.s) || mlx_slice_update(&v2, v, values, st, 4, sp, 4, one, 4, P.s)) return false;
        mlx_array_free(k); mlx_array_free(v);
        k = k2; v = v2;
        *ko = P.slice(k, {0, 0, 0, 0}, {1, Hkv, offset, D});
        *vo = P.slice(v, {0, 0, 0, 0}, {1, Hkv, offset, D});
        return !P.failed;
    }
};

// a Linear's tensors: the bf16 matrix, or the (packed weight, scales, biases) triplet of an MLX-quantized checkpoint (model.rs:621-727)
struct Lin {
    mlx_array w{nullptr}, sc{nullptr}, bi{nullptr};
    void release() { for (mlx_array* a : {&w, &sc, &bi}) if (a->ctx) { mlx_array_free(*a); a->ctx = nullptr; } }
};
struct LayerW { Lin q, k, v, o, gate, up, down; mlx_array in_ln, post_ln, q_norm, k_norm; };

}  // namespace

/* forced != null: the TEACHER-FORCED form for the oracle pins (tests/test_gpu_fullsize_pin.py) -- after the prompt, position i is fed
 * forced[i] instead of the route's own token (one pass at a time, the token from the host), tokens_out still holds the route's own greedy
 * choices, and the bf16 logits rows of the steps listed in logit_steps (0 = the prompt's last position) are copied to logits_out
 * [n_logit_steps, vocab].  forced == null: Generate::next's pipelined loop, timed (bench.py). */
extern "C" int omx_bench_qwen3_per_op_ex(omx_qwen3 model, const omx_qwen3_config* cfg, const uint32_t* prompt, int n_prompt, int n_new,
                                         uint32_t* tokens_out, double* prefill_ms, double* ms_per_token, double* calls_per_token,
                                         const uint32_t* forced, const int* logit_steps, int n_logit_steps, uint16_t* logits_out);
extern "C" int omx_bench_qwen3_per_op(omx_qwen3 model, const omx_qwen3_config* cfg, const uint32_t* prompt, int n_prompt, int n_new,
                                      uint32_t* tokens_out, double* prefill_ms, double* ms_per_token, double* calls_per_token) {
    return omx_bench_qwen3_per_op_ex(model, cfg, prompt, n_prompt, n_new, tokens_out, prefill_ms, ms_per_token, calls_per_token, nullptr, nullptr, 0,
                                     nullptr);
}
extern "C" int omx_bench_qwen3_per_op_ex(omx_qwen3 model, const omx_qwen3_config* cfg, const uint32_t* prompt, int n_prompt, int n_new,
                                         uint32_t* tokens_out, double* prefill_ms, double* ms_per_token, double* calls_per_token,
                                         const uint32_t* forced, const int* logit_steps, int n_logit_steps, uint16_t* logits_out) {
    using namespace omx;
    OMX_REQUIRE(model && cfg && prompt && n_prompt > 0 && n_new > 0 && tokens_out, "omx_bench_qwen3_per_op: bad arguments");
    OMX_REQUIRE(n_logit_steps == 0 || (forced && logit_steps && logits_out), "omx_bench_qwen3_per_op_ex: logits are captured in the forced form only");
    OMX_REQUIRE(cfg->num_experts == 0 && cfg->tp_size <= 1 && !cfg->quant_scales_f16,
                "omx_bench_qwen3_per_op: a dense model (bf16, or MLX-quantized with bf16 scales) on one rank");
    const bool quant = cfg->quant_bits != 0;
    const int hd = cfg->hidden_size, H = cfg->num_attention_heads, Hkv = cfg->num_key_value_heads, D = cfg->head_dim, I = cfg->intermediate_size,
              V = cfg->vocab_size, L = cfg->num_hidden_layers;
    auto borrow = [&](const std::string& name, std::vector<int> shape, mlx_array* out, mlx_dtype dt = MLX_BFLOAT16) -> int {
        const void* p = nullptr;
        size_t nb = 0;
        if (omx_qwen3_get_weight(model, name.c_str(), &p, &nb)) return 1;
        *out = omx_mlx_array_from_device(p, shape.data(), (int)shape.size(), dt);
        return out->ctx ? 0 : 1;
    };
    // a Linear [n, k] by its key prefix: `.weight` alone, or the quantized triplet
    auto borrow_lin = [&](const std::string& prefix, int n, int k, Lin* out) -> int {
        if (!quant) return borrow(prefix + ".weight", {n, k}, &out->w);
        return borrow(prefix + ".weight", {n, k * cfg->quant_bits / 32}, &out->w, MLX_UINT32) ||
               borrow(prefix + ".scales", {n, k / cfg->quant_group}, &out->sc) || borrow(prefix + ".biases", {n, k / cfg->quant_group}, &out->bi);
    };
    std::vector<LayerW> W(L);
    Lin embed, head;
    mlx_array final_norm{nullptr};
    int rc = borrow_lin("model.embed_tokens", V, hd, &embed) || borrow("model.norm.weight", {hd}, &final_norm);
    if (!rc) rc = cfg->tie_word_embeddings ? 0 : borrow_lin("lm_head", V, hd, &head);
    for (int l = 0; l < L && !rc; ++l) {
        const std::string p = "model.layers." + std::to_string(l) + ".";
        rc = borrow_lin(p + "self_attn.q_proj", H * D, hd, &W[l].q) || borrow_lin(p + "self_attn.k_proj", Hkv * D, hd, &W[l].k) ||
             borrow_lin(p + "self_attn.v_proj", Hkv * D, hd, &W[l].v) || borrow_lin(p + "self_attn.o_proj", hd, H * D, &W[l].o) ||
             borrow_lin(p + "mlp.gate_proj", I, hd, &W[l].gate) || borrow_lin(p + "mlp.up_proj", I, hd, &W[l].up) ||
             borrow_lin(p + "mlp.down_proj", hd, I, &W[l].down) || borrow(p + "input_layernorm.weight", {hd}, &W[l].in_ln) ||
             borrow(p + "post_attention_layernorm.weight", {hd}, &W[l].post_ln) || borrow(p + "self_attn.q_norm.weight", {D}, &W[l].q_norm) ||
             borrow(p + "self_attn.k_norm.weight", {D}, &W[l].k_norm);
    }
    if (rc) return 1;
    PerOp P;
    P.s = mlx_default_gpu_stream_new();
    if (quant) { P.q_group = cfg->quant_group; P.q_bits = cfg->quant_bits; }
    auto lin = [&](mlx_array x, const Lin& w) { return quant ? P.qmm(x, w.w, w.sc, w.bi) : P.linear(x, w.w); };
    std::vector<KvCache> cache(L);
    const float scale = 1.0f / sqrtf((float)D);

    // Model::forward on the token ids `idx` ([1, n], on the device) -> y = sample(logits of the LAST position) as a device array
    // (model.rs:387-433 + sampler at temperature 0).  forward's temporaries are dropped before this returns, like Rust drops them at
    // the end of their scopes -- i.e. BEFORE the caller's async_eval: an intermediate nobody holds may then be fused away (mlxc_lazy.hpp).
    auto forward_sample = [&](mlx_array idx, int n, mlx_array* y, mlx_array* keep_logits = nullptr) -> int {
        // Embedding::forward -> [1, n, hidden]; QuantizedEmbedding (quantized.rs:120-164): the picked rows of the triplet, dequantised
        mlx_array h = quant ? P.dequantize(P.take_axis(embed.w, idx, 0), P.take_axis(embed.sc, idx, 0), P.take_axis(embed.bi, idx, 0))
                            : P.take_axis(embed.w, idx, 0);
        const char* mode = n > 1 ? "causal" : "";                                    // create_attention_mask (utils.rs:156-188)
        for (int l = 0; l < L; ++l) {
            const LayerW& w = W[l];
            mlx_array xn = P.rms_norm(h, w.in_ln, cfg->rms_norm_eps);
            mlx_array q = P.transpose_axes(P.reshape(lin(xn, w.q), {1, n, H, D}), {0, 2, 1, 3});
            mlx_array k = P.transpose_axes(P.reshape(lin(xn, w.k), {1, n, Hkv, D}), {0, 2, 1, 3});
            mlx_array v = P.transpose_axes(P.reshape(lin(xn, w.v), {1, n, Hkv, D}), {0, 2, 1, 3});
            q = P.rms_norm(q, w.q_norm, cfg->rms_norm_eps);
            k = P.rms_norm(k, w.k_norm, cfg->rms_norm_eps);
            const int off = cache[l].offset;
            q = P.rope(q, D, cfg->rope_theta, off);
            k = P.rope(k, D, cfg->rope_theta, off);
            mlx_array kk, vv;
            if (!cache[l].update_and_fetch(P, k, v, Hkv, n, D, &kk, &vv)) return 1;
            mlx_array o = P.sdpa(q, kk, vv, scale, mode);
            o = P.reshape(P.transpose_axes(o, {0, 2, 1, 3}), {1, n, H * D});
            mlx_array h1 = P.add(h, lin(o, w.o));
            mlx_array hn = P.rms_norm(h1, w.post_ln, cfg->rms_norm_eps);
            mlx_array g = lin(hn, w.gate);
            mlx_array act = P.mul(P.mul(g, P.sigmoid(g)), lin(hn, w.up));       // nn::silu(gate) * up
            h = P.add(h1, lin(act, w.down));
            if (P.failed) return 1;
        }
        mlx_array hf = P.rms_norm(h, final_norm, cfg->rms_norm_eps);
        mlx_array logits = lin(hf, cfg->tie_word_embeddings ? embed : head);     // all n positions, like the reference (model.rs:815 keeps the last)
        if (n > 1) logits = P.slice(logits, {0, n - 1, 0}, {1, n, V});
        mlx_array t = P.argmax(logits);                                               // [1, 1]
        if (P.failed) return 1;
        *y = mlx_array_new();
        ++P.calls;
        if (mlx_array_set(y, t)) return 1;                                            // the one array that leaves forward
        if (keep_logits) {
            *keep_logits = mlx_array_new();
            if (mlx_array_set(keep_logits, logits)) return 1;
        }
        P.sweep();
        return 0;
    };
    auto async_eval = [&](mlx_array y) -> int {
        mlx_vector_array v = mlx_vector_array_new();
        mlx_vector_array_append_value(v, y);
        P.calls += 3;
        const int rc = mlx_async_eval(v);
        mlx_vector_array_free(v);
        return rc;
    };
    uint32_t tok = 0;
    auto t0 = std::chrono::steady_clock::now();
    mlx_array prefetched{nullptr};
    if (forced) {
        for (int step = 0; step <= n_new && !rc; ++step) {
            const int n = step == 0 ? n_prompt : 1;
            const int sh[2] = {1, n};
            mlx_array idx = mlx_array_new_data(step == 0 ? prompt : forced + (step - 1), sh, 2, MLX_UINT32);
            mlx_array y{nullptr}, lg{nullptr};
            int want = -1;
            for (int i = 0; i < n_logit_steps; ++i)
                if (logit_steps[i] == step) want = i;
            rc = forward_sample(idx, n, &y, want >= 0 ? &lg : nullptr) || async_eval(y);
            mlx_array_free(idx);
            if (!rc) rc = mlx_array_item_uint32(&tok, y);
            tokens_out[step] = tok;
            if (!rc && want >= 0) {
                const uint16_t* host = mlx_array_data_bfloat16(lg);
                if (host) memcpy(logits_out + (size_t)want * V, host, (size_t)V * 2); else rc = 1;
            }
            if (y.ctx) mlx_array_free(y);
            if (lg.ctx) mlx_array_free(lg);
        }
    }
    // Generate::next (model.rs:804-843): the step after the one being returned is recorded and sent before the caller reads a token
    if (!forced) {
        const int sh[2] = {1, n_prompt};
        mlx_array idx = mlx_array_new_data(prompt, sh, 2, MLX_UINT32);
        ++P.calls;
        mlx_array y{nullptr};
        rc = forward_sample(idx, n_prompt, &y) || async_eval(y);
        mlx_array_free(idx);
        if (!rc) rc = forward_sample(y, 1, &prefetched) || async_eval(prefetched);      // compute_next(&y): y is [1, 1] already
        ++P.calls;
        if (!rc) rc = mlx_array_item_uint32(&tok, y);                                   // eval([&y]) + the caller's token.item()
        if (y.ctx) mlx_array_free(y);
        tokens_out[0] = tok;
    }
    auto t1 = std::chrono::steady_clock::now();
    long calls0 = P.calls;
    for (int i = 1; i <= n_new && !rc && !forced; ++i) {
        mlx_array current = prefetched, next{nullptr};
        prefetched = mlx_array{nullptr};
        rc = forward_sample(current, 1, &next) || async_eval(next);
        prefetched = next;
        ++P.calls;
        if (!rc) rc = mlx_array_item_uint32(&tok, current);                             // the caller's token.item() on the step BEFORE the one just sent
        mlx_array_free(current);
        tokens_out[i] = tok;
    }
    if (prefetched.ctx) {
        mlx_vector_array v = mlx_vector_array_new();
        mlx_vector_array_append_value(v, prefetched);
        (void)mlx_eval(v);
        mlx_vector_array_free(v);
        mlx_array_free(prefetched);
    }
    auto t2 = std::chrono::steady_clock::now();
    if (prefill_ms) *prefill_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (ms_per_token) *ms_per_token = std::chrono::duration<double, std::milli>(t2 - t1).count() / n_new;
    if (calls_per_token) *calls_per_token = (double)(P.calls - calls0) / n_new;
    P.sweep();
    for (auto& c : cache) c.release();
    for (auto& w : W) {
        for (Lin* l : {&w.q, &w.k, &w.v, &w.o, &w.gate, &w.up, &w.down}) l->release();
        for (mlx_array a : {w.in_ln, w.post_ln, w.q_norm, w.k_norm}) mlx_array_free(a);
    }
    embed.release(); head.release();
    mlx_array_free(final_norm);
    if (rc) return omx::set_error("omx_bench_qwen3_per_op: an mlx_* call failed: %s", omx_last_error());
    return 0;
}
