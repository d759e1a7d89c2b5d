/*
 * omx.h -- C ABI of libomx_hip.so: the MI355X (gfx950) implementation of the
 * mlx-rs-core inference hot path of OminiX-MLX.
 *
 * This is the drop-in boundary.  The reference's Rust
 * crates reach their arithmetic through the mlx-c C shim (mlx-sys bindgen,
 * mlx-rs/mlx-sys/build.rs:390-399).  Two layers are exported:
 *
 *   1. omx_*  (this file): eager, raw-pointer kernels.  Plain device pointers,
 *      sizes and a hipStream_t (passed as void*; NULL = the null stream).  Each
 *      entry cites the mlx-c function (header:line under
 *      mlx-rs/mlx-sys/src/mlx-c/mlx/c/) and the Rust call site it replaces.
 *   2. mlx_*  (omx_mlx_c.h): the handle-based subset of the mlx-c ABI with the
 *      reference's ownership / status / error-handler conventions, implemented
 *      on top of layer 1, so `mlx-sys` can link against libomx_hip.so.
 *
 * Conventions (same as mlx-c, error.cpp:37-54, fast.cpp:614-632):
 *   - every function returns int: 0 = ok, 1 = error;
 *   - on error the registered handler (omx_set_error_handler == mlx_set_error_handler)
 *     is invoked on the calling thread with a NUL-terminated message; the default
 *     handler stores it in a thread-local slot readable with omx_last_error();
 *   - all tensor arguments are DEVICE pointers, row-major contiguous unless a
 *     stride argument says otherwise; dtype codes are the mlx_dtype values
 *     (array.h:37-52);
 *   - no function allocates, frees or synchronises unless its name says so:
 *     everything is enqueued on `stream` and is hipGraph-capturable.
 */
#ifndef OMX_H
#define OMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mlx_dtype numbering (mlx/c/array.h:37-52) */
typedef enum omx_dtype_ {
    OMX_BOOL = 0, OMX_UINT8 = 1, OMX_UINT16 = 2, OMX_UINT32 = 3, OMX_UINT64 = 4,
    OMX_INT8 = 5, OMX_INT16 = 6, OMX_INT32 = 7, OMX_INT64 = 8,
    OMX_FLOAT16 = 9, OMX_FLOAT32 = 10, OMX_FLOAT64 = 11, OMX_BFLOAT16 = 12, OMX_COMPLEX64 = 13
} omx_dtype;

typedef void* omx_stream; /* hipStream_t */

/* ---- library / error plumbing (mlx/c/error.h:15-23, stream.h:63, memory.h:30) ---- */
const char* omx_version(void);
int omx_experiments_built(void);           /* 1: built with `make EXPERIMENTS=1` (persistent step, AQL replay, 32x32 two-phase / stream-K attention) */
typedef void (*omx_error_handler_func)(const char* msg, void* data);
void omx_set_error_handler(omx_error_handler_func handler, void* data, void (*dtor)(void*));
const char* omx_last_error(void);          /* thread-local; "" when none */
void omx_clear_error(void);
int omx_device_count(int* count);
int omx_device_name(char* buf, size_t buflen);
int omx_synchronize(omx_stream stream);    /* mlx_synchronize */

/* device memory (what mlx_array_new_data / mlx_array_data_* do underneath) */
int omx_malloc(void** ptr, size_t bytes);
int omx_free(void* ptr);
int omx_memcpy_h2d(void* dst, const void* src, size_t bytes, omx_stream stream);
int omx_memcpy_d2h(void* dst, const void* src, size_t bytes, omx_stream stream);
int omx_memcpy_d2d(void* dst, const void* src, size_t bytes, omx_stream stream);
int omx_memset(void* dst, int value, size_t bytes, omx_stream stream);

/* synthetic tensors: value(i) = offset + amp*(2*u24(hash(seed,i))-1), RNE to dtype */
int omx_fill_uniform(void* dst, size_t n, uint32_t seed, float amp, float offset, omx_dtype dtype, omx_stream stream);

/* ---- a4: norms.  mlx_fast_rms_norm fast.h:163-168 (mlx-rs/src/fast.rs:165-180),
 *           mlx_fast_layer_norm fast.h:93-99 (fast.rs:204-218).  weight/bias may be NULL.
 *      x, out: [rows, dim] contiguous.                                                      */
int omx_rms_norm(void* out, const void* x, const void* weight, int64_t rows, int dim, float eps,
                 omx_dtype dtype, omx_stream stream);
int omx_layer_norm(void* out, const void* x, const void* weight, const void* bias, int64_t rows, int dim,
                   float eps, omx_dtype dtype, omx_stream stream);

/* ---- a3: RoPE.  mlx_fast_rope fast.h:169-178 (fast.rs:15-46; nn/positional_encoding.rs:112-137).
 *      x,out: [batch(=B*H), T, D] contiguous, position = axis -2, positions offset..offset+T-1.
 *      traditional=0: pairs (i, i+dims/2); =1: pairs (2i, 2i+1).  theta_i = base^(-2i/dims).  */
int omx_rope(void* out, const void* x, int64_t batch, int T, int D, int dims, int traditional, float base,
             float scale, int offset, omx_dtype dtype, omx_stream stream);
/* ... with custom frequencies instead of a base (mlx-rs/src/fast.rs:15-46 `freqs`, mlx-c fast.h mlx_fast_rope): freqs = dims / 2
 * float32 values on the device; the angle of pair j at position p is (p + offset) * scale / freqs[j] */
int omx_rope_freqs(void* out, const void* x, int64_t batch, int T, int D, int dims, int traditional, const float* freqs, float scale,
                   int offset, omx_dtype dtype, omx_stream stream);

/* ---- a8/a9: the two mlx-rs-core fused kernels (metal_kernels.rs:188-236, 260-339), which the
 *      reference JIT-compiles from Metal source through mlx_fast_metal_kernel_apply (fast.h:156).
 *      omx_fused_swiglu(out, x=up, gate): silu(gate)*x.   omx_fused_modulate: (1+scale)*LN(x)+shift,
 *      x [B,S,H], shift/scale [B,H].                                                           */
int omx_fused_swiglu(void* out, const void* x, const void* gate, int64_t n, omx_dtype dtype, omx_stream stream);
int omx_fused_modulate(void* out, const void* x, const void* shift, const void* scale, int B, int S, int H,
                       float eps, omx_dtype dtype, omx_stream stream);

/* ---- a5: Linear.  mlx_matmul ops.h:598-602 / mlx_addmm ops.h:36-43 as used by nn::Linear
 *      (mlx-rs/src/nn/linear.rs:87-92): out[M,N] = x[M,K] . W[N,K]^T (+ bias[N]).
 *      M <= 8 takes the HBM-streaming GEMV path, larger M the MFMA GEMM path.                  */
int omx_linear(void* out, const void* x, const void* w, const void* bias, int M, int N, int K,
               omx_dtype dtype, omx_stream stream);
/* Linear whose trailing 2*half output features are a [gate | up] pair consumed by fused_swiglu
 * (metal_kernels.rs:188-236) -- the FLUX blocks' mlp_in / to_qkv_mlp projections (klein_model.rs:489-493,
 * 905-916).  W [n_plain + 2*half, K].  out_plain [M, n_plain] = x . W[:n_plain]^T,
 * out_act [M, half] = silu(g) * u with g, u the bf16-rounded gate / up features: the same bits as
 * omx_linear followed by omx_fused_swiglu, without the 2*half wide intermediate in HBM.
 * n_plain % 4 == 0, half % 4 == 0, K % 64 == 0; other widths fail (use omx_linear + omx_fused_swiglu).   */
int omx_linear_swiglu(void* out_plain, void* out_act, const void* x, const void* w, int M, int n_plain,
                      int half, int K, omx_dtype dtype, omx_stream stream);

/* ---- a1: SDPA.  mlx_fast_scaled_dot_product_attention fast.h:189-198 (fast.rs:121-151;
 *      mlx-rs-core/src/utils.rs:191-209).  q [B,H,Tq,D]; k,v [B,Hkv,Tk,D] with explicit element
 *      strides so that the [..,:offset,:] views of the step-256 KV buffer (cache.rs:190-193) are
 *      passed without a copy: kv_batch_stride, kv_head_stride (row stride is D).
 *      mask_mode: 0 none, 1 "causal" (bottom-right aligned), 2 bool array [Tq,Tk] (keep-true),
 *      3 additive array [Tq,Tk] of `dtype`.  out [B,H,Tq,D].                                   */
enum { OMX_MASK_NONE = 0, OMX_MASK_CAUSAL = 1, OMX_MASK_BOOL = 2, OMX_MASK_ADDITIVE = 3 };
int omx_sdpa(void* out, const void* q, const void* k, const void* v, int B, int H, int Hkv, int Tq, int Tk,
             int D, int64_t kv_batch_stride, int64_t kv_head_stride, float scale, int mask_mode,
             const void* mask, omx_dtype dtype, omx_stream stream);
/* bytes of scratch omx_sdpa needs for the split-KV decode path (0 for prefill) */
size_t omx_sdpa_workspace_bytes(int B, int H, int Tq, int D);
int omx_set_workspace(void* ws, size_t bytes);   /* caller-provided scratch used by split kernels */

/* ---- a10: greedy sampler.  mlx_argmax_axis ops.h:106-111 (mlx-rs-core/src/sampler.rs:9-12):
 *      out[r] = first index of the maximum of logits[r, :]  (u32).                              */
int omx_argmax(uint32_t* out, const void* logits, int64_t rows, int n, omx_dtype dtype, omx_stream stream);

/* ---- a10, temperature branch: MLX's keyed generator + categorical sampler.
 *      mlx/c/random.h:37-72,129-139,149-157 (mlx_random_bits / _categorical* / _gumbel / _key / _split* / _uniform),
 *      called by mlx-rs/src/random.rs:98-115 (key, split), :397-414 (gumbel), :456-497 (categorical) and through
 *      them by mlx-rs-core/src/sampler.rs:13-16.  Keys are 2 x u32 in device memory; word i of an n-word draw is
 *      Threefry-2x32(key, counter pair of i) in MLX's layout, so results reproduce MLX's for the same key.
 *      categorical: out[r, s] = argmax_v( f32(logits[r, v]) * inv_temp + gumbel[r, v, s] ), first index on ties.   */
int omx_random_key(uint32_t* key, uint64_t seed, omx_stream stream);
int omx_random_split(uint32_t* out /*[num,2]*/, const uint32_t* key, int num, omx_stream stream);
int omx_random_bits(uint32_t* out, const uint32_t* key, int64_t n, omx_stream stream);
int omx_random_uniform(float* out, const uint32_t* key, int64_t n, float lo, float hi, omx_stream stream);
int omx_random_gumbel(float* out, const uint32_t* key, int64_t n, omx_stream stream);
/* mlx_random_normal (random.h:100-108; mlx-rs/src/random.rs:186-212): sqrt(2) * erfinv(uniform(-1, 1)) * scale + loc, f32 */
int omx_random_normal(float* out, const uint32_t* key, int64_t n, float loc, float scale, omx_stream stream);
int omx_random_categorical(uint32_t* out, const void* logits, int64_t rows, int n, int num_samples, float inv_temp,
                           const uint32_t* key, omx_dtype dtype, omx_stream stream);

/* embedding gather (mlx_take_axis ops.h:1109, nn/embedding.rs): out[r,:] = table[ids[r],:] */
int omx_take_rows(void* out, const void* table, const uint32_t* ids, int64_t n_ids, int dim, omx_dtype dtype,
                  omx_stream stream);
/* elementwise add (mlx_add): residual connections */
int omx_add(void* out, const void* a, const void* b, int64_t n, omx_dtype dtype, omx_stream stream);

/* 2-D synthetic fill of a shard of a logical [*, ld_full] tensor: element (r,c) of dst[rows,cols]
 * takes the value of logical index (row0+r)*ld_full + (col0+c) -- tensor-parallel shards of the
 * same logical weight are bit-identical to slices of the single-GPU tensor.                        */
int omx_fill_uniform_2d(void* dst, int64_t rows, int64_t cols, int64_t ld_full, int64_t row0, int64_t col0,
                        uint32_t seed, float amp, float offset, omx_dtype dtype, omx_stream stream);

/* =====================================================================================
 * Fused decode engine: the qwen3-mlx dense decoder + greedy Generate loop as ONE captured
 * hipGraph per token (qwen3-mlx/src/model.rs:161-215, 263-267, 321-332,
 * 394-424, 480-490, 733-741, 804-843).  Per layer: [RMSNorm+QKV GEMV] [q/k-norm+RoPE+cache
 * write+split-KV attention] [combine] [O GEMV+residual] [RMSNorm+gate/up GEMV+SwiGLU]
 * [down GEMV+residual]; then [RMSNorm+lm_head GEMV+argmax].  Weights are borrowed device
 * pointers registered under their HF checkpoint key names (model.rs:631-716).
 * ===================================================================================== */
typedef struct omx_qwen3_config_ {
    int hidden_size, num_hidden_layers, intermediate_size, num_attention_heads, num_key_value_heads, head_dim,
        vocab_size;
    float rms_norm_eps, rope_theta, rope_scale;   /* rope_scale = 1/factor for "linear" (utils.rs:70-85) */
    int tie_word_embeddings;
    int max_context;                              /* KV slab capacity (rounded up to the 256 step of cache.rs) */
    int tp_rank, tp_size;                         /* tensor parallel shard of this process (1 process per GPU) */
    /* config.json "quantization" {bits, group_size} (qwen3-mlx/src/model.rs:63, 621-727): 0 = bf16 checkpoint;
     * 4 / 8 = every Linear and the embedding are MLX (weight u32, scales, biases) triplets registered as
     * "<prefix>.weight" / ".scales" / ".biases"; the decode step then streams the PACKED weights (csrc/quant.hip) */
    int quant_bits, quant_group;
    /* sparse-MoE feed-forward in every layer (0 experts = dense MLP): Qwen3-MoE (qwen3-mlx/src/qwen3_moe.rs:440-508, mode 1,
     * weights "model.layers.N.mlp.gate.weight" [E, hidden] and "...mlp.switch_mlp.{gate,up,down}_proj.weight" stacked
     * [E, moe_intermediate_size, hidden] / [E, hidden, moe_intermediate_size]) or Mixtral (mixtral-mlx/src/model.rs:280-347,
     * mode 0, the same names under "block_sparse_moe.").  no_qk_norm: attention without q/k RMSNorm (Mixtral, :120-160). */
    int num_experts, num_experts_per_tok, moe_intermediate_size, moe_mode, norm_topk_prob, no_qk_norm;
    /* expert parallelism (one process per GPU): this rank holds experts [ep_rank*E/ep_size, ...) -- the stacked
     * expert tensors registered are ITS slices -- attention and router replicated; one all-reduce per layer (omx_qwen3_set_comm) */
    int ep_rank, ep_size;
    /* Qwen2 wiring (qwen3-mlx/src/qwen2.rs:100-218): q/k/v projections carry a bias ("self_attn.{q,k,v}_proj.bias") and the
     * attention has no q/k norm (set no_qk_norm too); bf16; under tensor parallelism the biases are sharded like their rows */
    int attention_bias;
    /* quantized checkpoints only: scales / biases are float16 (an MLX float16 checkpoint; nn/quantized.rs:361-385 takes any float).
     * They enter the arithmetic as their exact float32 values; activations and outputs stay bf16.  Dense decoders (not with experts). */
    int quant_scales_f16;
} omx_qwen3_config;
typedef struct omx_qwen3_* omx_qwen3;

int omx_qwen3_create(omx_qwen3* out, const omx_qwen3_config* cfg);
int omx_qwen3_destroy(omx_qwen3 m);
/* register a weight by HF key name; `ptr` is this rank's shard (column-split q/k/v/gate/up/lm_head rows,
 * row-split o/down columns), bf16 (packed u32 / bf16 scales, biases for a quantized checkpoint), contiguous, `nbytes` long: the
 * engine reads raw pointers, so a tensor whose size disagrees with the config is refused here ("ShapeMismatch", the reference's
 * load-time shape error) instead of being read past its end                                         */
int omx_qwen3_set_weight(omx_qwen3 m, const char* name, const void* ptr, size_t nbytes);
/* device pointer (+ expected byte length; 0 = unknown) of a registered or synthesised tensor by checkpoint name */
int omx_qwen3_get_weight(omx_qwen3 m, const char* name, const void** ptr, size_t* nbytes);
/* allocate + fill every weight with the seeded synthetic generator (seed = base ^ crc32(name))      */
int omx_qwen3_synth_weights(omx_qwen3 m, uint32_t base_seed);
/* ... with PEAKED logits: embedding std 64, lm_head[v] = table[(v + 1) mod V] at std 0.02 -- the greedy successor of token t is t - 1 with
 * a margin two orders of magnitude above the bf16 bound, so full-size parity tests can assert token EQUALITY (bf16, untied head) */
int omx_qwen3_synth_weights_peaked(omx_qwen3 m, uint32_t base_seed);
/* tensor-parallel hook: `allreduce` has the ncclAllReduce signature, `comm` is the ncclComm_t.      */
int omx_qwen3_set_comm(omx_qwen3 m, void* comm, void* allreduce_fn);
/* sampler (mlx-rs-core/src/sampler.rs:9-18, qwen3-mlx/src/model.rs:733-741): temperature 0 = argmax (default);
 * otherwise every sampled token is categorical(logits * (1/temperature)) with the next key of a RandomState
 * seeded like mlx_rs::random::seed(seed) (random.rs:21-41, :88-91).  The draw stays on the device, inside the step. */
int omx_qwen3_set_sampler(omx_qwen3 m, float temperature, uint64_t seed);
/* text-encoder use of the stack (flux-klein-mlx/src/qwen3_encoder.rs:403-455, `forward_with_hidden_states` + `encode`):
 * all n tokens through layers 0..tap_layers[n_taps-1]; out (DEVICE, bf16) [n, n_taps*hidden] = the raw hidden states
 * after the tapped layers, concatenated on the last axis (FLUX.2-klein: taps 8,17,26 -> 7680 of Qwen3-4B).
 * attention_mask (host, [n] of 0/1, may be NULL = causal only) adds the padding mask of qwen3_encoder.rs:172-198.  */
int omx_qwen3_encode(omx_qwen3 m, const uint32_t* ids, int n, const uint8_t* attention_mask, const int* tap_layers, int n_taps,
                     void* out_dev);
int omx_qwen3_reset(omx_qwen3 m);                                  /* KVCache::reset (cache.rs:130-132) */
int omx_qwen3_offset(omx_qwen3 m, int* offset);                    /* KeyValueCache::offset           */
/* Generate: prefill the prompt, return the first sampled token (model.rs:808-827)                   */
int omx_qwen3_prefill(omx_qwen3 m, const uint32_t* prompt, int n_prompt, uint32_t* first_token);
/* Generate: n further greedy tokens (model.rs:828-841); tokens_out is a HOST buffer of n entries    */
int omx_qwen3_decode(omx_qwen3 m, int n, uint32_t* tokens_out);
/* copy the logits of the last executed step ([vocab_local] bf16) to a host buffer                  */
int omx_qwen3_last_logits(omx_qwen3 m, void* host_bf16, int n);
/* Speculative decoding support (mlx-rs-core/src/speculative.rs).
 *   verify: verify_draft_tokens :132-161 -- the n tokens [last accepted, draft 1 .. draft n-1] in ONE batched pass on top of the cache
 *           (their K/V rows are appended), greedy_out[i] = argmax of the logits after token i; afterwards the pending input token is
 *           greedy_out[n-1].  Single-rank bf16 models.  verify_logits: bf16 logits [V] of one row of the last verify call.
 *   trim:   KeyValueCache::trim(n), the operation :165-169 notes the reference's cache trait lacks: forget the last n cached tokens
 *           and make next_token the pending input token (n = 0: only replace the token).                                             */
/*           With a sampler set (temperature != 0) each row's token is drawn categorical(logits / T) with the next key of the model's
 *           sequence (speculative.rs:104-109, :145-148) instead of the argmax.
 *   sampler_state: the two words of the key sequence (mlx-rs RandomState); set != 0 writes them.  The reference draws the draft's and
 *           the target's tokens from ONE global sequence: two models reproduce it by handing the state over.                        */
int omx_qwen3_verify(omx_qwen3 m, const uint32_t* tokens, int n, uint32_t* greedy_out);
int omx_qwen3_sampler_state(omx_qwen3 m, uint32_t* state2, int set);
int omx_qwen3_verify_logits(omx_qwen3 m, int row, void* host_bf16, int n);
int omx_qwen3_trim(omx_qwen3 m, int n, uint32_t next_token);
/* measurement hook (csrc/per_op_route.hip): qwen3-mlx's Model::forward + Generate::next replayed call for call through the mlx-c handle ABI
 * on this engine's weights (borrowed) -- what an UNMODIFIED crate gets.  tokens_out [n_new + 1]: the token sampled from the prompt, then n_new
 * greedy tokens; host wall-clock per decoded token and mlx_* calls per token. */
int omx_bench_qwen3_per_op(omx_qwen3 model, const omx_qwen3_config* cfg, const uint32_t* prompt, int n_prompt, int n_new,
                           uint32_t* tokens_out, double* prefill_ms, double* ms_per_token, double* calls_per_token);
/* timing of the last omx_qwen3_decode call measured with HIP events on the engine stream (ms)       */
int omx_qwen3_last_decode_ms(omx_qwen3 m, float* ms);
int omx_qwen3_last_prefill_ms(omx_qwen3 m, float* ms);   /* same for the last omx_qwen3_prefill call */
/* test hook: copy an internal bf16 buffer ("h","h2","qkv","attn_out","act","k<l>","v<l>") to the host */
int omx_qwen3_debug_read(omx_qwen3 m, const char* name, void* host, size_t n_elems);
int omx_qwen3_stream(omx_qwen3 m, omx_stream* s);
/* algorithmic HBM bytes of ONE decode step at context length ctx (SURVEY.md 8d formula)             */
int omx_qwen3_step_bytes(omx_qwen3 m, int ctx, double* bytes);
/* how the decode step is executed once built: 0 = not built yet, 1 = step graph (one launch per phase,
 * replayed as a hipGraph; one graph per 1024-token context bucket), 2 = the same launches issued eagerly */
int omx_qwen3_decode_path(omx_qwen3 m, int* path);
/* debug hook (tools/attn_step_trace.py): run ONE decode step eagerly with the attention launches stamping the
 * 100 MHz wall clock; host receives [layers][attention splits][kv heads][8], *blocks = splits * kv heads     */
int omx_qwen3_debug_trace_step(omx_qwen3 m, unsigned long long* host, size_t n_words, int* blocks);
/* measurement hook: `steps` real decode steps run eagerly, every launch of the per-layer kernels carrying its own HIP event pair
 * (the dispatch's begin / end timestamps on the step's stream); us[7] = average microseconds of {QKV GEMV, attention, O GEMV,
 * gate/up + SwiGLU GEMV, down GEMV, lm_head, persistent step launch (all layers; the five per-layer figures are 0 then)}.  bench.py's roofline.achieved is the gate/up figure.  Dense bf16 single-rank models. */
int omx_qwen3_time_step_kernels(omx_qwen3 m, int steps, float* us);
/* debug hook (tools/step_engine_trace.py): ONE eager decode step on the persistent engine (csrc/step_engine.hip, OMX_STEP_ENGINE=1)
 * with per-CU wall-clock stamps; host receives [CUs][64] words, *cus = CUs of the device */
int omx_qwen3_debug_trace_engine(omx_qwen3 m, unsigned long long* host, size_t n_words, int* cus);

/* =====================================================================================
 * MLX affine group quantisation (the reference's flagship checkpoint format).
 * mlx_rs::ops::{quantize, dequantize, quantized_matmul, gather_qmm} (mlx-rs/src/ops/quantization.rs:41-153,
 * 226-279) -> mlx_quantize / mlx_dequantize / mlx_quantized_matmul / mlx_gather_qmm (mlx-c ops.h:356-365,
 * 471-484, 793-810).  w [rows, cols] <-> packed u32 [rows, cols*bits/32] (LSB first) + scales, biases
 * [rows, cols/group_size] (dtype of w); group_size 32/64/128.  quantize / dequantize: bits 2, 4 or 8 on bf16 / f16 / f32 (the
 * reference's own value test loops [2, 4, 8] on f32, quantization.rs:289-305).  The matmuls: bits 4 or 8; dtype = the dtype of x,
 * out, scales and biases alike -- bf16, or f16 for a float16 MLX checkpoint, which then runs in float16 END TO END like in MLX
 * (nn/quantized.rs:361-385): float16 activations, float16 rounding points, f32 accumulation.
 * ===================================================================================== */
int omx_quantize(void* packed, void* scales, void* biases, const void* w, int64_t rows, int cols, int group_size, int bits,
                 omx_dtype dtype, omx_stream stream);
int omx_dequantize(void* out, const void* packed, const void* scales, const void* biases /* may be NULL */, int64_t rows,
                   int cols, int group_size, int bits, omx_dtype dtype, omx_stream stream);
/* out [M, N] = x [M, K] . dequant(W [N, K])^T (transpose = true, nn::QuantizedLinear::forward quantized.rs:366-375):
 * M <= 16 streams the packed weights (GEMV), larger M dequantises into the library workspace and runs the MFMA GEMM */
int omx_quantized_matmul(void* out, const void* x, const void* packed, const void* scales, const void* biases, int M, int N,
                         int K, int group_size, int bits, omx_dtype dtype, omx_stream stream);
/* out [n_rows, N]: row i = x[i / x_div] . dequant(W[rhs_indices[i]])^T over expert-stacked packed weights
 * [n_experts, N, K*bits/32] (QuantizedSwitchLinear::apply, mixtral-mlx/src/model.rs:195-201)                */
int omx_gather_qmm(void* out, const void* x, const void* packed, const void* scales, const void* biases,
                   const uint32_t* rhs_indices, int n_rows, int x_div, int N, int K, int n_experts, int group_size, int bits,
                   omx_dtype dtype, omx_stream stream);

/* dense sibling (mlx_gather_mm, ops.h:463; mlx-rs/src/ops/quantization.rs:169-203): out [n_rows, N], row i = x[i / x_div] . W[rhs_indices[i]]^T
 * over expert-stacked bf16 weights [n_experts, N, K]; expert-selected batched GEMV, K a multiple of 8                      */
int omx_gather_mm(void* out, const void* x, const void* w, const uint32_t* rhs_indices, int n_rows, int x_div, int N, int K,
                  int n_experts, omx_dtype dtype, omx_stream stream);

/* =====================================================================================
 * a6 + a7: sparse-MoE block = router + top-k + SwitchGLU + weighted sum.
 *   mode 0  MixtralSparseMoeBlock::forward (mixtral-mlx/src/model.rs:296-308): top-k of the gate logits,
 *           softmax (precise) over the SELECTED logits;
 *   mode 1  MoeBlock::forward of Qwen3-MoE (qwen3-mlx/src/qwen3_moe.rs:475-503): softmax over all experts,
 *           top-k, scores renormalised when norm_topk_prob.
 *   experts SwitchGLU::forward_experts (model.rs:243-274) = mlx_gather_qmm x3 (ops.h:471-484) + fused_swiglu;
 *           here with dense bf16 stacked weights w_gate/w_up [E, inter, hidden], w_down [E, hidden, inter]
 *           (mlx_gather_mm form, ops.h:463-470).
 * x [n_tokens, hidden] -> out [n_tokens, hidden]; inds_out [n_tokens, top_k] u32 and scores_out
 * [n_tokens, top_k] bf16 are optional device outputs (descending score order).  Scratch comes from the
 * library workspace (omx_moe_workspace_bytes tells how much; omx_set_workspace to provide it).
 * ===================================================================================== */
int omx_moe_workspace_bytes(int n_tokens, int hidden, int inter, int n_experts, int top_k, size_t* bytes);
int omx_moe_forward(void* out, const void* x, const void* gate_w, const void* w_gate, const void* w_up,
                    const void* w_down, int n_tokens, int hidden, int inter, int n_experts, int top_k, int mode,
                    int norm_topk_prob, uint32_t* inds_out, void* scores_out, omx_stream stream);
/* decoder-block form (mixtral model.rs:340-345, qwen3_moe.rs decoder layer): out = resid + moe(rmsnorm(x, norm_w, eps));
 * the norm is folded into the router launch, the residual into the combine launch; xn = [n_tokens, hidden] scratch */
int omx_moe_block_forward(void* out, const void* resid, const void* x, const void* norm_w, float eps, void* xn, const void* gate_w,
                          const void* w_gate, const void* w_up, const void* w_down, int n_tokens, int hidden, int inter,
                          int n_experts, int top_k, int mode, int norm_topk_prob, omx_stream stream);
/* expert-parallel decode form: this rank holds experts [e_lo, e_lo + e_n) (w_* = its stacks);
 * partial [n_tokens, hidden] f32 = its share of sum_j bf16(y_j * score_j), to be all-reduced over the ranks; the residual
 * h + bf16(sum) is the caller's.  n_tokens * top_k <= 32. */
/* tables of the batched expert-parallel block for an exchange-based combine (omx_moe_block_slots_ep -> omx_peer_moe_combine) */
typedef struct omx_moe_ep_slots_ {
    const void* y;                 /* [local slots (sorted by local expert), hidden] bf16 expert outputs */
    const uint32_t* pos_of_slot;   /* [n_tokens * top_k] slot -> row of y (slots of other ranks' experts: unused) */
    const uint32_t* inds;          /* [n_tokens * top_k] global expert id per slot */
    const void* scores;            /* [n_tokens * top_k] bf16 routing weights */
} omx_moe_ep_slots;
int omx_moe_block_partial_ep(float* partial, const void* x, const void* norm_w, float eps, void* xn, const void* gate_w,
                             const void* w_gate, const void* w_up, const void* w_down, int n_tokens, int hidden, int inter,
                             int n_experts, int top_k, int mode, int norm_topk_prob, int e_lo, int e_n, omx_stream stream);
/* the two sharded forms on MLX-packed expert stacks (the reference's own Mixtral format: mixtral-mlx/src/model.rs:466-615): this rank's packed
 * experts / columns -- the triplet of a slice is the slice of the triplet; bf16 scales.  Contracts of omx_moe_block_partial_ep / _tp. */
int omx_moe_block_partial_ep_q(float* partial, const void* x, const void* norm_w, float eps, void* xn, const void* q_router, const void* s_router,
                               const void* b_router, const void* q_gate, const void* s_gate, const void* b_gate, const void* q_up,
                               const void* s_up, const void* b_up, const void* q_down, const void* s_down, const void* b_down, int n_tokens,
                               int hidden, int inter, int n_experts, int top_k, int mode, int norm_topk_prob, int e_lo, int e_n, int group_size,
                               int bits, int f16, omx_stream stream);
int omx_moe_block_partial_tp_q(float* y_partial, uint32_t* route_inds, void* route_scores, const void* x, const void* norm_w, float eps,
                               const void* q_router, const void* s_router, const void* b_router, const void* q_gate, const void* s_gate,
                               const void* b_gate, const void* q_up, const void* s_up, const void* b_up, const void* q_down, const void* s_down,
                               const void* b_down, int n_tokens, int hidden, int inter, int n_experts, int top_k, int mode, int norm_topk_prob,
                               int group_size, int bits, int f16, omx_stream stream);
int omx_moe_block_slots_ep(omx_moe_ep_slots* out, const void* x /* normalised rows */, const void* gate_w, const void* w_gate, const void* w_up,
                           const void* w_down, int n_tokens, int hidden, int inter, int n_experts, int top_k, int mode, int norm_topk_prob,
                           int e_lo, int e_n, omx_stream stream);
/* expert TENSOR parallel decode form (<= 32 routed slots): this rank holds `inter` = I / tp intermediate columns of EVERY expert;
 * y_partial [slots, hidden] f32 = the unrounded partial down projections (to be all-reduced), the replicated router's choice goes to
 * route_inds / route_scores; omx_moe_combine_slots then forms bf16(resid + bf16(sum_j bf16(bf16(y_j) score_j))) (mixtral model.rs:296-308,
 * 343-344 with the single-device roundings).  No reference counterpart: mlx-c distributed.h:30-70 is declared, never bound. */
int omx_moe_block_partial_tp(float* y_partial, uint32_t* route_inds, void* route_scores, const void* x, const void* norm_w, float eps,
                             void* xn, const void* gate_w, const void* w_gate, const void* w_up, const void* w_down, int n_tokens,
                             int hidden, int inter, int n_experts, int top_k, int mode, int norm_topk_prob, omx_stream stream);
int omx_moe_combine_slots(void* out, const float* y_slots, const void* scores, const void* resid, int n_tokens, int hidden, int top_k,
                          omx_stream stream);
/* f16 != 0 (also the last int of the two packed-stack entry points above): a float16 checkpoint -- activations, scores, residual and every
 * rounding point float16 (nn/quantized.rs:361-385: the dequantised weight has the scales' dtype), f32 accumulation; decode form only */
int omx_moe_combine_slots_ex(void* out, const float* y_slots, const void* scores, const void* resid, int n_tokens, int hidden, int top_k,
                             int f16, omx_stream stream);
/* the same on a quantised checkpoint (mixtral-mlx/src/model.rs:560-600): router and expert stacks as MLX triplets */
/* one token, bf16 experts: the block WITHOUT its weighted sum -- partials[j, :] (f32) = bf16(bf16(y_j) * score_j) of the top_k routed
 * experts in slot order; the caller's next streaming GEMV folds x := bf16(resid + bf16(sum_j partials[j])) in its prologue (engine-internal:
 * csrc/gemv.hpp GemvArgs::x_partial_n).  Returns 2 when the shape does not take the batched-GEMV route.                              */
int omx_moe_block_partials(float* partials, const void* x, const void* norm_w, float eps, void* xn, const void* gate_w, const void* w_gate,
                           const void* w_up, const void* w_down, int hidden, int inter, int n_experts, int top_k, int mode,
                           int norm_topk_prob, omx_stream stream);
int omx_moe_block_forward_q(void* out, const void* resid, const void* x, const void* norm_w, float eps, void* xn,
                            const void* q_router, const void* s_router, const void* b_router, const void* q_gate, const void* s_gate,
                            const void* b_gate, const void* q_up, const void* s_up, const void* b_up, const void* q_down,
                            const void* s_down, const void* b_down, int n_tokens, int hidden, int inter, int n_experts, int top_k,
                            int mode, int norm_topk_prob, int group_size, int bits, omx_stream stream);
int omx_moe_block_forward_q_ex(void* out, const void* resid, const void* x, const void* norm_w, float eps, void* xn,
                            const void* q_router, const void* s_router, const void* b_router, const void* q_gate, const void* s_gate,
                            const void* b_gate, const void* q_up, const void* s_up, const void* b_up, const void* q_down,
                            const void* s_down, const void* b_down, int n_tokens, int hidden, int inter, int n_experts, int top_k,
                            int mode, int norm_topk_prob, int group_size, int bits, int f16 /* float16 checkpoint: x, norm weights, triplets and out are float16; <= 32 routed slots */, omx_stream stream);
/* the reference's own Mixtral format (mixtral-mlx/src/model.rs:182-274, QuantizedSwitchLinear -> mlx_gather_qmm x3):
 * expert stacks as MLX affine-quantised triplets, packed u32 [E, out, in*bits/32], scales / biases [E, out, in/group_size];
 * router gate bf16.  <= 32 routed slots: expert-selected GEMVs on the packed weights; more: dequantise + grouped GEMM. */
int omx_moe_forward_q(void* out, const void* x, const void* gate_w, const void* q_gate, const void* s_gate, const void* b_gate,
                      const void* q_up, const void* s_up, const void* b_up, const void* q_down, const void* s_down,
                      const void* b_down, int n_tokens, int hidden, int inter, int n_experts, int top_k, int mode,
                      int norm_topk_prob, int group_size, int bits, uint32_t* inds_out, void* scores_out, omx_stream stream);
/* The same block as three stages, for an expert-parallel host that exchanges token rows between them
 * (all-to-all dispatch + combine; ominix-mlx_amd/ep.py):
 *   route    x [n, hidden] -> inds [n, k] u32, scores [n, k] (dtype of x)            (model.rs:296-302)
 *   experts  rows [m, hidden] that already carry their (local) expert id -> y [m, hidden]
 *            = down_e(fused_swiglu(up_e x, gate_e x))                                 (model.rs:243-274)
 *   combine  y [n, k, hidden], scores [n, k] -> out [n, hidden]                      (model.rs:304-307) */
int omx_moe_route(uint32_t* inds_out, void* scores_out, const void* x, const void* gate_w, int n_tokens, int hidden,
                  int n_experts, int top_k, int mode, int norm_topk_prob, omx_stream stream);
int omx_moe_experts(void* y, const void* x_rows, const uint32_t* expert_ids, int n_rows, const void* w_gate,
                    const void* w_up, const void* w_down, int hidden, int inter, int n_experts, omx_stream stream);
int omx_moe_combine(void* out, const void* y_slots, const void* scores, int n_tokens, int hidden, int top_k,
                    omx_stream stream);

/* =====================================================================================
 * a14 (+ a9): FLUX.2-klein DiT / MMDiT forward -- FluxKlein::forward_with_rope
 * (flux-klein-mlx/src/klein_model.rs:799-854) with KleinDoubleBlock::forward (:399-522) and
 * KleinSingleBlock::forward (:603-674) as fused launch sequences.  Weights are registered under the
 * reference's internal names (double_blocks.{i}.img_to_q.weight, single_blocks.{i}.to_qkv_mlp.weight,
 * x_embedder.weight, ...; flux-klein-mlx/src/weights.rs:479-596), bf16, [out, in].
 * ===================================================================================== */
typedef struct omx_klein_config_ {   /* FluxKleinParams, klein_model.rs:166-196 */
    int in_channels, hidden_size, txt_embed_dim, num_heads, depth, depth_single, head_dim, mlp_hidden;
    int tp_rank, tp_size;   /* tensor parallel over one node (0/0 or 0/1 = single GPU): heads and MLP width sharded */
} omx_klein_config;
typedef struct omx_klein_* omx_klein;
int omx_klein_create(omx_klein* out, const omx_klein_config* cfg);
int omx_klein_destroy(omx_klein m);
/* `nbytes`: length of the tensor behind `ptr`; every use checks it against the rows x columns it is about to read */
int omx_klein_set_weight(omx_klein m, const char* name, const void* ptr, size_t nbytes);
int omx_klein_synth_weights(omx_klein m, uint32_t base_seed);   /* under TP: this rank's shards of the same logical tensors */
/* comm: ncclComm_t, allreduce_fn: address of ncclAllReduce (RCCL); one bf16 all-reduce after every row-split
 * projection (2 per double-block stream, 1 per single block) */
int omx_klein_set_comm(omx_klein m, void* comm, void* allreduce_fn);
/* latent [s_img, in_channels], txt_embed [s_txt, txt_embed_dim] (device bf16); timestep = t * 1000
 * (generate_klein.rs:434); rope_cos/rope_sin [s_txt + s_img, 128] device f32 with duplicated pair entries,
 * text rows first (compute_rope_freqs, klein_model.rs:53-110); out [s_img, in_channels] device bf16.      */
int omx_klein_forward_with_rope(omx_klein m, void* out, const void* latent, const void* txt_embed, int s_img, int s_txt,
                                float timestep, const float* rope_cos, const float* rope_sin);
/* ---- FLUX VAE decoder: flux-klein-mlx/src/autoencoder.rs Decoder::forward :375-412 with
 *      ResnetBlock :139-157, AttnBlock :195-232, GroupNorm(32, eps 1e-5, pytorch compatible).  NHWC bf16.
 *      Weights (bf16, device) under the names weights.rs:164-217 produces: conv_in.{weight,bias},
 *      mid_block_resnets_{0,1}.{norm1,conv1,norm2,conv2}.*, mid_block_attentions_0.{group_norm,to_q,to_k,to_v,to_out}.*,
 *      up_blocks.{b}.resnets.{j}.{norm1,conv1,norm2,conv2,conv_shortcut}.*, up_blocks.{b}.upsamplers_0_conv.*,
 *      conv_norm_out.*, conv_out.*, post_quant_conv.*; Conv2d weight [out, kH, kW, in], Linear [out, in].
 *      latent [h, w, z_channels] -> image [h*2^(n_mult-1), w*2^(n_mult-1), out_ch] in [-1, 1].                  ---- */
typedef struct omx_vae_config_ {     /* AutoEncoderConfig, autoencoder.rs:22-81 (flux2(): ch 128, mult 1,2,4,4, z 32) */
    int ch, ch_mult[8], n_mult, num_res_blocks, z_channels, out_ch;
    float scale_factor, shift_factor;
} omx_vae_config;
typedef struct omx_vae_decoder_* omx_vae_decoder;
int omx_vae_decoder_create(omx_vae_decoder* out, const omx_vae_config* cfg);
int omx_vae_decoder_destroy(omx_vae_decoder m);
int omx_vae_decoder_set_weight(omx_vae_decoder m, const char* name, const void* ptr);
int omx_vae_decoder_out_shape(omx_vae_decoder m, int h, int w, int* out_h, int* out_w, int* out_c);
int omx_vae_decode(omx_vae_decoder m, void* image, const void* latent, int h, int w);
int omx_vae_decoder_last_ms(omx_vae_decoder m, float* ms);

/* one Euler step of the rectified-flow sampler (flux-klein-mlx/examples/generate_klein.rs:441-443, src/sampler.rs:174-186):
 * latent_f32[i] += dt * v_bf16[i]; latent_bf16 (may be NULL) receives the rounded copy the next forward reads          */
int omx_klein_euler_step(void* latent_f32, const void* v_bf16, float dt, void* latent_bf16, int64_t n, omx_stream stream);
int omx_klein_last_ms(omx_klein m, float* ms);          /* HIP-event time of the last forward */
int omx_klein_debug_read(omx_klein m, const char* name, void* host, size_t n_elems);   /* test hook */

/* In-process stand-in for an RCCL communicator (csrc/loopback_comm.hip): `world` engine instances driven by
 * `world` host threads of one process exchange through it, so the tensor-parallel paths run with real shards
 * on a single-GPU box.  omx_loopback_allreduce has ncclAllReduce's signature.                             */
typedef struct omx_loopback_* omx_loopback;
int omx_loopback_create(omx_loopback* out, int world, size_t max_bytes);
int omx_loopback_destroy(omx_loopback g);
void* omx_loopback_rank_comm(omx_loopback g, int rank);
int omx_loopback_abort(omx_loopback g);

/* One-shot all-reduce over xGMI peer stores for the tensor-parallel decode step (csrc/peer_allreduce.hip): one
 * process per GPU, every rank's inbox (fine-grained device memory) mapped into every peer through HIP IPC handles, a call = ONE
 * kernel that stores tagged 8-byte granules into all inboxes and reduces its own in rank order.  f32 sum / u64 max of up to 8192
 * payload words.  Larger f32 / bf16 sums (16-byte multiples) take the TWO-SHOT path when no RCCL communicator was given at creation
 * (or OMX_PEER_LARGE=1): one kernel, slice s of every rank's input pushed into rank s's stage, summed there in rank order, pushed
 * back to all -- stages of OMX_PEER_STAGE_MB (64) behind every inbox, larger messages in chunks.  Anything else goes to the RCCL
 * communicator given at creation (may be NULL: such calls then fail).
 * omx_peer_allreduce has ncclAllReduce's signature: omx_qwen3_set_comm(model, comm, omx_peer_allreduce_fn()).
 * Host protocol: create -> handle (64 bytes) -> all-gather the handles rank-major -> connect.                                 */
int omx_peer_comm_create(void** out, int rank, int world, void* rccl_comm, void* rccl_allreduce_fn);
int omx_peer_comm_handle(void* comm, void* out64);
int omx_peer_comm_connect(void* comm, const void* handles);
int omx_peer_allreduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, omx_stream stream);
void* omx_peer_allreduce_fn(void);
int omx_peer_comm_counts(void* comm, unsigned long long* out4);   /* launches by path: one-shot, two-shot chunks, MoE combine, handed to RCCL */
size_t omx_peer_comm_stage_bytes(void* comm);             /* stage size of the two-shot / exchange path; 0 = off */
/* expert-parallel prompt: out [T, hidden] = resid + weighted expert outputs as an all-to-all combine of the routed slots' rows to their
 * tokens' owners + an all-gather of the finished rows (one kernel; BASELINE config 3's "expert-parallel all-to-all over xGMI").
 * Returns 2 when this communicator / size cannot take it (the caller keeps its all-reduce). */
int omx_peer_moe_combine(void* out, const void* resid, const omx_moe_ep_slots* slots, int T, int hidden, int top_k, int e_lo, int e_n, void* comm,
                         omx_stream stream);
const void* omx_peer_comm_device(void* comm);              /* device table for kernels that reduce their own output (engine-internal use) */
int omx_peer_comm_set_scope(void* comm, int system_scope); /* two-shot / exchange hand-offs: 1 system-scope release / acquire (default), 0 agent-scope fences (ranks on ONE GPU) */
int omx_peer_comm_scope(void* comm);
int omx_peer_device_id(char* out, int len);                /* PCI bus id of the current device (ranks compare them: same GPU or not) */
int omx_peer_comm_status(void* comm, unsigned* aborted);   /* 1: a wait gave up (a peer never arrived); results are void */
int omx_peer_comm_destroy(void* comm);

/* =====================================================================================
 * a12: Paraformer mel/STFT frontend (funasr-mlx/src/paraformer.rs:195-412), all on device:
 * x*32768 -> pre-emphasis 0.97 -> frames (n-400)/160+1 -> Hamming -> 400-pt DFT power -> 80 HTK
 * mel filters -> ln(max(.,1e-10)) -> LFR(7,6) -> CMVN.  Replaces MelFrontend::{new,set_cmvn,forward}.
 * ===================================================================================== */
typedef struct omx_mel_config_ {   /* ParaformerConfig frontend fields, paraformer.rs:110-145 */
    int sample_rate, n_mels, n_fft, hop_length, lfr_m, lfr_n;
} omx_mel_config;
typedef struct omx_mel_frontend_* omx_mel_frontend;
int omx_mel_frontend_create(omx_mel_frontend* out, const omx_mel_config* cfg);          /* MelFrontend::new  :195-222 */
int omx_mel_frontend_destroy(omx_mel_frontend f);
/* MelFrontend::set_cmvn :225-228; host vectors of lfr_m*n_mels floats (am.mvn <AddShift>/<Rescale>) */
int omx_mel_frontend_set_cmvn(omx_mel_frontend f, const float* addshift_host, const float* rescale_host, int dim);
/* frame counts for n_samples: STFT frames (1 when shorter than n_fft, :386-388) and LFR frames */
int omx_mel_frontend_frames(omx_mel_frontend f, int64_t n_samples, int* n_frames, int* n_lfr);
/* MelFrontend::forward :278-367.  audio: device f32 [n_samples]; feats: device f32 [n_lfr, lfr_m*n_mels];
 * logmel_out ([n_frames, n_mels]) and power_out ([n_frames, n_fft/2+1]) are optional device outputs.
 * Non-finite samples are an error ("Audio contains NaN or Inf values", :284-286).                     */
int omx_mel_frontend_forward(omx_mel_frontend f, const float* audio, int64_t n_samples, float* feats, float* logmel_out,
                             float* power_out, omx_stream stream);

/* sibling frontend: Fun-ASR-Nano / SenseVoice log-mel, funasr-nano-mlx/src/audio.rs:44-157 (`MelFrontend::{new,
 * compute_mel_spectrogram}`, defaults 16 kHz / 80 mels / n_fft 400 / hop 160 / 30 s): symmetric Hann, frames from sample frame * hop with
 * zero padding, n_frames = max(min(n, max_length * sr) / hop, 1), DFT power, FFT-bin triangles (:287-339), ln(max(., 1e-10)).
 * audio: device f32 [n_samples]; out: device f32 [n_mels, n_frames].  Errors as the reference: empty (:95-99), shorter than a hop (:105-110).
 * omx_apply_lfr = `apply_lfr` (:345-412): mel [n_mels, n_frames] -> out [ceil(n_frames / lfr_n), lfr_m * n_mels], centred, edge-clamped. */
int omx_sensevoice_mel_create(omx_mel_frontend* out, int sample_rate, int n_mels, int n_fft, int hop_length, float max_length_s);
int omx_sensevoice_mel_frames(omx_mel_frontend f, int64_t n_samples, int* n_frames);
int omx_sensevoice_mel_forward(omx_mel_frontend f, const float* audio, int64_t n_samples, float* out, omx_stream stream);
int omx_apply_lfr(float* out, const float* mel, int n_mels, int n_frames, int lfr_m, int lfr_n, omx_stream stream);

/* audio::resample (mlx-rs-core/src/audio.rs:178-277): windowed-sinc resampling with the reference's rubato configuration (sinc_len 256,
 * oversampling 256, cubic phase interpolation, BlackmanHarris2, cutoff 0.95) and its chunking / flush / truncation driver.
 * in: device f32 [n_in]; out: device f32 [out_cap], out_cap >= omx_resample_len(n_in, src, dst) = round(n_in * dst / src), the
 * reference's truncation bound; *n_out = samples written (fewer than the bound for inputs shorter than the filter, as in the
 * reference).  Equal rates / empty input: the input unchanged (:179-181).  Synchronises the stream before returning.             */
int64_t omx_resample_len(int64_t n_in, uint32_t src_rate, uint32_t dst_rate);
int omx_resample_sinc(const float* in, int64_t n_in, uint32_t src_rate, uint32_t dst_rate, float* out, int64_t out_cap, int64_t* n_out,
                      omx_stream stream);

/* sibling frontend: WhisperFeatureExtractor-compatible log-mel, qwen3-asr-mlx/src/audio.rs:24-128
 * (`MelFrontend::{new, compute_mel_spectrogram}`, defaults 16 kHz / 128 mels / n_fft 400 / hop 160): periodic Hann ->
 * 400-pt DFT power -> Slaney filters -> log10(max(., 1e-10)) -> max(., global max - 8) -> (x + 4) / 4.
 * audio: device f32 [n_samples]; out: device f32 [n_mels, n_frames], n_frames = 1 + (n_samples - n_fft) / hop.      */
int omx_whisper_mel_create(omx_mel_frontend* out, int sample_rate, int n_mels, int n_fft, int hop_length);
int omx_whisper_mel_frames(omx_mel_frontend f, int64_t n_samples, int* n_frames);
int omx_whisper_mel_forward(omx_mel_frontend f, const float* audio, int64_t n_samples, float* out, omx_stream stream);

/* =====================================================================================
 * a13: Paraformer body pieces (funasr-mlx/src/paraformer.rs).
 *   omx_sanm_encoder_layer  SanmEncoderLayer::forward :618-634 = LayerNorm(1e-5) -> SanmAttention (:496-532:
 *       fused qkv Linear, softmax(q k^T d^-1/2) v with heads x 128, FSMN depthwise Conv1d(k=11, pad 5) over v
 *       plus v, out_proj) -> residual (skipped when in_dim != dim, :625-629) -> LayerNorm -> Linear/ReLU/Linear
 *       (:560-570) -> residual.  x [T, in_dim] -> out [T, dim]; weights: Linear [out, in] + bias [out];
 *       fsmn_w [dim, kernel_size] (MLX Conv1d weight [C, k, 1], :1293-1298).
 *   Every entry point takes the arithmetic mode as `dtype` -- activations AND weights are of that type:
 *       OMX_FLOAT32   the reference's own (f32 checkpoint, f32 activations; exact-f32 matrix cores, explicit softmax);
 *       OMX_BFLOAT16  bf16 with fp32 accumulation (faster; narrower than the reference).
 *   omx_cif_fire            CIFPredictor::cif_fire :779-879 (threshold 1.0, tail 0.45): hidden [B, T, H] f32,
 *       alphas [B, T] f32 -> frames [B, max_frames, H] f32 (zero padded) and counts [B]; all device pointers.
 * ===================================================================================== */
typedef struct omx_sanm_layer_weights_ {
    const void *norm1_w, *norm1_b, *qkv_w, *qkv_b, *out_w, *out_b, *fsmn_w, *norm2_w, *norm2_b, *ffn_up_w, *ffn_up_b,
        *ffn_down_w, *ffn_down_b;
} omx_sanm_layer_weights;
int omx_sanm_encoder_layer(void* out, const void* x, const omx_sanm_layer_weights* w, int T, int in_dim, int dim,
                           int heads, int ffn_dim, int kernel_size, omx_dtype dtype, omx_stream stream);
/* SanmEncoder::forward's layer loop + after_norm (paraformer.rs:691-708) in ONE call (round 6): layers[0] maps in_dim -> dim, the others
 * keep dim; out [T, dim] = after_norm(layers(x)).  Scratch (device, caller-owned): act0 / act1 [T, dim], nrm0 / nrm1 [T, max(in_dim, dim)].
 * In float32 every layer's last launch also computes the next layer's norm1 (the last one: after_norm). */
int omx_sanm_encoder_stack(void* out, const void* x, const omx_sanm_layer_weights* layers, int n_layers, int T, int in_dim, int dim, int heads,
                           int ffn_dim, int kernel_size, const void* after_norm_w, const void* after_norm_b, void* act0, void* act1, void* nrm0,
                           void* nrm1, omx_dtype dtype, omx_stream stream);
int omx_cif_fire(float* frames, int* counts, const float* hidden, const float* alphas, int batch, int T, int H,
                 float threshold, float tail_threshold, int max_frames, omx_stream stream);
/* SanmEncoder::forward prologue (paraformer.rs:691-703): out [T, dim] = mel f32 [T, dim] * sqrt(512) + sinusoidal
 * position encoding (positions 1.., [sin | cos] halves, :418-439) */
int omx_paraformer_embed(void* out, const float* mel, int T, int dim, omx_dtype dtype, omx_stream stream);
/* CIFPredictor::compute_alphas (:761-768): alphas f32 [T] = sigmoid(Linear_{dim->1}(relu(Conv1d_{k}(enc)))) with the dense
 * convolution weight in MLX layout [dim_out, k, dim_in]; also writes enc as f32 (the hidden CIF integrates)           */
int omx_cif_alphas(float* alphas, float* hidden_f32, const void* enc, const void* conv_w, const void* conv_b,
                   const void* proj_w, const void* proj_b, int T, int dim, int kernel_size, omx_dtype dtype, omx_stream stream);
/* ParaformerDecoderLayer::forward (:1030-1053) incl. cross_attention (:981-1017): x [N, dim] acoustic embeddings,
 * enc [Ts, enc_dim] encoder output; FFN down has no bias (loader :1421)                                              */
typedef struct omx_paraformer_decoder_weights_ {
    const void *norm1_w, *norm1_b, *ffn_up_w, *ffn_up_b, *ffn_norm_w, *ffn_norm_b, *ffn_down_w, *norm2_w, *norm2_b, *fsmn_w,
        *norm3_w, *norm3_b, *q_w, *q_b, *kv_w, *kv_b, *out_w, *out_b;
} omx_paraformer_decoder_weights;
int omx_paraformer_decoder_layer(void* out, const void* x, const void* enc, const omx_paraformer_decoder_weights* w, int N,
                                 int Ts, int dim, int enc_dim, int heads, int ffn_dim, int kernel_size, omx_dtype dtype,
                                 omx_stream stream);
/* The float32 model's attention as one op (funasr-mlx/src/paraformer.rs:509-516 encoder self-attention, :1090-1102 decoder cross-attention):
 * out = softmax(q k^T / sqrt(128)) v per head of width 128, f32 throughout, no mask; rows ldq / ldkv / ldo floats apart, head h at column
 * 128 h (so q | k | v may be the three thirds of one fused projection). */
int omx_paraformer_attention_f32(float* out, const float* q, const float* k, const float* v, int64_t ldq, int64_t ldkv, int64_t ldo, int Tq,
                                 int Tk, int heads, omx_stream stream);
/* ParaformerDecoder::forward's layer loop (:1144-1156) in one call: out [N, dim] = layers(x).  Scratch: act0/act1 [N, dim], nrm0/nrm1 [N, dim],
 * kv_all [Ts, n_layers * 2 * dim] or null.  With kv_all given and the layers' linear_k_v weights / biases back to back in memory, the encoder
 * output's k | v projection for ALL layers is one GEMM up front; in float32 each layer's last launch also computes the next layer's norm1. */
int omx_paraformer_decoder_stack(void* out, const void* x, const void* enc, const omx_paraformer_decoder_weights* layers, int n_layers, int N,
                                 int Ts, int dim, int enc_dim, int heads, int ffn_dim, int kernel_size, void* act0, void* act1, void* nrm0,
                                 void* nrm1, void* kv_all, omx_dtype dtype, omx_stream stream);
/* ParaformerDecoder::forward tail (:1157-1165): LN -> Linear+ReLU -> LN(ffn) -> Linear(no bias) -> LN -> output_proj;
 * logits [N, vocab]                                                                                                    */
typedef struct omx_paraformer_tail_weights_ {
    const void *norm1_w, *norm1_b, *up_w, *up_b, *ffn_norm_w, *ffn_norm_b, *down_w, *after_norm_w, *after_norm_b, *out_w, *out_b;
} omx_paraformer_tail_weights;
int omx_paraformer_decoder_tail(void* logits, const void* x, const omx_paraformer_tail_weights* w, int N, int dim, int ffn_dim,
                                int vocab, omx_dtype dtype, omx_stream stream);
/* elementwise dtype conversion between bf16 / f16 / f32 device buffers (Array::as_dtype) */
int omx_cast(void* dst, omx_dtype dst_dtype, const void* src, omx_dtype src_dtype, int64_t n, omx_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* OMX_H */
