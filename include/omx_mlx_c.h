/*
 * omx_mlx_c.h -- the hot-path subset of the mlx-c C ABI, exported by libomx_hip.so with the
 * reference's exact symbol names, argument order, ownership and status conventions, so that
 * `mlx-sys` (bindgen over mlx/c/mlx.h, mlx-rs/mlx-sys/build.rs:390-399) can link the MI355X
 * library in place of libmlxc.a + libmlx.a for this path.
 *
 * Every declaration cites the header:line it mirrors under
 *   /root/reference/mlx-rs/mlx-sys/src/mlx-c/mlx/c/
 * and, where the hot path calls it, the Rust call site.
 *
 * Conventions kept:
 *   - handles are by-value structs { void* ctx }, ctx == NULL is "empty" (array.h:28-30);
 *   - ops take `mlx_array* res` and ASSIGN into it (free the old value if non-empty,
 *     private/array.h:24-41); inputs are borrowed; the caller frees with mlx_array_free;
 *   - "may be null" optional inputs are empty handles (fast.h:96-97,166,177,196-197);
 *   - every op returns int, 0 = ok, 1 = error after invoking the registered error handler on the
 *     calling thread (error.cpp:37-54); mlx_set_error_handler == omx_set_error_handler.
 * Deliberate differences (documented in INTEGRATION.md):
 *   - execution is EAGER on a HIP stream: ops enqueue kernels immediately; mlx_eval /
 *     mlx_async_eval / mlx_array_eval are ordering points (stream sync / no-op / sync);
 *   - mlx_array_data_* return a pointer to a HOST mirror refreshed at that call (device memory is
 *     not CPU-addressable on a discrete GPU); it stays valid until the array is freed or re-read;
 *   - there is no CPU backend: mlx_default_cpu_stream_new reports an error;
 *   - mlx_matmul is implemented for bfloat16 and float32 operands (N-D broadcast, MLX dtype promotion; float32 on the exact-f32
 *     matrix cores), mlx_quantized_matmul for bfloat16 / float16 / float32 activations; other dtypes report an error.
 */
#ifndef OMX_MLX_C_H
#define OMX_MLX_C_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* array.h:28-30, 37-52 */
typedef struct mlx_array_ { void* ctx; } mlx_array;
typedef enum mlx_dtype_ {
    MLX_BOOL, MLX_UINT8, MLX_UINT16, MLX_UINT32, MLX_UINT64, MLX_INT8, MLX_INT16, MLX_INT32, MLX_INT64,
    MLX_FLOAT16, MLX_FLOAT32, MLX_FLOAT64, MLX_BFLOAT16, MLX_COMPLEX64
} mlx_dtype;
/* stream.h:23-25, device.h:23-30, string.h:23-25, vector.h:25-27, map.h:25-27,85-87, closure.h:26-28, optional.h:24-43 */
typedef struct mlx_stream_ { void* ctx; } mlx_stream;
typedef struct mlx_device_ { void* ctx; } mlx_device;
typedef enum mlx_device_type_ { MLX_CPU, MLX_GPU } mlx_device_type;
typedef struct mlx_string_ { void* ctx; } mlx_string;
typedef struct mlx_vector_array_ { void* ctx; } mlx_vector_array;
typedef struct mlx_vector_string_ { void* ctx; } mlx_vector_string;
typedef struct mlx_map_string_to_array_ { void* ctx; } mlx_map_string_to_array;
typedef struct mlx_map_string_to_array_iterator_ { void* ctx; void* map_ctx; } mlx_map_string_to_array_iterator;
typedef struct mlx_map_string_to_string_ { void* ctx; } mlx_map_string_to_string;
typedef struct mlx_map_string_to_string_iterator_ { void* ctx; void* map_ctx; } mlx_map_string_to_string_iterator;
typedef struct mlx_closure_ { void* ctx; } mlx_closure;
typedef struct mlx_optional_int_ { int value; bool has_value; } mlx_optional_int;
typedef struct mlx_optional_float_ { float value; bool has_value; } mlx_optional_float;
typedef struct mlx_optional_dtype_ { mlx_dtype value; bool has_value; } mlx_optional_dtype;

/* error.h:15-23 (mlx-rs installs its handler per thread: utils/guard.rs:28-29, error.rs:241-259) */
typedef void (*mlx_error_handler_func)(const char* msg, void* data);
void mlx_set_error_handler(mlx_error_handler_func handler, void* data, void (*dtor)(void*));

/* ---- array lifecycle: array.h:57-340 ---- */
size_t mlx_dtype_size(mlx_dtype dtype);                                   /* :57  */
mlx_array mlx_array_new(void);                                            /* :67  */
int mlx_array_free(mlx_array arr);                                        /* :72  */
mlx_array mlx_array_new_bool(bool val);                                   /* :77  */
mlx_array mlx_array_new_int(int val);                                     /* :81  */
mlx_array mlx_array_new_float32(float val);                               /* :85  */
mlx_array mlx_array_new_float(float val);                                 /* :90  */
mlx_array mlx_array_new_data(const void* data, const int* shape, int dim, mlx_dtype dtype);   /* :111 copies */
int mlx_array_set(mlx_array* arr, const mlx_array src);                   /* :119 */
size_t mlx_array_itemsize(const mlx_array arr);                           /* :166 */
size_t mlx_array_size(const mlx_array arr);                               /* :170 */
size_t mlx_array_nbytes(const mlx_array arr);                             /* :174 */
size_t mlx_array_ndim(const mlx_array arr);                               /* :178 */
const int* mlx_array_shape(const mlx_array arr);                          /* :183 */
const size_t* mlx_array_strides(const mlx_array arr);                     /* :188 */
int mlx_array_dim(const mlx_array arr, int dim);                          /* :192 */
mlx_dtype mlx_array_dtype(const mlx_array arr);                           /* :196 */
int mlx_array_eval(mlx_array arr);                                        /* :201 */
int mlx_array_item_bool(bool* res, const mlx_array arr);                  /* :206 */
int mlx_array_item_uint32(uint32_t* res, const mlx_array arr);            /* :218 (sampler: token id) */
int mlx_array_item_int32(int32_t* res, const mlx_array arr);              /* :234 */
int mlx_array_item_float32(float* res, const mlx_array arr);              /* :242 */
const uint8_t* mlx_array_data_uint8(const mlx_array arr);                 /* :275 */
const uint16_t* mlx_array_data_uint16(const mlx_array arr);               /* :280 */
const uint32_t* mlx_array_data_uint32(const mlx_array arr);               /* :285 */
const int32_t* mlx_array_data_int32(const mlx_array arr);                 /* :305 */
const float* mlx_array_data_float32(const mlx_array arr);                 /* :315 */
const uint16_t* mlx_array_data_float16(const mlx_array arr);              /* :332 (float16_t == 16-bit storage) */
const uint16_t* mlx_array_data_bfloat16(const mlx_array arr);             /* :335 (bfloat16_t == 16-bit storage) */

/* ---- vector.h:28-47 ---- */
mlx_vector_array mlx_vector_array_new(void);
int mlx_vector_array_free(mlx_vector_array vec);
int mlx_vector_array_append_value(mlx_vector_array vec, const mlx_array val);
size_t mlx_vector_array_size(mlx_vector_array vec);
int mlx_vector_array_get(mlx_array* res, const mlx_vector_array vec, size_t idx);

/* ---- stream.h:30-80, transforms.h:30,42, memory.h:30-34 ---- */
mlx_stream mlx_stream_new(void);
int mlx_stream_free(mlx_stream stream);
bool mlx_stream_equal(mlx_stream lhs, mlx_stream rhs);
int mlx_synchronize(mlx_stream stream);                                   /* stream.h:63 */
mlx_stream mlx_default_cpu_stream_new(void);                              /* stream.h:75: no CPU backend -> error */
mlx_stream mlx_default_gpu_stream_new(void);                              /* stream.h:80 (metal_kernels.rs:196,286) */
int mlx_async_eval(const mlx_vector_array outputs);                       /* transforms.h:30 (model.rs:817-833) */
int mlx_eval(const mlx_vector_array outputs);                             /* transforms.h:42 */
int mlx_clear_cache(void);                                                /* memory.h:30 (model.rs:836-838) */
int mlx_get_active_memory(size_t* res);                                   /* memory.h:31 */
int mlx_get_peak_memory(size_t* res);                                     /* memory.h:34 */

/* ---- fast.h:93-99, 163-198 : the fused hot-path ops ---- */
int mlx_fast_layer_norm(mlx_array* res, const mlx_array x, const mlx_array weight /* may be null */,
                        const mlx_array bias /* may be null */, float eps, const mlx_stream s);
int mlx_fast_rms_norm(mlx_array* res, const mlx_array x, const mlx_array weight /* may be null */, float eps,
                      const mlx_stream s);
int mlx_fast_rope(mlx_array* res, const mlx_array x, int dims, bool traditional, mlx_optional_float base, float scale,
                  int offset, const mlx_array freqs /* may be null */, const mlx_stream s);
int mlx_fast_scaled_dot_product_attention(mlx_array* res, const mlx_array queries, const mlx_array keys,
                                          const mlx_array values, float scale, const char* mask_mode,
                                          const mlx_array mask_arr /* may be null */,
                                          const mlx_array sinks /* may be null */, const mlx_stream s);

/* ---- ops.h: GEMM + glue used by the four callers (line numbers per declaration) ---- */
int mlx_matmul(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                     /* :598 */
int mlx_addmm(mlx_array* res, const mlx_array c, const mlx_array a, const mlx_array b, float alpha, float beta,
              const mlx_stream s);
/* affine group quantisation (mlx-c ops.h:356-365, 471-484, 793-810; mlx-rs/src/ops/quantization.rs:41-153, 226-279).
 * mlx_quantize appends (w_q, scales, biases) to an existing vector; mode "" or "affine"; defaults group 64, bits 4 */
int mlx_quantize(mlx_vector_array* res, const mlx_array w, mlx_optional_int group_size, mlx_optional_int bits, const char* mode,
                 const mlx_stream s);
int mlx_dequantize(mlx_array* res, const mlx_array w, const mlx_array scales, const mlx_array biases /* may be null */,
                   mlx_optional_int group_size, mlx_optional_int bits, const char* mode, mlx_optional_dtype dtype,
                   const mlx_stream s);
int mlx_quantized_matmul(mlx_array* res, const mlx_array x, const mlx_array w, const mlx_array scales,
                         const mlx_array biases /* may be null */, bool transpose, mlx_optional_int group_size,
                         mlx_optional_int bits, const char* mode, const mlx_stream s);
int mlx_gather_qmm(mlx_array* res, const mlx_array x, const mlx_array w, const mlx_array scales,
                   const mlx_array biases /* may be null */, const mlx_array lhs_indices /* may be null */,
                   const mlx_array rhs_indices /* may be null */, bool transpose, mlx_optional_int group_size,
                   mlx_optional_int bits, const char* mode, bool sorted_indices, const mlx_stream s);
                                                                             /* :36  */
int mlx_add(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                        /* :31  */
int mlx_subtract(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                   /* :1080 */
int mlx_multiply(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                   /* :686 */
int mlx_divide(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                     /* :374 */
int mlx_sigmoid(mlx_array* res, const mlx_array a, const mlx_stream s);                                       /* :956 */
int mlx_exp(mlx_array* res, const mlx_array a, const mlx_stream s);                                           /* :396 */
int mlx_negative(mlx_array* res, const mlx_array a, const mlx_stream s);                                      /* :698 */
int mlx_astype(mlx_array* res, const mlx_array a, mlx_dtype dtype, const mlx_stream s);                       /* :160 */
int mlx_reshape(mlx_array* res, const mlx_array a, const int* shape, size_t shape_num, const mlx_stream s);   /* :830 */
int mlx_transpose_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, const mlx_stream s); /* :1165 */
int mlx_transpose(mlx_array* res, const mlx_array a, const mlx_stream s);                                     /* :1171 */
int mlx_expand_dims(mlx_array* res, const mlx_array a, int axis, const mlx_stream s);                         /* :403 */
int mlx_contiguous(mlx_array* res, const mlx_array a, bool allow_col_major, const mlx_stream s);              /* :220 */
int mlx_slice(mlx_array* res, const mlx_array a, const int* start, size_t start_num, const int* stop,
              size_t stop_num, const int* strides, size_t strides_num, const mlx_stream s);                   /* :960 */
/* Donation contract (eager stand-in for MLX's evaluation-time buffer donation): when `src` is the sole owner of a
 * contiguous buffer the update happens in place and the buffer moves to `*res`; `src` may then only be freed or
 * overwritten -- any op READING it returns an error ("donated").  Hold a second reference (mlx_array_set) before the
 * call to make it copy instead.  mlx-rs's index_mut (cache.rs:183-188) drops the old handle, i.e. never reads it. */
int mlx_slice_update(mlx_array* res, const mlx_array src, const mlx_array update, const int* start,
                     size_t start_num, const int* stop, size_t stop_num, const int* strides, size_t strides_num,
                     const mlx_stream s);                                                                      /* :979 (cache.rs:183-188) */
int mlx_concatenate_axis(mlx_array* res, const mlx_vector_array arrays, int axis, const mlx_stream s);        /* :210 (cache.rs:66-84,152-176) */
int mlx_zeros(mlx_array* res, const int* shape, size_t shape_num, mlx_dtype dtype, const mlx_stream s);       /* :1220 */
int mlx_take_axis(mlx_array* res, const mlx_array a, const mlx_array indices, int axis, const mlx_stream s);  /* :1109 (Embedding) */
int mlx_argmax_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream s);          /* :106 (sampler.rs:9-12) */
int mlx_softmax_axis(mlx_array* res, const mlx_array a, int axis, bool precise, const mlx_stream s);          /* :1005 */

/* ---- mlx/c/random.h (line numbers of that header): the keyed generator behind the sampler's temperature
 *      branch (mlx-rs-core/src/sampler.rs:13-16 -> mlx-rs/src/random.rs:98-115, 397-414, 456-497) ---- */
int mlx_random_seed(uint64_t seed);                                                                            /* :129 */
int mlx_random_key(mlx_array* res, uint64_t seed);                                                             /* :72 */
int mlx_random_split_num(mlx_array* res, const mlx_array key, int num, const mlx_stream s);                    /* :130 */
int mlx_random_split(mlx_array* res_0, mlx_array* res_1, const mlx_array key, const mlx_stream s);             /* :135 */
int mlx_random_bits(mlx_array* res, const int* shape, size_t shape_num, int width,
                    const mlx_array key /* may be null */, const mlx_stream s);                                /* :37 */
int mlx_random_uniform(mlx_array* res, const mlx_array low, const mlx_array high, const int* shape, size_t shape_num,
                       mlx_dtype dtype, const mlx_array key /* may be null */, const mlx_stream s);            /* :149 */
int mlx_random_gumbel(mlx_array* res, const int* shape, size_t shape_num, mlx_dtype dtype,
                      const mlx_array key /* may be null */, const mlx_stream s);                              /* :65 */
int mlx_random_normal(mlx_array* res, const int* shape, size_t shape_num, mlx_dtype dtype, float loc, float scale,
                      const mlx_array key /* may be null */, const mlx_stream s);                              /* :100 (sampler.rs:139-141 prior) */
int mlx_random_categorical(mlx_array* res, const mlx_array logits, int axis,
                           const mlx_array key /* may be null */, const mlx_stream s);                         /* :59 */
int mlx_random_categorical_num_samples(mlx_array* res, const mlx_array logits, int axis, int num_samples,
                                       const mlx_array key /* may be null */, const mlx_stream s);             /* :52 */
int mlx_random_categorical_shape(mlx_array* res, const mlx_array logits, int axis, const int* shape, size_t shape_num,
                                 const mlx_array key /* may be null */, const mlx_stream s);                   /* :44 */

/* ---- device.h:35-72, stream.h:35-67: Device / Stream objects as mlx-rs builds them (stream.rs:150-195, device.rs:40-90).
 *      There is ONE device (the MI355X the process selected) and ONE in-order stream behind every handle; a CPU device can be
 *      named and compared but not made the default nor given a stream (no CPU backend). ---- */
mlx_device mlx_device_new(void);
mlx_device mlx_device_new_type(mlx_device_type type, int index);
int mlx_device_free(mlx_device dev);
int mlx_device_set(mlx_device* dev, const mlx_device src);
int mlx_device_tostring(mlx_string* str, mlx_device dev);
bool mlx_device_equal(mlx_device lhs, mlx_device rhs);
int mlx_device_get_index(int* index, mlx_device dev);
int mlx_device_get_type(mlx_device_type* type, mlx_device dev);
int mlx_get_default_device(mlx_device* dev);
int mlx_set_default_device(mlx_device dev);
mlx_stream mlx_stream_new_device(mlx_device dev);
int mlx_stream_set(mlx_stream* stream, const mlx_stream src);
int mlx_stream_tostring(mlx_string* str, mlx_stream stream);
int mlx_stream_get_device(mlx_device* dev, mlx_stream stream);
int mlx_stream_get_index(int* index, mlx_stream stream);
int mlx_get_default_stream(mlx_stream* stream, mlx_device dev);
int mlx_set_default_stream(mlx_stream stream);

/* ---- string.h:30-48, vector.h (string vectors), array.h:62 ---- */
mlx_string mlx_string_new(void);
mlx_string mlx_string_new_data(const char* str);
int mlx_string_set(mlx_string* str, const mlx_string src);
const char* mlx_string_data(mlx_string str);
int mlx_string_free(mlx_string str);
mlx_vector_string mlx_vector_string_new(void);
int mlx_vector_string_set(mlx_vector_string* vec, const mlx_vector_string src);
int mlx_vector_string_free(mlx_vector_string vec);
mlx_vector_string mlx_vector_string_new_data(const char** data, size_t size);
mlx_vector_string mlx_vector_string_new_value(const char* val);
int mlx_vector_string_set_data(mlx_vector_string* vec, const char** data, size_t size);
int mlx_vector_string_set_value(mlx_vector_string* vec, const char* val);
int mlx_vector_string_append_data(mlx_vector_string vec, const char** data, size_t size);
int mlx_vector_string_append_value(mlx_vector_string vec, const char* val);
size_t mlx_vector_string_size(mlx_vector_string vec);
int mlx_vector_string_get(char** res, const mlx_vector_string vec, size_t idx);
int mlx_array_tostring(mlx_string* str, const mlx_array arr);
int mlx_vector_array_set(mlx_vector_array* vec, const mlx_vector_array src);
mlx_vector_array mlx_vector_array_new_data(const mlx_array* data, size_t size);
mlx_vector_array mlx_vector_array_new_value(const mlx_array val);

/* ---- map.h:32-140 and io.h:40-44: what `Array::load_safetensors` walks (mlx-rs/src/ops/io.rs:51-58) ---- */
mlx_map_string_to_array mlx_map_string_to_array_new(void);
int mlx_map_string_to_array_set(mlx_map_string_to_array* map, const mlx_map_string_to_array src);
int mlx_map_string_to_array_free(mlx_map_string_to_array map);
int mlx_map_string_to_array_insert(mlx_map_string_to_array map, const char* key, const mlx_array value);
int mlx_map_string_to_array_get(mlx_array* value, const mlx_map_string_to_array map, const char* key);
mlx_map_string_to_array_iterator mlx_map_string_to_array_iterator_new(mlx_map_string_to_array map);
int mlx_map_string_to_array_iterator_free(mlx_map_string_to_array_iterator it);
int mlx_map_string_to_array_iterator_next(const char** key, mlx_array* value, mlx_map_string_to_array_iterator it);
mlx_map_string_to_string mlx_map_string_to_string_new(void);
int mlx_map_string_to_string_set(mlx_map_string_to_string* map, const mlx_map_string_to_string src);
int mlx_map_string_to_string_free(mlx_map_string_to_string map);
int mlx_map_string_to_string_insert(mlx_map_string_to_string map, const char* key, const char* value);
int mlx_map_string_to_string_get(const char** value, const mlx_map_string_to_string map, const char* key);
mlx_map_string_to_string_iterator mlx_map_string_to_string_iterator_new(mlx_map_string_to_string map);
int mlx_map_string_to_string_iterator_free(mlx_map_string_to_string_iterator it);
int mlx_map_string_to_string_iterator_next(const char** key, const char** value, mlx_map_string_to_string_iterator it);
int mlx_load_safetensors(mlx_map_string_to_array* res_0, mlx_map_string_to_string* res_1, const char* file, const mlx_stream s);

/* ---- closure.h:33-50, compile.h:37-48: `nn::silu` and friends are wrapped in `compile` (nn/activation.rs:876-880,
 *      transforms/compile/compile.rs:334).  Execution here is eager, so compiling a closure returns the closure. ---- */
mlx_closure mlx_closure_new(void);
int mlx_closure_free(mlx_closure cls);
mlx_closure mlx_closure_new_func(int (*fun)(mlx_vector_array*, const mlx_vector_array));
mlx_closure mlx_closure_new_func_payload(int (*fun)(mlx_vector_array*, const mlx_vector_array, void*), void* payload, void (*dtor)(void*));
int mlx_closure_set(mlx_closure* cls, const mlx_closure src);
int mlx_closure_apply(mlx_vector_array* res, mlx_closure cls, const mlx_vector_array input);
mlx_closure mlx_closure_new_unary(int (*fun)(mlx_array*, const mlx_array));
int mlx_detail_compile(mlx_closure* res, const mlx_closure fun, uintptr_t fun_id, bool shapeless, const uint64_t* constants,
                       size_t constants_num);
int mlx_detail_compile_clear_cache(void);
int mlx_detail_compile_erase(uintptr_t fun_id);
int mlx_disable_compile(void);
int mlx_enable_compile(void);

/* ---- ops.h: the rest of the glue the four callers use -- Mixtral routing and gather_sort / scatter_unsort
 *      (mixtral-mlx/src/model.rs:204-228, 296-308), create_causal_mask (mlx-rs-core/src/utils.rs:134-153), Paraformer FSMN
 *      (funasr-mlx/src/paraformer.rs:496-532), Klein RoPE tables (flux-klein-mlx/src/klein_model.rs:53-162) ---- */
int mlx_arange(mlx_array* res, double start, double stop, double step, mlx_dtype dtype, const mlx_stream s);         /* :88  */
int mlx_greater(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                           /* :485 */
int mlx_greater_equal(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                     /* :490 */
int mlx_less(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                              /* :530 */
int mlx_less_equal(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                        /* :535 */
int mlx_equal(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                             /* :389 */
int mlx_logical_and(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                       /* :563 */
int mlx_maximum(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                           /* :621 */
int mlx_minimum(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                           /* :675 */
int mlx_floor_divide(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);                      /* :423 */
int mlx_cos(mlx_array* res, const mlx_array a, const mlx_stream s);                                                  /* :321 */
int mlx_sin(mlx_array* res, const mlx_array a, const mlx_stream s);                                                  /* :958 */
int mlx_sum_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream s);                    /* :1092 */
int mlx_argsort_axis(mlx_array* res, const mlx_array a, int axis, const mlx_stream s);                               /* :139 */
int mlx_argsort(mlx_array* res, const mlx_array a, const mlx_stream s);                                              /* :144 */
int mlx_argpartition_axis(mlx_array* res, const mlx_array a, int kth, int axis, const mlx_stream s);                 /* :128 */
int mlx_take(mlx_array* res, const mlx_array a, const mlx_array indices, const mlx_stream s);                        /* :1115 */
int mlx_take_along_axis(mlx_array* res, const mlx_array a, const mlx_array indices, int axis, const mlx_stream s);   /* :1120 */
int mlx_expand_dims_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, const mlx_stream s);   /* :397 */
int mlx_squeeze_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, const mlx_stream s);       /* :1037 */
int mlx_squeeze_axis(mlx_array* res, const mlx_array a, int axis, const mlx_stream s);                               /* :1043 */
int mlx_squeeze(mlx_array* res, const mlx_array a, const mlx_stream s);                                              /* :1048 */
int mlx_flatten(mlx_array* res, const mlx_array a, int start_axis, int end_axis, const mlx_stream s);                /* :416 */
int mlx_stack_axis(mlx_array* res, const mlx_vector_array arrays, int axis, const mlx_stream s);                     /* :1049 */
int mlx_stack(mlx_array* res, const mlx_vector_array arrays, const mlx_stream s);                                    /* :1054 */
int mlx_split(mlx_vector_array* res, const mlx_array a, int num_splits, int axis, const mlx_stream s);               /* :1022 */
int mlx_split_sections(mlx_vector_array* res, const mlx_array a, const int* indices, size_t indices_num, int axis,
                       const mlx_stream s);                                                                          /* :1028 */
int mlx_conv1d(mlx_array* res, const mlx_array input, const mlx_array weight, int stride, int padding, int dilation, int groups,
               const mlx_stream s);                                                                                  /* :225 */
/* nn::Conv2d (channels-last: input [B, H, W, C_in], weight [C_out, kH, kW, C_in / groups]); the FLUX autoencoder's convolution
 * (flux-klein-mlx/src/autoencoder.rs:110-131).  bfloat16 1x1 and 3x3 / stride 1 / padding 1 shapes run on the matrix-core GEMMs */
int mlx_conv2d(mlx_array* res, const mlx_array input, const mlx_array weight, int stride_0, int stride_1, int padding_0, int padding_1,
               int dilation_0, int dilation_1, int groups, const mlx_stream s);                                      /* :234 */
/* dense expert matmul (mlx-rs/src/ops/quantization.rs:169-203): a [..., 1, K] against the stacked b [E, K, N] picked per row by
 * rhs_indices; the SwitchLinear form (b = swap_axes(w [E, N, K]), no lhs_indices) runs on the expert-selected GEMV */
int mlx_gather_mm(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_array lhs_indices /* may be null */,
                  const mlx_array rhs_indices /* may be null */, bool sorted_indices, const mlx_stream s);           /* :463 */

/* ---- native replacements for the two JIT Metal kernels (mlx_fast_metal_kernel_apply, fast.h:156;
 *      mlx-rs-core/src/metal_kernels.rs:188-236, 260-339): the two Rust call sites switch to these ---- */
/* omx extension: an array over device memory the caller owns and keeps alive (row-major, 16-byte aligned; never written by an op) */
mlx_array omx_mlx_array_from_device(const void* device_ptr, const int* shape, int dim, mlx_dtype dtype);
/* omx extension (round 6): counters of the deferred op list behind this ABI (csrc/mlxc_lazy.hpp) -- out6[0] ops recorded, [1] launched as
 * recorded, [2] fused GEMV launches that replaced several of them, [3] flushes, [4] host ns inside the flushes, [5] of which in the rewrite pass.  OMX_MLX_LAZY=0 executes every call eagerly (round 5). */
void omx_mlx_lazy_stats(long* out6);
/* ... and its switches at run time: lazy 0 = every call launches as it is made, fuse 0 = recorded ops launch as recorded (default 1, 1;
 * OMX_MLX_LAZY / OMX_MLX_FUSE in the environment set the initial state) */
int omx_mlx_lazy_mode(int lazy, int fuse);
/* the launch worker behind mlx_async_eval: 1 = a worker thread rewrites and launches the recorded list while the caller records on
 * -- measured slower on this hardware, so 0 = the calling thread launches at the evaluation point is the default (OMX_MLX_ASYNC=1 starts with the worker) */
int omx_mlx_lazy_async(int on);
int omx_mlx_fused_swiglu(mlx_array* res, const mlx_array x, const mlx_array gate, const mlx_stream s);
int omx_mlx_fused_modulate(mlx_array* res, const mlx_array x, const mlx_array shift, const mlx_array scale,
                           const mlx_stream s);

/* ---- elementwise math, predicates, axis reductions, views and fills of ops.h beyond the four callers' path (round 4: these were
 *      error-returning link stubs before); same names and argument order as mlx/c/ops.h ---- */
int mlx_abs(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_sqrt(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_rsqrt(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_square(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_log(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_log2(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_log10(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_log1p(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_expm1(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_tanh(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_sinh(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_cosh(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_tan(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_arcsin(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_arccos(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_arctan(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_arcsinh(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_arccosh(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_arctanh(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_erf(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_reciprocal(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_floor(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_ceil(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_sign(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_isnan(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_isinf(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_isfinite(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_isposinf(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_isneginf(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_logical_not(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_round(mlx_array* res, const mlx_array a, int decimals, const mlx_stream s);   /* decimals = 0 */
int mlx_not_equal(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);
int mlx_logical_or(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);
int mlx_power(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);
int mlx_remainder(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);
int mlx_logaddexp(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);
int mlx_max_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream s);
int mlx_min_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream s);
int mlx_mean_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream s);
int mlx_all_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream s);
int mlx_any_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream s);
int mlx_logsumexp_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream s);
int mlx_max(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s);
int mlx_min(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s);
int mlx_mean(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s);
int mlx_sum(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s);
int mlx_all(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s);
int mlx_any(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s);
int mlx_logsumexp(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s);
int mlx_stop_gradient(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_sort_axis(mlx_array* res, const mlx_array a, int axis, const mlx_stream s);
int mlx_sort(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_broadcast_to(mlx_array* res, const mlx_array a, const int* shape, size_t shape_num, const mlx_stream s);
int mlx_concatenate(mlx_array* res, const mlx_vector_array arrays, const mlx_stream s);
int mlx_swapaxes(mlx_array* res, const mlx_array a, int axis1, int axis2, const mlx_stream s);
int mlx_moveaxis(mlx_array* res, const mlx_array a, int source, int destination, const mlx_stream s);
int mlx_full(mlx_array* res, const int* shape, size_t shape_num, const mlx_array vals, mlx_dtype dtype, const mlx_stream s);   /* scalar vals */
int mlx_ones(mlx_array* res, const int* shape, size_t shape_num, mlx_dtype dtype, const mlx_stream s);
int mlx_where(mlx_array* res, const mlx_array condition, const mlx_array x, const mlx_array y, const mlx_stream s);
int mlx_clip(mlx_array* res, const mlx_array a, const mlx_array a_min /* may be null */, const mlx_array a_max /* may be null */, const mlx_stream s);

/* ops.h, third batch (round 4, csrc/mlxc_glue2.hpp): composed from the ops above; signatures of mlx-rs/mlx-sys/src/mlx-c/mlx/c/ops.h */
int mlx_sum_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, const mlx_stream s);
int mlx_mean_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, const mlx_stream s);
int mlx_max_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, const mlx_stream s);
int mlx_min_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, const mlx_stream s);
int mlx_all_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, const mlx_stream s);
int mlx_any_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, const mlx_stream s);
int mlx_logsumexp_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, const mlx_stream s);
int mlx_prod_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, const mlx_stream s);
int mlx_prod_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream s);
int mlx_prod(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s);
int mlx_var_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, int ddof, const mlx_stream s);
int mlx_var_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, int ddof, const mlx_stream s);
int mlx_var(mlx_array* res, const mlx_array a, bool keepdims, int ddof, const mlx_stream s);
int mlx_std_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool keepdims, int ddof, const mlx_stream s);
int mlx_std_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, int ddof, const mlx_stream s);
int mlx_std(mlx_array* res, const mlx_array a, bool keepdims, int ddof, const mlx_stream s);
int mlx_softmax_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, bool precise, const mlx_stream s);
int mlx_softmax(mlx_array* res, const mlx_array a, bool precise, const mlx_stream s);   /* over all axes */
int mlx_argmax(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s);
int mlx_argmin(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s);
int mlx_argmin_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream s);
int mlx_cumsum(mlx_array* res, const mlx_array a, int axis, bool reverse, bool inclusive, const mlx_stream s);
int mlx_cumprod(mlx_array* res, const mlx_array a, int axis, bool reverse, bool inclusive, const mlx_stream s);
int mlx_cummax(mlx_array* res, const mlx_array a, int axis, bool reverse, bool inclusive, const mlx_stream s);
int mlx_cummin(mlx_array* res, const mlx_array a, int axis, bool reverse, bool inclusive, const mlx_stream s);
int mlx_partition_axis(mlx_array* res, const mlx_array a, int kth, int axis, const mlx_stream s);   /* a full sort: one valid partition */
int mlx_partition(mlx_array* res, const mlx_array a, int kth, const mlx_stream s);
int mlx_argpartition(mlx_array* res, const mlx_array a, int kth, const mlx_stream s);
int mlx_topk_axis(mlx_array* res, const mlx_array a, int k, int axis, const mlx_stream s);          /* the k largest, ascending */
int mlx_topk(mlx_array* res, const mlx_array a, int k, const mlx_stream s);
int mlx_tri(mlx_array* res, int n, int m, int k, mlx_dtype type, const mlx_stream s);
int mlx_tril(mlx_array* res, const mlx_array x, int k, const mlx_stream s);
int mlx_triu(mlx_array* res, const mlx_array x, int k, const mlx_stream s);
int mlx_eye(mlx_array* res, int n, int m, int k, mlx_dtype dtype, const mlx_stream s);
int mlx_identity(mlx_array* res, int n, mlx_dtype dtype, const mlx_stream s);
int mlx_linspace(mlx_array* res, double start, double stop, int num, mlx_dtype dtype, const mlx_stream s);
int mlx_outer(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);
int mlx_inner(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);
int mlx_atleast_1d(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_atleast_2d(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_atleast_3d(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_isclose(mlx_array* res, const mlx_array a, const mlx_array b, double rtol, double atol, bool equal_nan, const mlx_stream s);
int mlx_allclose(mlx_array* res, const mlx_array a, const mlx_array b, double rtol, double atol, bool equal_nan, const mlx_stream s);
int mlx_array_equal(mlx_array* res, const mlx_array a, const mlx_array b, bool equal_nan, const mlx_stream s);
int mlx_degrees(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_radians(mlx_array* res, const mlx_array a, const mlx_stream s);
int mlx_divmod(mlx_vector_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);
int mlx_unflatten(mlx_array* res, const mlx_array a, int axis, const int* shape, size_t shape_num, const mlx_stream s);
int mlx_pad(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, const int* low_pad_size, size_t low_pad_size_num,
            const int* high_pad_size, size_t high_pad_size_num, const mlx_array pad_value, const char* mode /* "constant" */, const mlx_stream s);
int mlx_repeat_axis(mlx_array* res, const mlx_array arr, int repeats, int axis, const mlx_stream s);
int mlx_repeat(mlx_array* res, const mlx_array arr, int repeats, const mlx_stream s);
int mlx_tile(mlx_array* res, const mlx_array arr, const int* reps, size_t reps_num, const mlx_stream s);
int mlx_diagonal(mlx_array* res, const mlx_array a, int offset, int axis1, int axis2, const mlx_stream s);   /* a view */
int mlx_diag(mlx_array* res, const mlx_array a, int k, const mlx_stream s);
int mlx_nan_to_num(mlx_array* res, const mlx_array a, float nan, mlx_optional_float posinf, mlx_optional_float neginf, const mlx_stream s);
int mlx_broadcast_arrays(mlx_vector_array* res, const mlx_vector_array inputs, const mlx_stream s);
const bool* mlx_array_data_bool(const mlx_array arr);
const int8_t* mlx_array_data_int8(const mlx_array arr);
const int16_t* mlx_array_data_int16(const mlx_array arr);
const int64_t* mlx_array_data_int64(const mlx_array arr);
const uint64_t* mlx_array_data_uint64(const mlx_array arr);
int mlx_as_strided(mlx_array* res, const mlx_array a, const int* shape, size_t shape_num, const int64_t* strides, size_t strides_num, size_t offset,
                   const mlx_stream s);   /* a view; strides / offset in elements */
int mlx_view(mlx_array* res, const mlx_array a, mlx_dtype dtype, const mlx_stream s);
int mlx_real(mlx_array* res, const mlx_array a, const mlx_stream s);   /* no complex dtype: the array itself */
int mlx_imag(mlx_array* res, const mlx_array a, const mlx_stream s);   /* ... and zeros */
int mlx_tensordot(mlx_array* res, const mlx_array a, const mlx_array b, const int* axes_a, size_t axes_a_num, const int* axes_b, size_t axes_b_num,
                  const mlx_stream s);
int mlx_tensordot_axis(mlx_array* res, const mlx_array a, const mlx_array b, int axis, const mlx_stream s);
int mlx_kron(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream s);
int mlx_random_bernoulli(mlx_array* res, const mlx_array p, const int* shape, size_t shape_num, const mlx_array key /* may be null */, const mlx_stream s);

#ifdef __cplusplus
}
#endif
#endif /* OMX_MLX_C_H */
