"""ctypes binding of the mlx-c compatible surface of libomx_hip.so (include/omx_mlx_c.h) plus a
thin `Array` wrapper that plays the role of mlx_rs::Array (mlx-rs/src/array/mod.rs): an owned
handle freed on drop, ops through `Guarded::try_from_op`-style status checks
(mlx-rs/src/utils/guard.rs:24-48)."""
from __future__ import annotations

import ctypes
import sys
from typing import Optional, Sequence

import numpy as np

from . import OmxError, lib, require_device

c_int, c_float, c_bool, c_size_t, c_void_p, c_char_p = (ctypes.c_int, ctypes.c_float, ctypes.c_bool, ctypes.c_size_t,
                                                         ctypes.c_void_p, ctypes.c_char_p)


class mlx_array(ctypes.Structure):
    _fields_ = [("ctx", c_void_p)]


class mlx_stream(ctypes.Structure):
    _fields_ = [("ctx", c_void_p)]


class mlx_vector_array(ctypes.Structure):
    _fields_ = [("ctx", c_void_p)]


class mlx_device(ctypes.Structure):
    _fields_ = [("ctx", c_void_p)]


class mlx_string(ctypes.Structure):
    _fields_ = [("ctx", c_void_p)]


class mlx_vector_string(ctypes.Structure):
    _fields_ = [("ctx", c_void_p)]


class mlx_map_string_to_array(ctypes.Structure):
    _fields_ = [("ctx", c_void_p)]


class mlx_map_string_to_string(ctypes.Structure):
    _fields_ = [("ctx", c_void_p)]


class mlx_map_iterator(ctypes.Structure):           # both iterator structs: { void* ctx; void* map_ctx; } (map.h:61-64, 121-124)
    _fields_ = [("ctx", c_void_p), ("map_ctx", c_void_p)]


class mlx_closure(ctypes.Structure):
    _fields_ = [("ctx", c_void_p)]


class mlx_optional_float(ctypes.Structure):
    _fields_ = [("value", c_float), ("has_value", c_bool)]


class mlx_optional_int(ctypes.Structure):
    _fields_ = [("value", c_int), ("has_value", c_bool)]


class mlx_optional_dtype(ctypes.Structure):
    _fields_ = [("value", c_int), ("has_value", c_bool)]


BOOL, UINT8, UINT16, UINT32, UINT64, INT8, INT16, INT32, INT64, FLOAT16, FLOAT32, FLOAT64, BFLOAT16, COMPLEX64 = range(14)
P_ARR, P_INT = ctypes.POINTER(mlx_array), ctypes.POINTER(c_int)

# name -> (restype, argtypes): must list every symbol include/omx_mlx_c.h declares (tests/test_abi.py)
SIGNATURES = {
    "mlx_set_error_handler": (None, [c_void_p, c_void_p, c_void_p]),
    "mlx_dtype_size": (c_size_t, [c_int]),
    "mlx_array_new": (mlx_array, []),
    "mlx_array_free": (c_int, [mlx_array]),
    "mlx_array_new_bool": (mlx_array, [c_bool]),
    "mlx_array_new_int": (mlx_array, [c_int]),
    "mlx_array_new_float32": (mlx_array, [c_float]),
    "mlx_array_new_float": (mlx_array, [c_float]),
    "mlx_array_new_data": (mlx_array, [c_void_p, P_INT, c_int, c_int]),
    "mlx_array_set": (c_int, [P_ARR, mlx_array]),
    "mlx_array_itemsize": (c_size_t, [mlx_array]),
    "mlx_array_size": (c_size_t, [mlx_array]),
    "mlx_array_nbytes": (c_size_t, [mlx_array]),
    "mlx_array_ndim": (c_size_t, [mlx_array]),
    "mlx_array_shape": (P_INT, [mlx_array]),
    "mlx_array_strides": (ctypes.POINTER(c_size_t), [mlx_array]),
    "mlx_array_dim": (c_int, [mlx_array, c_int]),
    "mlx_array_dtype": (c_int, [mlx_array]),
    "mlx_array_eval": (c_int, [mlx_array]),
    "mlx_array_item_bool": (c_int, [ctypes.POINTER(c_bool), mlx_array]),
    "mlx_array_item_uint32": (c_int, [ctypes.POINTER(ctypes.c_uint32), mlx_array]),
    "mlx_array_item_int32": (c_int, [ctypes.POINTER(ctypes.c_int32), mlx_array]),
    "mlx_array_item_float32": (c_int, [ctypes.POINTER(c_float), mlx_array]),
    "mlx_array_data_uint8": (c_void_p, [mlx_array]),
    "mlx_array_data_uint16": (c_void_p, [mlx_array]),
    "mlx_array_data_uint32": (c_void_p, [mlx_array]),
    "mlx_array_data_int32": (c_void_p, [mlx_array]),
    "mlx_array_data_float32": (c_void_p, [mlx_array]),
    "mlx_array_data_bfloat16": (c_void_p, [mlx_array]),
    "mlx_array_data_float16": (c_void_p, [mlx_array]),
    "mlx_vector_array_new": (mlx_vector_array, []),
    "mlx_vector_array_free": (c_int, [mlx_vector_array]),
    "mlx_vector_array_append_value": (c_int, [mlx_vector_array, mlx_array]),
    "mlx_vector_array_size": (c_size_t, [mlx_vector_array]),
    "mlx_vector_array_get": (c_int, [P_ARR, mlx_vector_array, c_size_t]),
    "mlx_stream_new": (mlx_stream, []),
    "mlx_stream_free": (c_int, [mlx_stream]),
    "mlx_stream_equal": (c_bool, [mlx_stream, mlx_stream]),
    "mlx_synchronize": (c_int, [mlx_stream]),
    "mlx_default_cpu_stream_new": (mlx_stream, []),
    "mlx_default_gpu_stream_new": (mlx_stream, []),
    "mlx_async_eval": (c_int, [mlx_vector_array]),
    "mlx_eval": (c_int, [mlx_vector_array]),
    "mlx_clear_cache": (c_int, []),
    "mlx_get_active_memory": (c_int, [ctypes.POINTER(c_size_t)]),
    "mlx_get_peak_memory": (c_int, [ctypes.POINTER(c_size_t)]),
    "mlx_fast_layer_norm": (c_int, [P_ARR, mlx_array, mlx_array, mlx_array, c_float, mlx_stream]),
    "mlx_fast_rms_norm": (c_int, [P_ARR, mlx_array, mlx_array, c_float, mlx_stream]),
    "mlx_fast_rope": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_optional_float, c_float, c_int, mlx_array, mlx_stream]),
    "mlx_fast_scaled_dot_product_attention": (c_int, [P_ARR, mlx_array, mlx_array, mlx_array, c_float, c_char_p,
                                                      mlx_array, mlx_array, mlx_stream]),
    "mlx_matmul": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_addmm": (c_int, [P_ARR, mlx_array, mlx_array, mlx_array, c_float, c_float, mlx_stream]),
    "mlx_quantize": (c_int, [ctypes.POINTER(mlx_vector_array), mlx_array, mlx_optional_int, mlx_optional_int, ctypes.c_char_p, mlx_stream]),
    "mlx_dequantize": (c_int, [P_ARR, mlx_array, mlx_array, mlx_array, mlx_optional_int, mlx_optional_int, ctypes.c_char_p,
                               mlx_optional_dtype, mlx_stream]),
    "mlx_quantized_matmul": (c_int, [P_ARR, mlx_array, mlx_array, mlx_array, mlx_array, c_bool, mlx_optional_int, mlx_optional_int,
                                     ctypes.c_char_p, mlx_stream]),
    "mlx_gather_qmm": (c_int, [P_ARR, mlx_array, mlx_array, mlx_array, mlx_array, mlx_array, mlx_array, c_bool, mlx_optional_int,
                               mlx_optional_int, ctypes.c_char_p, c_bool, mlx_stream]),
    "mlx_add": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_subtract": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_multiply": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_divide": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_sigmoid": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_exp": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_negative": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_astype": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_reshape": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, mlx_stream]),
    "mlx_transpose_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, mlx_stream]),
    "mlx_transpose": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_expand_dims": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_contiguous": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_slice": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, P_INT, c_size_t, P_INT, c_size_t, mlx_stream]),
    "mlx_slice_update": (c_int, [P_ARR, mlx_array, mlx_array, P_INT, c_size_t, P_INT, c_size_t, P_INT, c_size_t, mlx_stream]),
    "mlx_concatenate_axis": (c_int, [P_ARR, mlx_vector_array, c_int, mlx_stream]),
    "mlx_concatenate": (c_int, [P_ARR, mlx_vector_array, mlx_stream]),
    "mlx_zeros": (c_int, [P_ARR, P_INT, c_size_t, c_int, mlx_stream]),
    "mlx_take_axis": (c_int, [P_ARR, mlx_array, mlx_array, c_int, mlx_stream]),
    "mlx_argmax_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_stream]),
    "mlx_softmax_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_stream]),
    "mlx_random_seed": (c_int, [ctypes.c_uint64]),
    "mlx_random_key": (c_int, [P_ARR, ctypes.c_uint64]),
    "mlx_random_split_num": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_random_split": (c_int, [P_ARR, P_ARR, mlx_array, mlx_stream]),
    "mlx_random_bits": (c_int, [P_ARR, P_INT, c_size_t, c_int, mlx_array, mlx_stream]),
    "mlx_random_uniform": (c_int, [P_ARR, mlx_array, mlx_array, P_INT, c_size_t, c_int, mlx_array, mlx_stream]),
    "mlx_random_gumbel": (c_int, [P_ARR, P_INT, c_size_t, c_int, mlx_array, mlx_stream]),
    "mlx_random_normal": (c_int, [P_ARR, P_INT, c_size_t, c_int, c_float, c_float, mlx_array, mlx_stream]),
    "mlx_random_categorical": (c_int, [P_ARR, mlx_array, c_int, mlx_array, mlx_stream]),
    "mlx_random_categorical_num_samples": (c_int, [P_ARR, mlx_array, c_int, c_int, mlx_array, mlx_stream]),
    "mlx_random_categorical_shape": (c_int, [P_ARR, mlx_array, c_int, P_INT, c_size_t, mlx_array, mlx_stream]),
    # device.h / stream.h
    "mlx_device_new": (mlx_device, []),
    "mlx_device_new_type": (mlx_device, [c_int, c_int]),
    "mlx_device_free": (c_int, [mlx_device]),
    "mlx_device_set": (c_int, [ctypes.POINTER(mlx_device), mlx_device]),
    "mlx_device_tostring": (c_int, [ctypes.POINTER(mlx_string), mlx_device]),
    "mlx_device_equal": (c_bool, [mlx_device, mlx_device]),
    "mlx_device_get_index": (c_int, [P_INT, mlx_device]),
    "mlx_device_get_type": (c_int, [P_INT, mlx_device]),
    "mlx_get_default_device": (c_int, [ctypes.POINTER(mlx_device)]),
    "mlx_set_default_device": (c_int, [mlx_device]),
    "mlx_stream_new_device": (mlx_stream, [mlx_device]),
    "mlx_stream_set": (c_int, [ctypes.POINTER(mlx_stream), mlx_stream]),
    "mlx_stream_tostring": (c_int, [ctypes.POINTER(mlx_string), mlx_stream]),
    "mlx_stream_get_device": (c_int, [ctypes.POINTER(mlx_device), mlx_stream]),
    "mlx_stream_get_index": (c_int, [P_INT, mlx_stream]),
    "mlx_get_default_stream": (c_int, [ctypes.POINTER(mlx_stream), mlx_device]),
    "mlx_set_default_stream": (c_int, [mlx_stream]),
    # string.h / vector.h / array.h:62
    "mlx_string_new": (mlx_string, []),
    "mlx_string_new_data": (mlx_string, [c_char_p]),
    "mlx_string_set": (c_int, [ctypes.POINTER(mlx_string), mlx_string]),
    "mlx_string_data": (c_char_p, [mlx_string]),
    "mlx_string_free": (c_int, [mlx_string]),
    "mlx_vector_string_new": (mlx_vector_string, []),
    "mlx_vector_string_set": (c_int, [ctypes.POINTER(mlx_vector_string), mlx_vector_string]),
    "mlx_vector_string_free": (c_int, [mlx_vector_string]),
    "mlx_vector_string_new_data": (mlx_vector_string, [ctypes.POINTER(c_char_p), c_size_t]),
    "mlx_vector_string_new_value": (mlx_vector_string, [c_char_p]),
    "mlx_vector_string_set_data": (c_int, [ctypes.POINTER(mlx_vector_string), ctypes.POINTER(c_char_p), c_size_t]),
    "mlx_vector_string_set_value": (c_int, [ctypes.POINTER(mlx_vector_string), c_char_p]),
    "mlx_vector_string_append_data": (c_int, [mlx_vector_string, ctypes.POINTER(c_char_p), c_size_t]),
    "mlx_vector_string_append_value": (c_int, [mlx_vector_string, c_char_p]),
    "mlx_vector_string_size": (c_size_t, [mlx_vector_string]),
    "mlx_vector_string_get": (c_int, [ctypes.POINTER(c_char_p), mlx_vector_string, c_size_t]),
    "mlx_array_tostring": (c_int, [ctypes.POINTER(mlx_string), mlx_array]),
    "mlx_vector_array_set": (c_int, [ctypes.POINTER(mlx_vector_array), mlx_vector_array]),
    "mlx_vector_array_new_data": (mlx_vector_array, [P_ARR, c_size_t]),
    "mlx_vector_array_new_value": (mlx_vector_array, [mlx_array]),
    # map.h / io.h
    "mlx_map_string_to_array_new": (mlx_map_string_to_array, []),
    "mlx_map_string_to_array_set": (c_int, [ctypes.POINTER(mlx_map_string_to_array), mlx_map_string_to_array]),
    "mlx_map_string_to_array_free": (c_int, [mlx_map_string_to_array]),
    "mlx_map_string_to_array_insert": (c_int, [mlx_map_string_to_array, c_char_p, mlx_array]),
    "mlx_map_string_to_array_get": (c_int, [P_ARR, mlx_map_string_to_array, c_char_p]),
    "mlx_map_string_to_array_iterator_new": (mlx_map_iterator, [mlx_map_string_to_array]),
    "mlx_map_string_to_array_iterator_free": (c_int, [mlx_map_iterator]),
    "mlx_map_string_to_array_iterator_next": (c_int, [ctypes.POINTER(c_char_p), P_ARR, mlx_map_iterator]),
    "mlx_map_string_to_string_new": (mlx_map_string_to_string, []),
    "mlx_map_string_to_string_set": (c_int, [ctypes.POINTER(mlx_map_string_to_string), mlx_map_string_to_string]),
    "mlx_map_string_to_string_free": (c_int, [mlx_map_string_to_string]),
    "mlx_map_string_to_string_insert": (c_int, [mlx_map_string_to_string, c_char_p, c_char_p]),
    "mlx_map_string_to_string_get": (c_int, [ctypes.POINTER(c_char_p), mlx_map_string_to_string, c_char_p]),
    "mlx_map_string_to_string_iterator_new": (mlx_map_iterator, [mlx_map_string_to_string]),
    "mlx_map_string_to_string_iterator_free": (c_int, [mlx_map_iterator]),
    "mlx_map_string_to_string_iterator_next": (c_int, [ctypes.POINTER(c_char_p), ctypes.POINTER(c_char_p), mlx_map_iterator]),
    "mlx_load_safetensors": (c_int, [ctypes.POINTER(mlx_map_string_to_array), ctypes.POINTER(mlx_map_string_to_string), c_char_p, mlx_stream]),
    # closure.h / compile.h
    "mlx_closure_new": (mlx_closure, []),
    "mlx_closure_free": (c_int, [mlx_closure]),
    "mlx_closure_new_func": (mlx_closure, [c_void_p]),
    "mlx_closure_new_func_payload": (mlx_closure, [c_void_p, c_void_p, c_void_p]),
    "mlx_closure_set": (c_int, [ctypes.POINTER(mlx_closure), mlx_closure]),
    "mlx_closure_apply": (c_int, [ctypes.POINTER(mlx_vector_array), mlx_closure, mlx_vector_array]),
    "mlx_closure_new_unary": (mlx_closure, [c_void_p]),
    "mlx_detail_compile": (c_int, [ctypes.POINTER(mlx_closure), mlx_closure, c_size_t, c_bool, ctypes.POINTER(ctypes.c_uint64), c_size_t]),
    "mlx_detail_compile_clear_cache": (c_int, []),
    "mlx_detail_compile_erase": (c_int, [c_size_t]),
    "mlx_disable_compile": (c_int, []),
    "mlx_enable_compile": (c_int, []),
    # ops.h glue
    "mlx_arange": (c_int, [P_ARR, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_int, mlx_stream]),
    "mlx_greater": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_greater_equal": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_less": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_less_equal": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_equal": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_logical_and": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_maximum": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_minimum": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_floor_divide": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_cos": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_abs": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_sqrt": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_rsqrt": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_square": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_log": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_log2": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_log10": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_log1p": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_expm1": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_tanh": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_sinh": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_cosh": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_tan": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_arcsin": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_arccos": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_arctan": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_arcsinh": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_arccosh": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_arctanh": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_erf": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_reciprocal": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_floor": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_ceil": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_sign": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_isnan": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_isinf": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_isfinite": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_isposinf": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_isneginf": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_logical_not": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_round": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_not_equal": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_logical_or": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_power": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_remainder": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_logaddexp": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_max_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_stream]),
    "mlx_all_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_stream]),
    "mlx_any_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_stream]),
    "mlx_logsumexp_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_stream]),
    "mlx_max": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_min": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_mean": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_sum": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_all": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_any": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_logsumexp": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_stop_gradient": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_sort_axis": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_sort": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_broadcast_to": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, mlx_stream]),
    "mlx_min_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_stream]),
    "mlx_mean_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_stream]),
    "mlx_swapaxes": (c_int, [P_ARR, mlx_array, c_int, c_int, mlx_stream]),
    "mlx_moveaxis": (c_int, [P_ARR, mlx_array, c_int, c_int, mlx_stream]),
    "mlx_full": (c_int, [P_ARR, P_INT, c_size_t, mlx_array, c_int, mlx_stream]),
    "mlx_ones": (c_int, [P_ARR, P_INT, c_size_t, c_int, mlx_stream]),
    "mlx_where": (c_int, [P_ARR, mlx_array, mlx_array, mlx_array, mlx_stream]),
    "mlx_clip": (c_int, [P_ARR, mlx_array, mlx_array, mlx_array, mlx_stream]),
    # ops.h, third batch (csrc/mlxc_glue2.hpp)
    "mlx_array_data_bool": (c_void_p, [mlx_array]),
    "mlx_array_data_int8": (c_void_p, [mlx_array]),
    "mlx_array_data_int16": (c_void_p, [mlx_array]),
    "mlx_array_data_int64": (c_void_p, [mlx_array]),
    "mlx_array_data_uint64": (c_void_p, [mlx_array]),
    "mlx_as_strided": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, ctypes.POINTER(ctypes.c_int64), c_size_t, c_size_t, mlx_stream]),
    "mlx_view": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_real": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_imag": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_tensordot": (c_int, [P_ARR, mlx_array, mlx_array, P_INT, c_size_t, P_INT, c_size_t, mlx_stream]),
    "mlx_tensordot_axis": (c_int, [P_ARR, mlx_array, mlx_array, c_int, mlx_stream]),
    "mlx_kron": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_random_bernoulli": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, mlx_array, mlx_stream]),
    "mlx_sum_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, c_bool, mlx_stream]),
    "mlx_mean_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, c_bool, mlx_stream]),
    "mlx_max_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, c_bool, mlx_stream]),
    "mlx_min_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, c_bool, mlx_stream]),
    "mlx_all_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, c_bool, mlx_stream]),
    "mlx_any_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, c_bool, mlx_stream]),
    "mlx_logsumexp_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, c_bool, mlx_stream]),
    "mlx_prod_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, c_bool, mlx_stream]),
    "mlx_prod_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_stream]),
    "mlx_prod": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_var_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, c_bool, c_int, mlx_stream]),
    "mlx_var_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, c_int, mlx_stream]),
    "mlx_var": (c_int, [P_ARR, mlx_array, c_bool, c_int, mlx_stream]),
    "mlx_std_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, c_bool, c_int, mlx_stream]),
    "mlx_std_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, c_int, mlx_stream]),
    "mlx_std": (c_int, [P_ARR, mlx_array, c_bool, c_int, mlx_stream]),
    "mlx_softmax_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, c_bool, mlx_stream]),
    "mlx_softmax": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_argmax": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_argmin": (c_int, [P_ARR, mlx_array, c_bool, mlx_stream]),
    "mlx_argmin_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_stream]),
    "mlx_cumsum": (c_int, [P_ARR, mlx_array, c_int, c_bool, c_bool, mlx_stream]),
    "mlx_cumprod": (c_int, [P_ARR, mlx_array, c_int, c_bool, c_bool, mlx_stream]),
    "mlx_cummax": (c_int, [P_ARR, mlx_array, c_int, c_bool, c_bool, mlx_stream]),
    "mlx_cummin": (c_int, [P_ARR, mlx_array, c_int, c_bool, c_bool, mlx_stream]),
    "mlx_partition_axis": (c_int, [P_ARR, mlx_array, c_int, c_int, mlx_stream]),
    "mlx_partition": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_argpartition": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_topk_axis": (c_int, [P_ARR, mlx_array, c_int, c_int, mlx_stream]),
    "mlx_topk": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_tri": (c_int, [P_ARR, c_int, c_int, c_int, c_int, mlx_stream]),
    "mlx_tril": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_triu": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_eye": (c_int, [P_ARR, c_int, c_int, c_int, c_int, mlx_stream]),
    "mlx_identity": (c_int, [P_ARR, c_int, c_int, mlx_stream]),
    "mlx_linspace": (c_int, [P_ARR, ctypes.c_double, ctypes.c_double, c_int, c_int, mlx_stream]),
    "mlx_outer": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_inner": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_atleast_1d": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_atleast_2d": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_atleast_3d": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_isclose": (c_int, [P_ARR, mlx_array, mlx_array, ctypes.c_double, ctypes.c_double, c_bool, mlx_stream]),
    "mlx_allclose": (c_int, [P_ARR, mlx_array, mlx_array, ctypes.c_double, ctypes.c_double, c_bool, mlx_stream]),
    "mlx_array_equal": (c_int, [P_ARR, mlx_array, mlx_array, c_bool, mlx_stream]),
    "mlx_degrees": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_radians": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_divmod": (c_int, [ctypes.POINTER(mlx_vector_array), mlx_array, mlx_array, mlx_stream]),
    "mlx_unflatten": (c_int, [P_ARR, mlx_array, c_int, P_INT, c_size_t, mlx_stream]),
    "mlx_pad": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, P_INT, c_size_t, P_INT, c_size_t, mlx_array, c_char_p, mlx_stream]),
    "mlx_repeat_axis": (c_int, [P_ARR, mlx_array, c_int, c_int, mlx_stream]),
    "mlx_repeat": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_tile": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, mlx_stream]),
    "mlx_diagonal": (c_int, [P_ARR, mlx_array, c_int, c_int, c_int, mlx_stream]),
    "mlx_diag": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_nan_to_num": (c_int, [P_ARR, mlx_array, c_float, mlx_optional_float, mlx_optional_float, mlx_stream]),
    "mlx_broadcast_arrays": (c_int, [ctypes.POINTER(mlx_vector_array), mlx_vector_array, mlx_stream]),
    "mlx_sin": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_sum_axis": (c_int, [P_ARR, mlx_array, c_int, c_bool, mlx_stream]),
    "mlx_argsort_axis": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_argsort": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_argpartition_axis": (c_int, [P_ARR, mlx_array, c_int, c_int, mlx_stream]),
    "mlx_take": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "mlx_take_along_axis": (c_int, [P_ARR, mlx_array, mlx_array, c_int, mlx_stream]),
    "mlx_expand_dims_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, mlx_stream]),
    "mlx_squeeze_axes": (c_int, [P_ARR, mlx_array, P_INT, c_size_t, mlx_stream]),
    "mlx_squeeze_axis": (c_int, [P_ARR, mlx_array, c_int, mlx_stream]),
    "mlx_squeeze": (c_int, [P_ARR, mlx_array, mlx_stream]),
    "mlx_flatten": (c_int, [P_ARR, mlx_array, c_int, c_int, mlx_stream]),
    "mlx_stack_axis": (c_int, [P_ARR, mlx_vector_array, c_int, mlx_stream]),
    "mlx_stack": (c_int, [P_ARR, mlx_vector_array, mlx_stream]),
    "mlx_split": (c_int, [ctypes.POINTER(mlx_vector_array), mlx_array, c_int, c_int, mlx_stream]),
    "mlx_split_sections": (c_int, [ctypes.POINTER(mlx_vector_array), mlx_array, P_INT, c_size_t, c_int, mlx_stream]),
    "mlx_conv1d": (c_int, [P_ARR, mlx_array, mlx_array, c_int, c_int, c_int, c_int, mlx_stream]),
    "mlx_conv2d": (c_int, [P_ARR, mlx_array, mlx_array, c_int, c_int, c_int, c_int, c_int, c_int, c_int, mlx_stream]),
    "mlx_gather_mm": (c_int, [P_ARR, mlx_array, mlx_array, mlx_array, mlx_array, c_bool, mlx_stream]),
    "omx_mlx_array_from_device": (mlx_array, [ctypes.c_void_p, P_INT, c_int, c_int]),
    "omx_mlx_lazy_stats": (None, [ctypes.POINTER(ctypes.c_long)]),
    "omx_mlx_lazy_mode": (c_int, [c_int, c_int]),
    "omx_mlx_lazy_async": (c_int, [c_int]),
    "omx_mlx_fused_swiglu": (c_int, [P_ARR, mlx_array, mlx_array, mlx_stream]),
    "omx_mlx_fused_modulate": (c_int, [P_ARR, mlx_array, mlx_array, mlx_array, mlx_stream]),
}
for _n, (_r, _a) in SIGNATURES.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a


def _check(status: int) -> None:
    if status != 0:
        msg = lib.omx_last_error().decode("utf-8", "replace")
        lib.omx_clear_error()
        raise OmxError(msg or "unknown mlx_* error")


_NP = {FLOAT32: np.float32, UINT32: np.uint32, INT32: np.int32, BOOL: np.bool_, UINT8: np.uint8, FLOAT16: np.float16}
_STREAM = None


def default_stream() -> mlx_stream:
    global _STREAM
    if _STREAM is None:
        require_device()
        _STREAM = lib.mlx_default_gpu_stream_new()
    return _STREAM


def _ints(seq):
    arr = (c_int * len(seq))(*[int(v) for v in seq])
    return arr, len(seq)


class Array:
    """Owned mlx_array handle (mlx_rs::Array)."""

    def __init__(self, handle: mlx_array):
        self.h = handle

    def __del__(self):
        if sys is not None and not sys.is_finalizing() and getattr(self, "h", None) is not None and self.h.ctx:
            lib.mlx_array_free(self.h)
            self.h = mlx_array(None)

    # -- construction --
    @staticmethod
    def from_numpy(a, dtype=BFLOAT16) -> "Array":
        require_device()
        a = np.asarray(a)
        if dtype == BFLOAT16:
            u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
            host = ((u + ((u >> np.uint64(16)) & np.uint64(1)) + np.uint64(0x7FFF)) >> np.uint64(16)).astype(np.uint16)
        else:
            host = np.ascontiguousarray(a, _NP[dtype])
        shape, n = _ints(a.shape)
        h = lib.mlx_array_new_data(host.ctypes.data, shape, n, dtype)
        if not h.ctx:
            _check(1)
        return Array(h)

    @staticmethod
    def op(fn, *args) -> "Array":
        """Guarded::try_from_op: fresh empty out-handle, adopt on status 0, free + raise otherwise."""
        res = lib.mlx_array_new()
        status = fn(ctypes.byref(res), *args)
        if status != 0:
            lib.mlx_array_free(res)
            _check(status)
        return Array(res)

    # -- inspection --
    @property
    def shape(self):
        n = lib.mlx_array_ndim(self.h)
        p = lib.mlx_array_shape(self.h)
        return tuple(p[i] for i in range(n))

    @property
    def strides(self):
        n = lib.mlx_array_ndim(self.h)
        p = lib.mlx_array_strides(self.h)
        return tuple(p[i] for i in range(n))

    @property
    def dtype(self) -> int:
        return lib.mlx_array_dtype(self.h)

    def eval(self) -> None:
        _check(lib.mlx_array_eval(self.h))

    def item(self):
        dt = self.dtype
        if dt == UINT32:
            v = ctypes.c_uint32(); _check(lib.mlx_array_item_uint32(ctypes.byref(v), self.h)); return v.value
        if dt == INT32:
            v = ctypes.c_int32(); _check(lib.mlx_array_item_int32(ctypes.byref(v), self.h)); return v.value
        if dt == FLOAT32:
            v = c_float(); _check(lib.mlx_array_item_float32(ctypes.byref(v), self.h)); return v.value
        raise OmxError(f"item(): unsupported dtype {dt}")

    def numpy(self) -> np.ndarray:
        """as_slice(): goes through mlx_array_data_* (host mirror); bf16 widened to float32."""
        dt, shape = self.dtype, self.shape
        n = int(np.prod(shape, dtype=np.int64))
        if dt == BFLOAT16:
            p = lib.mlx_array_data_bfloat16(self.h)
            raw = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint16)), (max(n, 1),))[:n].copy()
            return (raw.astype(np.uint32) << np.uint32(16)).view(np.float32).reshape(shape)
        if dt == FLOAT16:      # as_slice::<f16>(): mlx_array_data_float16
            p = lib.mlx_array_data_float16(self.h)
            if not p:
                _check(1)
            return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint16)), (max(n, 1),))[:n].copy().view(np.float16).reshape(shape)
        fn = {FLOAT32: lib.mlx_array_data_float32, UINT32: lib.mlx_array_data_uint32, INT32: lib.mlx_array_data_int32,
              BOOL: lib.mlx_array_data_bool, UINT8: lib.mlx_array_data_uint8, INT8: lib.mlx_array_data_int8, INT16: lib.mlx_array_data_int16,
              UINT16: lib.mlx_array_data_uint16, INT64: lib.mlx_array_data_int64, UINT64: lib.mlx_array_data_uint64}[dt]
        ct = {FLOAT32: c_float, UINT32: ctypes.c_uint32, INT32: ctypes.c_int32, BOOL: ctypes.c_uint8, UINT8: ctypes.c_uint8, INT8: ctypes.c_int8,
              INT16: ctypes.c_int16, UINT16: ctypes.c_uint16, INT64: ctypes.c_int64, UINT64: ctypes.c_uint64}[dt]
        p = fn(self.h)
        if not p:
            _check(1)
        out = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ct)), (max(n, 1),))[:n].copy().reshape(shape)
        return out.astype(bool) if dt == BOOL else out


_EMPTY = mlx_array(None)


def _h(a: Optional[Array]) -> mlx_array:
    return _EMPTY if a is None else a.h


# ---- mlx_rs::ops / mlx_rs::fast, one function per C entry point ----
def rms_norm(x, weight, eps):
    return Array.op(lib.mlx_fast_rms_norm, x.h, _h(weight), eps, default_stream())


def layer_norm(x, weight, bias, eps):
    return Array.op(lib.mlx_fast_layer_norm, x.h, _h(weight), _h(bias), eps, default_stream())


def rope(x, dims, traditional, base, scale, offset, freqs=None):
    return Array.op(lib.mlx_fast_rope, x.h, dims, traditional, mlx_optional_float(base if base is not None else 0.0,
                    base is not None), scale, offset, _h(freqs), default_stream())


def scaled_dot_product_attention(q, k, v, scale, mask=None):
    """mask: None | "causal" | Array (bool or additive) -- fast.rs:88-108 (mode "" + array, or "causal")."""
    mode, arr = b"", None
    if isinstance(mask, str):
        mode = mask.encode()
    elif mask is not None:
        arr = mask
    return Array.op(lib.mlx_fast_scaled_dot_product_attention, q.h, k.h, v.h, scale, mode, _h(arr), _EMPTY, default_stream())


def matmul(a, b):
    return Array.op(lib.mlx_matmul, a.h, b.h, default_stream())


def addmm(c, a, b, alpha=1.0, beta=1.0):
    return Array.op(lib.mlx_addmm, c.h, a.h, b.h, alpha, beta, default_stream())


def add(a, b): return Array.op(lib.mlx_add, a.h, b.h, default_stream())
def subtract(a, b): return Array.op(lib.mlx_subtract, a.h, b.h, default_stream())
def multiply(a, b): return Array.op(lib.mlx_multiply, a.h, b.h, default_stream())
def divide(a, b): return Array.op(lib.mlx_divide, a.h, b.h, default_stream())
def sigmoid(a): return Array.op(lib.mlx_sigmoid, a.h, default_stream())
def exp(a): return Array.op(lib.mlx_exp, a.h, default_stream())
def negative(a): return Array.op(lib.mlx_negative, a.h, default_stream())
def astype(a, dtype): return Array.op(lib.mlx_astype, a.h, dtype, default_stream())
def transpose(a): return Array.op(lib.mlx_transpose, a.h, default_stream())
def expand_dims(a, axis): return Array.op(lib.mlx_expand_dims, a.h, axis, default_stream())
def contiguous(a): return Array.op(lib.mlx_contiguous, a.h, False, default_stream())
def take_axis(a, indices, axis): return Array.op(lib.mlx_take_axis, a.h, indices.h, axis, default_stream())
def argmax_axis(a, axis, keepdims=False): return Array.op(lib.mlx_argmax_axis, a.h, axis, keepdims, default_stream())
def softmax_axis(a, axis, precise=True): return Array.op(lib.mlx_softmax_axis, a.h, axis, precise, default_stream())


# ---- mlx_rs::random (random.rs): keys are [2] u32 arrays; key=None draws from the library's global sequence ----
def random_seed(seed: int) -> None:
    _check(lib.mlx_random_seed(int(seed) & 0xFFFFFFFFFFFFFFFF))


def random_key(seed: int) -> Array:
    require_device()
    return Array.op(lambda res: lib.mlx_random_key(res, int(seed) & 0xFFFFFFFFFFFFFFFF))


def random_split(key: Array, num: int = 2):
    """random.rs:103-115: split_num then index rows 0 and 1."""
    keys = Array.op(lib.mlx_random_split_num, key.h, num, default_stream())
    return tuple(reshape(slice(keys, [i, 0], [i + 1, 2]), [2]) for i in range(num))


def random_bits(shape, key: Optional[Array] = None, width: int = 4):
    s, n = _ints(shape)
    return Array.op(lib.mlx_random_bits, s, n, width, _h(key), default_stream())


def random_uniform(low, high, shape, key: Optional[Array] = None, dtype=FLOAT32):
    lo, hi = Array.from_numpy(np.float32(low), FLOAT32), Array.from_numpy(np.float32(high), FLOAT32)
    s, n = _ints(shape)
    return Array.op(lib.mlx_random_uniform, lo.h, hi.h, s, n, dtype, _h(key), default_stream())


def random_gumbel(shape, key: Optional[Array] = None, dtype=FLOAT32):
    s, n = _ints(shape)
    return Array.op(lib.mlx_random_gumbel, s, n, dtype, _h(key), default_stream())


def random_normal(shape, key: Optional[Array] = None, loc: float = 0.0, scale: float = 1.0, dtype=FLOAT32):
    s, n = _ints(shape)
    return Array.op(lib.mlx_random_normal, s, n, dtype, float(loc), float(scale), _h(key), default_stream())


def random_categorical(logits: Array, axis: int = -1, num_samples: Optional[int] = None, key: Optional[Array] = None):
    if num_samples is None:
        return Array.op(lib.mlx_random_categorical, logits.h, axis, _h(key), default_stream())
    return Array.op(lib.mlx_random_categorical_num_samples, logits.h, axis, int(num_samples), _h(key), default_stream())


def fused_swiglu(x, gate): return Array.op(lib.omx_mlx_fused_swiglu, x.h, gate.h, default_stream())
def fused_modulate(x, shift, scale): return Array.op(lib.omx_mlx_fused_modulate, x.h, shift.h, scale.h, default_stream())


def reshape(a, shape):
    s, n = _ints(shape)
    return Array.op(lib.mlx_reshape, a.h, s, n, default_stream())


def transpose_axes(a, axes):
    s, n = _ints(axes)
    return Array.op(lib.mlx_transpose_axes, a.h, s, n, default_stream())


def zeros(shape, dtype):
    s, n = _ints(shape)
    return Array.op(lib.mlx_zeros, s, n, dtype, default_stream())


def slice(a, start, stop, strides=None):
    st, n = _ints(start)
    sp, _ = _ints(stop)
    sd, _ = _ints(strides if strides is not None else [1] * n)
    return Array.op(lib.mlx_slice, a.h, st, n, sp, n, sd, n, default_stream())


def slice_update(src, update, start, stop, strides=None):
    st, n = _ints(start)
    sp, _ = _ints(stop)
    sd, _ = _ints(strides if strides is not None else [1] * n)
    return Array.op(lib.mlx_slice_update, src.h, update.h, st, n, sp, n, sd, n, default_stream())


def concatenate_axis(arrays: Sequence[Array], axis: int):
    vec = lib.mlx_vector_array_new()
    try:
        for a in arrays:
            _check(lib.mlx_vector_array_append_value(vec, a.h))
        return Array.op(lib.mlx_concatenate_axis, vec, axis, default_stream())
    finally:
        lib.mlx_vector_array_free(vec)


def _oi(v):
    return mlx_optional_int(int(v) if v is not None else 0, v is not None)


def quantize(w: "Array", group_size=None, bits=None):
    """mlx_rs::ops::quantize -> (w_q, scales, biases) (ops/quantization.rs:41-84)."""
    vec = lib.mlx_vector_array_new()
    try:
        _check(lib.mlx_quantize(ctypes.byref(vec), w.h, _oi(group_size), _oi(bits), b"affine", default_stream()))
        out = []
        for i in range(3):
            h = lib.mlx_array_new()
            _check(lib.mlx_vector_array_get(ctypes.byref(h), vec, i))
            out.append(Array(h))
        return tuple(out)
    finally:
        lib.mlx_vector_array_free(vec)


def dequantize(w, scales, biases, group_size=None, bits=None):
    return Array.op(lib.mlx_dequantize, w.h, scales.h, _h(biases), _oi(group_size), _oi(bits), b"affine",
                    mlx_optional_dtype(0, False), default_stream())


def quantized_matmul(x, w, scales, biases, transpose=True, group_size=None, bits=None):
    return Array.op(lib.mlx_quantized_matmul, x.h, w.h, scales.h, _h(biases), transpose, _oi(group_size), _oi(bits), b"affine",
                    default_stream())


def gather_qmm(x, w, scales, biases, rhs_indices, transpose=True, group_size=None, bits=None, sorted_indices=False):
    return Array.op(lib.mlx_gather_qmm, x.h, w.h, scales.h, _h(biases), mlx_array(None), rhs_indices.h, transpose, _oi(group_size),
                    _oi(bits), b"affine", sorted_indices, default_stream())


def eval(*arrays) -> None:
    vec = lib.mlx_vector_array_new()
    try:
        for a in arrays:
            _check(lib.mlx_vector_array_append_value(vec, a.h))
        _check(lib.mlx_eval(vec))
    finally:
        lib.mlx_vector_array_free(vec)


# ---- the rest of mlx_rs::ops used by the four callers (round 2) ----
def _bin(fn):
    return lambda a, b: Array.op(fn, a.h, b.h, default_stream())


greater, greater_equal, less, less_equal, equal = (_bin(lib.mlx_greater), _bin(lib.mlx_greater_equal), _bin(lib.mlx_less),
                                                   _bin(lib.mlx_less_equal), _bin(lib.mlx_equal))
logical_and, maximum, minimum, floor_divide = _bin(lib.mlx_logical_and), _bin(lib.mlx_maximum), _bin(lib.mlx_minimum), _bin(lib.mlx_floor_divide)
def cos(a): return Array.op(lib.mlx_cos, a.h, default_stream())


def unary_op(name, a):
    """ops.h elementwise math / predicates by name: abs sqrt rsqrt square log log2 log10 log1p expm1 tanh ... isnan ... logical_not."""
    return Array.op(getattr(lib, "mlx_" + name), a.h, default_stream())


def binary_op(name, a, b):
    """not_equal logical_or power remainder logaddexp (and the comparison / max / min family)."""
    return Array.op(getattr(lib, "mlx_" + name), a.h, b.h, default_stream())


def reduce_axis_op(name, a, axis, keepdims=False):
    """all_axis any_axis logsumexp_axis (and max / min / mean / sum _axis) by name."""
    return Array.op(getattr(lib, "mlx_" + name), a.h, axis, keepdims, default_stream())


def reduce_all_op(name, a, keepdims=False):
    """max min mean sum all any logsumexp over the whole array."""
    return Array.op(getattr(lib, "mlx_" + name), a.h, keepdims, default_stream())


def sort_axis(a, axis): return Array.op(lib.mlx_sort_axis, a.h, axis, default_stream())
def sort(a): return Array.op(lib.mlx_sort, a.h, default_stream())
def stop_gradient(a): return Array.op(lib.mlx_stop_gradient, a.h, default_stream())


def broadcast_to(a, shape):
    s, n = _ints(shape)
    return Array.op(lib.mlx_broadcast_to, a.h, s, n, default_stream())


def round_(a, decimals=0): return Array.op(lib.mlx_round, a.h, decimals, default_stream())
def max_axis(a, axis, keepdims=False): return Array.op(lib.mlx_max_axis, a.h, axis, keepdims, default_stream())
def min_axis(a, axis, keepdims=False): return Array.op(lib.mlx_min_axis, a.h, axis, keepdims, default_stream())
def mean_axis(a, axis, keepdims=False): return Array.op(lib.mlx_mean_axis, a.h, axis, keepdims, default_stream())
def swapaxes(a, axis1, axis2): return Array.op(lib.mlx_swapaxes, a.h, axis1, axis2, default_stream())
def moveaxis(a, source, destination): return Array.op(lib.mlx_moveaxis, a.h, source, destination, default_stream())
def where(cond, x, y): return Array.op(lib.mlx_where, cond.h, x.h, y.h, default_stream())
def clip(a, lo=None, hi=None): return Array.op(lib.mlx_clip, a.h, _h(lo), _h(hi), default_stream())


def full(shape, value, dtype):
    s, n = _ints(shape)
    return Array.op(lib.mlx_full, s, n, value.h, dtype, default_stream())


def ones(shape, dtype):
    s, n = _ints(shape)
    return Array.op(lib.mlx_ones, s, n, dtype, default_stream())
def sin(a): return Array.op(lib.mlx_sin, a.h, default_stream())
def arange(start, stop, step=1.0, dtype=INT32): return Array.op(lambda res: lib.mlx_arange(res, float(start), float(stop), float(step), dtype, default_stream()))
def sum_axis(a, axis, keepdims=False): return Array.op(lib.mlx_sum_axis, a.h, axis, keepdims, default_stream())
def argsort(a): return Array.op(lib.mlx_argsort, a.h, default_stream())
def argsort_axis(a, axis): return Array.op(lib.mlx_argsort_axis, a.h, axis, default_stream())
def argpartition_axis(a, kth, axis): return Array.op(lib.mlx_argpartition_axis, a.h, kth, axis, default_stream())
def take(a, indices): return Array.op(lib.mlx_take, a.h, indices.h, default_stream())
def take_along_axis(a, indices, axis): return Array.op(lib.mlx_take_along_axis, a.h, indices.h, axis, default_stream())
def squeeze(a): return Array.op(lib.mlx_squeeze, a.h, default_stream())
def squeeze_axis(a, axis): return Array.op(lib.mlx_squeeze_axis, a.h, axis, default_stream())
def flatten(a, start_axis=0, end_axis=-1): return Array.op(lib.mlx_flatten, a.h, start_axis, end_axis, default_stream())
def conv1d(x, w, stride=1, padding=0, dilation=1, groups=1): return Array.op(lib.mlx_conv1d, x.h, w.h, stride, padding, dilation, groups, default_stream())
def conv2d(x, w, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1):
    return Array.op(lib.mlx_conv2d, x.h, w.h, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1], groups, default_stream())


def expand_dims_axes(a, axes):
    s, n = _ints(axes)
    return Array.op(lib.mlx_expand_dims_axes, a.h, s, n, default_stream())


def squeeze_axes(a, axes):
    s, n = _ints(axes)
    return Array.op(lib.mlx_squeeze_axes, a.h, s, n, default_stream())


def _vec(arrays):
    vec = lib.mlx_vector_array_new()
    for a in arrays:
        _check(lib.mlx_vector_array_append_value(vec, a.h))
    return vec


def _unvec(vec):
    out = []
    for i in range(lib.mlx_vector_array_size(vec)):
        h = lib.mlx_array_new()
        _check(lib.mlx_vector_array_get(ctypes.byref(h), vec, i))
        out.append(Array(h))
    return out


def stack_axis(arrays, axis=0):
    vec = _vec(arrays)
    try:
        return Array.op(lib.mlx_stack_axis, vec, axis, default_stream())
    finally:
        lib.mlx_vector_array_free(vec)


def split(a, num_splits, axis=0):
    vec = lib.mlx_vector_array_new()
    try:
        _check(lib.mlx_split(ctypes.byref(vec), a.h, num_splits, axis, default_stream()))
        return _unvec(vec)
    finally:
        lib.mlx_vector_array_free(vec)


def split_sections(a, indices, axis=0):
    vec = lib.mlx_vector_array_new()
    s, n = _ints(indices)
    try:
        _check(lib.mlx_split_sections(ctypes.byref(vec), a.h, s, n, axis, default_stream()))
        return _unvec(vec)
    finally:
        lib.mlx_vector_array_free(vec)


def gather_mm(a, b, rhs_indices, sorted_indices=False):
    return Array.op(lib.mlx_gather_mm, a.h, b.h, mlx_array(None), rhs_indices.h, sorted_indices, default_stream())


def tostring(a) -> str:
    st = lib.mlx_string_new()
    try:
        _check(lib.mlx_array_tostring(ctypes.byref(st), a.h))
        return lib.mlx_string_data(st).decode()
    finally:
        lib.mlx_string_free(st)


def load_safetensors(path: str):
    """mlx_rs::Array::load_safetensors (utils/io.rs:40-120): (name -> Array, metadata name -> str) walked with the map iterators."""
    require_device()
    arrays, meta = lib.mlx_map_string_to_array_new(), lib.mlx_map_string_to_string_new()
    try:
        _check(lib.mlx_load_safetensors(ctypes.byref(arrays), ctypes.byref(meta), path.encode(), default_stream()))
        out, md = {}, {}
        it = lib.mlx_map_string_to_array_iterator_new(arrays)
        while True:
            key, h = c_char_p(), lib.mlx_array_new()
            st = lib.mlx_map_string_to_array_iterator_next(ctypes.byref(key), ctypes.byref(h), it)
            if st == 2:
                break
            _check(st)
            out[key.value.decode()] = Array(h)
        lib.mlx_map_string_to_array_iterator_free(it)
        it = lib.mlx_map_string_to_string_iterator_new(meta)
        while True:
            key, val = c_char_p(), c_char_p()
            st = lib.mlx_map_string_to_string_iterator_next(ctypes.byref(key), ctypes.byref(val), it)
            if st == 2:
                break
            _check(st)
            md[key.value.decode()] = val.value.decode()
        lib.mlx_map_string_to_string_iterator_free(it)
        return out, md
    finally:
        lib.mlx_map_string_to_array_free(arrays)
        lib.mlx_map_string_to_string_free(meta)


UNARY_FN = ctypes.CFUNCTYPE(c_int, P_ARR, mlx_array)


def compile_unary(fn):
    """mlx_rs::transforms::compile over a one-array closure, the way nn::silu is wrapped (nn/activation.rs:876-880 ->
    compile.rs:334: mlx_closure_new_unary / payload closure -> mlx_detail_compile -> mlx_closure_apply).  `fn(Array) -> Array`."""
    def tramp(res_p, x):
        wrap = Array(mlx_array(x.ctx))              # borrowed input: the wrapper must not free it
        try:
            y = fn(wrap)
            _check(lib.mlx_array_set(res_p, y.h))
            return 0
        except Exception:                           # noqa: BLE001 -- the C side reports status 1
            return 1
        finally:
            wrap.h = mlx_array(None)
    cb = UNARY_FN(tramp)
    plain = lib.mlx_closure_new_unary(ctypes.cast(cb, c_void_p))
    compiled = lib.mlx_closure_new()
    _check(lib.mlx_detail_compile(ctypes.byref(compiled), plain, id(fn), False, None, 0))
    lib.mlx_closure_free(plain)

    def call(x):
        vin = _vec([x])
        vout = lib.mlx_vector_array_new()
        try:
            _check(lib.mlx_closure_apply(ctypes.byref(vout), compiled, vin))
            return _unvec(vout)[0]
        finally:
            lib.mlx_vector_array_free(vin)
            lib.mlx_vector_array_free(vout)
    call._keep = (cb, compiled)
    return call


# ---- ops.h, third batch (csrc/mlxc_glue2.hpp): thin wrappers, numpy-like argument order ----
def _axes_form(fn_axes, fn_all):
    def f(a, axes=None, keepdims=False):
        if axes is None:
            return Array.op(fn_all, a.h, keepdims, default_stream())
        p, n = _ints([axes] if isinstance(axes, int) else axes)
        return Array.op(fn_axes, a.h, p, n, keepdims, default_stream())
    return f


sum_axes = _axes_form(lib.mlx_sum_axes, lib.mlx_sum)
mean_axes = _axes_form(lib.mlx_mean_axes, lib.mlx_mean)
max_axes = _axes_form(lib.mlx_max_axes, lib.mlx_max)
min_axes = _axes_form(lib.mlx_min_axes, lib.mlx_min)
all_axes = _axes_form(lib.mlx_all_axes, lib.mlx_all)
any_axes = _axes_form(lib.mlx_any_axes, lib.mlx_any)
logsumexp_axes = _axes_form(lib.mlx_logsumexp_axes, lib.mlx_logsumexp)
prod_axes = _axes_form(lib.mlx_prod_axes, lib.mlx_prod)


def var(a, axes=None, keepdims=False, ddof=0, std=False):
    if axes is None:
        return Array.op(lib.mlx_std if std else lib.mlx_var, a.h, keepdims, ddof, default_stream())
    p, n = _ints([axes] if isinstance(axes, int) else axes)
    return Array.op(lib.mlx_std_axes if std else lib.mlx_var_axes, a.h, p, n, keepdims, ddof, default_stream())


def softmax_axes(a, axes=None, precise=False):
    if axes is None:
        return Array.op(lib.mlx_softmax, a.h, precise, default_stream())
    p, n = _ints(axes)
    return Array.op(lib.mlx_softmax_axes, a.h, p, n, precise, default_stream())


def argmax_all(a, keepdims=False): return Array.op(lib.mlx_argmax, a.h, keepdims, default_stream())
def argmin(a, axis=None, keepdims=False):
    if axis is None:
        return Array.op(lib.mlx_argmin, a.h, keepdims, default_stream())
    return Array.op(lib.mlx_argmin_axis, a.h, axis, keepdims, default_stream())


def scan(kind, a, axis, reverse=False, inclusive=True):
    return Array.op(getattr(lib, "mlx_cum" + kind), a.h, axis, reverse, inclusive, default_stream())


def topk(a, k, axis=-1): return Array.op(lib.mlx_topk_axis, a.h, k, axis, default_stream())
def partition(a, kth, axis=None):
    if axis is None:
        return Array.op(lib.mlx_partition, a.h, kth, default_stream())
    return Array.op(lib.mlx_partition_axis, a.h, kth, axis, default_stream())


def argpartition_flat(a, kth): return Array.op(lib.mlx_argpartition, a.h, kth, default_stream())
def tri(n, m, k, dtype): return Array.op(lib.mlx_tri, n, m, k, dtype, default_stream())
def tril(a, k=0): return Array.op(lib.mlx_tril, a.h, k, default_stream())
def triu(a, k=0): return Array.op(lib.mlx_triu, a.h, k, default_stream())
def eye(n, m, k, dtype): return Array.op(lib.mlx_eye, n, m, k, dtype, default_stream())
def identity(n, dtype): return Array.op(lib.mlx_identity, n, dtype, default_stream())
def linspace(start, stop, num, dtype): return Array.op(lib.mlx_linspace, float(start), float(stop), num, dtype, default_stream())
def outer(a, b): return Array.op(lib.mlx_outer, a.h, b.h, default_stream())
def inner(a, b): return Array.op(lib.mlx_inner, a.h, b.h, default_stream())
def atleast(a, nd): return Array.op(getattr(lib, f"mlx_atleast_{nd}d"), a.h, default_stream())
def isclose(a, b, rtol=1e-5, atol=1e-8, equal_nan=False): return Array.op(lib.mlx_isclose, a.h, b.h, rtol, atol, equal_nan, default_stream())
def allclose(a, b, rtol=1e-5, atol=1e-8, equal_nan=False): return Array.op(lib.mlx_allclose, a.h, b.h, rtol, atol, equal_nan, default_stream())
def array_equal(a, b, equal_nan=False): return Array.op(lib.mlx_array_equal, a.h, b.h, equal_nan, default_stream())
def degrees(a): return Array.op(lib.mlx_degrees, a.h, default_stream())
def radians(a): return Array.op(lib.mlx_radians, a.h, default_stream())


def divmod_(a, b):
    vec = lib.mlx_vector_array_new()
    try:
        _check(lib.mlx_divmod(ctypes.byref(vec), a.h, b.h, default_stream()))
        return _unvec(vec)
    finally:
        lib.mlx_vector_array_free(vec)


def unflatten(a, axis, shape):
    p, n = _ints(shape)
    return Array.op(lib.mlx_unflatten, a.h, axis, p, n, default_stream())


def pad(a, axes, low, high, value):
    pa, na = _ints(axes)
    pl, nl = _ints(low)
    ph, nh = _ints(high)
    return Array.op(lib.mlx_pad, a.h, pa, na, pl, nl, ph, nh, value.h, b"constant", default_stream())


def repeat(a, repeats, axis=None):
    if axis is None:
        return Array.op(lib.mlx_repeat, a.h, repeats, default_stream())
    return Array.op(lib.mlx_repeat_axis, a.h, repeats, axis, default_stream())


def tile(a, reps):
    p, n = _ints(reps)
    return Array.op(lib.mlx_tile, a.h, p, n, default_stream())


def diagonal(a, offset=0, axis1=0, axis2=1): return Array.op(lib.mlx_diagonal, a.h, offset, axis1, axis2, default_stream())
def diag(a, k=0): return Array.op(lib.mlx_diag, a.h, k, default_stream())
def nan_to_num(a, nan=0.0, posinf=None, neginf=None):
    opt = lambda v: mlx_optional_float(float(v) if v is not None else 0.0, v is not None)
    return Array.op(lib.mlx_nan_to_num, a.h, float(nan), opt(posinf), opt(neginf), default_stream())


def broadcast_arrays(arrays):
    vec, out = _vec(arrays), lib.mlx_vector_array_new()
    try:
        _check(lib.mlx_broadcast_arrays(ctypes.byref(out), vec, default_stream()))
        return _unvec(out)
    finally:
        lib.mlx_vector_array_free(vec)
        lib.mlx_vector_array_free(out)


def as_strided(a, shape, strides, offset=0):
    p, n = _ints(shape)
    st = (ctypes.c_int64 * len(strides))(*[int(v) for v in strides])
    return Array.op(lib.mlx_as_strided, a.h, p, n, st, len(strides), offset, default_stream())


def view(a, dtype): return Array.op(lib.mlx_view, a.h, dtype, default_stream())
def real(a): return Array.op(lib.mlx_real, a.h, default_stream())
def imag(a): return Array.op(lib.mlx_imag, a.h, default_stream())
def tensordot(a, b, axes):
    if isinstance(axes, int):
        return Array.op(lib.mlx_tensordot_axis, a.h, b.h, axes, default_stream())
    pa, na = _ints(axes[0])
    pb, nb = _ints(axes[1])
    return Array.op(lib.mlx_tensordot, a.h, b.h, pa, na, pb, nb, default_stream())


def kron(a, b): return Array.op(lib.mlx_kron, a.h, b.h, default_stream())
def random_bernoulli(p, shape, key):
    ps, n = _ints(shape)
    return Array.op(lib.mlx_random_bernoulli, p.h, ps, n, key.h, default_stream())


def lazy_stats() -> dict:
    """Counters of the deferred op list behind the ABI (csrc/mlxc_lazy.hpp)."""
    v = (ctypes.c_long * 6)()
    lib.omx_mlx_lazy_stats(v)
    return {"recorded": v[0], "launched_as_recorded": v[1], "fused_launches": v[2], "flushes": v[3], "flush_host_ns": v[4], "rewrite_ns": v[5]}


def lazy_mode(lazy: bool = True, fuse: bool = True, worker: bool = False) -> None:
    _check(lib.omx_mlx_lazy_mode(1 if lazy else 0, 1 if fuse else 0))
    _check(lib.omx_mlx_lazy_async(1 if worker else 0))


def async_eval(*arrays) -> None:
    vec = lib.mlx_vector_array_new()
    try:
        for a in arrays:
            _check(lib.mlx_vector_array_append_value(vec, a.h))
        _check(lib.mlx_async_eval(vec))
    finally:
        lib.mlx_vector_array_free(vec)
