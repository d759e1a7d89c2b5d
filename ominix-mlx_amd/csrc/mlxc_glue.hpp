// Second half of the mlx-c handle layer (included at the end of mlxc.hip: it uses that file's Arr / Vec / assign / Contig).
// Devices, streams, strings, maps, closures and the glue ops of SURVEY.md section 8b that round 1 left out -- what the four
// callers touch outside the fused hot-path ops:
//   Mixtral routing + gather_sort / scatter_unsort     mixtral-mlx/src/model.rs:204-228, 296-308
//   create_causal_mask                                 mlx-rs-core/src/utils.rs:134-153
//   Stream::default / Device                           mlx-rs/src/stream.rs:150-195, device.rs
//   compile-wrapped activations                        mlx-rs/src/nn/activation.rs:876-880, transforms/compile/compile.rs:334
//   Paraformer FSMN conv1d                             funasr-mlx/src/paraformer.rs:496-532
//   safetensors loading                                mlx-rs/src/utils/io.rs:40-120 -> mlx-c io.h:40-44
// Reference shim files mirrored (ownership, status codes): mlx-c/mlx/c/{device,stream,string,vector,map,closure,compile,io,ops}.cpp
#pragma once

#include <stdio.h>

#include <cmath>
#include <fstream>
#include <functional>
#include <string>

namespace {

// ---------------------------------------------------------------- handles
struct Dev { mlx_device_type type; int index; };
struct MapArr { std::map<std::string, Arr*> m; ~MapArr() { for (auto& kv : m) delete kv.second; } };
struct MapStr { std::map<std::string, std::string> m; };
struct Closure {
    std::function<int(mlx_vector_array*, const mlx_vector_array)> fn;
    std::shared_ptr<void> payload;   // keeps the payload's destructor alive with the last copy
};
Dev g_default_dev = {MLX_GPU, 0};

Arr* clone_handle(const Arr* a) {
    Arr* n = new Arr(*a);
    n->host.clear();
    return n;
}

// ---------------------------------------------------------------- kernels

enum { B2_GT, B2_GE, B2_LT, B2_LE, B2_EQ, B2_AND, B2_NE, B2_OR, B2_LAST_BOOL = B2_OR, B2_MAX, B2_MIN, B2_FLOORDIV, B2_POW, B2_REM, B2_LOGADDEXP };

// comparison / logical / max / min / floor_divide with broadcasting; integer operands are combined in 64-bit integers
// (indices beyond 2^24 would not survive a float round trip)
__global__ void binary2_kernel(char* out, int odt, const char* a, int adt, const char* b, int bdt, Idx ix, size_t n, int op, bool int_math) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        long long oa = 0, ob = 0;
        for (int d = ix.nd - 1; d >= 0; --d) {
            const int c = (int)(r % ix.shape[d]);
            r /= ix.shape[d];
            oa += c * ix.sa[d];
            ob += c * ix.sb[d];
        }
        if (int_math) {
            const long long x = ld_i(a, adt, oa), y = ld_i(b, bdt, ob);
            long long v;
            switch (op) {
                case B2_GT: v = x > y; break;
                case B2_GE: v = x >= y; break;
                case B2_LT: v = x < y; break;
                case B2_LE: v = x <= y; break;
                case B2_EQ: v = x == y; break;
                case B2_AND: v = (x != 0) && (y != 0); break;
                case B2_NE: v = x != y; break;
                case B2_OR: v = (x != 0) || (y != 0); break;
                case B2_MAX: v = x > y ? x : y; break;
                case B2_MIN: v = x < y ? x : y; break;
                case B2_POW: {   // integer power by squaring; a negative exponent gives 0 (1 for a base of 1), as integer division would
                    long long base = x, e = y;
                    v = 1;
                    if (e < 0) { v = (x == 1) ? 1 : (x == -1 ? ((e & 1) ? -1 : 1) : 0); break; }
                    while (e) { if (e & 1) v *= base; base *= base; e >>= 1; }
                    break;
                }
                case B2_REM: {   // sign of the divisor (numpy / MLX remainder); division by zero gives 0
                    if (y == 0) { v = 0; break; }
                    v = x % y;
                    if (v != 0 && ((v < 0) != (y < 0))) v += y;
                    break;
                }
                case B2_LOGADDEXP: v = 0; break;   // (float-only op: the host wrapper never takes the integer route for it)
                default: {   // floor division (numpy / MLX semantics for negative operands); division by zero gives 0
                    if (y == 0) { v = 0; break; }
                    v = x / y;
                    if ((x % y != 0) && ((x < 0) != (y < 0))) --v;
                }
            }
            st_i(out, odt, i, v);
        } else {
            const float x = ld_f(a, adt, oa), y = ld_f(b, bdt, ob);
            float v;
            switch (op) {
                case B2_GT: v = x > y; break;
                case B2_GE: v = x >= y; break;
                case B2_LT: v = x < y; break;
                case B2_LE: v = x <= y; break;
                case B2_EQ: v = x == y; break;
                case B2_AND: v = (x != 0.f) && (y != 0.f); break;
                case B2_NE: v = x != y; break;
                case B2_OR: v = (x != 0.f) || (y != 0.f); break;
                case B2_MAX: v = (x != x || y != y) ? NAN : fmaxf(x, y); break;   // NaN propagates like MLX's maximum
                case B2_MIN: v = (x != x || y != y) ? NAN : fminf(x, y); break;
                case B2_POW: v = powf(x, y); break;
                case B2_REM: { v = fmodf(x, y); if (v != 0.f && ((v < 0.f) != (y < 0.f))) v += y; break; }
                case B2_LOGADDEXP: { const float mx = fmaxf(x, y), mn = fminf(x, y); v = (mx == -INFINITY) ? -INFINITY : mx + log1pf(expf(mn - mx)); break; }
                default: v = floorf(x / y); break;
            }
            st_f(out, odt, i, v);
        }
    }
}
enum { U2_COS, U2_SIN, U2_ABS, U2_SQRT, U2_RSQRT, U2_SQUARE, U2_LOG, U2_LOG2, U2_LOG10, U2_LOG1P, U2_EXPM1, U2_TANH, U2_SINH, U2_COSH, U2_TAN,
       U2_ARCSIN, U2_ARCCOS, U2_ARCTAN, U2_ARCSINH, U2_ARCCOSH, U2_ARCTANH, U2_ERF, U2_RECIP, U2_FLOOR, U2_CEIL, U2_ROUND, U2_SIGN,
       U2_FIRST_PRED, U2_ISNAN = U2_FIRST_PRED, U2_ISINF, U2_ISFINITE, U2_ISPOSINF, U2_ISNEGINF, U2_NOT };
// elementwise math of ops.h beyond the hot path (float32 inside, one rounding to the output dtype; MLX: float results for float
// inputs, float32 for integer inputs; predicates and logical_not give bool)
__device__ inline float unary2_apply(int op, float x) {
    switch (op) {
        case U2_COS: return cosf(x);
        case U2_SIN: return sinf(x);
        case U2_ABS: return fabsf(x);
        case U2_SQRT: return sqrtf(x);
        case U2_RSQRT: return 1.0f / sqrtf(x);
        case U2_SQUARE: return x * x;
        case U2_LOG: return logf(x);
        case U2_LOG2: return log2f(x);
        case U2_LOG10: return log10f(x);
        case U2_LOG1P: return log1pf(x);
        case U2_EXPM1: return expm1f(x);
        case U2_TANH: return tanhf(x);
        case U2_SINH: return sinhf(x);
        case U2_COSH: return coshf(x);
        case U2_TAN: return tanf(x);
        case U2_ARCSIN: return asinf(x);
        case U2_ARCCOS: return acosf(x);
        case U2_ARCTAN: return atanf(x);
        case U2_ARCSINH: return asinhf(x);
        case U2_ARCCOSH: return acoshf(x);
        case U2_ARCTANH: return atanhf(x);
        case U2_ERF: return erff(x);
        case U2_RECIP: return 1.0f / x;
        case U2_FLOOR: return floorf(x);
        case U2_CEIL: return ceilf(x);
        case U2_ROUND: return rintf(x);                       // round half to even, as MLX (numpy) does
        case U2_SIGN: return (float)((x > 0.f) - (x < 0.f));
        case U2_ISNAN: return x != x;
        case U2_ISINF: return isinf(x);
        case U2_ISFINITE: return isfinite(x);
        case U2_ISPOSINF: return isinf(x) && x > 0.f;
        case U2_ISNEGINF: return isinf(x) && x < 0.f;
        default: return x == 0.f;                               // U2_NOT
    }
}
__global__ void unary2_kernel(char* out, int odt, const char* a, int adt, Idx ix, size_t n, int op) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        long long oa = 0;
        for (int d = ix.nd - 1; d >= 0; --d) {
            const int c = (int)(r % ix.shape[d]);
            r /= ix.shape[d];
            oa += c * ix.sa[d];
        }
        const float x = ld_f(a, adt, oa);
        st_f(out, odt, i, unary2_apply(op, x));
    }
}
__global__ void arange_kernel(char* out, int dt, size_t n, double start, double step) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double v = start + (double)i * step;
        if (is_int_dt(dt)) st_i(out, dt, i, (long long)v);
        else st_f(out, dt, i, (float)v);
    }
}
// out[o, j] = sum_k in[o, k, j] over a contiguous [outer, n, inner]; one thread per output, k ascending (deterministic)
__global__ void sum_axis_kernel(char* out, int odt, const char* in, int idt, size_t outer, int n, size_t inner) {
    const size_t total = outer * inner;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = i / inner, j = i % inner;
        if (is_int_dt(idt)) {
            long long acc = 0;
            for (int k = 0; k < n; ++k) acc += ld_i(in, idt, (o * n + k) * inner + j);
            st_i(out, odt, i, acc);
        } else {
            float acc = 0.f;
            for (int k = 0; k < n; ++k) acc += ld_f(in, idt, (o * n + k) * inner + j);
            st_f(out, odt, i, acc);
        }
    }
}
// max / min / mean over the middle axis of a contiguous [outer, n, inner] (mode 0 max, 1 min, 2 mean); NaN propagates in max / min
__global__ void reduce_axis_kernel(char* out, int odt, const char* in, int idt, size_t outer, int n, size_t inner, int mode) {
    const size_t total = outer * inner;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = i / inner, j = i % inner;
        if (is_int_dt(idt) && mode < 2) {
            long long acc = ld_i(in, idt, (o * n) * inner + j);
            for (int k = 1; k < n; ++k) {
                const long long v = ld_i(in, idt, (o * n + k) * inner + j);
                acc = mode == 0 ? (v > acc ? v : acc) : (v < acc ? v : acc);
            }
            st_i(out, odt, i, acc);
        } else {
            if (mode >= 3) {   // 3: all, 4: any (bool), 5: logsumexp (max-shifted, f32)
                float mx = -INFINITY;
                bool all = true, any = false;
                for (int k = 0; k < n; ++k) {
                    const float v = ld_f(in, idt, (o * n + k) * inner + j);
                    all &= v != 0.f; any |= v != 0.f; mx = fmaxf(mx, v);
                }
                if (mode == 5) {
                    float sm = 0.f;
                    for (int k = 0; k < n; ++k) sm += expf(ld_f(in, idt, (o * n + k) * inner + j) - mx);
                    st_f(out, odt, i, mx == -INFINITY ? -INFINITY : mx + logf(sm));
                } else {
                    st_f(out, odt, i, mode == 3 ? (float)all : (float)any);
                }
                continue;
            }
            float acc = mode == 2 ? 0.f : ld_f(in, idt, (o * n) * inner + j);
            bool nan = acc != acc;
            for (int k = mode == 2 ? 0 : 1; k < n; ++k) {
                const float v = ld_f(in, idt, (o * n + k) * inner + j);
                nan |= v != v;
                acc = mode == 2 ? acc + v : mode == 0 ? fmaxf(acc, v) : fminf(acc, v);
            }
            st_f(out, odt, i, mode == 2 ? acc / (float)n : (nan ? NAN : acc));
        }
    }
}
// out = cond ? x : y with all three broadcast to one shape
__global__ void where_kernel(char* out, int odt, const char* c, int cdt, const char* x, int xdt, const char* y, int ydt, Idx ixc, Idx ixy, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        long long oc = 0, ox = 0, oy = 0;
        for (int d = ixc.nd - 1; d >= 0; --d) {
            const int k = (int)(r % ixc.shape[d]);
            r /= ixc.shape[d];
            oc += k * ixc.sa[d];
            ox += k * ixc.sb[d];
            oy += k * ixy.sb[d];
        }
        const bool take = is_int_dt(cdt) ? ld_i(c, cdt, oc) != 0 : ld_f(c, cdt, oc) != 0.f;
        if (is_int_dt(odt)) st_i(out, odt, i, take ? ld_i(x, xdt, ox) : ld_i(y, ydt, oy));
        else st_f(out, odt, i, take ? ld_f(x, xdt, ox) : ld_f(y, ydt, oy));
    }
}
__global__ void fill_value_kernel(char* out, int odt, const char* v, int vdt, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (is_int_dt(odt) && is_int_dt(vdt)) st_i(out, odt, i, ld_i(v, vdt, 0));
        else st_f(out, odt, i, ld_f(v, vdt, 0));
    }
}
// stable ascending argsort along the middle axis of a contiguous [outer, n, inner]: the rank of element k is the number of
// elements that sort before it (smaller, or equal with a smaller index; NaN last) -- O(n^2) per line, every line independent;
// the routed-token counts this serves (N*k <= a few thousand) make that cheaper than a sort network's launches
__global__ void argsort_kernel(uint32_t* out, const char* in, int dt, size_t outer, int n, size_t inner) {
    const size_t total = outer * (size_t)n * inner;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = i % inner, k = (i / inner) % n, o = i / (inner * n);
        int rank = 0;
        if (is_int_dt(dt)) {
            const long long x = ld_i(in, dt, i);
            for (int q = 0; q < n; ++q) {
                const long long y = ld_i(in, dt, (o * n + q) * inner + j);
                rank += (y < x) || (y == x && (size_t)q < k);
            }
        } else {
            const float x = ld_f(in, dt, i);
            for (int q = 0; q < n; ++q) {
                const float y = ld_f(in, dt, (o * n + q) * inner + j);
                const bool before = (x != x) ? (y == y || (size_t)q < k) : (y < x || (y == x && (size_t)q < k));
                rank += before;
            }
        }
        out[(o * n + rank) * inner + j] = (uint32_t)k;
    }
}
// out[idx] = a[..., index at `axis` replaced by indices[idx], ...]; `ix.shape` is the output (= broadcast indices) shape,
// sa the strides of `a` (0 where a broadcasts), sb the strides of `indices`
template <int ES>
__global__ void take_along_kernel(char* out, const char* a, const char* ind, int idt, Idx ix, size_t n, int axis, int axis_len,
                                  long long a_axis_stride) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        long long oa = 0, ob = 0;
        for (int d = ix.nd - 1; d >= 0; --d) {
            const int c = (int)(r % ix.shape[d]);
            r /= ix.shape[d];
            if (d != axis) oa += c * ix.sa[d];
            ob += c * ix.sb[d];
        }
        long long t = ld_i(ind, idt, ob);
        if (t < 0) t += axis_len;
        t = t < 0 ? 0 : (t >= axis_len ? axis_len - 1 : t);
        const char* s = a + (oa + t * a_axis_stride) * ES;
        char* d = out + i * ES;
        if (ES == 1) *d = *s;
        else if (ES == 2) *(uint16_t*)d = *(const uint16_t*)s;
        else if (ES == 4) *(uint32_t*)d = *(const uint32_t*)s;
        else *(uint64_t*)d = *(const uint64_t*)s;
    }
}
// take along `axis` with an index array of any shape: out[pre..., idx..., post...] = a[pre..., ind[idx...], post...] over a
// contiguous a = [outer, n, inner] and contiguous indices (gather_sort: take_axis(x [N,1,d], order, 0), model.rs:212-213)
template <int ES>
__global__ void take_axis_kernel(char* out, const char* a, const char* ind, int idt, size_t outer, int n, size_t inner, size_t n_idx) {
    const size_t total = outer * n_idx * inner;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = i % inner, q = (i / inner) % n_idx, o = i / (inner * n_idx);
        long long t = ld_i(ind, idt, q);
        if (t < 0) t += n;
        t = t < 0 ? 0 : (t >= n ? n - 1 : t);
        const char* s = a + ((o * n + (size_t)t) * inner + j) * ES;
        char* d = out + i * ES;
        if (ES == 1) *d = *s;
        else if (ES == 2) *(uint16_t*)d = *(const uint16_t*)s;
        else if (ES == 4) *(uint32_t*)d = *(const uint32_t*)s;
        else *(uint64_t*)d = *(const uint64_t*)s;
    }
}
// Conv1d, channels-last like MLX: input [B, L, Cin], weight [Cout, Kw, Cin / groups], output [B, Lout, Cout]; f32 accumulation
// in (kw, ci) order.  One thread per output element: the Paraformer FSMN it serves is depthwise (groups == channels, Kw = 11)
__global__ void conv1d_kernel(char* out, const char* x, const char* w, int dt, int B, int L, int Cin, int Lout, int Cout, int Kw, int stride,
                              int padding, int dilation, int groups) {
    const size_t total = (size_t)B * Lout * Cout;
    const int cig = Cin / groups, cog = Cout / groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout), t = (int)((i / Cout) % Lout), b = (int)(i / ((size_t)Cout * Lout));
        const int g = co / cog;
        float acc = 0.f;
        for (int k = 0; k < Kw; ++k) {
            const int p = t * stride - padding + k * dilation;
            if (p < 0 || p >= L) continue;
            for (int ci = 0; ci < cig; ++ci)
                acc = fmaf(ld_f(x, dt, ((size_t)b * L + p) * Cin + g * cig + ci), ld_f(w, dt, ((size_t)co * Kw + k) * cig + ci), acc);
        }
        st_f(out, dt, i, acc);
    }
}

// ---------------------------------------------------------------- host helpers
int broadcast2(const Arr& a, const Arr& b, const char* name, std::vector<int>* shape, Idx* ix) {
    const int nd = (int)std::max(a.shape.size(), b.shape.size());
    shape->assign(nd, 1);
    if (fill_idx(*ix, *shape)) return 1;
    for (int i = 0; i < nd; ++i) {
        const int ia = i - (nd - (int)a.shape.size()), ib = i - (nd - (int)b.shape.size());
        const int da = ia >= 0 ? a.shape[ia] : 1, db = ib >= 0 ? b.shape[ib] : 1;
        OMX_REQUIRE(da == db || da == 1 || db == 1, "%s: shapes are not broadcastable (dim %d: %d vs %d)", name, i, da, db);
        (*shape)[i] = da == 1 ? db : da;
        ix->shape[i] = (*shape)[i];
        ix->sa[i] = (ia >= 0 && da != 1) ? (long long)a.strides[ia] : 0;
        ix->sb[i] = (ib >= 0 && db != 1) ? (long long)b.strides[ib] : 0;
    }
    return 0;
}
int binary2(mlx_array* res, const mlx_array ha, const mlx_array hb, int op, const char* name) {
    REQ_ARR(ha, name); REQ_ARR(hb, name);
    const Arr &a = *A(ha), &b = *A(hb);
    std::vector<int> shape;
    Idx ix;
    if (broadcast2(a, b, name, &shape, &ix)) return 1;
    const bool boolean = op <= B2_LAST_BOOL;
    mlx_dtype odt = boolean ? MLX_BOOL : promote(a.dt, b.dt);
    if (op == B2_LOGADDEXP && !is_float(odt)) odt = MLX_FLOAT32;
    NEW_OR_FAIL(r, shape, odt);
    const size_t n = r->size();
    if (n) {
        binary2_kernel<<<grid_for(n), 256, 0, g_stream>>>(r->ptr(), odt, a.ptr(), a.dt, b.ptr(), b.dt, ix, n, op, is_int_dt(a.dt) && is_int_dt(b.dt) && op != B2_LOGADDEXP);
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
int unary2(mlx_array* res, const mlx_array ha, int op, const char* name) {
    REQ_ARR(ha, name);
    const Arr& a = *A(ha);
    // abs / square / sign / floor / ceil / round keep an integer input's dtype in MLX; the float32 round trip is exact below 2^24
    const bool keeps = op == U2_ABS || op == U2_SQUARE || op == U2_SIGN || op == U2_FLOOR || op == U2_CEIL || op == U2_ROUND;
    const mlx_dtype odt = op >= U2_FIRST_PRED ? MLX_BOOL : (is_float(a.dt) || keeps) ? a.dt : MLX_FLOAT32;
    NEW_OR_FAIL(r, a.shape, odt);
    Idx ix;
    if (fill_idx(ix, a.shape)) { delete r; return 1; }
    for (int i = 0; i < ix.nd; ++i) ix.sa[i] = (long long)a.strides[i];
    const size_t n = r->size();
    if (n) {
        unary2_kernel<<<grid_for(n), 256, 0, g_stream>>>(r->ptr(), odt, a.ptr(), a.dt, ix, n, op);
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
}  // namespace
int take_axis_general(mlx_array* res, const Arr& a, const Arr& ind, int ax) {
    std::vector<int> shape(a.shape.begin(), a.shape.begin() + ax);
    shape.insert(shape.end(), ind.shape.begin(), ind.shape.end());
    shape.insert(shape.end(), a.shape.begin() + ax + 1, a.shape.end());
    NEW_OR_FAIL(r, shape, a.dt);
    Rec rec;                       // deferred like the row gather of a floating table (a QuantizedEmbedding takes rows of packed words)
    rec.a[0] = *r; rec.a[1] = a; rec.a[2] = ind; rec.na = 3;
    rec.i0 = ax;
    rec.run = [](Rec& q) -> int {
        const Arr &a = q.a[1], &ind = q.a[2];
        const int ax = q.i0;
        Contig ca, ci;
        if (ca.init(a) || ci.init(ind)) return 1;
        size_t outer = 1, inner = 1;
        for (int i = 0; i < ax; ++i) outer *= (size_t)a.shape[i];
        for (size_t i = ax + 1; i < a.shape.size(); ++i) inner *= (size_t)a.shape[i];
        const size_t n = q.a[0].size(), n_idx = ind.size();
        if (n) {
            switch (dsize(a.dt)) {
                case 1: take_axis_kernel<1><<<grid_for(n), 256, 0, g_stream>>>(q.a[0].ptr(), ca.a->ptr(), ci.a->ptr(), ind.dt, outer, a.shape[ax], inner, n_idx); break;
                case 2: take_axis_kernel<2><<<grid_for(n), 256, 0, g_stream>>>(q.a[0].ptr(), ca.a->ptr(), ci.a->ptr(), ind.dt, outer, a.shape[ax], inner, n_idx); break;
                case 4: take_axis_kernel<4><<<grid_for(n), 256, 0, g_stream>>>(q.a[0].ptr(), ca.a->ptr(), ci.a->ptr(), ind.dt, outer, a.shape[ax], inner, n_idx); break;
                default: take_axis_kernel<8><<<grid_for(n), 256, 0, g_stream>>>(q.a[0].ptr(), ca.a->ptr(), ci.a->ptr(), ind.dt, outer, a.shape[ax], inner, n_idx); break;
            }
            OMX_LAUNCH_CHECK();
        }
        return 0;
    };
    if (record(std::move(rec))) { delete r; return 1; }
    return assign(res, r);
}
namespace {
// [outer, n, inner] factorisation of a shape around `ax`
void around_axis(const std::vector<int>& shape, int ax, size_t* outer, int* n, size_t* inner) {
    *outer = 1; *inner = 1;
    for (int i = 0; i < ax; ++i) *outer *= (size_t)shape[i];
    for (size_t i = ax + 1; i < shape.size(); ++i) *inner *= (size_t)shape[i];
    *n = shape[ax];
}
Arr* view_with_shape(const Arr& s, const std::vector<int>& shape, const std::vector<size_t>& strides) {
    Arr* r = clone_handle(&s);
    r->shape = shape;
    r->strides = strides;
    return r;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- device.h / stream.h
mlx_device mlx_device_new(void) { return mlx_device{new Dev(g_default_dev)}; }
mlx_device mlx_device_new_type(mlx_device_type type, int index) { return mlx_device{new Dev{type, index}}; }
int mlx_device_free(mlx_device dev) { delete reinterpret_cast<Dev*>(dev.ctx); return 0; }
int mlx_device_set(mlx_device* dev, const mlx_device src) {
    OMX_REQUIRE(dev, "mlx_device_set: null destination");
    Dev* n = src.ctx ? new Dev(*reinterpret_cast<Dev*>(src.ctx)) : nullptr;
    delete reinterpret_cast<Dev*>(dev->ctx);
    dev->ctx = n;
    return 0;
}
bool mlx_device_equal(mlx_device lhs, mlx_device rhs) {
    if (!lhs.ctx || !rhs.ctx) return lhs.ctx == rhs.ctx;
    const Dev &a = *reinterpret_cast<Dev*>(lhs.ctx), &b = *reinterpret_cast<Dev*>(rhs.ctx);
    return a.type == b.type && a.index == b.index;
}
int mlx_device_get_index(int* index, mlx_device dev) {
    OMX_REQUIRE(index && dev.ctx, "mlx_device_get_index: null argument");
    *index = reinterpret_cast<Dev*>(dev.ctx)->index;
    return 0;
}
int mlx_device_get_type(mlx_device_type* type, mlx_device dev) {
    OMX_REQUIRE(type && dev.ctx, "mlx_device_get_type: null argument");
    *type = reinterpret_cast<Dev*>(dev.ctx)->type;
    return 0;
}
int mlx_device_tostring(mlx_string* str, mlx_device dev) {
    OMX_REQUIRE(str && dev.ctx, "mlx_device_tostring: null argument");
    const Dev& d = *reinterpret_cast<Dev*>(dev.ctx);
    char buf[64];
    snprintf(buf, sizeof buf, "Device(%s, %d)", d.type == MLX_GPU ? "gpu" : "cpu", d.index);
    delete reinterpret_cast<std::string*>(str->ctx);
    str->ctx = new std::string(buf);
    return 0;
}
int mlx_get_default_device(mlx_device* dev) {
    OMX_REQUIRE(dev, "mlx_get_default_device: null result");
    delete reinterpret_cast<Dev*>(dev->ctx);
    dev->ctx = new Dev(g_default_dev);
    return 0;
}
int mlx_set_default_device(mlx_device dev) {
    OMX_REQUIRE(dev.ctx, "mlx_set_default_device: empty device");
    OMX_REQUIRE(reinterpret_cast<Dev*>(dev.ctx)->type == MLX_GPU, "mlx_set_default_device: libomx_hip has no CPU backend (MI355X only)");
    g_default_dev = *reinterpret_cast<Dev*>(dev.ctx);
    return 0;
}
mlx_stream mlx_stream_new_device(mlx_device dev) {
    if (dev.ctx && reinterpret_cast<Dev*>(dev.ctx)->type == MLX_CPU) {
        set_error("mlx_stream_new_device: libomx_hip has no CPU backend (MI355X only)");
        return mlx_stream{nullptr};
    }
    return mlx_stream{new Str{false}};
}
int mlx_stream_set(mlx_stream* stream, const mlx_stream src) {
    OMX_REQUIRE(stream, "mlx_stream_set: null destination");
    Str* n = src.ctx ? new Str(*reinterpret_cast<Str*>(src.ctx)) : nullptr;
    delete reinterpret_cast<Str*>(stream->ctx);
    stream->ctx = n;
    return 0;
}
int mlx_stream_tostring(mlx_string* str, mlx_stream) {
    OMX_REQUIRE(str, "mlx_stream_tostring: null result");
    delete reinterpret_cast<std::string*>(str->ctx);
    str->ctx = new std::string("Stream(Device(gpu, 0), 0)");
    return 0;
}
int mlx_stream_get_device(mlx_device* dev, mlx_stream) {
    OMX_REQUIRE(dev, "mlx_stream_get_device: null result");
    delete reinterpret_cast<Dev*>(dev->ctx);
    dev->ctx = new Dev{MLX_GPU, 0};
    return 0;
}
int mlx_stream_get_index(int* index, mlx_stream) {
    OMX_REQUIRE(index, "mlx_stream_get_index: null result");
    *index = 0;   // one in-order stream behind every handle
    return 0;
}
int mlx_get_default_stream(mlx_stream* stream, mlx_device dev) {
    OMX_REQUIRE(stream, "mlx_get_default_stream: null result");
    OMX_REQUIRE(!dev.ctx || reinterpret_cast<Dev*>(dev.ctx)->type == MLX_GPU, "mlx_get_default_stream: libomx_hip has no CPU backend (MI355X only)");
    delete reinterpret_cast<Str*>(stream->ctx);
    stream->ctx = new Str{false};
    return 0;
}
int mlx_set_default_stream(mlx_stream stream) {
    OMX_REQUIRE(stream.ctx, "mlx_set_default_stream: empty stream");
    return 0;   // every stream handle is the same device stream
}

// ---------------------------------------------------------------- string.h / vector.h (strings) / array.h:62
mlx_string mlx_string_new(void) { return mlx_string{new std::string()}; }
mlx_string mlx_string_new_data(const char* str) { return mlx_string{new std::string(str ? str : "")}; }
int mlx_string_set(mlx_string* str, const mlx_string src) {
    OMX_REQUIRE(str, "mlx_string_set: null destination");
    std::string* n = src.ctx ? new std::string(*reinterpret_cast<std::string*>(src.ctx)) : nullptr;
    delete reinterpret_cast<std::string*>(str->ctx);
    str->ctx = n;
    return 0;
}
const char* mlx_string_data(mlx_string str) { return str.ctx ? reinterpret_cast<std::string*>(str.ctx)->c_str() : nullptr; }
int mlx_string_free(mlx_string str) { delete reinterpret_cast<std::string*>(str.ctx); return 0; }

typedef std::vector<std::string> VecStr;
mlx_vector_string mlx_vector_string_new(void) { return mlx_vector_string{new VecStr()}; }
int mlx_vector_string_free(mlx_vector_string vec) { delete reinterpret_cast<VecStr*>(vec.ctx); return 0; }
int mlx_vector_string_set(mlx_vector_string* vec, const mlx_vector_string src) {
    OMX_REQUIRE(vec, "mlx_vector_string_set: null destination");
    VecStr* n = src.ctx ? new VecStr(*reinterpret_cast<VecStr*>(src.ctx)) : nullptr;
    delete reinterpret_cast<VecStr*>(vec->ctx);
    vec->ctx = n;
    return 0;
}
mlx_vector_string mlx_vector_string_new_data(const char** data, size_t size) {
    VecStr* v = new VecStr();
    for (size_t i = 0; i < size; ++i) v->emplace_back(data[i] ? data[i] : "");
    return mlx_vector_string{v};
}
mlx_vector_string mlx_vector_string_new_value(const char* val) { return mlx_vector_string_new_data(&val, 1); }
int mlx_vector_string_set_data(mlx_vector_string* vec, const char** data, size_t size) {
    OMX_REQUIRE(vec, "mlx_vector_string_set_data: null destination");
    delete reinterpret_cast<VecStr*>(vec->ctx);
    vec->ctx = mlx_vector_string_new_data(data, size).ctx;
    return 0;
}
int mlx_vector_string_set_value(mlx_vector_string* vec, const char* val) { return mlx_vector_string_set_data(vec, &val, 1); }
int mlx_vector_string_append_data(mlx_vector_string vec, const char** data, size_t size) {
    OMX_REQUIRE(vec.ctx, "mlx_vector_string_append_data: empty vector");
    for (size_t i = 0; i < size; ++i) reinterpret_cast<VecStr*>(vec.ctx)->emplace_back(data[i] ? data[i] : "");
    return 0;
}
int mlx_vector_string_append_value(mlx_vector_string vec, const char* val) { return mlx_vector_string_append_data(vec, &val, 1); }
size_t mlx_vector_string_size(mlx_vector_string vec) { return vec.ctx ? reinterpret_cast<VecStr*>(vec.ctx)->size() : 0; }
int mlx_vector_string_get(char** res, const mlx_vector_string vec, size_t idx) {
    OMX_REQUIRE(res && vec.ctx && idx < reinterpret_cast<VecStr*>(vec.ctx)->size(), "mlx_vector_string_get: index %zu out of range", idx);
    *res = const_cast<char*>((*reinterpret_cast<VecStr*>(vec.ctx))[idx].c_str());   // borrowed, like the reference (vector.cpp)
    return 0;
}
int mlx_vector_array_set(mlx_vector_array* vec, const mlx_vector_array src) {
    OMX_REQUIRE(vec, "mlx_vector_array_set: null destination");
    Vec* n = nullptr;
    if (src.ctx) {
        n = new Vec();
        for (Arr* a : reinterpret_cast<Vec*>(src.ctx)->v) n->v.push_back(clone_handle(a));
    }
    delete reinterpret_cast<Vec*>(vec->ctx);
    vec->ctx = n;
    return 0;
}
mlx_vector_array mlx_vector_array_new_data(const mlx_array* data, size_t size) {
    Vec* v = new Vec();
    for (size_t i = 0; i < size; ++i)
        if (data[i].ctx) v->v.push_back(clone_handle(A(data[i])));
    return mlx_vector_array{v};
}
mlx_vector_array mlx_vector_array_new_value(const mlx_array val) { return mlx_vector_array_new_data(&val, 1); }

int mlx_array_tostring(mlx_string* str, const mlx_array arr) {
    OMX_REQUIRE(str, "mlx_array_tostring: null result");
    REQ_ARR(arr, "mlx_array_tostring");
    const Arr& a = *A(arr);
    static const char* names[] = {"bool", "uint8", "uint16", "uint32", "uint64", "int8", "int16", "int32", "int64", "float16", "float32",
                                  "float64", "bfloat16", "complex64"};
    std::string out = "array(";
    const size_t n = a.size(), show = n < 16 ? n : 16;
    Contig c;
    if (c.init(a)) return 1;
    std::vector<uint8_t> host(show * dsize(a.dt) + 8);
    if (show) {
        OMX_HIP_CHECK(hipMemcpyAsync(host.data(), c.a->ptr(), show * dsize(a.dt), hipMemcpyDeviceToHost, g_stream));
        OMX_HIP_CHECK(hipStreamSynchronize(g_stream));
    }
    if (!a.shape.empty()) out += "[";
    char buf[48];
    for (size_t i = 0; i < show; ++i) {
        double v = 0;
        switch (a.dt) {
            case MLX_FLOAT32: v = ((float*)host.data())[i]; break;
            case MLX_BFLOAT16: v = omx::bf16_to_f32(((uint16_t*)host.data())[i]); break;
            case MLX_FLOAT16: v = (double)(float)((_Float16*)host.data())[i]; break;
            case MLX_INT32: v = ((int32_t*)host.data())[i]; break;
            case MLX_UINT32: v = ((uint32_t*)host.data())[i]; break;
            case MLX_BOOL: case MLX_UINT8: v = host[i]; break;
            default: v = 0; break;
        }
        if (a.dt == MLX_BOOL) snprintf(buf, sizeof buf, "%s", v != 0 ? "True" : "False");
        else snprintf(buf, sizeof buf, "%g", v);
        out += buf;
        if (i + 1 < show) out += ", ";
    }
    if (show < n) out += ", ...";
    if (!a.shape.empty()) out += "]";
    out += std::string(", dtype=") + names[(int)a.dt] + ")";
    delete reinterpret_cast<std::string*>(str->ctx);
    str->ctx = new std::string(out);
    return 0;
}

// ---------------------------------------------------------------- map.h
mlx_map_string_to_array mlx_map_string_to_array_new(void) { return mlx_map_string_to_array{new MapArr()}; }
int mlx_map_string_to_array_free(mlx_map_string_to_array map) { delete reinterpret_cast<MapArr*>(map.ctx); return 0; }
int mlx_map_string_to_array_set(mlx_map_string_to_array* map, const mlx_map_string_to_array src) {
    OMX_REQUIRE(map, "mlx_map_string_to_array_set: null destination");
    MapArr* n = nullptr;
    if (src.ctx) {
        n = new MapArr();
        for (auto& kv : reinterpret_cast<MapArr*>(src.ctx)->m) n->m[kv.first] = clone_handle(kv.second);
    }
    delete reinterpret_cast<MapArr*>(map->ctx);
    map->ctx = n;
    return 0;
}
int mlx_map_string_to_array_insert(mlx_map_string_to_array map, const char* key, const mlx_array value) {
    OMX_REQUIRE(map.ctx && key && value.ctx, "mlx_map_string_to_array_insert: null argument");
    Arr*& slot = reinterpret_cast<MapArr*>(map.ctx)->m[key];
    delete slot;
    slot = clone_handle(A(value));
    return 0;
}
int mlx_map_string_to_array_get(mlx_array* value, const mlx_map_string_to_array map, const char* key) {
    OMX_REQUIRE(value && map.ctx && key, "mlx_map_string_to_array_get: null argument");
    auto& m = reinterpret_cast<MapArr*>(map.ctx)->m;
    auto it = m.find(key);
    if (it == m.end()) return 2;   // map.cpp: "not found" is status 2, not an error
    return assign(value, clone_handle(it->second));
}
struct MapArrIt { std::map<std::string, Arr*>::iterator it; };
mlx_map_string_to_array_iterator mlx_map_string_to_array_iterator_new(mlx_map_string_to_array map) {
    if (!map.ctx) return mlx_map_string_to_array_iterator{nullptr, nullptr};
    return mlx_map_string_to_array_iterator{new MapArrIt{reinterpret_cast<MapArr*>(map.ctx)->m.begin()}, map.ctx};
}
int mlx_map_string_to_array_iterator_free(mlx_map_string_to_array_iterator it) { delete reinterpret_cast<MapArrIt*>(it.ctx); return 0; }
int mlx_map_string_to_array_iterator_next(const char** key, mlx_array* value, mlx_map_string_to_array_iterator it) {
    OMX_REQUIRE(key && value && it.ctx && it.map_ctx, "mlx_map_string_to_array_iterator_next: null argument");
    MapArrIt* s = reinterpret_cast<MapArrIt*>(it.ctx);
    if (s->it == reinterpret_cast<MapArr*>(it.map_ctx)->m.end()) return 2;
    *key = s->it->first.c_str();
    if (assign(value, clone_handle(s->it->second))) return 1;
    ++s->it;
    return 0;
}
mlx_map_string_to_string mlx_map_string_to_string_new(void) { return mlx_map_string_to_string{new MapStr()}; }
int mlx_map_string_to_string_free(mlx_map_string_to_string map) { delete reinterpret_cast<MapStr*>(map.ctx); return 0; }
int mlx_map_string_to_string_set(mlx_map_string_to_string* map, const mlx_map_string_to_string src) {
    OMX_REQUIRE(map, "mlx_map_string_to_string_set: null destination");
    MapStr* n = src.ctx ? new MapStr(*reinterpret_cast<MapStr*>(src.ctx)) : nullptr;
    delete reinterpret_cast<MapStr*>(map->ctx);
    map->ctx = n;
    return 0;
}
int mlx_map_string_to_string_insert(mlx_map_string_to_string map, const char* key, const char* value) {
    OMX_REQUIRE(map.ctx && key && value, "mlx_map_string_to_string_insert: null argument");
    reinterpret_cast<MapStr*>(map.ctx)->m[key] = value;
    return 0;
}
int mlx_map_string_to_string_get(const char** value, const mlx_map_string_to_string map, const char* key) {
    OMX_REQUIRE(value && map.ctx && key, "mlx_map_string_to_string_get: null argument");
    auto& m = reinterpret_cast<MapStr*>(map.ctx)->m;
    auto it = m.find(key);
    if (it == m.end()) return 2;
    *value = it->second.c_str();
    return 0;
}
struct MapStrIt { std::map<std::string, std::string>::iterator it; };
mlx_map_string_to_string_iterator mlx_map_string_to_string_iterator_new(mlx_map_string_to_string map) {
    if (!map.ctx) return mlx_map_string_to_string_iterator{nullptr, nullptr};
    return mlx_map_string_to_string_iterator{new MapStrIt{reinterpret_cast<MapStr*>(map.ctx)->m.begin()}, map.ctx};
}
int mlx_map_string_to_string_iterator_free(mlx_map_string_to_string_iterator it) { delete reinterpret_cast<MapStrIt*>(it.ctx); return 0; }
int mlx_map_string_to_string_iterator_next(const char** key, const char** value, mlx_map_string_to_string_iterator it) {
    OMX_REQUIRE(key && value && it.ctx && it.map_ctx, "mlx_map_string_to_string_iterator_next: null argument");
    MapStrIt* s = reinterpret_cast<MapStrIt*>(it.ctx);
    if (s->it == reinterpret_cast<MapStr*>(it.map_ctx)->m.end()) return 2;
    *key = s->it->first.c_str();
    *value = s->it->second.c_str();
    ++s->it;
    return 0;
}

// ---------------------------------------------------------------- io.h:40-44
// safetensors: 8-byte little-endian header length, JSON header {name: {dtype, shape, data_offsets}, "__metadata__": {...}}, raw
// tensor bytes.  The JSON subset a safetensors header uses is parsed here (objects, strings, integer arrays).
namespace {
struct JsonCur {
    const char* p; const char* e;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < e && *p == c) { ++p; return true; } return false; }
    bool str(std::string* out) {
        ws();
        if (p >= e || *p != '"') return false;
        ++p;
        out->clear();
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) {
                ++p;
                switch (*p) {
                    case 'n': out->push_back('\n'); break;
                    case 't': out->push_back('\t'); break;
                    case 'u': out->push_back('?'); p += 4; break;   // non-ASCII key: kept recognisable, not decoded
                    default: out->push_back(*p); break;
                }
                ++p;
            } else {
                out->push_back(*p++);
            }
        }
        if (p >= e) return false;
        ++p;
        return true;
    }
    bool num(long long* v) {
        ws();
        char* end = nullptr;
        *v = strtoll(p, &end, 10);
        if (end == p) return false;
        p = end;
        return true;
    }
};
}  // namespace
int mlx_load_safetensors(mlx_map_string_to_array* res_0, mlx_map_string_to_string* res_1, const char* file, const mlx_stream) {
    OMX_REQUIRE(res_0 && res_1 && file, "mlx_load_safetensors: null argument");
    std::ifstream f(file, std::ios::binary);
    OMX_REQUIRE(f.good(), "mlx_load_safetensors: cannot open %s", file);
    uint8_t lenb[8];
    f.read((char*)lenb, 8);
    OMX_REQUIRE(f.gcount() == 8, "mlx_load_safetensors: %s is not a safetensors file (short header)", file);
    uint64_t hlen = 0;
    for (int i = 7; i >= 0; --i) hlen = (hlen << 8) | lenb[i];
    OMX_REQUIRE(hlen > 0 && hlen < (100ull << 20), "mlx_load_safetensors: %s: implausible header length %llu", file, (unsigned long long)hlen);
    std::string header(hlen, '\0');
    f.read(&header[0], (std::streamsize)hlen);
    OMX_REQUIRE((uint64_t)f.gcount() == hlen, "mlx_load_safetensors: %s: truncated header", file);
    MapArr* arrays = new MapArr();
    MapStr* meta = new MapStr();
    auto fail = [&](const char* what) { delete arrays; delete meta; return set_error("mlx_load_safetensors: %s: %s", file, what); };
    JsonCur c{header.data(), header.data() + header.size()};
    if (!c.eat('{')) return fail("header is not a JSON object");
    std::vector<char> blob;
    while (true) {
        std::string name;
        if (c.eat('}')) break;
        if (!c.str(&name) || !c.eat(':') || !c.eat('{')) return fail("malformed header entry");
        if (name == "__metadata__") {
            while (!c.eat('}')) {
                std::string k, v;
                if (!c.str(&k) || !c.eat(':') || !c.str(&v)) return fail("malformed __metadata__");
                meta->m[k] = v;
                c.eat(',');
            }
        } else {
            std::string dtype;
            std::vector<int> shape;
            long long off0 = -1, off1 = -1;
            while (!c.eat('}')) {
                std::string k;
                if (!c.str(&k) || !c.eat(':')) return fail("malformed tensor entry");
                if (k == "dtype") {
                    if (!c.str(&dtype)) return fail("malformed dtype");
                } else if (k == "shape" || k == "data_offsets") {
                    if (!c.eat('[')) return fail("malformed array");
                    std::vector<long long> vals;
                    while (!c.eat(']')) {
                        long long v;
                        if (!c.num(&v)) return fail("malformed number");
                        vals.push_back(v);
                        c.eat(',');
                    }
                    if (k == "shape") for (long long v : vals) shape.push_back((int)v);
                    else if (vals.size() == 2) { off0 = vals[0]; off1 = vals[1]; }
                } else {
                    return fail("unknown tensor field");
                }
                c.eat(',');
            }
            static const std::map<std::string, mlx_dtype> kinds = {
                {"F32", MLX_FLOAT32}, {"F16", MLX_FLOAT16}, {"BF16", MLX_BFLOAT16}, {"U32", MLX_UINT32}, {"I32", MLX_INT32}, {"U8", MLX_UINT8},
                {"I8", MLX_INT8}, {"BOOL", MLX_BOOL}, {"U16", MLX_UINT16}, {"I16", MLX_INT16}, {"I64", MLX_INT64}, {"U64", MLX_UINT64}, {"F64", MLX_FLOAT64}};
            auto kd = kinds.find(dtype);
            if (kd == kinds.end()) return fail("unsupported tensor dtype");
            Arr* a = new_arr(shape, kd->second);
            if (!a) return fail("out of device memory");
            const size_t bytes = a->size() * dsize(kd->second);
            if (off0 < 0 || off1 < off0 || (size_t)(off1 - off0) != bytes) { delete a; return fail("data_offsets disagree with shape and dtype"); }
            if (bytes) {
                blob.resize(bytes);
                f.seekg((std::streamoff)(8 + hlen + (uint64_t)off0));
                f.read(blob.data(), (std::streamsize)bytes);
                if ((size_t)f.gcount() != bytes) { delete a; return fail("truncated tensor data"); }
                if (hipMemcpyAsync(a->ptr(), blob.data(), bytes, hipMemcpyHostToDevice, g_stream) != hipSuccess ||
                    hipStreamSynchronize(g_stream) != hipSuccess) { delete a; return fail("host to device copy failed"); }
            }
            Arr*& slot = arrays->m[name];
            delete slot;
            slot = a;
        }
        c.eat(',');
    }
    delete reinterpret_cast<MapArr*>(res_0->ctx);
    delete reinterpret_cast<MapStr*>(res_1->ctx);
    res_0->ctx = arrays;
    res_1->ctx = meta;
    return 0;
}

// ---------------------------------------------------------------- closure.h / compile.h
mlx_closure mlx_closure_new(void) { return mlx_closure{nullptr}; }
int mlx_closure_free(mlx_closure cls) { delete reinterpret_cast<Closure*>(cls.ctx); return 0; }
mlx_closure mlx_closure_new_func(int (*fun)(mlx_vector_array*, const mlx_vector_array)) {
    Closure* c = new Closure();
    c->fn = fun;
    return mlx_closure{c};
}
mlx_closure mlx_closure_new_func_payload(int (*fun)(mlx_vector_array*, const mlx_vector_array, void*), void* payload, void (*dtor)(void*)) {
    Closure* c = new Closure();
    c->payload = std::shared_ptr<void>(payload, [dtor](void* p) { if (dtor) dtor(p); });
    void* raw = payload;
    c->fn = [fun, raw](mlx_vector_array* res, const mlx_vector_array in) { return fun(res, in, raw); };
    return mlx_closure{c};
}
mlx_closure mlx_closure_new_unary(int (*fun)(mlx_array*, const mlx_array)) {
    Closure* c = new Closure();
    c->fn = [fun](mlx_vector_array* res, const mlx_vector_array in) -> int {
        OMX_REQUIRE(res && in.ctx && reinterpret_cast<Vec*>(in.ctx)->v.size() == 1, "unary closure: expected exactly one input");
        mlx_array x{reinterpret_cast<Vec*>(in.ctx)->v[0]}, y{nullptr};
        if (fun(&y, x)) { delete A(y); return 1; }
        Vec* out = new Vec();
        out->v.push_back(A(y));
        delete reinterpret_cast<Vec*>(res->ctx);
        res->ctx = out;
        return 0;
    };
    return mlx_closure{c};
}
int mlx_closure_set(mlx_closure* cls, const mlx_closure src) {
    OMX_REQUIRE(cls, "mlx_closure_set: null destination");
    Closure* n = src.ctx ? new Closure(*reinterpret_cast<Closure*>(src.ctx)) : nullptr;
    delete reinterpret_cast<Closure*>(cls->ctx);
    cls->ctx = n;
    return 0;
}
int mlx_closure_apply(mlx_vector_array* res, mlx_closure cls, const mlx_vector_array input) {
    OMX_REQUIRE(res && cls.ctx && reinterpret_cast<Closure*>(cls.ctx)->fn, "mlx_closure_apply: empty closure");
    return reinterpret_cast<Closure*>(cls.ctx)->fn(res, input);
}
// compile (compile.h:37-48): there is no graph to trace -- every op already runs as its own tuned launch -- so the "compiled"
// closure is the closure itself; the cache entry points have nothing to clear
int mlx_detail_compile(mlx_closure* res, const mlx_closure fun, uintptr_t, bool, const uint64_t*, size_t) { return mlx_closure_set(res, fun); }
int mlx_detail_compile_clear_cache(void) { return 0; }
int mlx_detail_compile_erase(uintptr_t) { return 0; }
int mlx_disable_compile(void) { return 0; }
int mlx_enable_compile(void) { return 0; }

// ---------------------------------------------------------------- ops.h glue
int mlx_greater(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_GT, "mlx_greater"); }
int mlx_greater_equal(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_GE, "mlx_greater_equal"); }
int mlx_less(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_LT, "mlx_less"); }
int mlx_less_equal(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_LE, "mlx_less_equal"); }
int mlx_equal(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_EQ, "mlx_equal"); }
int mlx_logical_and(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_AND, "mlx_logical_and"); }
int mlx_maximum(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_MAX, "mlx_maximum"); }
int mlx_minimum(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_MIN, "mlx_minimum"); }
int mlx_floor_divide(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_FLOORDIV, "mlx_floor_divide"); }
int mlx_cos(mlx_array* res, const mlx_array a, const mlx_stream) { return unary2(res, a, U2_COS, "mlx_cos"); }
int mlx_sin(mlx_array* res, const mlx_array a, const mlx_stream) { return unary2(res, a, U2_SIN, "mlx_sin"); }
// the rest of ops.h's elementwise math (round 4; not on the four callers' path, but `mlx-rs` names them)
#define OMX_UNARY2(NAME, OP) int NAME(mlx_array* res, const mlx_array a, const mlx_stream) { return unary2(res, a, OP, #NAME); }
OMX_UNARY2(mlx_abs, U2_ABS) OMX_UNARY2(mlx_sqrt, U2_SQRT) OMX_UNARY2(mlx_rsqrt, U2_RSQRT) OMX_UNARY2(mlx_square, U2_SQUARE)
OMX_UNARY2(mlx_log, U2_LOG) OMX_UNARY2(mlx_log2, U2_LOG2) OMX_UNARY2(mlx_log10, U2_LOG10) OMX_UNARY2(mlx_log1p, U2_LOG1P)
OMX_UNARY2(mlx_expm1, U2_EXPM1) OMX_UNARY2(mlx_tanh, U2_TANH) OMX_UNARY2(mlx_sinh, U2_SINH) OMX_UNARY2(mlx_cosh, U2_COSH)
OMX_UNARY2(mlx_tan, U2_TAN) OMX_UNARY2(mlx_arcsin, U2_ARCSIN) OMX_UNARY2(mlx_arccos, U2_ARCCOS) OMX_UNARY2(mlx_arctan, U2_ARCTAN)
OMX_UNARY2(mlx_arcsinh, U2_ARCSINH) OMX_UNARY2(mlx_arccosh, U2_ARCCOSH) OMX_UNARY2(mlx_arctanh, U2_ARCTANH) OMX_UNARY2(mlx_erf, U2_ERF)
OMX_UNARY2(mlx_reciprocal, U2_RECIP) OMX_UNARY2(mlx_floor, U2_FLOOR) OMX_UNARY2(mlx_ceil, U2_CEIL) OMX_UNARY2(mlx_sign, U2_SIGN)
OMX_UNARY2(mlx_isnan, U2_ISNAN) OMX_UNARY2(mlx_isinf, U2_ISINF) OMX_UNARY2(mlx_isfinite, U2_ISFINITE) OMX_UNARY2(mlx_isposinf, U2_ISPOSINF)
OMX_UNARY2(mlx_isneginf, U2_ISNEGINF) OMX_UNARY2(mlx_logical_not, U2_NOT)
#undef OMX_UNARY2
int mlx_round(mlx_array* res, const mlx_array a, int decimals, const mlx_stream) {
    OMX_REQUIRE(decimals == 0, "mlx_round: only decimals = 0 is supported (got %d)", decimals);
    return unary2(res, a, U2_ROUND, "mlx_round");
}
int mlx_not_equal(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_NE, "mlx_not_equal"); }
int mlx_logical_or(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_OR, "mlx_logical_or"); }
int mlx_power(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_POW, "mlx_power"); }
int mlx_remainder(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_REM, "mlx_remainder"); }
int mlx_logaddexp(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary2(res, a, b, B2_LOGADDEXP, "mlx_logaddexp"); }

int mlx_arange(mlx_array* res, double start, double stop, double step, mlx_dtype dtype, const mlx_stream) {
    OMX_REQUIRE(step != 0.0 && step == step && start == start && stop == stop, "mlx_arange: step must be non-zero and the bounds finite");
    const double cnt = std::ceil((stop - start) / step);
    OMX_REQUIRE(cnt < 2147483647.0, "mlx_arange: too many elements");
    std::vector<int> shape = {cnt > 0 ? (int)cnt : 0};
    NEW_OR_FAIL(r, shape, dtype);
    if (r->size()) {
        arange_kernel<<<grid_for(r->size()), 256, 0, g_stream>>>(r->ptr(), dtype, r->size(), start, step);
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
int mlx_sum_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream) {
    REQ_ARR(a, "mlx_sum_axis");
    const Arr& s = *A(a);
    int ax;
    if (norm_axis(axis, (int)s.shape.size(), "mlx_sum_axis", &ax)) return 1;
    Contig c;
    if (c.init(s)) return 1;
    size_t outer, inner; int n;
    around_axis(s.shape, ax, &outer, &n, &inner);
    std::vector<int> shape = s.shape;
    if (keepdims) shape[ax] = 1; else shape.erase(shape.begin() + ax);
    const mlx_dtype odt = s.dt == MLX_BOOL ? MLX_INT32 : s.dt;   // sums of booleans count
    NEW_OR_FAIL(r, shape, odt);
    if (r->size()) {
        sum_axis_kernel<<<grid_for(r->size()), 256, 0, g_stream>>>(r->ptr(), odt, c.a->ptr(), s.dt, outer, n, inner);
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
static int reduce_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, int mode, const char* name) {
    REQ_ARR(a, name);
    const Arr& s = *A(a);
    int ax;
    if (norm_axis(axis, (int)s.shape.size(), name, &ax)) return 1;
    OMX_REQUIRE(s.shape[ax] > 0, "%s: empty reduction axis", name);
    Contig c;
    if (c.init(s)) return 1;
    size_t outer, inner; int n;
    around_axis(s.shape, ax, &outer, &n, &inner);
    std::vector<int> shape = s.shape;
    if (keepdims) shape[ax] = 1; else shape.erase(shape.begin() + ax);
    const mlx_dtype odt = (mode == 3 || mode == 4) ? MLX_BOOL : ((mode == 2 || mode == 5) && !is_float(s.dt)) ? MLX_FLOAT32 : s.dt;    // the mean of integers is float32 in MLX
    NEW_OR_FAIL(r, shape, odt);
    if (r->size()) {
        reduce_axis_kernel<<<grid_for(r->size()), 256, 0, g_stream>>>(r->ptr(), odt, c.a->ptr(), s.dt, outer, n, inner, mode);
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
int mlx_max_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream) { return reduce_axis(res, a, axis, keepdims, 0, "mlx_max_axis"); }
int mlx_min_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream) { return reduce_axis(res, a, axis, keepdims, 1, "mlx_min_axis"); }
int mlx_mean_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream) { return reduce_axis(res, a, axis, keepdims, 2, "mlx_mean_axis"); }
int mlx_all_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream) { return reduce_axis(res, a, axis, keepdims, 3, "mlx_all_axis"); }
int mlx_any_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream) { return reduce_axis(res, a, axis, keepdims, 4, "mlx_any_axis"); }
int mlx_logsumexp_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream) { return reduce_axis(res, a, axis, keepdims, 5, "mlx_logsumexp_axis"); }
// the whole-array forms: the same reduction over the flattened array (keepdims: every axis kept as 1)
static int reduce_all(mlx_array* res, const mlx_array a, bool keepdims, int mode, const char* name, const mlx_stream s) {
    REQ_ARR(a, name);
    const int nd = (int)A(a)->shape.size();
    mlx_array flat = mlx_array_new(), red = mlx_array_new();
    int rc = nd <= 1 ? mlx_array_set(&flat, a) : mlx_flatten(&flat, a, 0, -1, s);
    if (!rc && nd == 0) rc = mlx_reshape(&flat, a, std::vector<int>{1}.data(), 1, s);
    if (!rc) rc = mode == 6 ? mlx_sum_axis(&red, flat, 0, false, s) : reduce_axis(&red, flat, 0, false, mode, name);
    if (!rc && keepdims && nd > 0) {
        std::vector<int> ones((size_t)nd, 1);
        rc = mlx_reshape(res, red, ones.data(), ones.size(), s);
    } else if (!rc) {
        rc = mlx_array_set(res, red);
    }
    mlx_array_free(flat); mlx_array_free(red);
    return rc;
}
int mlx_max(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s) { return reduce_all(res, a, keepdims, 0, "mlx_max", s); }
int mlx_min(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s) { return reduce_all(res, a, keepdims, 1, "mlx_min", s); }
int mlx_mean(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s) { return reduce_all(res, a, keepdims, 2, "mlx_mean", s); }
int mlx_all(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s) { return reduce_all(res, a, keepdims, 3, "mlx_all", s); }
int mlx_any(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s) { return reduce_all(res, a, keepdims, 4, "mlx_any", s); }
int mlx_logsumexp(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s) { return reduce_all(res, a, keepdims, 5, "mlx_logsumexp", s); }
int mlx_sum(mlx_array* res, const mlx_array a, bool keepdims, const mlx_stream s) { return reduce_all(res, a, keepdims, 6, "mlx_sum", s); }
int mlx_stop_gradient(mlx_array* res, const mlx_array a, const mlx_stream) { REQ_ARR(a, "mlx_stop_gradient"); return mlx_array_set(res, a); }   // inference only: the identity
int mlx_sort_axis(mlx_array* res, const mlx_array a, int axis, const mlx_stream s) {   // values in the order of the stable argsort
    REQ_ARR(a, "mlx_sort_axis");
    mlx_array idx = mlx_array_new();
    int rc = mlx_argsort_axis(&idx, a, axis, s);
    if (!rc) rc = mlx_take_along_axis(res, a, idx, axis, s);
    mlx_array_free(idx);
    return rc;
}
int mlx_sort(mlx_array* res, const mlx_array a, const mlx_stream s) {
    REQ_ARR(a, "mlx_sort");
    mlx_array flat = mlx_array_new();
    int rc = A(a)->shape.size() <= 1 ? mlx_array_set(&flat, a) : mlx_flatten(&flat, a, 0, -1, s);
    if (!rc) rc = mlx_sort_axis(res, flat, 0, s);
    mlx_array_free(flat);
    return rc;
}
int mlx_broadcast_to(mlx_array* res, const mlx_array a, const int* shape, size_t shape_num, const mlx_stream) {   // a view: stride 0 along the broadcast axes
    REQ_ARR(a, "mlx_broadcast_to");
    const Arr& s = *A(a);
    OMX_REQUIRE(shape_num >= s.shape.size(), "mlx_broadcast_to: cannot broadcast %zu dimensions to %zu", s.shape.size(), shape_num);
    Arr* r = new Arr(s);
    r->host.clear();
    r->shape.assign(shape, shape + shape_num);
    r->strides.assign(shape_num, 0);
    const int lead = (int)shape_num - (int)s.shape.size();
    for (int i = 0; i < (int)shape_num; ++i) {
        const int is = i - lead;
        const int d = is >= 0 ? s.shape[is] : 1;
        if (d != shape[i] && d != 1) { delete r; return set_error("mlx_broadcast_to: dimension %d of size %d does not broadcast to %d", i, d, shape[i]); }
        r->strides[i] = (is >= 0 && d != 1) ? s.strides[is] : 0;
    }
    return assign(res, r);
}
int mlx_concatenate(mlx_array* res, const mlx_vector_array arrays, const mlx_stream s) { return mlx_concatenate_axis(res, arrays, 0, s); }
// axis permutations as views (ops.h mlx_swapaxes / mlx_moveaxis -> the transpose view of mlxc.hip)
int mlx_swapaxes(mlx_array* res, const mlx_array a, int axis1, int axis2, const mlx_stream s) {
    REQ_ARR(a, "mlx_swapaxes");
    const int nd = (int)A(a)->shape.size();
    int a1, a2;
    if (norm_axis(axis1, nd, "mlx_swapaxes", &a1) || norm_axis(axis2, nd, "mlx_swapaxes", &a2)) return 1;
    std::vector<int> axes(nd);
    for (int i = 0; i < nd; ++i) axes[i] = i;
    std::swap(axes[a1], axes[a2]);
    return mlx_transpose_axes(res, a, axes.data(), axes.size(), s);
}
int mlx_moveaxis(mlx_array* res, const mlx_array a, int source, int destination, const mlx_stream s) {
    REQ_ARR(a, "mlx_moveaxis");
    const int nd = (int)A(a)->shape.size();
    int src, dst;
    if (norm_axis(source, nd, "mlx_moveaxis", &src) || norm_axis(destination, nd, "mlx_moveaxis", &dst)) return 1;
    std::vector<int> axes;
    for (int i = 0; i < nd; ++i)
        if (i != src) axes.push_back(i);
    axes.insert(axes.begin() + dst, src);
    return mlx_transpose_axes(res, a, axes.data(), axes.size(), s);
}
int mlx_full(mlx_array* res, const int* shape, size_t shape_num, const mlx_array vals, mlx_dtype dtype, const mlx_stream) {
    REQ_ARR(vals, "mlx_full");
    OMX_REQUIRE(A(vals)->size() == 1, "mlx_full: only a scalar fill value is supported (got %zu elements)", A(vals)->size());
    std::vector<int> sh(shape, shape + shape_num);
    NEW_OR_FAIL(r, sh, dtype);
    if (r->size()) {
        fill_value_kernel<<<grid_for(r->size()), 256, 0, g_stream>>>(r->ptr(), dtype, A(vals)->ptr(), A(vals)->dt, r->size());
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
int mlx_ones(mlx_array* res, const int* shape, size_t shape_num, mlx_dtype dtype, const mlx_stream s) {
    const float one = 1.0f;
    mlx_array v = mlx_array_new_data(&one, nullptr, 0, MLX_FLOAT32);
    OMX_REQUIRE(v.ctx, "mlx_ones: out of device memory");
    const int rc = mlx_full(res, shape, shape_num, v, dtype, s);
    mlx_array_free(v);
    return rc;
}
int mlx_where(mlx_array* res, const mlx_array condition, const mlx_array x, const mlx_array y, const mlx_stream) {
    REQ_ARR(condition, "mlx_where"); REQ_ARR(x, "mlx_where"); REQ_ARR(y, "mlx_where");
    const Arr &c = *A(condition), &a = *A(x), &b = *A(y);
    // broadcast (condition, x) and (that shape, y) to one shape: the two index tables share the output shape
    std::vector<int> sh1, shape;
    Idx t1;
    if (broadcast2(c, a, "mlx_where", &sh1, &t1)) return 1;
    Arr probe;
    probe.shape = sh1; probe.dt = c.dt;
    probe.strides.assign(sh1.size(), 0);
    Idx t2;
    if (broadcast2(probe, b, "mlx_where", &shape, &t2)) return 1;
    // re-derive the strides of condition and x against the FINAL shape
    Idx ixc, ixy;
    if (fill_idx(ixc, shape) || fill_idx(ixy, shape)) return 1;
    const int nd = (int)shape.size();
    auto stride_of = [&](const Arr& t, int i) -> long long {
        const int it = i - (nd - (int)t.shape.size());
        return (it >= 0 && t.shape[it] != 1) ? (long long)t.strides[it] : 0;
    };
    for (int i = 0; i < nd; ++i) { ixc.sa[i] = stride_of(c, i); ixc.sb[i] = stride_of(a, i); ixy.sb[i] = stride_of(b, i); }
    const mlx_dtype odt = promote(a.dt, b.dt);
    NEW_OR_FAIL(r, shape, odt);
    if (r->size()) {
        where_kernel<<<grid_for(r->size()), 256, 0, g_stream>>>(r->ptr(), odt, c.ptr(), c.dt, a.ptr(), a.dt, b.ptr(), b.dt, ixc, ixy, r->size());
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
int mlx_clip(mlx_array* res, const mlx_array a, const mlx_array a_min, const mlx_array a_max, const mlx_stream s) {
    REQ_ARR(a, "mlx_clip");
    OMX_REQUIRE(a_min.ctx || a_max.ctx, "mlx_clip: at least one of a_min and a_max must be given");
    mlx_array lo = mlx_array_new();
    if (a_min.ctx) {
        if (mlx_maximum(&lo, a, a_min, s)) { mlx_array_free(lo); return 1; }
    }
    const mlx_array mid = a_min.ctx ? lo : a;
    const int rc = a_max.ctx ? mlx_minimum(res, mid, a_max, s) : mlx_array_set(res, mid);
    mlx_array_free(lo);
    return rc;
}
int mlx_argsort_axis(mlx_array* res, const mlx_array a, int axis, const mlx_stream) {
    REQ_ARR(a, "mlx_argsort_axis");
    const Arr& s = *A(a);
    int ax;
    if (norm_axis(axis, (int)s.shape.size(), "mlx_argsort_axis", &ax)) return 1;
    Contig c;
    if (c.init(s)) return 1;
    size_t outer, inner; int n;
    around_axis(s.shape, ax, &outer, &n, &inner);
    OMX_REQUIRE(n <= 65536, "mlx_argsort_axis: %d elements along the sorted axis (at most 65536)", n);
    NEW_OR_FAIL(r, s.shape, MLX_UINT32);
    if (r->size()) {
        argsort_kernel<<<grid_for(r->size()), 256, 0, g_stream>>>((uint32_t*)r->ptr(), c.a->ptr(), s.dt, outer, n, inner);
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
int mlx_argsort(mlx_array* res, const mlx_array a, const mlx_stream s) {
    REQ_ARR(a, "mlx_argsort");
    return mlx_argsort_axis(res, a, -1, s);
}
// argpartition (ops.h:128): any arrangement with the kth element in its sorted place, smaller before, larger after, is a
// valid answer -- MLX does not promise more -- and a full stable argsort is one (Mixtral takes [..., :k], model.rs:296-302)
int mlx_argpartition_axis(mlx_array* res, const mlx_array a, int kth, int axis, const mlx_stream s) {
    REQ_ARR(a, "mlx_argpartition_axis");
    int ax;
    if (norm_axis(axis, (int)A(a)->shape.size(), "mlx_argpartition_axis", &ax)) return 1;
    const int n = A(a)->shape[ax], k = kth < 0 ? kth + n : kth;
    OMX_REQUIRE(k >= 0 && k < n, "mlx_argpartition_axis: kth %d out of range for %d elements", kth, n);
    return mlx_argsort_axis(res, a, ax, s);
}
static int take_along_impl(mlx_array* res, const Arr& a, const Arr& ind, int ax, const char* name) {
    OMX_REQUIRE(is_int_dt(ind.dt), "%s: indices must be integers", name);
    OMX_REQUIRE(a.shape.size() == ind.shape.size(), "%s: indices must have as many dimensions as the array (%zu vs %zu)", name, ind.shape.size(), a.shape.size());
    const int nd = (int)a.shape.size();
    std::vector<int> shape(nd);
    Idx ix;
    if (fill_idx(ix, shape)) return 1;
    for (int i = 0; i < nd; ++i) {
        if (i == ax) { shape[i] = ind.shape[i]; ix.sa[i] = 0; ix.sb[i] = (long long)ind.strides[i]; }
        else {
            const int da = a.shape[i], db = ind.shape[i];
            OMX_REQUIRE(da == db || da == 1 || db == 1, "%s: shapes are not broadcastable (dim %d: %d vs %d)", name, i, da, db);
            shape[i] = da == 1 ? db : da;
            ix.sa[i] = da != 1 ? (long long)a.strides[i] : 0;
            ix.sb[i] = db != 1 ? (long long)ind.strides[i] : 0;
        }
        ix.shape[i] = shape[i];
    }
    NEW_OR_FAIL(r, shape, a.dt);
    const size_t n = r->size();
    if (n) {
        const long long ast = (long long)a.strides[ax];
        switch (dsize(a.dt)) {
            case 1: take_along_kernel<1><<<grid_for(n), 256, 0, g_stream>>>(r->ptr(), a.ptr(), ind.ptr(), ind.dt, ix, n, ax, a.shape[ax], ast); break;
            case 2: take_along_kernel<2><<<grid_for(n), 256, 0, g_stream>>>(r->ptr(), a.ptr(), ind.ptr(), ind.dt, ix, n, ax, a.shape[ax], ast); break;
            case 4: take_along_kernel<4><<<grid_for(n), 256, 0, g_stream>>>(r->ptr(), a.ptr(), ind.ptr(), ind.dt, ix, n, ax, a.shape[ax], ast); break;
            default: take_along_kernel<8><<<grid_for(n), 256, 0, g_stream>>>(r->ptr(), a.ptr(), ind.ptr(), ind.dt, ix, n, ax, a.shape[ax], ast); break;
        }
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
int mlx_take_along_axis(mlx_array* res, const mlx_array a, const mlx_array indices, int axis, const mlx_stream) {
    REQ_ARR(a, "mlx_take_along_axis"); REQ_ARR(indices, "mlx_take_along_axis");
    int ax;
    if (norm_axis(axis, (int)A(a)->shape.size(), "mlx_take_along_axis", &ax)) return 1;
    return take_along_impl(res, *A(a), *A(indices), ax, "mlx_take_along_axis");
}
// take without an axis (ops.h:1115): indexes the FLATTENED array, the result has the shape of the indices
int mlx_take(mlx_array* res, const mlx_array a, const mlx_array indices, const mlx_stream) {
    REQ_ARR(a, "mlx_take"); REQ_ARR(indices, "mlx_take");
    Contig ca, ci;
    if (ca.init(*A(a)) || ci.init(*A(indices))) return 1;
    Arr flat_a = *ca.a, flat_i = *ci.a;
    flat_a.host.clear(); flat_i.host.clear();
    flat_a.shape = {(int)ca.a->size()}; flat_a.strides = {1};
    flat_i.shape = {(int)ci.a->size()}; flat_i.strides = {1};
    mlx_array tmp{nullptr};
    if (take_along_impl(&tmp, flat_a, flat_i, 0, "mlx_take")) return 1;
    A(tmp)->shape = ci.a->shape;
    A(tmp)->strides = row_major(ci.a->shape);
    return assign(res, A(tmp));
}
int mlx_expand_dims_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, const mlx_stream) {
    REQ_ARR(a, "mlx_expand_dims_axes");
    const Arr& s = *A(a);
    const int nd = (int)(s.shape.size() + axes_num);
    std::vector<bool> is_new(nd, false);
    for (size_t i = 0; i < axes_num; ++i) {
        int ax;
        if (norm_axis(axes[i], nd, "mlx_expand_dims_axes", &ax)) return 1;
        OMX_REQUIRE(!is_new[ax], "mlx_expand_dims_axes: repeated axis %d", axes[i]);
        is_new[ax] = true;
    }
    std::vector<int> shape(nd);
    std::vector<size_t> strides(nd);
    for (int i = nd - 1, j = (int)s.shape.size() - 1; i >= 0; --i) {
        if (is_new[i]) { shape[i] = 1; strides[i] = (i + 1 < nd) ? strides[i + 1] * (size_t)shape[i + 1] : 1; }
        else { shape[i] = s.shape[j]; strides[i] = s.strides[j]; --j; }
    }
    return assign(res, view_with_shape(s, shape, strides));
}
int mlx_squeeze_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, const mlx_stream) {
    REQ_ARR(a, "mlx_squeeze_axes");
    const Arr& s = *A(a);
    const int nd = (int)s.shape.size();
    std::vector<bool> drop(nd, false);
    for (size_t i = 0; i < axes_num; ++i) {
        int ax;
        if (norm_axis(axes[i], nd, "mlx_squeeze_axes", &ax)) return 1;
        OMX_REQUIRE(s.shape[ax] == 1, "mlx_squeeze_axes: cannot squeeze axis %d of size %d", axes[i], s.shape[ax]);
        drop[ax] = true;
    }
    std::vector<int> shape;
    std::vector<size_t> strides;
    for (int i = 0; i < nd; ++i)
        if (!drop[i]) { shape.push_back(s.shape[i]); strides.push_back(s.strides[i]); }
    return assign(res, view_with_shape(s, shape, strides));
}
int mlx_squeeze_axis(mlx_array* res, const mlx_array a, int axis, const mlx_stream s) { return mlx_squeeze_axes(res, a, &axis, 1, s); }
int mlx_squeeze(mlx_array* res, const mlx_array a, const mlx_stream s) {
    REQ_ARR(a, "mlx_squeeze");
    std::vector<int> axes;
    for (size_t i = 0; i < A(a)->shape.size(); ++i)
        if (A(a)->shape[i] == 1) axes.push_back((int)i);
    return mlx_squeeze_axes(res, a, axes.data(), axes.size(), s);
}
int mlx_flatten(mlx_array* res, const mlx_array a, int start_axis, int end_axis, const mlx_stream s) {
    REQ_ARR(a, "mlx_flatten");
    const Arr& src = *A(a);
    const int nd = (int)src.shape.size();
    if (nd == 0) { const int one = 1; return mlx_reshape(res, a, &one, 1, s); }
    int b = start_axis < 0 ? start_axis + nd : start_axis, e = end_axis < 0 ? end_axis + nd : end_axis;
    b = b < 0 ? 0 : b; e = e >= nd ? nd - 1 : e;           // MLX clamps, it does not fail
    OMX_REQUIRE(b <= e && b < nd, "mlx_flatten: start_axis %d must not come after end_axis %d", start_axis, end_axis);
    std::vector<int> shape(src.shape.begin(), src.shape.begin() + b);
    int prod = 1;
    for (int i = b; i <= e; ++i) prod *= src.shape[i];
    shape.push_back(prod);
    shape.insert(shape.end(), src.shape.begin() + e + 1, src.shape.end());
    return mlx_reshape(res, a, shape.data(), shape.size(), s);
}
int mlx_stack_axis(mlx_array* res, const mlx_vector_array arrays, int axis, const mlx_stream s) {
    OMX_REQUIRE(arrays.ctx && !reinterpret_cast<Vec*>(arrays.ctx)->v.empty(), "mlx_stack_axis: no arrays to stack");
    Vec* in = reinterpret_cast<Vec*>(arrays.ctx);
    Vec* expanded = new Vec();
    mlx_vector_array ev{expanded};
    int rc = 0;
    for (Arr* a : in->v) {
        mlx_array h{a}, x{nullptr};
        if (mlx_expand_dims(&x, h, axis, s)) { rc = 1; break; }
        expanded->v.push_back(A(x));
    }
    if (!rc) rc = mlx_concatenate_axis(res, ev, axis < 0 ? axis + (int)in->v[0]->shape.size() + 1 : axis, s);
    delete expanded;
    return rc;
}
int mlx_stack(mlx_array* res, const mlx_vector_array arrays, const mlx_stream s) { return mlx_stack_axis(res, arrays, 0, s); }
int mlx_split_sections(mlx_vector_array* res, const mlx_array a, const int* indices, size_t indices_num, int axis, const mlx_stream) {
    OMX_REQUIRE(res, "mlx_split_sections: null result");
    REQ_ARR(a, "mlx_split_sections");
    const Arr& s = *A(a);
    int ax;
    if (norm_axis(axis, (int)s.shape.size(), "mlx_split_sections", &ax)) return 1;
    Vec* out = new Vec();
    int prev = 0;
    for (size_t i = 0; i <= indices_num; ++i) {
        int stop = i < indices_num ? indices[i] : s.shape[ax];
        stop = stop < prev ? prev : (stop > s.shape[ax] ? s.shape[ax] : stop);
        Arr* v = clone_handle(&s);
        v->off += (size_t)prev * s.strides[ax] * dsize(s.dt);
        v->shape[ax] = stop - prev;
        out->v.push_back(v);
        prev = stop;
    }
    delete reinterpret_cast<Vec*>(res->ctx);
    res->ctx = out;
    return 0;
}
int mlx_split(mlx_vector_array* res, const mlx_array a, int num_splits, int axis, const mlx_stream s) {
    REQ_ARR(a, "mlx_split");
    int ax;
    if (norm_axis(axis, (int)A(a)->shape.size(), "mlx_split", &ax)) return 1;
    const int n = A(a)->shape[ax];
    OMX_REQUIRE(num_splits > 0 && n % num_splits == 0, "mlx_split: axis of size %d does not split into %d equal parts", n, num_splits);
    std::vector<int> idx;
    for (int i = 1; i < num_splits; ++i) idx.push_back(i * (n / num_splits));
    return mlx_split_sections(res, a, idx.data(), idx.size(), ax, s);
}
int mlx_conv1d(mlx_array* res, const mlx_array input, const mlx_array weight, int stride, int padding, int dilation, int groups, const mlx_stream) {
    REQ_ARR(input, "mlx_conv1d"); REQ_ARR(weight, "mlx_conv1d");
    Contig cx, cw;
    if (cx.init(*A(input)) || cw.init(*A(weight))) return 1;
    OMX_REQUIRE(cx.a->shape.size() == 3 && cw.a->shape.size() == 3, "mlx_conv1d: input [B, L, C_in] and weight [C_out, K, C_in / groups] expected");
    OMX_REQUIRE(cx.a->dt == cw.a->dt && is_float(cx.a->dt), "mlx_conv1d: input and weight must share a floating dtype");
    const int B = cx.a->shape[0], L = cx.a->shape[1], Cin = cx.a->shape[2], Cout = cw.a->shape[0], Kw = cw.a->shape[1];
    OMX_REQUIRE(stride >= 1 && dilation >= 1 && padding >= 0 && groups >= 1 && Cin % groups == 0 && Cout % groups == 0 && cw.a->shape[2] == Cin / groups,
                "mlx_conv1d: bad stride / dilation / groups (C_in %d, C_out %d, groups %d, weight C_in %d)", Cin, Cout, groups, cw.a->shape[2]);
    const int span = dilation * (Kw - 1) + 1, Lout = (L + 2 * padding - span) / stride + 1;
    OMX_REQUIRE(L + 2 * padding >= span, "mlx_conv1d: kernel span %d exceeds the padded input length %d", span, L + 2 * padding);
    std::vector<int> shape = {B, Lout, Cout};
    NEW_OR_FAIL(r, shape, cx.a->dt);
    if (r->size()) {
        conv1d_kernel<<<grid_for(r->size()), 256, 0, g_stream>>>(r->ptr(), cx.a->ptr(), cw.a->ptr(), cx.a->dt, B, L, Cin, Lout, Cout, Kw, stride, padding,
                                                               dilation, groups);
        OMX_LAUNCH_CHECK();
    }
    return assign(res, r);
}
int mlx_gather_mm(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_array lhs_indices, const mlx_array rhs_indices, bool,
                  const mlx_stream) {
    REQ_ARR(a, "mlx_gather_mm"); REQ_ARR(b, "mlx_gather_mm");
    OMX_REQUIRE(!lhs_indices.ctx && rhs_indices.ctx, "mlx_gather_mm: supported form is rhs_indices only (SwitchLinear, ops/quantization.rs:169-203)");
    const Arr& bs = *A(b);
    OMX_REQUIRE(bs.shape.size() == 3 && bs.dt == MLX_BFLOAT16 && A(a)->dt == MLX_BFLOAT16, "mlx_gather_mm: bfloat16 a [..., 1, K] and b [E, K, N] expected");
    const int E = bs.shape[0], K = bs.shape[1], N = bs.shape[2];
    // the weights as [E, N, K] row-major: what b IS underneath when it is swap_axes(w, -1, -2) of a stacked nn::Linear weight
    // (strides [N*K, 1, K]); anything else is transposed into that form once
    Arr wt = bs;
    wt.host.clear();
    wt.shape = {E, N, K};
    wt.strides = {bs.strides[0], bs.strides[2], bs.strides[1]};
    Contig cw, cx, ci;
    if (cw.init(wt) || cx.init(*A(a)) || ci.init(*A(rhs_indices))) return 1;
    OMX_REQUIRE(ci.a->dt == MLX_UINT32 || ci.a->dt == MLX_INT32, "mlx_gather_mm: rhs_indices must be (u)int32");
    OMX_REQUIRE(cx.a->shape.size() >= 2 && cx.a->shape.back() == K && cx.a->shape[cx.a->shape.size() - 2] == 1, "mlx_gather_mm: a must be [..., 1, K=%d]", K);
    const size_t n_x = cx.a->size() / K, n = ci.a->size();
    OMX_REQUIRE(n_x > 0 && n % n_x == 0, "mlx_gather_mm: %zu indices do not broadcast over %zu activation rows", n, n_x);
    std::vector<int> shape = ci.a->shape;
    shape.push_back(1);
    shape.push_back(N);
    NEW_OR_FAIL(r, shape, MLX_BFLOAT16);
    if (n && omx_gather_mm(r->ptr(), cx.a->ptr(), cw.a->ptr(), (const uint32_t*)ci.a->ptr(), (int)n, (int)(n / n_x), N, K, E, OMX_BFLOAT16, g_stream)) {
        delete r;
        return 1;
    }
    return assign(res, r);
}

}  // extern "C"
