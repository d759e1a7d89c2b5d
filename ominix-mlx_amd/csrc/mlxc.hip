// mlx-c compatible handle layer (include/omx_mlx_c.h) over the omx_* kernels.
//
// Mirrors the vendored shim mlx-rs/mlx-sys/src/mlx-c/mlx/c/{array,vector,stream,transforms,memory,
// fast,ops}.cpp in ownership and error behaviour (private/array.h:12-53, error.cpp:12-54), but is
// EAGER: there is no lazy graph, an op enqueues its kernels on the device stream when called.
// Arrays are ref-counted device buffers + (shape, strides, offset): reshape / transpose / slice are
// views, ops that need contiguous data materialise them with a strided-copy kernel.
#include <limits.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/omx_mlx_c.h"
#include "common.hpp"
#include "gemm.hpp"
#include "vec.hpp"

namespace {

using omx::bf16_t;
using omx::set_error;

constexpr int kMaxDim = 8;
hipStream_t g_stream = nullptr;   // one in-order device stream for the whole handle layer

// ---- caching allocator (mlx_clear_cache / mlx_get_{active,peak}_memory) ----
struct Pool {
    std::mutex mu;
    std::multimap<size_t, void*> free_list;
    size_t active = 0, peak = 0, cached = 0;
    void* get(size_t bytes) {
        bytes = (bytes + 511) & ~(size_t)511;
        if (bytes == 0) bytes = 512;
        std::lock_guard<std::mutex> lk(mu);
        auto it = free_list.find(bytes);
        void* p = nullptr;
        if (it != free_list.end()) {
            p = it->second;
            free_list.erase(it);
            cached -= bytes;
        } else if (hipMalloc(&p, bytes) != hipSuccess) {
            return nullptr;
        }
        active += bytes;
        if (active > peak) peak = active;
        return p;
    }
    void put(void* p, size_t bytes) {
        bytes = (bytes + 511) & ~(size_t)511;
        if (bytes == 0) bytes = 512;
        std::lock_guard<std::mutex> lk(mu);
        free_list.emplace(bytes, p);
        active -= bytes;
        cached += bytes;
    }
    void clear() {
        std::lock_guard<std::mutex> lk(mu);
        (void)hipStreamSynchronize(g_stream);
        for (auto& kv : free_list) (void)hipFree(kv.second);
        free_list.clear();
        cached = 0;
    }
} g_pool;

struct Buf {
    void* p;
    size_t bytes;
    bool owned;                     // false: device memory BORROWED from the caller (omx_mlx_array_from_device): never returned to the pool
    uint64_t seq = 0;               // the flush (mlxc_lazy.hpp) whose launches last wrote it; 0: written by launches outside any flush
    std::shared_ptr<void> aux;      // derived forms of a packed weight the deferred list built on first use (scale | bias words, matrix-core tiles)
    std::atomic<int> inflight{0};   // references held by recorded / submitted ops (not by anything a caller can reach): what "solely owned" ignores
    Buf(void* p_, size_t b, bool owned_ = true) : p(p_), bytes(b), owned(owned_) {}
    ~Buf() { if (p && owned) g_pool.put(p, bytes); }
};

// deferred execution (mlxc_lazy.hpp): ops may be RECORDED instead of launched; ptr() -- the only way to a buffer's address -- executes
// the recorded list first, unless the caller is a recorded launch itself
}  // namespace
int flush_pending();                // everything recorded so far has been LAUNCHED when this returns
int submit_pending();               // ... has been handed to the launch worker (mlx_async_eval)
int wait_issued(uint64_t seq);      // the launch worker has sent batch `seq` (and recorded its event)
namespace {
std::atomic<int> g_batches_in_flight{0};   // batches handed to the worker and not yet launched
thread_local int g_lazy_busy = 0;  // > 0: this thread is executing recorded ops (the launch worker, or an immediate run)
size_t g_n_pending = 0;
bool g_deferred_failed = false;    // a flush forced from inside ptr() failed: reported by the next evaluation point
uint64_t g_flush_seq = 0;          // batches handed over so far; every batch records an event behind its launches: item() / data() wait for the
constexpr int kEvRing = 64;        // producing batch only, not for the step the caller has queued behind it (Generate::next, model.rs:804-843)
hipEvent_t g_flush_ev[kEvRing] = {};
hipStream_t g_copy_stream = nullptr;

struct Arr {
    std::shared_ptr<Buf> buf;
    size_t off = 0;                 // byte offset of element 0
    std::vector<int> shape;
    std::vector<size_t> strides;    // in elements
    mlx_dtype dt = MLX_FLOAT32;
    std::vector<uint8_t> host;      // mirror handed out by mlx_array_data_*
    bool donated = false;           // its buffer was updated in place on behalf of a slice_update result (see there)
    Arr() = default;
    // a copy shares the buffer and the view, never the host mirror (that belongs to the handle mlx_array_data_* was called on; recorded ops
    // copy their operands' Arr, and a mirror of a logits row would travel with every record)
    Arr(const Arr& o) : buf(o.buf), off(o.off), shape(o.shape), strides(o.strides), dt(o.dt), donated(o.donated) {}
    Arr& operator=(const Arr& o) {
        if (this != &o) { buf = o.buf; off = o.off; shape = o.shape; strides = o.strides; dt = o.dt; donated = o.donated; host.clear(); }
        return *this;
    }
    Arr(Arr&&) = default;
    Arr& operator=(Arr&&) = default;
    size_t size() const { size_t n = 1; for (int d : shape) n *= (size_t)d; return n; }
    char* ptr() const {
        if (!g_lazy_busy && (g_n_pending || g_batches_in_flight.load(std::memory_order_acquire)) && flush_pending())
            g_deferred_failed = true;   // (no status to return here: the next evaluation point reports it)
        return (char*)buf->p + off;
    }
};

size_t dsize(mlx_dtype d) {
    switch (d) {
        case MLX_BOOL: case MLX_UINT8: case MLX_INT8: return 1;
        case MLX_UINT16: case MLX_INT16: case MLX_FLOAT16: case MLX_BFLOAT16: return 2;
        case MLX_UINT32: case MLX_INT32: case MLX_FLOAT32: return 4;
        default: return 8;
    }
}
bool is_float(mlx_dtype d) { return d == MLX_FLOAT16 || d == MLX_FLOAT32 || d == MLX_BFLOAT16; }

std::vector<size_t> row_major(const std::vector<int>& shape) {
    std::vector<size_t> st(shape.size());
    size_t s = 1;
    for (int i = (int)shape.size() - 1; i >= 0; --i) { st[i] = s; s *= (size_t)shape[i]; }
    return st;
}
bool is_contig(const Arr& a) {
    size_t s = 1;
    for (int i = (int)a.shape.size() - 1; i >= 0; --i) {
        if (a.shape[i] != 1 && a.strides[i] != s) return false;
        s *= (size_t)a.shape[i];
    }
    return true;
}

Arr* A(const mlx_array h) { return reinterpret_cast<Arr*>(h.ctx); }

Arr* new_arr(const std::vector<int>& shape, mlx_dtype dt) {
    Arr* a = new Arr();
    a->shape = shape;
    a->strides = row_major(shape);
    a->dt = dt;
    const size_t bytes = a->size() * dsize(dt);
    void* p = g_pool.get(bytes);
    if (!p) { delete a; return nullptr; }
    a->buf = std::make_shared<Buf>(p, bytes);
    return a;
}
int assign(mlx_array* res, Arr* n) {
    if (!res) { delete n; return set_error("null result pointer"); }
    if (res->ctx) delete A(*res);
    res->ctx = n;
    return 0;
}
#define NEW_OR_FAIL(var, shape, dt) Arr* var = new_arr(shape, dt); if (!var) return set_error("out of device memory allocating %zu-element array", (size_t)0)
#define REQ_ARR(h, name)                                                                                              \
    do {                                                                                                              \
        OMX_REQUIRE((h).ctx != nullptr, "%s: empty array handle", name);                                              \
        OMX_REQUIRE(!A(h)->donated, "%s: this array's buffer was donated to the result of an earlier mlx_slice_update " \
                    "(it was the sole owner); keep another reference before the update to preserve it", name);        \
    } while (0)

// ---- generic strided kernels ----
struct Idx {
    int nd;
    int shape[kMaxDim];
    long long sa[kMaxDim], sb[kMaxDim];   // element strides of up to two operands (0 = broadcast)
};

template <int ES>
__global__ void strided_copy_kernel(char* __restrict__ dst, const char* __restrict__ src, Idx ix, size_t n, bool dst_strided) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        long long o = 0;
        for (int d = ix.nd - 1; d >= 0; --d) {
            const int c = (int)(r % ix.shape[d]);
            r /= ix.shape[d];
            o += c * ix.sa[d];
        }
        const char* s = dst_strided ? src + i * ES : src + o * ES;
        char* t = dst_strided ? dst + o * ES : dst + i * ES;
        if (ES == 1) *t = *s;
        else if (ES == 2) *(uint16_t*)t = *(const uint16_t*)s;
        else if (ES == 4) *(uint32_t*)t = *(const uint32_t*)s;
        else *(uint64_t*)t = *(const uint64_t*)s;
    }
}

__device__ __forceinline__ float ld_f(const char* p, int dt, size_t i) {
    switch (dt) {
        case MLX_BFLOAT16: return omx::bf16_to_f32(((const bf16_t*)p)[i]);
        case MLX_FLOAT16: return (float)((const _Float16*)p)[i];
        case MLX_FLOAT32: return ((const float*)p)[i];
        case MLX_INT32: return (float)((const int32_t*)p)[i];
        case MLX_UINT32: return (float)((const uint32_t*)p)[i];
        case MLX_BOOL: case MLX_UINT8: return (float)((const uint8_t*)p)[i];
        case MLX_INT8: return (float)((const int8_t*)p)[i];
        case MLX_INT16: return (float)((const int16_t*)p)[i];
        case MLX_UINT16: return (float)((const uint16_t*)p)[i];
        case MLX_INT64: return (float)((const long long*)p)[i];
        case MLX_UINT64: return (float)((const unsigned long long*)p)[i];
        case MLX_FLOAT64: return (float)((const double*)p)[i];
        default: return 0.f;
    }
}
__device__ __forceinline__ void st_f(char* p, int dt, size_t i, float v) {
    switch (dt) {
        case MLX_BFLOAT16: ((bf16_t*)p)[i] = omx::f32_to_bf16(v); break;
        case MLX_FLOAT16: ((_Float16*)p)[i] = (_Float16)v; break;
        case MLX_FLOAT32: ((float*)p)[i] = v; break;
        case MLX_INT32: ((int32_t*)p)[i] = (int32_t)v; break;
        case MLX_UINT32: ((uint32_t*)p)[i] = (uint32_t)v; break;
        case MLX_BOOL: ((uint8_t*)p)[i] = v != 0.f; break;
        case MLX_UINT8: ((uint8_t*)p)[i] = (uint8_t)v; break;
        case MLX_INT8: ((int8_t*)p)[i] = (int8_t)v; break;
        case MLX_INT16: ((int16_t*)p)[i] = (int16_t)v; break;
        case MLX_UINT16: ((uint16_t*)p)[i] = (uint16_t)v; break;
        case MLX_INT64: ((long long*)p)[i] = (long long)v; break;
        case MLX_UINT64: ((unsigned long long*)p)[i] = (unsigned long long)v; break;
        case MLX_FLOAT64: ((double*)p)[i] = (double)v; break;
        default: break;
    }
}
// the integer twins (exact for every integer width; the glue's integer arithmetic and integer -> integer casts go through them)
__device__ __forceinline__ long long ld_i(const char* p, int dt, size_t i) {
    switch (dt) {
        case MLX_INT32: return ((const int32_t*)p)[i];
        case MLX_UINT32: return ((const uint32_t*)p)[i];
        case MLX_INT64: return ((const long long*)p)[i];
        case MLX_UINT64: return (long long)((const unsigned long long*)p)[i];
        case MLX_INT16: return ((const int16_t*)p)[i];
        case MLX_UINT16: return ((const uint16_t*)p)[i];
        case MLX_INT8: return ((const int8_t*)p)[i];
        case MLX_BOOL: case MLX_UINT8: return ((const uint8_t*)p)[i];
        default: return (long long)ld_f(p, dt, i);
    }
}
__device__ __forceinline__ void st_i(char* p, int dt, size_t i, long long v) {
    switch (dt) {
        case MLX_INT32: ((int32_t*)p)[i] = (int32_t)v; break;
        case MLX_UINT32: ((uint32_t*)p)[i] = (uint32_t)v; break;
        case MLX_INT64: case MLX_UINT64: ((long long*)p)[i] = v; break;
        case MLX_INT16: case MLX_UINT16: ((uint16_t*)p)[i] = (uint16_t)v; break;
        case MLX_INT8: case MLX_UINT8: ((uint8_t*)p)[i] = (uint8_t)v; break;
        case MLX_BOOL: ((uint8_t*)p)[i] = v != 0; break;
        default: st_f(p, dt, i, (float)v); break;
    }
}
__host__ __device__ inline bool is_int_dt(int d) { return d != MLX_FLOAT16 && d != MLX_FLOAT32 && d != MLX_BFLOAT16 && d != MLX_FLOAT64 && d != MLX_COMPLEX64; }


}  // namespace
#include "mlxc_lazy.hpp"
namespace {

enum { OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_SIGMOID, OP_EXP, OP_NEG, OP_CAST };

__global__ void binary_kernel(char* out, int odt, const char* a, int adt, const char* b, int bdt, Idx ix, size_t n, int op) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        long long oa = 0, ob = 0;
        for (int d = ix.nd - 1; d >= 0; --d) {
            const int c = (int)(r % ix.shape[d]);
            r /= ix.shape[d];
            oa += c * ix.sa[d];
            ob += c * ix.sb[d];
        }
        const float x = ld_f(a, adt, oa), y = ld_f(b, bdt, ob);
        float v;
        switch (op) {
            case OP_ADD: v = x + y; break;
            case OP_SUB: v = x - y; break;
            case OP_MUL: v = x * y; break;
            default: v = x / y; break;
        }
        st_f(out, odt, i, v);
    }
}
__global__ void unary_kernel(char* out, int odt, const char* a, int adt, Idx ix, size_t n, int op) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        long long oa = 0;
        for (int d = ix.nd - 1; d >= 0; --d) {
            const int c = (int)(r % ix.shape[d]);
            r /= ix.shape[d];
            oa += c * ix.sa[d];
        }
        if (op == OP_CAST && is_int_dt(adt) && is_int_dt(odt)) {   // integer -> integer: exact at every width
            st_i(out, odt, i, ld_i(a, adt, oa));
            continue;
        }
        const float x = ld_f(a, adt, oa);
        float v;
        switch (op) {
            case OP_SIGMOID: v = 1.0f / (1.0f + expf(-x)); break;
            case OP_EXP: v = expf(x); break;
            case OP_NEG: v = -x; break;
            default: v = x; break;
        }
        st_f(out, odt, i, v);
    }
}
// softmax over the last axis of a contiguous [rows, n]; one wave per row, fp32 inside (precise)
__global__ __launch_bounds__(256) void softmax_kernel(char* out, const char* in, int dt, size_t rows, int n) {
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, ld_f(in, dt, row * n + i));
    mx = omx::wave_max(mx);
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += expf(ld_f(in, dt, row * n + i) - mx);
    s = omx::wave_sum(s);
    for (int i = lane; i < n; i += 64) st_f(out, dt, row * n + i, expf(ld_f(in, dt, row * n + i) - mx) / s);
}

unsigned grid_for(size_t n) { size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

int fill_idx(Idx& ix, const std::vector<int>& shape) {
    OMX_REQUIRE(shape.size() <= (size_t)kMaxDim, "arrays of more than %d dimensions are not supported", kMaxDim);
    ix.nd = (int)shape.size();
    for (int i = 0; i < ix.nd; ++i) { ix.shape[i] = shape[i]; ix.sa[i] = ix.sb[i] = 0; }
    return 0;
}

// materialise a view as a fresh row-major array
int contiguous(const Arr& a, Arr** out) {
    NEW_OR_FAIL(r, a.shape, a.dt);
    const size_t n = a.size();
    if (n) {
        if (is_contig(a)) {
            OMX_HIP_CHECK(hipMemcpyAsync(r->ptr(), a.ptr(), n * dsize(a.dt), hipMemcpyDeviceToDevice, g_stream));
        } else {
            Idx ix;
            if (fill_idx(ix, a.shape)) { delete r; return 1; }
            for (int i = 0; i < ix.nd; ++i) ix.sa[i] = (long long)a.strides[i];
            switch (dsize(a.dt)) {
                case 1: strided_copy_kernel<1><<<grid_for(n), 256, 0, g_stream>>>(r->ptr(), a.ptr(), ix, n, false); break;
                case 2: strided_copy_kernel<2><<<grid_for(n), 256, 0, g_stream>>>(r->ptr(), a.ptr(), ix, n, false); break;
                case 4: strided_copy_kernel<4><<<grid_for(n), 256, 0, g_stream>>>(r->ptr(), a.ptr(), ix, n, false); break;
                default: strided_copy_kernel<8><<<grid_for(n), 256, 0, g_stream>>>(r->ptr(), a.ptr(), ix, n, false); break;
            }
            OMX_LAUNCH_CHECK();
        }
    }
    *out = r;
    return 0;
}
// contiguous view holder: either aliases `a` or owns a materialised copy
struct Contig {
    const Arr* a = nullptr;
    Arr* owned = nullptr;
    ~Contig() { delete owned; }
    int init(const Arr& src) {
        if (is_contig(src)) { a = &src; return 0; }
        if (contiguous(src, &owned)) return 1;
        a = owned;
        return 0;
    }
};
// scatter a contiguous array into a strided region of dst (slice_update / concatenate)
int scatter_into(char* dst_base, const std::vector<size_t>& dst_strides, const Arr& src_contig, mlx_dtype dt) {
    const size_t n = src_contig.size();
    if (!n) return 0;
    Idx ix;
    if (fill_idx(ix, src_contig.shape)) return 1;
    for (int i = 0; i < ix.nd; ++i) ix.sa[i] = (long long)dst_strides[i];
    switch (dsize(dt)) {
        case 1: strided_copy_kernel<1><<<grid_for(n), 256, 0, g_stream>>>(dst_base, src_contig.ptr(), ix, n, true); break;
        case 2: strided_copy_kernel<2><<<grid_for(n), 256, 0, g_stream>>>(dst_base, src_contig.ptr(), ix, n, true); break;
        case 4: strided_copy_kernel<4><<<grid_for(n), 256, 0, g_stream>>>(dst_base, src_contig.ptr(), ix, n, true); break;
        default: strided_copy_kernel<8><<<grid_for(n), 256, 0, g_stream>>>(dst_base, src_contig.ptr(), ix, n, true); break;
    }
    OMX_LAUNCH_CHECK();
    return 0;
}

// same-shape contiguous float operands (the residual adds and the SwiGLU products of a prompt pass: [2048, 4096 .. 12288] elements): 16 bytes
// per lane and no index arithmetic -- the general broadcast kernel above spent 41 us on 8 M bf16 elements, this form 6 (round 6).  Same
// arithmetic: float32 inside, one rounding to the output dtype.
template <int DT, int OP>
__global__ __launch_bounds__(256) void binary_vec_kernel(typename omx::Elem<DT>::T* __restrict__ out, const typename omx::Elem<DT>::T* __restrict__ a,
                                                         const typename omx::Elem<DT>::T* __restrict__ b, size_t n_vec) {
    constexpr int N = omx::Vec16<DT>::N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 256) {
        float x[N], y[N], r[N];
        omx::Vec16<DT>::ld(a + i * N, x);
        omx::Vec16<DT>::ld(b + i * N, y);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (OP == OP_ADD) r[j] = x[j] + y[j];
            else if (OP == OP_SUB) r[j] = x[j] - y[j];
            else if (OP == OP_MUL) r[j] = x[j] * y[j];
            else r[j] = x[j] / y[j];
        }
        omx::Vec16<DT>::st(out + i * N, r);
    }
}
template <int DT, int OP>
__global__ __launch_bounds__(256) void unary_vec_kernel(typename omx::Elem<DT>::T* __restrict__ out, const typename omx::Elem<DT>::T* __restrict__ a,
                                                        size_t n_vec) {
    constexpr int N = omx::Vec16<DT>::N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 256) {
        float x[N], r[N];
        omx::Vec16<DT>::ld(a + i * N, x);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (OP == OP_SIGMOID) r[j] = 1.0f / (1.0f + expf(-x[j]));
            else if (OP == OP_EXP) r[j] = expf(x[j]);
            else r[j] = -x[j];
        }
        omx::Vec16<DT>::st(out + i * N, r);
    }
}
template <int DT>
int launch_binary_vec(void* out, const void* a, const void* b, size_t nv, int op) {
    typedef typename omx::Elem<DT>::T T;
    const unsigned grid = (unsigned)std::min<size_t>((nv + 255) / 256, 8192);
    if (op == OP_ADD) binary_vec_kernel<DT, OP_ADD><<<grid, 256, 0, g_stream>>>((T*)out, (const T*)a, (const T*)b, nv);
    else if (op == OP_SUB) binary_vec_kernel<DT, OP_SUB><<<grid, 256, 0, g_stream>>>((T*)out, (const T*)a, (const T*)b, nv);
    else if (op == OP_MUL) binary_vec_kernel<DT, OP_MUL><<<grid, 256, 0, g_stream>>>((T*)out, (const T*)a, (const T*)b, nv);
    else binary_vec_kernel<DT, OP_DIV><<<grid, 256, 0, g_stream>>>((T*)out, (const T*)a, (const T*)b, nv);
    OMX_LAUNCH_CHECK();
    return 0;
}
template <int DT>
int launch_unary_vec(void* out, const void* a, size_t nv, int op) {
    typedef typename omx::Elem<DT>::T T;
    const unsigned grid = (unsigned)std::min<size_t>((nv + 255) / 256, 8192);
    if (op == OP_SIGMOID) unary_vec_kernel<DT, OP_SIGMOID><<<grid, 256, 0, g_stream>>>((T*)out, (const T*)a, nv);
    else if (op == OP_EXP) unary_vec_kernel<DT, OP_EXP><<<grid, 256, 0, g_stream>>>((T*)out, (const T*)a, nv);
    else unary_vec_kernel<DT, OP_NEG><<<grid, 256, 0, g_stream>>>((T*)out, (const T*)a, nv);
    OMX_LAUNCH_CHECK();
    return 0;
}
bool vec_ok(const Arr& t, size_t n) {       // a whole contiguous array of 16-byte vectors
    const size_t per = 16 / dsize(t.dt);
    return is_contig(t) && t.size() == n && n % per == 0 && ((uintptr_t)t.buf->p + t.off) % 16 == 0 &&
           (t.dt == MLX_BFLOAT16 || t.dt == MLX_FLOAT16 || t.dt == MLX_FLOAT32);
}

mlx_dtype promote(mlx_dtype a, mlx_dtype b) {
    if (a == b) return a;
    if (is_float(a) && !is_float(b)) return a;
    if (is_float(b) && !is_float(a)) return b;
    if (is_float(a) && is_float(b)) return MLX_FLOAT32;   // bf16 x f16, half x f32
    return dsize(a) >= dsize(b) ? a : b;
}

int binary(mlx_array* res, const mlx_array ha, const mlx_array hb, int op, const char* name) {
    REQ_ARR(ha, name); REQ_ARR(hb, name);
    const Arr &a = *A(ha), &b = *A(hb);
    const int nd = (int)std::max(a.shape.size(), b.shape.size());
    std::vector<int> shape(nd);
    Idx ix;
    if (fill_idx(ix, shape)) return 1;
    for (int i = 0; i < nd; ++i) {
        const int ia = i - (nd - (int)a.shape.size()), ib = i - (nd - (int)b.shape.size());
        const int da = ia >= 0 ? a.shape[ia] : 1, db = ib >= 0 ? b.shape[ib] : 1;
        OMX_REQUIRE(da == db || da == 1 || db == 1, "%s: shapes are not broadcastable (dim %d: %d vs %d)", name, i, da, db);
        shape[i] = da == 1 ? db : da;
        ix.shape[i] = shape[i];
        ix.sa[i] = (ia >= 0 && da != 1) ? (long long)a.strides[ia] : 0;
        ix.sb[i] = (ib >= 0 && db != 1) ? (long long)b.strides[ib] : 0;
    }
    const mlx_dtype odt = promote(a.dt, b.dt);
    NEW_OR_FAIL(r, shape, odt);
    const size_t n_ = r->size();
    Rec rec;
    rec.kind = op == OP_ADD ? RK_ADD : op == OP_MUL ? RK_MUL : RK_GENERIC;
    rec.a[0] = *r; rec.a[1] = a; rec.a[2] = b; rec.na = 3;
    rec.i0 = op;
    // the elementwise form the GEMV epilogues can absorb: two whole bf16 rows of the result's size
    rec.flag = odt == MLX_BFLOAT16 && a.dt == odt && b.dt == odt && a.size() == r->size() && b.size() == r->size() && is_contig(a) && is_contig(b);
    rec.i1 = (n_ && a.dt == odt && b.dt == odt && vec_ok(a, n_) && vec_ok(b, n_)) ? 1 : 0;   // the vector form applies (decided on shapes; the result is fresh and aligned)
    rec.run = [ix](Rec& q) -> int {
        const size_t n = q.a[0].size();
        if (n && q.i1) {
            const size_t nv = n * dsize(q.a[0].dt) / 16;
            if (q.a[0].dt == MLX_BFLOAT16) return launch_binary_vec<OMX_BFLOAT16>(q.a[0].ptr(), q.a[1].ptr(), q.a[2].ptr(), nv, q.i0);
            if (q.a[0].dt == MLX_FLOAT16) return launch_binary_vec<OMX_FLOAT16>(q.a[0].ptr(), q.a[1].ptr(), q.a[2].ptr(), nv, q.i0);
            return launch_binary_vec<OMX_FLOAT32>(q.a[0].ptr(), q.a[1].ptr(), q.a[2].ptr(), nv, q.i0);
        } else if (n) {
            binary_kernel<<<grid_for(n), 256, 0, g_stream>>>(q.a[0].ptr(), q.a[0].dt, q.a[1].ptr(), q.a[1].dt, q.a[2].ptr(), q.a[2].dt, ix, n, q.i0);
            OMX_LAUNCH_CHECK();
        }
        return 0;
    };
    if (record(std::move(rec))) { delete r; return 1; }
    return assign(res, r);
}
int unary(mlx_array* res, const mlx_array ha, int op, mlx_dtype odt, const char* name) {
    REQ_ARR(ha, name);
    const Arr& a = *A(ha);
    NEW_OR_FAIL(r, a.shape, odt);
    Idx ix;
    if (fill_idx(ix, a.shape)) { delete r; return 1; }
    for (int i = 0; i < ix.nd; ++i) ix.sa[i] = (long long)a.strides[i];
    Rec rec;
    rec.kind = op == OP_SIGMOID ? RK_SIGMOID : RK_GENERIC;
    rec.a[0] = *r; rec.a[1] = a; rec.na = 2;
    rec.i0 = op;
    rec.i1 = (r->size() && a.dt == odt && (op == OP_SIGMOID || op == OP_EXP || op == OP_NEG) && vec_ok(a, r->size())) ? 1 : 0;
    rec.run = [ix](Rec& q) -> int {
        const size_t n = q.a[0].size();
        if (n && q.i1) {
            const size_t nv = n * dsize(q.a[0].dt) / 16;
            if (q.a[0].dt == MLX_BFLOAT16) return launch_unary_vec<OMX_BFLOAT16>(q.a[0].ptr(), q.a[1].ptr(), nv, q.i0);
            if (q.a[0].dt == MLX_FLOAT16) return launch_unary_vec<OMX_FLOAT16>(q.a[0].ptr(), q.a[1].ptr(), nv, q.i0);
            return launch_unary_vec<OMX_FLOAT32>(q.a[0].ptr(), q.a[1].ptr(), nv, q.i0);
        } else if (n) {
            unary_kernel<<<grid_for(n), 256, 0, g_stream>>>(q.a[0].ptr(), q.a[0].dt, q.a[1].ptr(), q.a[1].dt, ix, n, q.i0);
            OMX_LAUNCH_CHECK();
        }
        return 0;
    };
    if (record(std::move(rec))) { delete r; return 1; }
    return assign(res, r);
}
int norm_axis(int axis, int nd, const char* name, int* out) {
    const int ax = axis < 0 ? axis + nd : axis;
    OMX_REQUIRE(ax >= 0 && ax < nd, "%s: axis %d out of range for %d dimensions", name, axis, nd);
    *out = ax;
    return 0;
}
omx_dtype to_omx(mlx_dtype d) { return (omx_dtype)(int)d; }

// an evaluation point that found the deferred list failing earlier reports it now
int take_deferred_error() {
    if (!g_deferred_failed) return 0;
    g_deferred_failed = false;
    return 1;     // (the failing op set the message)
}
// device -> host for item() / data(): waits for the flush that produced the buffer (a side stream behind that flush's event) when it is
// known and recent, for the whole stream otherwise.  `src` was obtained through ptr(): nothing is pending any more.
int read_back(void* dst, const Arr& a, const char* src, size_t bytes) {
    if (take_deferred_error()) return 1;
    const uint64_t seq = a.buf->seq;
    if (seq && wait_issued(seq)) return 1;
    hipEvent_t ev = (seq && g_flush_seq - seq < (uint64_t)kEvRing - 1) ? g_flush_ev[seq % kEvRing] : nullptr;
    if (ev) {
        if (!g_copy_stream) OMX_HIP_CHECK(hipStreamCreateWithFlags(&g_copy_stream, hipStreamNonBlocking));
        OMX_HIP_CHECK(hipStreamWaitEvent(g_copy_stream, ev, 0));
        OMX_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g_copy_stream));
        OMX_HIP_CHECK(hipStreamSynchronize(g_copy_stream));
        return 0;
    }
    OMX_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g_stream));
    OMX_HIP_CHECK(hipStreamSynchronize(g_stream));
    return 0;
}
int item_host(const mlx_array h, void* dst, mlx_dtype want, const char* name) {
    REQ_ARR(h, name);
    const Arr& a = *A(h);
    OMX_REQUIRE(a.size() == 1, "%s: item() needs a size-1 array (size %zu)", name, a.size());
    OMX_REQUIRE(a.dt == want || dsize(a.dt) == dsize(want), "%s: dtype mismatch", name);
    // (not through ptr(): that would wait until EVERYTHING recorded has been launched, the step the caller queued behind this value included)
    if (g_n_pending && a.buf->seq > g_flush_seq && submit_pending()) return 1;
    return read_back(dst, a, (const char*)a.buf->p + a.off, dsize(a.dt));
}
const void* data_host(const mlx_array h) {
    if (!h.ctx) return nullptr;
    Arr* a = A(h);
    Contig c;
    if (c.init(*a)) return nullptr;
    const size_t bytes = a->size() * dsize(a->dt);
    a->host.resize(bytes ? bytes : 1);
    if (bytes) {
        if (c.owned) {    // a strided view was gathered by a launch of this call: on the stream, behind everything
            if (hipMemcpyAsync(a->host.data(), c.a->ptr(), bytes, hipMemcpyDeviceToHost, g_stream) != hipSuccess) return nullptr;
            if (hipStreamSynchronize(g_stream) != hipSuccess || take_deferred_error()) return nullptr;
        } else if ((g_n_pending && a->buf->seq > g_flush_seq && submit_pending()) || read_back(a->host.data(), *a, (const char*)a->buf->p + a->off, bytes)) {
            return nullptr;
        }
    } else if (hipStreamSynchronize(g_stream) != hipSuccess) {
        return nullptr;
    }
    return a->host.data();
}

struct Vec { std::vector<Arr*> v; ~Vec() { for (Arr* a : v) delete a; } };
struct Str { bool cpu; };

}  // namespace

extern "C" {

void mlx_set_error_handler(mlx_error_handler_func handler, void* data, void (*dtor)(void*)) {
    omx_set_error_handler(handler, data, dtor);
}

size_t mlx_dtype_size(mlx_dtype dtype) { return dsize(dtype); }
mlx_array mlx_array_new(void) { return mlx_array{nullptr}; }
int mlx_array_free(mlx_array arr) { delete A(arr); return 0; }

static mlx_array scalar(const void* v, mlx_dtype dt) {
    int shape0 = 0;
    return mlx_array_new_data(v, &shape0, 0, dt);
}
mlx_array mlx_array_new_bool(bool val) { uint8_t v = val; return scalar(&v, MLX_BOOL); }
mlx_array mlx_array_new_int(int val) { int32_t v = val; return scalar(&v, MLX_INT32); }
mlx_array mlx_array_new_float32(float val) { return scalar(&val, MLX_FLOAT32); }
mlx_array mlx_array_new_float(float val) { return scalar(&val, MLX_FLOAT32); }

mlx_array mlx_array_new_data(const void* data, const int* shape, int dim, mlx_dtype dtype) {
    std::vector<int> sh(shape, shape + (dim > 0 ? dim : 0));
    Arr* a = new_arr(sh, dtype);
    if (!a) { set_error("mlx_array_new_data: out of device memory"); return mlx_array{nullptr}; }
    const size_t bytes = a->size() * dsize(dtype);
    if (bytes && data) {
        // pageable host memory: hipMemcpyAsync stages it before returning, so `data` may be released
        if (hipMemcpyAsync(a->ptr(), data, bytes, hipMemcpyHostToDevice, g_stream) != hipSuccess ||
            hipStreamSynchronize(g_stream) != hipSuccess) {
            delete a;
            set_error("mlx_array_new_data: host to device copy failed");
            return mlx_array{nullptr};
        }
    }
    return mlx_array{a};
}
// omx extension (no mlx-c counterpart: MLX owns its buffers): an array over device memory the CALLER owns and keeps alive -- the weights
// an engine already holds, shared with the per-op route without a copy.  Row-major, 16-byte aligned.  Never written by an op (results are
// fresh arrays; mlx_slice_update copies when its source is not solely owned -- a borrowed buffer never counts as solely owned).
mlx_array omx_mlx_array_from_device(const void* device_ptr, const int* shape, int dim, mlx_dtype dtype) {
    if (!device_ptr || ((uintptr_t)device_ptr & 15u) != 0 || dim < 0 || (dim > 0 && !shape)) {
        set_error("omx_mlx_array_from_device: null / unaligned pointer or bad shape");
        return mlx_array{nullptr};
    }
    Arr* a = new Arr();
    a->shape.assign(shape, shape + dim);
    a->strides = row_major(a->shape);
    a->dt = dtype;
    a->buf = std::make_shared<Buf>(const_cast<void*>(device_ptr), a->size() * dsize(dtype), false);
    return mlx_array{a};
}
int mlx_array_set(mlx_array* arr, const mlx_array src) {
    OMX_REQUIRE(arr != nullptr, "mlx_array_set: null destination");
    Arr* n = src.ctx ? new Arr(*A(src)) : nullptr;   // shares the buffer (ref-counted), like mlx::core::array copy
    if (n) n->host.clear();
    if (arr->ctx) delete A(*arr);
    arr->ctx = n;
    return 0;
}
size_t mlx_array_itemsize(const mlx_array arr) { return arr.ctx ? dsize(A(arr)->dt) : 0; }
size_t mlx_array_size(const mlx_array arr) { return arr.ctx ? A(arr)->size() : 0; }
size_t mlx_array_nbytes(const mlx_array arr) { return arr.ctx ? A(arr)->size() * dsize(A(arr)->dt) : 0; }
size_t mlx_array_ndim(const mlx_array arr) { return arr.ctx ? A(arr)->shape.size() : 0; }
const int* mlx_array_shape(const mlx_array arr) { return arr.ctx ? A(arr)->shape.data() : nullptr; }
const size_t* mlx_array_strides(const mlx_array arr) { return arr.ctx ? A(arr)->strides.data() : nullptr; }
int mlx_array_dim(const mlx_array arr, int dim) {
    if (!arr.ctx) return 0;
    const int nd = (int)A(arr)->shape.size();
    const int d = dim < 0 ? dim + nd : dim;
    return (d >= 0 && d < nd) ? A(arr)->shape[d] : 0;
}
mlx_dtype mlx_array_dtype(const mlx_array arr) { return arr.ctx ? A(arr)->dt : MLX_FLOAT32; }
int mlx_array_eval(mlx_array arr) {
    REQ_ARR(arr, "mlx_array_eval");
    if (flush_pending() || take_deferred_error()) return 1;
    OMX_HIP_CHECK(hipStreamSynchronize(g_stream));
    return 0;
}
int mlx_array_item_bool(bool* res, const mlx_array arr) { uint8_t v = 0; if (item_host(arr, &v, MLX_BOOL, "mlx_array_item_bool")) return 1; *res = v != 0; return 0; }
int mlx_array_item_uint32(uint32_t* res, const mlx_array arr) { return item_host(arr, res, MLX_UINT32, "mlx_array_item_uint32"); }
int mlx_array_item_int32(int32_t* res, const mlx_array arr) { return item_host(arr, res, MLX_INT32, "mlx_array_item_int32"); }
int mlx_array_item_float32(float* res, const mlx_array arr) { return item_host(arr, res, MLX_FLOAT32, "mlx_array_item_float32"); }
const uint8_t* mlx_array_data_uint8(const mlx_array arr) { return (const uint8_t*)data_host(arr); }
const uint16_t* mlx_array_data_uint16(const mlx_array arr) { return (const uint16_t*)data_host(arr); }
const uint32_t* mlx_array_data_uint32(const mlx_array arr) { return (const uint32_t*)data_host(arr); }
const int32_t* mlx_array_data_int32(const mlx_array arr) { return (const int32_t*)data_host(arr); }
const float* mlx_array_data_float32(const mlx_array arr) { return (const float*)data_host(arr); }
const uint16_t* mlx_array_data_bfloat16(const mlx_array arr) { return (const uint16_t*)data_host(arr); }
const uint16_t* mlx_array_data_float16(const mlx_array arr) { return (const uint16_t*)data_host(arr); }   // (float16_t spelled as its bits, like bfloat16)

mlx_vector_array mlx_vector_array_new(void) { return mlx_vector_array{new Vec()}; }
int mlx_vector_array_free(mlx_vector_array vec) { delete reinterpret_cast<Vec*>(vec.ctx); return 0; }
int mlx_vector_array_append_value(mlx_vector_array vec, const mlx_array val) {
    OMX_REQUIRE(vec.ctx && val.ctx, "mlx_vector_array_append_value: empty handle");
    Arr* c = new Arr(*A(val));
    c->host.clear();
    reinterpret_cast<Vec*>(vec.ctx)->v.push_back(c);
    return 0;
}
size_t mlx_vector_array_size(mlx_vector_array vec) { return vec.ctx ? reinterpret_cast<Vec*>(vec.ctx)->v.size() : 0; }
int mlx_vector_array_get(mlx_array* res, const mlx_vector_array vec, size_t idx) {
    OMX_REQUIRE(vec.ctx, "mlx_vector_array_get: empty vector");
    Vec* v = reinterpret_cast<Vec*>(vec.ctx);
    OMX_REQUIRE(idx < v->v.size(), "mlx_vector_array_get: index %zu out of range (size %zu)", idx, v->v.size());
    return mlx_array_set(res, mlx_array{v->v[idx]});
}

mlx_stream mlx_stream_new(void) { return mlx_stream{new Str{false}}; }
int mlx_stream_free(mlx_stream stream) { delete reinterpret_cast<Str*>(stream.ctx); return 0; }
bool mlx_stream_equal(mlx_stream lhs, mlx_stream rhs) {
    if (!lhs.ctx || !rhs.ctx) return lhs.ctx == rhs.ctx;
    return reinterpret_cast<Str*>(lhs.ctx)->cpu == reinterpret_cast<Str*>(rhs.ctx)->cpu;
}
int mlx_synchronize(mlx_stream) {
    if (flush_pending() || take_deferred_error()) return 1;
    OMX_HIP_CHECK(hipStreamSynchronize(g_stream));
    return 0;
}
mlx_stream mlx_default_cpu_stream_new(void) {
    set_error("mlx_default_cpu_stream_new: libomx_hip has no CPU backend (MI355X only)");
    return mlx_stream{nullptr};
}
mlx_stream mlx_default_gpu_stream_new(void) { return mlx_stream{new Str{false}}; }
// transforms.h:30,42 -- the evaluation points of the deferred list (mlxc_lazy.hpp): async_eval sends the recorded launches and returns,
// eval waits for them.  Everything recorded goes out, not only what the given arrays depend on (one in-order stream; a superset is allowed).
int mlx_async_eval(const mlx_vector_array) { return (submit_pending() || take_deferred_error()) ? 1 : 0; }
int mlx_eval(const mlx_vector_array) {
    if (flush_pending() || take_deferred_error()) return 1;
    OMX_HIP_CHECK(hipStreamSynchronize(g_stream));
    return 0;
}
/* omx extension: counters of the deferred list -- [0] ops recorded, [1] launched as recorded, [2] fused GEMV launches, [3] flushes, [4] host ns in flushes, [5] ns of the rewrite pass
 * (bench.py per_op_route; tests assert that the decode idioms really fuse) */
/* omx extension: switch the deferred list at run time (tests compare the three forms in one process).  lazy 0: every call launches as it
 * is made (round 5); fuse 0: recorded ops are launched as recorded.  Whatever is pending is executed first. */
int omx_mlx_lazy_mode(int lazy, int fuse) {
    if (flush_pending()) return 1;
    g_lazy_on = lazy != 0;
    g_lazy_env_read = true;
    g_fuse_mode = fuse != 0;
    return 0;
}
/* ... and the launch worker: 1 = mlx_async_eval hands the recorded list to a worker thread, 0 = the caller's thread launches it (default) */
int omx_mlx_lazy_async(int on) {
    if (flush_pending()) return 1;
    g_async_mode = on != 0;
    return 0;
}
void omx_mlx_lazy_stats(long* out6) { for (int i = 0; i < 6; ++i) out6[i] = g_lazy_stats[i]; }
int mlx_clear_cache(void) {
    if (flush_pending()) return 1;
    g_pool.clear();
    return 0;
}
int mlx_get_active_memory(size_t* res) { OMX_REQUIRE(res, "null result"); *res = g_pool.active; return 0; }
int mlx_get_peak_memory(size_t* res) { OMX_REQUIRE(res, "null result"); *res = g_pool.peak; return 0; }

// ---- fused hot-path ops ----
int mlx_fast_rms_norm(mlx_array* res, const mlx_array x, const mlx_array weight, float eps, const mlx_stream) {
    REQ_ARR(x, "mlx_fast_rms_norm");
    const Arr& x0 = *A(x);
    OMX_REQUIRE(!x0.shape.empty(), "mlx_fast_rms_norm: input must have at least 1 dimension");
    const int dim = x0.shape.back();
    if (weight.ctx) {
        REQ_ARR(weight, "mlx_fast_rms_norm");
        OMX_REQUIRE(A(weight)->shape.size() == 1 && A(weight)->shape[0] == dim && A(weight)->dt == x0.dt,
                    "mlx_fast_rms_norm: weight must be 1-D of size %d and of the input dtype", dim);
    }
    NEW_OR_FAIL(r, x0.shape, x0.dt);
    Rec rec;
    rec.kind = RK_RMSNORM;
    rec.a[0] = *r; rec.a[1] = x0; rec.na = 2;
    if (weight.ctx) { rec.a[2] = *A(weight); rec.na = 3; }
    rec.f0 = eps; rec.i0 = dim;
    rec.flag = weight.ctx && x0.dt == MLX_BFLOAT16 && is_contig(x0) && is_contig(*A(weight));   // what a GEMV prologue reproduces
    rec.run = [](Rec& q) -> int {
        Contig cx, cw;
        if (cx.init(q.a[1])) return 1;
        if (q.na > 2 && cw.init(q.a[2])) return 1;
        const int dim = q.i0;
        return omx_rms_norm(q.a[0].ptr(), cx.a->ptr(), q.na > 2 ? cw.a->ptr() : nullptr, dim ? (int64_t)(q.a[0].size() / dim) : 0, dim, q.f0,
                            to_omx(q.a[0].dt), g_stream);
    };
    if (record(std::move(rec))) { delete r; return 1; }
    return assign(res, r);
}
int mlx_fast_layer_norm(mlx_array* res, const mlx_array x, const mlx_array weight, const mlx_array bias, float eps,
                        const mlx_stream) {
    REQ_ARR(x, "mlx_fast_layer_norm");
    Contig cx, cw, cb;
    if (cx.init(*A(x))) return 1;
    OMX_REQUIRE(!cx.a->shape.empty(), "mlx_fast_layer_norm: input must have at least 1 dimension");
    const int dim = cx.a->shape.back();
    if (weight.ctx && cw.init(*A(weight))) return 1;
    if (bias.ctx && cb.init(*A(bias))) return 1;
    NEW_OR_FAIL(r, cx.a->shape, cx.a->dt);
    if (omx_layer_norm(r->ptr(), cx.a->ptr(), weight.ctx ? cw.a->ptr() : nullptr, bias.ctx ? cb.a->ptr() : nullptr,
                       dim ? (int64_t)(r->size() / dim) : 0, dim, eps, to_omx(r->dt), g_stream)) { delete r; return 1; }
    return assign(res, r);
}
int mlx_fast_rope(mlx_array* res, const mlx_array x, int dims, bool traditional, mlx_optional_float base, float scale,
                  int offset, const mlx_array freqs, const mlx_stream) {
    REQ_ARR(x, "mlx_fast_rope");
    // MLX core: exactly one of `base` and `freqs`
    OMX_REQUIRE(base.has_value != (freqs.ctx != nullptr), "mlx_fast_rope: exactly one of `base` and `freqs` must be given");
    const Arr& x0 = *A(x);
    if (freqs.ctx) {
        REQ_ARR(freqs, "mlx_fast_rope");
        OMX_REQUIRE(A(freqs)->dt == MLX_FLOAT32 && A(freqs)->shape.size() == 1 && A(freqs)->shape[0] == dims / 2,
                    "mlx_fast_rope: `freqs` must be a float32 vector of dims / 2 = %d entries", dims / 2);
    }
    const int nd = (int)x0.shape.size();
    OMX_REQUIRE(nd >= 2, "mlx_fast_rope: input must have at least 2 dimensions");   // same check as MLX core
    NEW_OR_FAIL(r, x0.shape, x0.dt);
    Rec rec;
    rec.kind = RK_ROPE;
    rec.a[0] = *r; rec.a[1] = x0; rec.na = 2;
    if (freqs.ctx) { rec.a[2] = *A(freqs); rec.na = 3; }
    rec.flag = !freqs.ctx && base.has_value && x0.dt == MLX_BFLOAT16;
    rec.i0 = dims; rec.i1 = traditional ? 1 : 0; rec.i2 = offset;
    rec.f0 = base.has_value ? base.value : 0.f; rec.f1 = scale;
    rec.run = [](Rec& q) -> int {
        Contig cx, cf;
        if (cx.init(q.a[1])) return 1;
        if (q.na > 2 && cf.init(q.a[2])) return 1;
        const int nd = (int)cx.a->shape.size();
        const int T = cx.a->shape[nd - 2], D = cx.a->shape[nd - 1];
        const int64_t batch = (T && D) ? (int64_t)(q.a[0].size() / ((size_t)T * D)) : 0;
        return q.na > 2 ? omx_rope_freqs(q.a[0].ptr(), cx.a->ptr(), batch, T, D, q.i0, q.i1 != 0, (const float*)cf.a->ptr(), q.f1, q.i2,
                                         to_omx(q.a[0].dt), g_stream)
                        : omx_rope(q.a[0].ptr(), cx.a->ptr(), batch, T, D, q.i0, q.i1 != 0, q.f0, q.f1, q.i2, to_omx(q.a[0].dt), g_stream);
    };
    if (record(std::move(rec))) { delete r; return 1; }
    return assign(res, r);
}
int mlx_fast_scaled_dot_product_attention(mlx_array* res, const mlx_array queries, const mlx_array keys,
                                          const mlx_array values, float scale, const char* mask_mode,
                                          const mlx_array mask_arr, const mlx_array sinks, const mlx_stream) {
    REQ_ARR(queries, "mlx_fast_scaled_dot_product_attention");
    REQ_ARR(keys, "mlx_fast_scaled_dot_product_attention");
    REQ_ARR(values, "mlx_fast_scaled_dot_product_attention");
    OMX_REQUIRE(!sinks.ctx, "mlx_fast_scaled_dot_product_attention: attention sinks are not supported");
    const Arr &q0 = *A(queries), &k0 = *A(keys), &v0 = *A(values);
    OMX_REQUIRE(q0.shape.size() == 4 && k0.shape.size() == 4 && v0.shape.size() == 4,
                "mlx_fast_scaled_dot_product_attention: queries, keys, values must be 4-D [B, H, T, D]");
    OMX_REQUIRE(k0.shape == v0.shape && q0.shape[0] == k0.shape[0] && q0.shape[3] == k0.shape[3],
                "mlx_fast_scaled_dot_product_attention: incompatible shapes");
    OMX_REQUIRE(q0.dt == k0.dt && q0.dt == v0.dt, "mlx_fast_scaled_dot_product_attention: dtype mismatch");
    const int Tq = q0.shape[2], Tk = k0.shape[2];
    int mode = OMX_MASK_NONE;
    const bool causal = mask_mode && strcmp(mask_mode, "causal") == 0;
    OMX_REQUIRE(!mask_mode || mask_mode[0] == 0 || causal || strcmp(mask_mode, "array") == 0,
                "mlx_fast_scaled_dot_product_attention: Invalid mask mode '%s'", mask_mode);
    if (causal) {
        mode = OMX_MASK_CAUSAL;
    } else if (mask_arr.ctx) {
        REQ_ARR(mask_arr, "mlx_fast_scaled_dot_product_attention");
        OMX_REQUIRE(A(mask_arr)->size() == (size_t)Tq * Tk, "mlx_fast_scaled_dot_product_attention: mask must broadcast from [Tq=%d, Tk=%d]", Tq, Tk);
        if (A(mask_arr)->dt == MLX_BOOL) mode = OMX_MASK_BOOL;
        else {
            OMX_REQUIRE(A(mask_arr)->dt == q0.dt, "mlx_fast_scaled_dot_product_attention: additive mask must have the query dtype");
            mode = OMX_MASK_ADDITIVE;
        }
    }
    NEW_OR_FAIL(r, q0.shape, q0.dt);
    Rec rec;
    rec.kind = RK_SDPA;
    rec.a[0] = *r; rec.a[1] = q0; rec.a[2] = k0; rec.a[3] = v0; rec.na = 4;
    if (mode == OMX_MASK_BOOL || mode == OMX_MASK_ADDITIVE) { rec.a[4] = *A(mask_arr); rec.na = 5; }
    rec.f0 = scale; rec.i0 = mode;
    rec.run = [](Rec& q) -> int {
        const Arr &q0 = q.a[1], &k0 = q.a[2], &v0 = q.a[3];
        const int B = q0.shape[0], H = q0.shape[1], Tq = q0.shape[2], D = q0.shape[3], Hkv = k0.shape[1], Tk = k0.shape[2];
        Contig cq, ck, cv, cm;
        if (cq.init(q0)) return 1;
        // K/V: the [.., :offset, :] views of the step-256 cache buffers are consumed in place (cache.rs:190-193)
        const Arr *k = &k0, *v = &v0;
        auto kv_ok = [&](const Arr& t) { return t.strides[3] == 1 && (t.strides[2] == (size_t)D || Tk == 1); };
        if (!kv_ok(k0) || !kv_ok(v0) || k0.strides[0] != v0.strides[0] || k0.strides[1] != v0.strides[1]) {
            if (ck.init(k0) || cv.init(v0)) return 1;   // row-major copies (or aliases when already contiguous)
            k = ck.a;
            v = cv.a;
        }
        const void* mptr = nullptr;
        if (q.na > 4) {
            if (cm.init(q.a[4])) return 1;
            mptr = cm.a->ptr();
        }
        return omx_sdpa(q.a[0].ptr(), cq.a->ptr(), k->ptr(), v->ptr(), B, H, Hkv, Tq, Tk, D, (int64_t)k->strides[0], (int64_t)k->strides[1],
                        q.f0, q.i0, mptr, to_omx(q0.dt), g_stream);
    };
    if (record(std::move(rec))) { delete r; return 1; }
    return assign(res, r);
}

// ---- GEMM ----
// a[..., M, K] @ b[..., K, N] with broadcast batch dimensions (mlx::core::matmul, ops.h:598-602): the form of the reference's explicit
// attention -- q.matmul(&k_t) and attn.matmul(&v) on [B, H, T, D] (funasr-mlx/src/paraformer.rs:517-520, 981-1017;
// flux-klein-mlx/src/klein_model.rs:474-483, 656-659).  b^T is taken in place when b is a transposed view of row-major [.., N, K] (k_t),
// otherwise materialised once for the whole batch; float32 runs on the exact-f32 matrix cores (one batched launch when nothing is
// broadcast), bfloat16 one GEMM / GEMV launch per batch entry.
static int matmul_batched(mlx_array* res, const Arr& a_in, const Arr& b_in, const char* name) {
    Arr av = a_in, bv = b_in;
    bool drop_m = false, drop_n = false;
    if (av.shape.size() == 1) { av.shape.insert(av.shape.begin(), 1); av.strides.insert(av.strides.begin(), 0); drop_m = true; }
    if (bv.shape.size() == 1) { bv.shape.push_back(1); bv.strides.push_back(0); drop_n = true; }
    const int na = (int)av.shape.size(), nb = (int)bv.shape.size();
    const int M = av.shape[na - 2], K = av.shape[na - 1], N = bv.shape[nb - 1];
    OMX_REQUIRE(bv.shape[nb - 2] == K, "%s: inner dimensions differ (%d vs %d)", name, K, bv.shape[nb - 2]);
    OMX_REQUIRE(av.dt == bv.dt, "%s: dtype mismatch", name);
    OMX_REQUIRE(av.dt == MLX_BFLOAT16 || av.dt == MLX_FLOAT32, "%s: batched operands in bfloat16 or float32", name);
    // broadcast batch shape, right-aligned
    const int nbat = std::max(na, nb) - 2;
    std::vector<int> bs(nbat), da(nbat, 1), db(nbat, 1);
    for (int i = 0; i < nbat; ++i) {
        const int ia = i - (nbat - (na - 2)), ib = i - (nbat - (nb - 2));
        if (ia >= 0) da[i] = av.shape[ia];
        if (ib >= 0) db[i] = bv.shape[ib];
        OMX_REQUIRE(da[i] == db[i] || da[i] == 1 || db[i] == 1, "%s: batch dimensions %d and %d do not broadcast", name, da[i], db[i]);
        bs[i] = std::max(da[i], db[i]);
    }
    Contig ca;
    if (ca.init(av)) return 1;                                   // [da..., M, K] row-major
    Arr bt = bv;                                                 // b^T: [db..., N, K]
    std::swap(bt.shape[nb - 1], bt.shape[nb - 2]);
    std::swap(bt.strides[nb - 1], bt.strides[nb - 2]);
    Contig cb;
    if (cb.init(bt)) return 1;
    std::vector<int> oshape = bs;
    oshape.push_back(M); oshape.push_back(N);
    Arr* r = new_arr(oshape, av.dt);
    if (!r) return set_error("%s: out of device memory", name);
    size_t nbatch = 1;
    for (int d : bs) nbatch *= (size_t)d;
    const size_t es = dsize(av.dt);
    bool plain = true;                                           // no broadcasting: uniform batch strides
    for (int i = 0; i < nbat; ++i) plain = plain && da[i] == bs[i] && db[i] == bs[i];
    int rc = 0;
    if (M && N && K && nbatch) {
        if (av.dt == MLX_FLOAT32 && plain) {
            omx::GemmF32 g = {};
            g.a = (const float*)ca.a->ptr(); g.b = (const float*)cb.a->ptr(); g.out = (float*)r->ptr();
            g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
            g.sa = (int64_t)M * K; g.sb = (int64_t)N * K; g.sc = (int64_t)M * N; g.batch = (int)nbatch; g.alpha = 1.0f;
            rc = omx::launch_gemm_f32(g, g_stream);
        } else {
            for (size_t i = 0; i < nbatch && !rc; ++i) {
                size_t rem = i, oa = 0, ob = 0, sa = 1, sb = 1;   // offsets of this entry in the (un-broadcast) operands, in matrices
                for (int d = nbat - 1; d >= 0; --d) {
                    const size_t id = rem % (size_t)bs[d];
                    rem /= (size_t)bs[d];
                    if (da[d] != 1) oa += id * sa;
                    if (db[d] != 1) ob += id * sb;
                    sa *= (size_t)da[d]; sb *= (size_t)db[d];
                }
                rc = omx_linear(r->ptr() + i * (size_t)M * N * es, ca.a->ptr() + oa * (size_t)M * K * es, cb.a->ptr() + ob * (size_t)N * K * es,
                                nullptr, M, N, K, to_omx(av.dt), g_stream);
            }
        }
    }
    if (rc) { delete r; return 1; }
    if (drop_m) { r->shape.erase(r->shape.end() - 2); r->strides = row_major(r->shape); }
    if (drop_n) { r->shape.pop_back(); r->strides = row_major(r->shape); }
    return assign(res, r);
}

int mlx_astype(mlx_array* res, const mlx_array a, mlx_dtype dtype, const mlx_stream);
static int matmul_impl(mlx_array* res, const mlx_array ha_in, const mlx_array hb_in, const Arr* bias, const char* name) {
    REQ_ARR(ha_in, name); REQ_ARR(hb_in, name);
    // MLX promotes mixed floating operands before the product (bf16 probabilities x f32 anything -> f32: the Klein blocks divide their
    // bf16 scores by an f32 scalar array, so their softmax and attn @ v run in float32, klein_model.rs:653-656)
    mlx_array ha = ha_in, hb = hb_in, tmp = {nullptr};
    struct Drop { mlx_array* t; ~Drop() { if (t->ctx) delete A(*t); } } drop{&tmp};
    if (A(ha)->dt != A(hb)->dt && is_float(A(ha)->dt) && is_float(A(hb)->dt)) {
        const bool cast_a = A(ha)->dt != MLX_FLOAT32;
        if (mlx_astype(&tmp, cast_a ? ha : hb, MLX_FLOAT32, mlx_stream{nullptr})) return 1;
        if (cast_a) ha = tmp; else hb = tmp;
        if (A(ha)->dt != A(hb)->dt) {   // bf16 x f16: both to float32
            mlx_array t2 = {nullptr};
            if (mlx_astype(&t2, cast_a ? hb : ha, MLX_FLOAT32, mlx_stream{nullptr})) return 1;
            mlx_array r2 = {nullptr};
            const int rc = matmul_impl(&r2, cast_a ? ha : t2, cast_a ? t2 : hb, bias, name);
            delete A(t2);
            if (rc) return 1;
            return assign(res, A(r2));
        }
    }
    const Arr &a0 = *A(ha), &b0 = *A(hb);
    OMX_REQUIRE(a0.shape.size() >= 1 && b0.shape.size() >= 1, "%s: scalar operand", name);
    if (b0.shape.size() != 2) {
        OMX_REQUIRE(bias == nullptr, "%s: a bias needs a 2-D weight", name);
        return matmul_batched(res, a0, b0, name);
    }
    const int K = a0.shape.back(), N = b0.shape[1];
    OMX_REQUIRE(b0.shape[0] == K, "%s: inner dimensions differ (%d vs %d)", name, K, b0.shape[0]);
    OMX_REQUIRE(a0.dt == b0.dt, "%s: dtype mismatch", name);
    std::vector<int> oshape(a0.shape.begin(), a0.shape.end() - 1);
    oshape.push_back(N);
    Arr* r = new_arr(oshape, a0.dt);
    if (!r) return set_error("%s: out of device memory", name);
    const int M = K ? (int)(a0.size() / K) : 0;
    // nn::Linear passes w.t(): a [N,K] row-major weight seen as [K,N] with strides (1, K) -- the NT operand
    const bool nt = b0.strides[0] == 1 && b0.strides[1] == (size_t)K;
    Rec rec;
    rec.kind = RK_MATMUL;
    rec.a[0] = *r; rec.a[1] = a0; rec.a[2] = b0; rec.na = 3;
    if (bias) { rec.a[3] = *bias; rec.na = 4; }
    rec.i0 = N; rec.i1 = K; rec.i2 = M;
    // the decode form (one bf16 row against an NT weight): what the deferred list may rewrite onto the fused GEMV family
    rec.flag = M == 1 && !bias && nt && a0.dt == MLX_BFLOAT16 && is_contig(a0) && K % 8 == 0 && K <= 65536;
    // ... and the prompt form (many bf16 rows): residual / SwiGLU epilogues and segment stacking of the 256^2 GEMM family (gemm.hpp)
    rec.i3 = (M > 8 && !bias && nt && a0.dt == MLX_BFLOAT16 && is_contig(a0) && K % 64 == 0 && N % 4 == 0) ? 1 : 0;
    rec.run = [](Rec& q) -> int {
        const Arr &a0 = q.a[1], &b0 = q.a[2];
        const int N = q.i0, K = q.i1, M = q.i2;
        Contig ca;
        if (ca.init(a0)) return 1;
        const Arr* w = &b0;
        Arr* wt = nullptr;
        if (!(b0.strides[0] == 1 && b0.strides[1] == (size_t)K)) {   // materialise b^T as [N, K]
            Arr view = b0;
            view.shape = {N, K};
            view.strides = {b0.strides[1], b0.strides[0]};
            if (contiguous(view, &wt)) return 1;
            w = wt;
        }
        const int rc = omx_linear(q.a[0].ptr(), ca.a->ptr(), w->ptr(), q.na > 3 ? q.a[3].ptr() : nullptr, M, N, K, to_omx(a0.dt), g_stream);
        delete wt;
        return rc;
    };
    if (record(std::move(rec))) { delete r; return 1; }
    return assign(res, r);
}
int mlx_matmul(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) {
    return matmul_impl(res, a, b, nullptr, "mlx_matmul");
}
int mlx_addmm(mlx_array* res, const mlx_array c, const mlx_array a, const mlx_array b, float alpha, float beta,
              const mlx_stream) {
    REQ_ARR(c, "mlx_addmm");
    REQ_ARR(b, "mlx_addmm");
    // nn::Linear's call (linear.rs:88-90): alpha = beta = 1, c a bias of size N, b a 2-D weight view -> the bias rides in the GEMM epilogue
    if (alpha == 1.0f && beta == 1.0f && A(b)->shape.size() == 2 && A(c)->size() == (size_t)A(b)->shape[1] && A(c)->dt == A(b)->dt) {
        Contig cc;
        if (cc.init(*A(c))) return 1;
        return matmul_impl(res, a, b, cc.a, "mlx_addmm");
    }
    // general form beta * c + alpha * (a @ b) with c broadcast to the product: composed from the product and two elementwise
    // ops in the product's dtype (one more rounding per op than MLX's fused epilogue when the dtype is bf16 / f16)
    mlx_array prod = {nullptr}, sa = {nullptr}, sb = {nullptr}, t1 = {nullptr}, t2 = {nullptr};
    struct Drop { std::vector<mlx_array*> v; ~Drop() { for (auto* h : v) if (h->ctx) delete A(*h); } } drop{{&prod, &sa, &sb, &t1, &t2}};
    if (matmul_impl(&prod, a, b, nullptr, "mlx_addmm")) return 1;
    const mlx_dtype dt = A(prod)->dt;
    auto scalar = [&](float v, mlx_array* out) -> int {
        mlx_array f = mlx_array_new_float32(v);
        const int rc = mlx_astype(out, f, dt, mlx_stream{nullptr});
        delete A(f);
        return rc;
    };
    const mlx_array* lhs = &prod;
    if (alpha != 1.0f) {
        if (scalar(alpha, &sa) || mlx_multiply(&t1, prod, sa, mlx_stream{nullptr})) return 1;
        lhs = &t1;
    }
    mlx_array cc = c;
    if (beta != 1.0f) {
        if (scalar(beta, &sb) || mlx_multiply(&t2, c, sb, mlx_stream{nullptr})) return 1;
        cc = t2;
    }
    return mlx_add(res, cc, *lhs, mlx_stream{nullptr});
}

// ---- affine group quantisation (ops.h:356-365, 471-484, 793-810; ops/quantization.rs:41-153, 226-279) ----
static int quant_params(const char* name, mlx_optional_int group_size, mlx_optional_int bits, const char* mode, int* g, int* b) {
    *g = group_size.has_value ? group_size.value : 64;    // quantized.rs:330-333 defaults
    *b = bits.has_value ? bits.value : 4;
    OMX_REQUIRE(!mode || mode[0] == 0 || strcmp(mode, "affine") == 0, "%s: only the affine mode is supported (got '%s')", name, mode);
    return 0;
}
int mlx_quantize(mlx_vector_array* res, const mlx_array w, mlx_optional_int group_size, mlx_optional_int bits, const char* mode,
                 const mlx_stream) {
    REQ_ARR(w, "mlx_quantize");
    OMX_REQUIRE(res && res->ctx, "mlx_quantize: result vector must be created with mlx_vector_array_new");
    int g, b;
    if (quant_params("mlx_quantize", group_size, bits, mode, &g, &b)) return 1;
    Contig cw;
    if (cw.init(*A(w))) return 1;
    OMX_REQUIRE(!cw.a->shape.empty(), "mlx_quantize: scalar input");
    const int K = cw.a->shape.back();
    OMX_REQUIRE(K % g == 0 && (K * b) % 32 == 0, "mlx_quantize: the last dimension (%d) must be divisible by the group size (%d)", K, g);
    std::vector<int> ps = cw.a->shape, ss = cw.a->shape;
    ps.back() = K * b / 32;
    ss.back() = K / g;
    Arr *pk = new_arr(ps, MLX_UINT32), *sc = new_arr(ss, cw.a->dt), *bi = new_arr(ss, cw.a->dt);
    if (!pk || !sc || !bi) { delete pk; delete sc; delete bi; return set_error("mlx_quantize: out of device memory"); }
    const int rc = omx_quantize(pk->ptr(), sc->ptr(), bi->ptr(), cw.a->ptr(), K ? (int64_t)(cw.a->size() / K) : 0, K, g, b,
                                to_omx(cw.a->dt), g_stream);
    if (!rc) {
        Vec* v = reinterpret_cast<Vec*>(res->ctx);
        v->v.push_back(pk); v->v.push_back(sc); v->v.push_back(bi);
        return 0;
    }
    delete pk; delete sc; delete bi;
    return 1;
}
int mlx_dequantize(mlx_array* res, const mlx_array w, const mlx_array scales, const mlx_array biases, mlx_optional_int group_size,
                   mlx_optional_int bits, const char* mode, mlx_optional_dtype dtype, const mlx_stream) {
    REQ_ARR(w, "mlx_dequantize"); REQ_ARR(scales, "mlx_dequantize");
    int g, b;
    if (quant_params("mlx_dequantize", group_size, bits, mode, &g, &b)) return 1;
    const Arr &w0 = *A(w), &s0 = *A(scales);
    if (biases.ctx) REQ_ARR(biases, "mlx_dequantize");
    OMX_REQUIRE(w0.dt == MLX_UINT32 && !w0.shape.empty(), "mlx_dequantize: w must be a packed uint32 array");
    OMX_REQUIRE(!dtype.has_value || dtype.value == s0.dt, "mlx_dequantize: output dtype must be the dtype of scales");
    const int K = w0.shape.back() * 32 / b;
    OMX_REQUIRE(s0.size() * (size_t)g == w0.size() * 32 / b, "mlx_dequantize: scales shape does not match w / group_size");
    std::vector<int> shape = w0.shape;
    shape.back() = K;
    NEW_OR_FAIL(r, shape, s0.dt);
    Rec rec;
    rec.a[0] = *r; rec.a[1] = w0; rec.a[2] = s0; rec.na = 3;
    if (biases.ctx) { rec.a[3] = *A(biases); rec.na = 4; }
    rec.i0 = K; rec.i1 = g; rec.i2 = b;
    rec.run = [](Rec& q) -> int {
        Contig cw, cs, cb;
        if (cw.init(q.a[1]) || cs.init(q.a[2]) || (q.na > 3 && cb.init(q.a[3]))) return 1;
        const int K = q.i0;
        return omx_dequantize(q.a[0].ptr(), cw.a->ptr(), cs.a->ptr(), q.na > 3 ? cb.a->ptr() : nullptr, K ? (int64_t)(q.a[0].size() / K) : 0, K, q.i1,
                              q.i2, to_omx(cs.a->dt), g_stream);
    };
    if (record(std::move(rec))) { delete r; return 1; }
    return assign(res, r);
}
int mlx_quantized_matmul(mlx_array* res, const mlx_array x, const mlx_array w, const mlx_array scales, const mlx_array biases,
                         bool transpose, mlx_optional_int group_size, mlx_optional_int bits, const char* mode, const mlx_stream) {
    REQ_ARR(x, "mlx_quantized_matmul"); REQ_ARR(w, "mlx_quantized_matmul"); REQ_ARR(scales, "mlx_quantized_matmul");
    int g, b;
    if (quant_params("mlx_quantized_matmul", group_size, bits, mode, &g, &b)) return 1;
    OMX_REQUIRE(transpose, "mlx_quantized_matmul: only transpose = true (QuantizedLinear, quantized.rs:366-375) is supported");
    const Arr &x0 = *A(x), &w0 = *A(w), &s0 = *A(scales);
    if (biases.ctx) REQ_ARR(biases, "mlx_quantized_matmul");
    OMX_REQUIRE(w0.dt == MLX_UINT32 && w0.shape.size() == 2, "mlx_quantized_matmul: w must be a 2-D packed uint32 array");
    const int N = w0.shape[0], K = w0.shape[1] * 32 / b;
    OMX_REQUIRE(!x0.shape.empty() && x0.shape.back() == K, "mlx_quantized_matmul: x has %d features, packed w has %d",
                x0.shape.empty() ? 0 : x0.shape.back(), K);
    OMX_REQUIRE(s0.size() == (size_t)N * (K / g), "mlx_quantized_matmul: scales shape does not match w / group_size");
    std::vector<int> shape(x0.shape.begin(), x0.shape.end() - 1);
    shape.push_back(N);
    // x, scales and biases share one dtype: bfloat16, or float16 (a float16 checkpoint runs in float16 end to end, quantized.rs:361-385);
    // a mix is MLX's promotion to float32, which this path does not implement
    OMX_REQUIRE((x0.dt == MLX_BFLOAT16 || x0.dt == MLX_FLOAT16) && x0.dt == s0.dt,
                "mlx_quantized_matmul: x and scales / biases must both be bfloat16 or both float16");
    NEW_OR_FAIL(r, shape, x0.dt);
    const int M = K ? (int)(x0.size() / K) : 0;
    Rec rec;
    rec.kind = RK_QMM;
    rec.a[0] = *r; rec.a[1] = x0; rec.a[2] = w0; rec.a[3] = s0; rec.na = 4;
    if (biases.ctx) { rec.a[4] = *A(biases); rec.na = 5; }
    rec.i0 = N; rec.i1 = K; rec.i2 = M; rec.i3 = g | (b << 16);
    // the decode form (one bf16 row against a whole packed matrix): what the deferred list may rewrite onto the fused packed-GEMV family
    rec.flag = M == 1 && x0.dt == MLX_BFLOAT16 && (b == 4 || b == 8) && K % 512 == 0 && is_contig(x0) && is_contig(w0) && is_contig(s0) &&
               (!biases.ctx || is_contig(*A(biases)));
    rec.run = [](Rec& q) -> int {
        Contig cx, cw, cs, cb;
        if (cx.init(q.a[1]) || cw.init(q.a[2]) || cs.init(q.a[3]) || (q.na > 4 && cb.init(q.a[4]))) return 1;
        return omx_quantized_matmul(q.a[0].ptr(), cx.a->ptr(), cw.a->ptr(), cs.a->ptr(), q.na > 4 ? cb.a->ptr() : nullptr, q.i2, q.i0, q.i1,
                                    q.i3 & 0xFFFF, q.i3 >> 16, to_omx(cs.a->dt), g_stream);
    };
    if (record(std::move(rec))) { delete r; return 1; }
    return assign(res, r);
}
int mlx_gather_qmm(mlx_array* res, const mlx_array x, const mlx_array w, const mlx_array scales, const mlx_array biases,
                   const mlx_array lhs_indices, const mlx_array rhs_indices, bool transpose, mlx_optional_int group_size,
                   mlx_optional_int bits, const char* mode, bool, const mlx_stream) {
    REQ_ARR(x, "mlx_gather_qmm"); REQ_ARR(w, "mlx_gather_qmm"); REQ_ARR(scales, "mlx_gather_qmm"); REQ_ARR(rhs_indices, "mlx_gather_qmm");
    int g, b;
    if (quant_params("mlx_gather_qmm", group_size, bits, mode, &g, &b)) return 1;
    OMX_REQUIRE(transpose && !lhs_indices.ctx, "mlx_gather_qmm: supported form is transpose = true, no lhs_indices (SwitchLinear, model.rs:195-201)");
    Contig cx, cw, cs, cb, ci;
    if (cx.init(*A(x)) || cw.init(*A(w)) || cs.init(*A(scales)) || (biases.ctx && cb.init(*A(biases))) || ci.init(*A(rhs_indices))) return 1;
    OMX_REQUIRE(cw.a->dt == MLX_UINT32 && cw.a->shape.size() == 3, "mlx_gather_qmm: w must be [E, N, K*bits/32] uint32");
    OMX_REQUIRE(ci.a->dt == MLX_UINT32 || ci.a->dt == MLX_INT32, "mlx_gather_qmm: rhs_indices must be (u)int32");
    const int E = cw.a->shape[0], N = cw.a->shape[1], K = cw.a->shape[2] * 32 / b;
    // x [..., 1, K] broadcast against rhs_indices [..., k]: row i of the result uses x row i / x_div
    OMX_REQUIRE(cx.a->shape.size() >= 2 && cx.a->shape.back() == K && cx.a->shape[cx.a->shape.size() - 2] == 1,
                "mlx_gather_qmm: x must be [..., 1, K=%d]", K);
    const size_t n_x = cx.a->size() / K, n = ci.a->size();
    OMX_REQUIRE(n_x > 0 && n % n_x == 0, "mlx_gather_qmm: %zu indices do not broadcast over %zu activation rows", n, n_x);
    std::vector<int> shape = ci.a->shape;
    shape.push_back(1);
    shape.push_back(N);
    OMX_REQUIRE((cx.a->dt == MLX_BFLOAT16 || cx.a->dt == MLX_FLOAT16) && cx.a->dt == cs.a->dt,
                "mlx_gather_qmm: x and scales / biases must both be bfloat16 or both float16");
    NEW_OR_FAIL(r, shape, cx.a->dt);
    if (omx_gather_qmm(r->ptr(), cx.a->ptr(), cw.a->ptr(), cs.a->ptr(), biases.ctx ? cb.a->ptr() : nullptr, (const uint32_t*)ci.a->ptr(),
                       (int)n, (int)(n / n_x), N, K, E, g, b, to_omx(cs.a->dt), g_stream)) { delete r; return 1; }
    return assign(res, r);
}

// ---- elementwise ----
int mlx_add(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary(res, a, b, OP_ADD, "mlx_add"); }
int mlx_subtract(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary(res, a, b, OP_SUB, "mlx_subtract"); }
int mlx_multiply(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary(res, a, b, OP_MUL, "mlx_multiply"); }
int mlx_divide(mlx_array* res, const mlx_array a, const mlx_array b, const mlx_stream) { return binary(res, a, b, OP_DIV, "mlx_divide"); }
int mlx_sigmoid(mlx_array* res, const mlx_array a, const mlx_stream) { return unary(res, a, OP_SIGMOID, a.ctx ? A(a)->dt : MLX_FLOAT32, "mlx_sigmoid"); }
int mlx_exp(mlx_array* res, const mlx_array a, const mlx_stream) { return unary(res, a, OP_EXP, a.ctx ? A(a)->dt : MLX_FLOAT32, "mlx_exp"); }
int mlx_negative(mlx_array* res, const mlx_array a, const mlx_stream) { return unary(res, a, OP_NEG, a.ctx ? A(a)->dt : MLX_FLOAT32, "mlx_negative"); }
int mlx_astype(mlx_array* res, const mlx_array a, mlx_dtype dtype, const mlx_stream) { return unary(res, a, OP_CAST, dtype, "mlx_astype"); }

// ---- shape ops (views where possible) ----
int mlx_reshape(mlx_array* res, const mlx_array a, const int* shape, size_t shape_num, const mlx_stream) {
    REQ_ARR(a, "mlx_reshape");
    const Arr& s = *A(a);
    std::vector<int> sh(shape, shape + shape_num);
    size_t known = 1;
    int infer = -1;
    for (size_t i = 0; i < sh.size(); ++i) {
        if (sh[i] == -1) { OMX_REQUIRE(infer < 0, "mlx_reshape: can only infer one dimension"); infer = (int)i; }
        else known *= (size_t)sh[i];
    }
    if (infer >= 0) {
        OMX_REQUIRE(known != 0 && s.size() % known == 0, "mlx_reshape: cannot infer the missing dimension");
        sh[infer] = (int)(s.size() / known);
        known *= (size_t)sh[infer];
    }
    OMX_REQUIRE(known == s.size(), "mlx_reshape: cannot reshape array of size %zu into the requested shape", s.size());
    Arr* r = nullptr;
    if (is_contig(s)) r = new Arr(s); else if (contiguous(s, &r)) return 1;
    r->host.clear();
    r->shape = sh;
    r->strides = row_major(sh);
    return assign(res, r);
}
int mlx_transpose_axes(mlx_array* res, const mlx_array a, const int* axes, size_t axes_num, const mlx_stream) {
    REQ_ARR(a, "mlx_transpose_axes");
    const Arr& s = *A(a);
    OMX_REQUIRE(axes_num == s.shape.size(), "mlx_transpose_axes: expected %zu axes, got %zu", s.shape.size(), axes_num);
    Arr* r = new Arr(s);
    r->host.clear();
    std::vector<bool> seen(axes_num, false);
    for (size_t i = 0; i < axes_num; ++i) {
        int ax;
        if (norm_axis(axes[i], (int)axes_num, "mlx_transpose_axes", &ax)) { delete r; return 1; }
        if (seen[ax]) { delete r; return set_error("mlx_transpose_axes: repeated axis %d", ax); }
        seen[ax] = true;
        r->shape[i] = s.shape[ax];
        r->strides[i] = s.strides[ax];
    }
    return assign(res, r);
}
int mlx_transpose(mlx_array* res, const mlx_array a, const mlx_stream s) {
    REQ_ARR(a, "mlx_transpose");
    const int nd = (int)A(a)->shape.size();
    std::vector<int> axes(nd);
    for (int i = 0; i < nd; ++i) axes[i] = nd - 1 - i;
    return mlx_transpose_axes(res, a, axes.data(), axes.size(), s);
}
int mlx_expand_dims(mlx_array* res, const mlx_array a, int axis, const mlx_stream) {
    REQ_ARR(a, "mlx_expand_dims");
    const Arr& s = *A(a);
    int ax;
    if (norm_axis(axis, (int)s.shape.size() + 1, "mlx_expand_dims", &ax)) return 1;
    Arr* r = new Arr(s);
    r->host.clear();
    r->shape.insert(r->shape.begin() + ax, 1);
    r->strides.insert(r->strides.begin() + ax, ax < (int)s.shape.size() ? s.strides[ax] * (size_t)s.shape[ax] : 1);
    return assign(res, r);
}
int mlx_contiguous(mlx_array* res, const mlx_array a, bool, const mlx_stream) {
    REQ_ARR(a, "mlx_contiguous");
    Arr* r = nullptr;
    if (is_contig(*A(a))) { r = new Arr(*A(a)); r->host.clear(); }
    else if (contiguous(*A(a), &r)) return 1;
    return assign(res, r);
}
static int slice_view(const Arr& s, const int* start, size_t ns, const int* stop, size_t ne, const int* strides, size_t nst,
                      const char* name, Arr* view) {
    const size_t nd = s.shape.size();
    OMX_REQUIRE(ns == nd && ne == nd && (nst == nd || nst == 0), "%s: start/stop/strides must have one entry per dimension (%zu)", name, nd);
    *view = s;
    view->host.clear();
    for (size_t i = 0; i < nd; ++i) {
        const int st = nst ? strides[i] : 1;
        OMX_REQUIRE(st >= 1, "%s: only positive strides are supported", name);
        int b = start[i] < 0 ? start[i] + s.shape[i] : start[i];
        int e = stop[i] < 0 ? stop[i] + s.shape[i] : stop[i];
        b = b < 0 ? 0 : (b > s.shape[i] ? s.shape[i] : b);
        e = e < b ? b : (e > s.shape[i] ? s.shape[i] : e);
        view->off += (size_t)b * s.strides[i] * dsize(s.dt);
        view->shape[i] = (e - b + st - 1) / st;
        view->strides[i] = s.strides[i] * (size_t)st;
    }
    return 0;
}
int mlx_slice(mlx_array* res, const mlx_array a, const int* start, size_t start_num, const int* stop, size_t stop_num,
              const int* strides, size_t strides_num, const mlx_stream) {
    REQ_ARR(a, "mlx_slice");
    Arr* r = new Arr();
    if (slice_view(*A(a), start, start_num, stop, stop_num, strides, strides_num, "mlx_slice", r)) { delete r; return 1; }
    return assign(res, r);
}
int mlx_slice_update(mlx_array* res, const mlx_array src, const mlx_array update, const int* start, size_t start_num,
                     const int* stop, size_t stop_num, const int* strides, size_t strides_num, const mlx_stream) {
    REQ_ARR(src, "mlx_slice_update"); REQ_ARR(update, "mlx_slice_update");
    const Arr& s = *A(src);
    OMX_REQUIRE(s.dt == A(update)->dt, "mlx_slice_update: dtype mismatch");
    // Functional semantics: the result is a new array.  MLX reaches the KVCache pattern `k = k.slice_update(..)`
    // (cache.rs:183-188; mlx-rs's index_mut writes the result into a fresh handle and drops the old one afterwards)
    // without copying through buffer donation at evaluation time.  This eager implementation decides at call time:
    // when `src` is the ONLY owner of a contiguous buffer the update is done in place and the buffer moves to the
    // result; `src` is then marked donated -- freeing or overwriting it is fine, READING it again is an error, never a
    // silently mutated value.  A caller that wants to keep `src` holds a second reference (mlx_array_set) first,
    // which makes the update copy.  (include/omx_mlx_c.h states the contract.)
    Arr* r = nullptr;
    bool donate = false;
    // solely owned: no other handle, view or vector entry shares the buffer.  References held by recorded ops do not count -- they run in call
    // order, an earlier reader still sees the old rows -- or every step of Generate::next would copy the cache while the previous step is in flight
    if (is_contig(s) && s.buf.use_count() - s.buf->inflight.load() == 1 && s.buf->owned) {   // (a borrowed buffer is the caller's: never updated in place)
        r = new Arr(s);
        r->host.clear();
        donate = !(res && res->ctx == src.ctx);   // the same handle is replaced by assign() below: nothing left to mark
    } else if (contiguous(s, &r)) {
        return 1;
    }
    Arr region;
    if (slice_view(*r, start, start_num, stop, stop_num, strides, strides_num, "mlx_slice_update", &region)) { delete r; return 1; }
    // broadcast-free form: update must have the region's element count (leading 1s allowed)
    if (A(update)->size() != region.size()) { delete r; return set_error("mlx_slice_update: update has %zu elements, the slice has %zu", A(update)->size(), region.size()); }
    Rec rec;
    rec.kind = RK_SLICE_UPDATE;
    rec.a[0] = region; rec.a[1] = *A(update); rec.na = 2;     // (a[0] shares the result's buffer: the write is what this record is)
    rec.run = [](Rec& q) -> int {
        Contig cu;
        if (cu.init(q.a[1])) return 1;
        Arr upd = *cu.a;
        upd.shape = q.a[0].shape;
        upd.strides = row_major(q.a[0].shape);
        return scatter_into(q.a[0].ptr(), q.a[0].strides, upd, q.a[0].dt);
    };
    if (record(std::move(rec))) { delete r; return 1; }
    if (donate) A(src)->donated = true;
    return assign(res, r);
}
int mlx_concatenate_axis(mlx_array* res, const mlx_vector_array arrays, int axis, const mlx_stream) {
    OMX_REQUIRE(arrays.ctx, "mlx_concatenate_axis: empty vector");
    Vec* v = reinterpret_cast<Vec*>(arrays.ctx);
    OMX_REQUIRE(!v->v.empty(), "mlx_concatenate_axis: no arrays to concatenate");
    const Arr& f = *v->v[0];
    int ax;
    if (norm_axis(axis, (int)f.shape.size(), "mlx_concatenate_axis", &ax)) return 1;
    std::vector<int> shape = f.shape;
    shape[ax] = 0;
    for (Arr* a : v->v) {
        OMX_REQUIRE(a->shape.size() == f.shape.size() && a->dt == f.dt, "mlx_concatenate_axis: rank or dtype mismatch");
        for (size_t i = 0; i < f.shape.size(); ++i)
            OMX_REQUIRE((int)i == ax || a->shape[i] == f.shape[i], "mlx_concatenate_axis: shapes differ outside the concatenation axis");
        shape[ax] += a->shape[ax];
    }
    NEW_OR_FAIL(r, shape, f.dt);
    size_t at = 0;
    for (Arr* a : v->v) {
        Contig c;
        if (c.init(*a)) { delete r; return 1; }
        char* base = r->ptr() + at * r->strides[ax] * dsize(f.dt);
        if (scatter_into(base, r->strides, *c.a, f.dt)) { delete r; return 1; }
        at += (size_t)a->shape[ax];
    }
    return assign(res, r);
}
int mlx_zeros(mlx_array* res, const int* shape, size_t shape_num, mlx_dtype dtype, const mlx_stream) {
    std::vector<int> sh(shape, shape + shape_num);
    NEW_OR_FAIL(r, sh, dtype);
    if (r->size()) OMX_HIP_CHECK(hipMemsetAsync(r->ptr(), 0, r->size() * dsize(dtype), g_stream));
    return assign(res, r);
}
int take_axis_general(mlx_array* res, const Arr& a, const Arr& ind, int ax);   // mlxc_glue.hpp
int mlx_take_axis(mlx_array* res, const mlx_array a, const mlx_array indices, int axis, const mlx_stream) {
    REQ_ARR(a, "mlx_take_axis"); REQ_ARR(indices, "mlx_take_axis");
    int ax;
    if (norm_axis(axis, (int)A(a)->shape.size(), "mlx_take_axis", &ax)) return 1;
    OMX_REQUIRE(A(indices)->dt == MLX_UINT32 || A(indices)->dt == MLX_INT32, "mlx_take_axis: indices must be (u)int32");
    if (!(ax == 0 && A(a)->shape.size() == 2 && is_float(A(a)->dt))) return take_axis_general(res, *A(a), *A(indices), ax);
    // row gather from a 2-D floating table (Embedding): 16-byte vector rows
    std::vector<int> shape = A(indices)->shape;
    shape.push_back(A(a)->shape[1]);
    NEW_OR_FAIL(r, shape, A(a)->dt);
    Rec rec;
    rec.a[0] = *r; rec.a[1] = *A(a); rec.a[2] = *A(indices); rec.na = 3;
    rec.run = [](Rec& q) -> int {
        Contig ct, ci;
        if (ct.init(q.a[1]) || ci.init(q.a[2])) return 1;
        return omx_take_rows(q.a[0].ptr(), ct.a->ptr(), (const uint32_t*)ci.a->ptr(), (int64_t)ci.a->size(), ct.a->shape[1], to_omx(ct.a->dt), g_stream);
    };
    if (record(std::move(rec))) { delete r; return 1; }
    return assign(res, r);
}
int mlx_argmax_axis(mlx_array* res, const mlx_array a, int axis, bool keepdims, const mlx_stream) {
    REQ_ARR(a, "mlx_argmax_axis");
    const Arr& s = *A(a);
    int ax;
    if (norm_axis(axis, (int)s.shape.size(), "mlx_argmax_axis", &ax)) return 1;
    // any axis: reduce over a VIEW with that axis moved last (the hot path's call is the last axis, sampler.rs:11)
    Arr moved = s;
    moved.shape.erase(moved.shape.begin() + ax); moved.shape.push_back(s.shape[ax]);
    moved.strides.erase(moved.strides.begin() + ax); moved.strides.push_back(s.strides[ax]);
    std::vector<int> shape(moved.shape.begin(), moved.shape.end() - 1);
    if (keepdims) shape.insert(shape.begin() + ax, 1);
    NEW_OR_FAIL(r, shape, MLX_UINT32);
    Rec rec;
    rec.kind = RK_ARGMAX;
    rec.a[0] = *r; rec.a[1] = moved; rec.na = 2;
    // one contiguous bf16 row: what the lm_head GEMV's argmax epilogue produces
    rec.flag = s.dt == MLX_BFLOAT16 && is_contig(moved) && moved.shape.back() > 0 && s.size() == (size_t)moved.shape.back();
    rec.run = [](Rec& q) -> int {
        Contig c;
        if (c.init(q.a[1])) return 1;
        const int n = q.a[1].shape.back();
        return omx_argmax((uint32_t*)q.a[0].ptr(), c.a->ptr(), n ? (int64_t)(q.a[1].size() / n) : 0, n, to_omx(q.a[1].dt), g_stream);
    };
    if (record(std::move(rec))) { delete r; return 1; }
    return assign(res, r);
}
// ---- random.h: keyed generator + categorical (sampler.rs:13-16 through mlx-rs/src/random.rs) ----
// `key` may be an empty handle: MLX then draws the next key of its own global sequence (seeded by mlx_random_seed,
// default seed 0 here; MLX seeds it from the clock).  mlx-rs never relies on that: it keeps its RandomState on
// the Rust side and always passes a key (random.rs:57-68).
static Arr* g_rng_state = nullptr;
static int seed_global(uint64_t seed) {
    if (!g_rng_state) {
        g_rng_state = new_arr({2}, MLX_UINT32);
        if (!g_rng_state) return set_error("out of device memory allocating the random state");
    }
    return omx_random_key((uint32_t*)g_rng_state->ptr(), seed, g_stream);
}
struct KeyArg {
    Contig c;
    Arr* drawn = nullptr;
    const uint32_t* ptr = nullptr;
    ~KeyArg() { delete drawn; }
    int init(const mlx_array key, const char* name) {
        if (key.ctx) {
            const Arr& k = *A(key);
            OMX_REQUIRE(k.dt == MLX_UINT32 && k.size() == 2, "%s: a PRNG key is 2 x uint32", name);
            if (c.init(k)) return 1;
            ptr = (const uint32_t*)c.a->ptr();
            return 0;
        }
        if (!g_rng_state && seed_global(0)) return 1;
        drawn = new_arr({2, 2}, MLX_UINT32);
        if (!drawn) return set_error("%s: out of device memory", name);
        if (omx_random_split((uint32_t*)drawn->ptr(), (const uint32_t*)g_rng_state->ptr(), 2, g_stream)) return 1;
        OMX_HIP_CHECK(hipMemcpyAsync(g_rng_state->ptr(), drawn->ptr(), 8, hipMemcpyDeviceToDevice, g_stream));
        ptr = (const uint32_t*)drawn->ptr() + 2;
        return 0;
    }
};
int mlx_random_seed(uint64_t seed) { return seed_global(seed); }
int mlx_random_key(mlx_array* res, uint64_t seed) {
    NEW_OR_FAIL(r, std::vector<int>({2}), MLX_UINT32);
    if (omx_random_key((uint32_t*)r->ptr(), seed, g_stream)) { delete r; return 1; }
    return assign(res, r);
}
int mlx_random_split_num(mlx_array* res, const mlx_array key, int num, const mlx_stream) {
    REQ_ARR(key, "mlx_random_split_num");
    OMX_REQUIRE(num >= 1, "mlx_random_split_num: num=%d must be positive", num);
    KeyArg k;
    if (k.init(key, "mlx_random_split_num")) return 1;
    NEW_OR_FAIL(r, std::vector<int>({num, 2}), MLX_UINT32);
    if (omx_random_split((uint32_t*)r->ptr(), k.ptr, num, g_stream)) { delete r; return 1; }
    return assign(res, r);
}
int mlx_random_split(mlx_array* res_0, mlx_array* res_1, const mlx_array key, const mlx_stream s) {
    mlx_array both = mlx_array_new();
    if (mlx_random_split_num(&both, key, 2, s)) return 1;
    const int rc = [&]() {
        for (int i = 0; i < 2; ++i) {
            Arr* r = new Arr(*A(both));          // view of row i
            r->host.clear();
            r->shape = {2};
            r->strides = {1};
            r->off = A(both)->off + (size_t)i * 8;
            if (assign(i == 0 ? res_0 : res_1, r)) return 1;
        }
        return 0;
    }();
    mlx_array_free(both);
    return rc;
}
int mlx_random_bits(mlx_array* res, const int* shape, size_t shape_num, int width, const mlx_array key, const mlx_stream) {
    OMX_REQUIRE(width == 4, "mlx_random_bits: only 4-byte words are supported (width %d)", width);
    KeyArg k;
    if (k.init(key, "mlx_random_bits")) return 1;
    NEW_OR_FAIL(r, std::vector<int>(shape, shape + shape_num), MLX_UINT32);
    if (omx_random_bits((uint32_t*)r->ptr(), k.ptr, (int64_t)r->size(), g_stream)) { delete r; return 1; }
    return assign(res, r);
}
int mlx_random_uniform(mlx_array* res, const mlx_array low, const mlx_array high, const int* shape, size_t shape_num,
                       mlx_dtype dtype, const mlx_array key, const mlx_stream) {
    REQ_ARR(low, "mlx_random_uniform"); REQ_ARR(high, "mlx_random_uniform");
    OMX_REQUIRE(dtype == MLX_FLOAT32, "mlx_random_uniform: float32 only");
    OMX_REQUIRE(A(low)->size() == 1 && A(high)->size() == 1, "mlx_random_uniform: scalar bounds only");
    float lo = 0.f, hi = 1.f;
    mlx_array lf = mlx_array_new(), hf = mlx_array_new();
    int rc = mlx_astype(&lf, low, MLX_FLOAT32, mlx_stream{nullptr}) || mlx_astype(&hf, high, MLX_FLOAT32, mlx_stream{nullptr}) ||
             item_host(lf, &lo, MLX_FLOAT32, "mlx_random_uniform") || item_host(hf, &hi, MLX_FLOAT32, "mlx_random_uniform");
    mlx_array_free(lf); mlx_array_free(hf);
    if (rc) return 1;
    KeyArg k;
    if (k.init(key, "mlx_random_uniform")) return 1;
    NEW_OR_FAIL(r, std::vector<int>(shape, shape + shape_num), MLX_FLOAT32);
    if (omx_random_uniform((float*)r->ptr(), k.ptr, (int64_t)r->size(), lo, hi, g_stream)) { delete r; return 1; }
    return assign(res, r);
}
int mlx_random_gumbel(mlx_array* res, const int* shape, size_t shape_num, mlx_dtype dtype, const mlx_array key, const mlx_stream) {
    OMX_REQUIRE(dtype == MLX_FLOAT32, "mlx_random_gumbel: float32 only");
    KeyArg k;
    if (k.init(key, "mlx_random_gumbel")) return 1;
    NEW_OR_FAIL(r, std::vector<int>(shape, shape + shape_num), MLX_FLOAT32);
    if (omx_random_gumbel((float*)r->ptr(), k.ptr, (int64_t)r->size(), g_stream)) { delete r; return 1; }
    return assign(res, r);
}
int mlx_random_normal(mlx_array* res, const int* shape, size_t shape_num, mlx_dtype dtype, float loc, float scale, const mlx_array key,
                      const mlx_stream) {
    OMX_REQUIRE(dtype == MLX_FLOAT32, "mlx_random_normal: float32 only");
    KeyArg k;
    if (k.init(key, "mlx_random_normal")) return 1;
    NEW_OR_FAIL(r, std::vector<int>(shape, shape + shape_num), MLX_FLOAT32);
    if (omx_random_normal((float*)r->ptr(), k.ptr, (int64_t)r->size(), loc, scale, g_stream)) { delete r; return 1; }
    return assign(res, r);
}
static int categorical_impl(mlx_array* res, const mlx_array logits, int axis, int num_samples, bool keep_samples_dim,
                            const mlx_array key, const char* name) {
    REQ_ARR(logits, name);
    const Arr& s = *A(logits);
    OMX_REQUIRE(is_float(s.dt), "%s: logits must be floating point", name);
    OMX_REQUIRE(!s.shape.empty(), "%s: logits need at least one axis", name);
    int ax;
    if (norm_axis(axis, (int)s.shape.size(), name, &ax)) return 1;
    OMX_REQUIRE(ax == (int)s.shape.size() - 1, "%s: only the last axis is supported (sampler.rs:15)", name);
    Contig c;
    if (c.init(s)) return 1;
    KeyArg k;
    if (k.init(key, name)) return 1;
    std::vector<int> shape(s.shape.begin(), s.shape.end() - 1);
    if (keep_samples_dim) shape.push_back(num_samples);
    NEW_OR_FAIL(r, shape, MLX_UINT32);
    const int n = s.shape.back();
    OMX_REQUIRE(n > 0, "%s: empty distribution axis", name);
    if (omx_random_categorical((uint32_t*)r->ptr(), c.a->ptr(), (int64_t)(s.size() / n), n, num_samples, 1.0f, k.ptr, to_omx(s.dt),
                               g_stream)) { delete r; return 1; }
    return assign(res, r);
}
int mlx_random_categorical(mlx_array* res, const mlx_array logits, int axis, const mlx_array key, const mlx_stream) {
    return categorical_impl(res, logits, axis, 1, false, key, "mlx_random_categorical");
}
int mlx_random_categorical_num_samples(mlx_array* res, const mlx_array logits, int axis, int num_samples, const mlx_array key,
                                       const mlx_stream) {
    OMX_REQUIRE(num_samples >= 1, "mlx_random_categorical_num_samples: num_samples=%d must be positive", num_samples);
    return categorical_impl(res, logits, axis, num_samples, true, key, "mlx_random_categorical_num_samples");
}
int mlx_random_categorical_shape(mlx_array* res, const mlx_array logits, int axis, const int* shape, size_t shape_num,
                                 const mlx_array key, const mlx_stream) {
    REQ_ARR(logits, "mlx_random_categorical_shape");
    const Arr& s = *A(logits);
    bool same = shape_num + 1 == s.shape.size();
    for (size_t i = 0; same && i < shape_num; ++i) same = shape[i] == s.shape[i];
    OMX_REQUIRE(same, "mlx_random_categorical_shape: only shape == logits.shape without the axis is supported");
    return categorical_impl(res, logits, axis, 1, false, key, "mlx_random_categorical_shape");
}
int mlx_softmax_axis(mlx_array* res, const mlx_array a, int axis, bool, const mlx_stream) {
    REQ_ARR(a, "mlx_softmax_axis");
    const Arr& s = *A(a);
    int ax;
    if (norm_axis(axis, (int)s.shape.size(), "mlx_softmax_axis", &ax)) return 1;
    OMX_REQUIRE(is_float(s.dt), "mlx_softmax_axis: floating arrays only");
    const int nd = (int)s.shape.size();
    // any axis: normalise over a VIEW with that axis moved last; the result is handed back as the inverse view of the row-major
    // buffer (MLX returns a fresh array either way, strides are not part of the contract)
    Arr moved = s;
    moved.shape.erase(moved.shape.begin() + ax); moved.shape.push_back(s.shape[ax]);
    moved.strides.erase(moved.strides.begin() + ax); moved.strides.push_back(s.strides[ax]);
    Contig c;
    if (c.init(moved)) return 1;
    NEW_OR_FAIL(r, moved.shape, s.dt);
    const int n = moved.shape.back();
    const size_t rows = n ? s.size() / n : 0;
    if (rows) {
        softmax_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, g_stream>>>(r->ptr(), c.a->ptr(), s.dt, rows, n);
        OMX_LAUNCH_CHECK();
    }
    if (ax != nd - 1) {   // view the [.., others.., axis] buffer as the input's shape again
        const std::vector<size_t> st = r->strides;
        r->shape = s.shape;
        r->strides.assign(nd, 0);
        for (int d = 0, k = 0; d < nd; ++d) r->strides[d] = d == ax ? st[nd - 1] : st[k++];
    }
    return assign(res, r);
}

int omx_mlx_fused_swiglu(mlx_array* res, const mlx_array x, const mlx_array gate, const mlx_stream) {
    REQ_ARR(x, "fused_swiglu"); REQ_ARR(gate, "fused_swiglu");
    OMX_REQUIRE(A(x)->shape == A(gate)->shape && A(x)->dt == A(gate)->dt, "fused_swiglu: x and gate must match in shape and dtype");
    Contig cx, cg;
    if (cx.init(*A(x)) || cg.init(*A(gate))) return 1;
    NEW_OR_FAIL(r, cx.a->shape, cx.a->dt);
    if (omx_fused_swiglu(r->ptr(), cx.a->ptr(), cg.a->ptr(), (int64_t)r->size(), to_omx(r->dt), g_stream)) { delete r; return 1; }
    return assign(res, r);
}
int omx_mlx_fused_modulate(mlx_array* res, const mlx_array x, const mlx_array shift, const mlx_array scale, const mlx_stream) {
    REQ_ARR(x, "fused_modulate"); REQ_ARR(shift, "fused_modulate"); REQ_ARR(scale, "fused_modulate");
    OMX_REQUIRE(A(x)->shape.size() == 3, "fused_modulate: x must be [B, S, H]");
    Contig cx, csh, csc;
    if (cx.init(*A(x)) || csh.init(*A(shift)) || csc.init(*A(scale))) return 1;
    const int B = cx.a->shape[0], S = cx.a->shape[1], H = cx.a->shape[2];
    OMX_REQUIRE(csh.a->size() == (size_t)B * H && csc.a->size() == (size_t)B * H, "fused_modulate: shift/scale must be [B, H]");
    NEW_OR_FAIL(r, cx.a->shape, cx.a->dt);
    if (omx_fused_modulate(r->ptr(), cx.a->ptr(), csh.a->ptr(), csc.a->ptr(), B, S, H, 1e-6f, to_omx(r->dt), g_stream)) { delete r; return 1; }
    return assign(res, r);
}

}  // extern "C"

#include "mlxc_glue.hpp"   // devices, strings, maps, closures and the remaining glue ops (same translation unit: shares Arr / Vec)
#include "mlxc_glue2.hpp"  // ops.h, third batch: composed from the ops above
