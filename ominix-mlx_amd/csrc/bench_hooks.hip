// Measurement hooks (not part of the drop-in surface): time one kernel family with HIP events on
// the stream it is launched on.  Weight buffers are rotated through `n_copies` distinct
// allocations so that the 256 MiB Infinity Cache cannot serve re-reads (MI355X_MICROARCH.md).
#include <vector>

#include "gemm.hpp"
#include "gemv.hpp"

extern "C" int omx_bench_gemv(int N, int K, int pro, int epi, int rows_per_wave, int n_copies, int iters,
                              float* avg_ms) {
    using namespace omx;
    OMX_REQUIRE(avg_ms && n_copies > 0 && iters > 0, "omx_bench_gemv: bad arguments");
    const int mats = (epi == EPI_SWIGLU) ? 2 : 1;
    const size_t wbytes = (size_t)N * K * 2 * mats;
    std::vector<void*> w(n_copies, nullptr);
    void *x = nullptr, *nw = nullptr, *out = nullptr, *resid = nullptr, *slot = nullptr;
    for (auto& p : w) {
        OMX_HIP_CHECK(hipMalloc(&p, wbytes));
        if (omx_fill_uniform(p, wbytes / 2, 17u + (uint32_t)(&p - &w[0]), 0.03f, 0.f, OMX_BFLOAT16, nullptr)) return 1;
    }
    OMX_HIP_CHECK(hipMalloc(&x, (size_t)K * 2));
    OMX_HIP_CHECK(hipMalloc(&nw, (size_t)K * 2));
    OMX_HIP_CHECK(hipMalloc(&out, (size_t)N * 4));
    OMX_HIP_CHECK(hipMalloc(&resid, (size_t)N * 2));
    OMX_HIP_CHECK(hipMalloc(&slot, 8 * 65536));
    omx_fill_uniform(x, K, 3, 1.0f, 0.f, OMX_BFLOAT16, nullptr);
    omx_fill_uniform(nw, K, 4, 0.1f, 1.f, OMX_BFLOAT16, nullptr);
    omx_fill_uniform(resid, N, 5, 1.0f, 0.f, OMX_BFLOAT16, nullptr);
    OMX_HIP_CHECK(hipMemset(slot, 0, 8 * 65536));
    hipStream_t s;
    OMX_HIP_CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    OMX_HIP_CHECK(hipEventCreate(&e0));
    OMX_HIP_CHECK(hipEventCreate(&e1));
    auto run = [&](int i) {
        GemvArgs a = {};
        const bf16_t* base = (const bf16_t*)w[i % n_copies];
        a.w0 = base;
        a.w1 = base + (size_t)N * K;
        a.n0 = N; a.N = N; a.K = K;
        a.x = (const bf16_t*)x;
        a.norm_w = (const bf16_t*)nw;
        a.eps = 1e-6f;
        a.resid = (const bf16_t*)resid;
        a.out = out;
        a.argmax_slot = (unsigned long long*)slot;
        a.rows_per_wave = rows_per_wave;
        return launch_gemv(a, pro, epi, s);
    };
    for (int i = 0; i < n_copies + 2; ++i)
        if (run(i)) return 1;
    OMX_HIP_CHECK(hipStreamSynchronize(s));
    OMX_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        if (run(i)) return 1;
    OMX_HIP_CHECK(hipEventRecord(e1, s));
    OMX_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    OMX_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    for (auto p : w) (void)hipFree(p);
    (void)hipFree(x); (void)hipFree(nw); (void)hipFree(out); (void)hipFree(resid); (void)hipFree(slot);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    return 0;
}

// per-block timeline of `chain` back-to-back launches of one GEMV shape on distinct weight buffers (tools/gemv_trace.py):
// host receives [chain][blocks][4] wall-clock stamps (gemv.hpp GemvArgs::trace); *blocks = blocks per launch
extern "C" int omx_bench_gemv_trace(int N, int K, int pro, int epi, int rows_per_wave, int chain, unsigned long long* host,
                                    size_t n_words, int* blocks) {
    using namespace omx;
    OMX_REQUIRE(host && blocks && chain > 0, "omx_bench_gemv_trace: bad arguments");
    const int mats = (epi == EPI_SWIGLU) ? 2 : 1;
    const size_t wbytes = (size_t)N * K * 2 * mats;
    const int nb = gemv_grid(N, K, epi, rows_per_wave);
    OMX_REQUIRE(n_words >= (size_t)chain * nb * 4, "omx_bench_gemv_trace: buffer too small (%d blocks)", nb);
    std::vector<void*> w(chain + 2, nullptr);
    void *x = nullptr, *nw = nullptr, *out = nullptr, *resid = nullptr, *slot = nullptr;
    unsigned long long* tr = nullptr;
    for (auto& p : w) {
        OMX_HIP_CHECK(hipMalloc(&p, wbytes));
        if (omx_fill_uniform(p, wbytes / 2, 17u + (uint32_t)(&p - &w[0]), 0.03f, 0.f, OMX_BFLOAT16, nullptr)) return 1;
    }
    OMX_HIP_CHECK(hipMalloc(&x, (size_t)K * 2));
    OMX_HIP_CHECK(hipMalloc(&nw, (size_t)K * 2));
    OMX_HIP_CHECK(hipMalloc(&out, (size_t)N * 4));
    OMX_HIP_CHECK(hipMalloc(&resid, (size_t)N * 2));
    OMX_HIP_CHECK(hipMalloc(&slot, 8 * 65536));
    OMX_HIP_CHECK(hipMalloc((void**)&tr, (size_t)chain * nb * 4 * 8));
    omx_fill_uniform(x, K, 3, 1.0f, 0.f, OMX_BFLOAT16, nullptr);
    omx_fill_uniform(nw, K, 4, 0.1f, 1.f, OMX_BFLOAT16, nullptr);
    omx_fill_uniform(resid, N, 5, 1.0f, 0.f, OMX_BFLOAT16, nullptr);
    OMX_HIP_CHECK(hipMemset(slot, 0, 8 * 65536));
    OMX_HIP_CHECK(hipMemset(tr, 0, (size_t)chain * nb * 4 * 8));
    OMX_HIP_CHECK(hipDeviceSynchronize());
    hipStream_t s;
    OMX_HIP_CHECK(hipStreamCreate(&s));
    auto run = [&](int i, unsigned long long* t) {
        GemvArgs a = {};
        const bf16_t* base = (const bf16_t*)w[i % w.size()];
        a.w0 = base; a.w1 = base + (size_t)N * K;
        a.n0 = N; a.N = N; a.K = K;
        a.x = (const bf16_t*)x; a.norm_w = (const bf16_t*)nw; a.eps = 1e-6f;
        a.resid = (const bf16_t*)resid; a.out = out;
        a.argmax_slot = (unsigned long long*)slot;
        a.rows_per_wave = rows_per_wave;
        a.trace = t;
        return launch_gemv(a, pro, epi, s);
    };
    for (int i = 0; i < 2; ++i)
        if (run(i, nullptr)) return 1;
    for (int i = 0; i < chain; ++i)
        if (run(2 + i, tr + (size_t)i * nb * 4)) return 1;
    OMX_HIP_CHECK(hipStreamSynchronize(s));
    OMX_HIP_CHECK(hipMemcpy(host, tr, (size_t)chain * nb * 4 * 8, hipMemcpyDeviceToHost));
    *blocks = nb;
    for (auto p : w) (void)hipFree(p);
    (void)hipFree(x); (void)hipFree(nw); (void)hipFree(out); (void)hipFree(resid); (void)hipFree(slot); (void)hipFree(tr);
    (void)hipStreamDestroy(s);
    return 0;
}

// time the bf16 MFMA GEMM out[M,N] = x[M,K] . W[N,K]^T with HIP events (operands rotated over n_copies buffers)
extern "C" int omx_bench_gemm(int M, int N, int K, int n_copies, int iters, float* avg_ms) {
    using namespace omx;
    OMX_REQUIRE(avg_ms && n_copies > 0 && iters > 0, "omx_bench_gemm: bad arguments");
    std::vector<void*> w(n_copies, nullptr), x(n_copies, nullptr);
    void* out = nullptr;
    for (int i = 0; i < n_copies; ++i) {
        OMX_HIP_CHECK(hipMalloc(&w[i], (size_t)N * K * 2));
        OMX_HIP_CHECK(hipMalloc(&x[i], (size_t)M * K * 2));
        // OMX_BENCH_GEMM_DATA=zero | one: constant operands -- what the same instruction stream costs when no operand bit toggles (the chip's
        // power limit, not the kernel, sets the pace of a full-chip GEMM: EXPERIMENTS.md R5-4)
        const char* de = getenv("OMX_BENCH_GEMM_DATA");
        if (de && (de[0] == 'z' || de[0] == 'o')) {
            const int v = de[0] == 'z' ? 0 : 0x3f;      // bytes 0x3f3f = bf16 0.746
            OMX_HIP_CHECK(hipMemset(w[i], v, (size_t)N * K * 2));
            OMX_HIP_CHECK(hipMemset(x[i], v, (size_t)M * K * 2));
            continue;
        }
        if (omx_fill_uniform(w[i], (size_t)N * K, 100 + i, 0.05f, 0.f, OMX_BFLOAT16, nullptr)) return 1;
        if (omx_fill_uniform(x[i], (size_t)M * K, 200 + i, 1.0f, 0.f, OMX_BFLOAT16, nullptr)) return 1;
    }
    OMX_HIP_CHECK(hipMalloc(&out, (size_t)M * N * 2));
    hipStream_t s;
    OMX_HIP_CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    OMX_HIP_CHECK(hipEventCreate(&e0));
    OMX_HIP_CHECK(hipEventCreate(&e1));
    OMX_HIP_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < 3; ++i)
        if (launch_gemm_bf16((bf16_t*)out, (const bf16_t*)x[i % n_copies], (const bf16_t*)w[i % n_copies], nullptr, M, N, K, s)) return 1;
    OMX_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        if (launch_gemm_bf16((bf16_t*)out, (const bf16_t*)x[i % n_copies], (const bf16_t*)w[i % n_copies], nullptr, M, N, K, s)) return 1;
    OMX_HIP_CHECK(hipEventRecord(e1, s));
    OMX_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    OMX_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    for (int i = 0; i < n_copies; ++i) { (void)hipFree(w[i]); (void)hipFree(x[i]); }
    (void)hipFree(out);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    return 0;
}

// ---- device-wide barrier probe (gridsync.hpp): average cost of one exchange + barrier over `iters` ----
#include "gridsync.hpp"
namespace {
// variant 0: gridsync.hpp as shipped (flags + coherent accessors, no fences)
// variant 1: the textbook form it replaced: one atomic counter, agent-scope release/acquire fences
template <int VAR>
__global__ __launch_bounds__(256) void grid_barrier_probe_kernel(unsigned* words, unsigned* counter, unsigned* mismatches,
                                                                 int iters, float* sink) {
    omx::GridSync g{words, 1u, gridDim.x, false};
    const unsigned nb = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        // double-buffered exchange: read what the neighbour wrote LAST round, write this round's value
        float* src = sink + ((size_t)((i + 1) & 1) * nb + (blockIdx.x + 1) % nb) * 256 + threadIdx.x;
        float* dst = sink + ((size_t)(i & 1) * nb + blockIdx.x) * 256 + threadIdx.x;
        if (VAR == 0) {
            if (omx::ld_coh_f32(src) != (float)i) atomicAdd(mismatches, 1u);
            omx::st_coh_f32(dst, (float)(i + 1));
            omx::grid_sync(g);
        } else {
            if (*src != (float)i) atomicAdd(mismatches, 1u);
            *dst = (float)(i + 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(i + 1) * nb)
                    __builtin_amdgcn_s_sleep(2);
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
}
// variant 2: XCD-LOCAL groups.  Blocks are dispatched to XCDs round-robin (block b runs on XCD b % 8), so the blocks with
// the same b % 8 share one L2.  They exchange and synchronise through that L2 only: workgroup-scope relaxed atomics lower to
// `global_load/store ... sc0`, which miss the per-CU vector cache but are served by the XCD's L2 -- no write-through to
// memory, no re-validation against the other XCDs.  The question this answers: what does a barrier + a 1 KiB hand-over cost
// when a dependent phase is kept inside one XCD (a tensor-parallel-over-XCDs decode step would need only two device-wide
// exchanges per layer instead of five)?  A wrong placement assumption shows up as stale reads (failed & 2) or a timeout.
// (measured: an sc0 LOAD may still be served by the CU's vector cache -- polls never saw the flag.  What works: poll with a
//  read-modify-write atomic, which always executes in the L2, then drop the vector cache with `buffer_inv sc0` before the
//  plain data loads.)
__device__ __forceinline__ unsigned ld_l2_32(const void* p) {
    // written as asm: the compiler folds an idempotent `fetch_or(p, 0)` back into an (sc0) load
    unsigned r;
    const unsigned zero = 0u;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p), "v"(zero) : "memory");
    return r;
}
__device__ __forceinline__ void l1_invalidate() { asm volatile("buffer_inv sc0" ::: "memory"); }
__device__ __forceinline__ void st_l2_32(void* p, unsigned v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ __launch_bounds__(256) void xcd_barrier_probe_kernel(unsigned* words, unsigned* mismatches, int iters, float* sink) {
    const unsigned nb = gridDim.x, xcd = blockIdx.x % 8, member = blockIdx.x / 8, members = nb / 8;
    unsigned* flags = words + 64 + 16 * nb + 32;           // after the device-wide area and the counters: [8 groups][members] flags, 64 B apart
    unsigned* my_group = flags + (size_t)xcd * members * 16;
    unsigned* release = words + 64 + 16 * xcd;             // one release word per group
    bool dead = false;
    for (int i = 0; i < iters; ++i) {
        const unsigned epoch = (unsigned)i + 1;
        const unsigned peer = xcd + 8 * ((member + 1) % members);   // the next block of the same XCD
        float* src = sink + ((size_t)((i + 1) & 1) * nb + peer) * 256 + threadIdx.x;
        float* dst = sink + ((size_t)(i & 1) * nb + blockIdx.x) * 256 + threadIdx.x;
        l1_invalidate();
        if (*reinterpret_cast<volatile float*>(src) != (float)i) atomicAdd(mismatches, 1u);
        *reinterpret_cast<volatile float*>(dst) = (float)(i + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) st_l2_32(my_group + 16 * member, epoch);
        if (member == 0) {
            for (unsigned b = threadIdx.x; b < members; b += blockDim.x) {
                unsigned it = 0;
                while (!dead && (int)(ld_l2_32(my_group + 16 * b) - epoch) < 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++it > (1u << 22)) { dead = true; atomicAdd(mismatches + 1, 1u); }
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) st_l2_32(release, epoch);
        } else if (threadIdx.x == 0) {
            unsigned it = 0;
            while (!dead && (int)(ld_l2_32(release) - epoch) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++it > (1u << 22)) { dead = true; atomicAdd(mismatches + 1, 1u); }
            }
        }
        __syncthreads();
    }
}
}  // namespace

// which XCD (XCC_ID hardware register, gfx942+: hwreg 20, bits 3:0) and CU each block of a plain launch runs on
namespace {
__global__ void xcc_id_kernel(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
}
}  // namespace
extern "C" int omx_bench_xcc_ids(int nblocks, unsigned* host_out) {
    OMX_REQUIRE(nblocks > 0 && host_out, "omx_bench_xcc_ids: bad arguments");
    unsigned* d = nullptr;
    OMX_HIP_CHECK(hipMalloc(&d, (size_t)nblocks * 4));
    xcc_id_kernel<<<nblocks, 256>>>(d);
    OMX_LAUNCH_CHECK();
    OMX_HIP_CHECK(hipMemcpy(host_out, d, (size_t)nblocks * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return 0;
}

extern "C" int omx_bench_grid_barrier(int nblocks, int iters, int variant, float* us_per_barrier, int* failed) {
    using namespace omx;
    OMX_REQUIRE(nblocks > 0 && iters > 0 && us_per_barrier && failed, "omx_bench_grid_barrier: bad arguments");
    unsigned* words = nullptr;
    float* sink = nullptr;
    OMX_REQUIRE(variant != 2 || (nblocks % 8 == 0 && nblocks <= 512), "omx_bench_grid_barrier: the XCD-local variant needs a multiple of 8 blocks (<= 512)");
    const size_t wbytes = (grid_sync_words(nblocks) + 32 + 16 * (size_t)nblocks + 64) * 4;   // + counter, mismatch count, XCD-group flags
    OMX_HIP_CHECK(hipMalloc(&words, wbytes));
    OMX_HIP_CHECK(hipMalloc(&sink, (size_t)nblocks * 512 * 4));
    OMX_HIP_CHECK(hipMemset(words, 0, wbytes));
    OMX_HIP_CHECK(hipMemset(sink, 0, (size_t)nblocks * 512 * 4));
    unsigned* extra = words + grid_sync_words(nblocks);
    hipEvent_t e0, e1;
    OMX_HIP_CHECK(hipEventCreate(&e0));
    OMX_HIP_CHECK(hipEventCreate(&e1));
    OMX_HIP_CHECK(hipDeviceSynchronize());
    OMX_HIP_CHECK(hipEventRecord(e0, nullptr));
    if (variant == 2) xcd_barrier_probe_kernel<<<nblocks, 256>>>(words, extra + 16, iters, sink);
    else if (variant == 0) grid_barrier_probe_kernel<0><<<nblocks, 256>>>(words, extra, extra + 16, iters, sink);
    else grid_barrier_probe_kernel<1><<<nblocks, 256>>>(words, extra, extra + 16, iters, sink);
    OMX_LAUNCH_CHECK();
    OMX_HIP_CHECK(hipEventRecord(e1, nullptr));
    OMX_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    OMX_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned abort_word = 0, mism = 0;
    OMX_HIP_CHECK(hipMemcpy(&abort_word, words + 16, 4, hipMemcpyDeviceToHost));
    OMX_HIP_CHECK(hipMemcpy(&mism, extra + 16, 4, hipMemcpyDeviceToHost));
    *us_per_barrier = ms * 1000.f / iters;
    unsigned timeouts = 0;
    OMX_HIP_CHECK(hipMemcpy(&timeouts, extra + 17, 4, hipMemcpyDeviceToHost));
    *failed = ((abort_word || timeouts) ? 1 : 0) | (mism ? 2 : 0);   // 1: a waiter timed out; 2: an exchange read a stale value
    (void)hipFree(words); (void)hipFree(sink);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

// ---- packed-weight GEMV (quant.hip): one shape, weights rotated over n_copies buffers, HIP events ----
#include "quant.hpp"
extern "C" int omx_bench_qgemv(int N, int K, int bits, int pro, int epi, int n_copies, int iters, float* avg_ms) {
    using namespace omx;
    OMX_REQUIRE(avg_ms && n_copies > 0 && iters > 0, "omx_bench_qgemv: bad arguments");
    const int mats = (epi == EPI_SWIGLU) ? 2 : 1;
    const size_t wwords = (size_t)N * K * bits / 32, ngroups = (size_t)N * K / 64;
    std::vector<uint32_t*> w(n_copies * mats, nullptr);
    std::vector<bf16_t*> sc(n_copies * mats, nullptr), bi(n_copies * mats, nullptr);
    for (size_t i = 0; i < w.size(); ++i) {
        OMX_HIP_CHECK(hipMalloc((void**)&w[i], wwords * 4));
        OMX_HIP_CHECK(hipMalloc((void**)&sc[i], ngroups * 2));
        OMX_HIP_CHECK(hipMalloc((void**)&bi[i], ngroups * 2));
        if (omx_fill_uniform(w[i], wwords * 2, 70u + (uint32_t)i, 1.0f, 0.f, OMX_BFLOAT16, nullptr)) return 1;   // arbitrary nibbles
        if (omx_fill_uniform(sc[i], ngroups, 90u + (uint32_t)i, 0.002f, 0.004f, OMX_BFLOAT16, nullptr)) return 1;
        if (omx_fill_uniform(bi[i], ngroups, 110u + (uint32_t)i, 0.03f, 0.f, OMX_BFLOAT16, nullptr)) return 1;
    }
    void *x = nullptr, *nw = nullptr, *out = nullptr, *resid = nullptr, *slot = nullptr;
    OMX_HIP_CHECK(hipMalloc(&x, (size_t)K * 2));
    OMX_HIP_CHECK(hipMalloc(&nw, (size_t)K * 2));
    OMX_HIP_CHECK(hipMalloc(&out, (size_t)N * 4));
    OMX_HIP_CHECK(hipMalloc(&resid, (size_t)N * 2));
    OMX_HIP_CHECK(hipMalloc(&slot, 8 * 65536));
    omx_fill_uniform(x, K, 3, 1.0f, 0.f, OMX_BFLOAT16, nullptr);
    omx_fill_uniform(nw, K, 4, 0.1f, 1.f, OMX_BFLOAT16, nullptr);
    omx_fill_uniform(resid, N, 5, 1.0f, 0.f, OMX_BFLOAT16, nullptr);
    hipStream_t s;
    OMX_HIP_CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    OMX_HIP_CHECK(hipEventCreate(&e0));
    OMX_HIP_CHECK(hipEventCreate(&e1));
    std::vector<uint32_t*> tiles(w.size(), nullptr);       // the matrix-core kernel's form of the same matrices (qgemv_mfma.hip)
    const char* mfma_env = getenv("OMX_QGEMV_MFMA");
    if (qgemv4m_shape_ok(K, 64, bits) && !(mfma_env && mfma_env[0] == '0'))
        for (size_t i = 0; i < w.size(); ++i) {
            OMX_HIP_CHECK(hipMalloc((void**)&tiles[i], qgemv4m_tile_words(N, K) * 4));
            if (launch_qgemv4m_repack(tiles[i], w[i], sc[i], bi[i], N, K, nullptr)) return 1;
        }
    auto run = [&](int i) {
        QGemvArgs a = {};
        const int c = (i % n_copies) * mats;
        a.m[0] = QMat{w[c], sc[c], bi[c], N};
        a.m[0].tiles = tiles[c];
        if (mats == 2) { a.m[1] = QMat{w[c + 1], sc[c + 1], bi[c + 1], N}; a.m[1].tiles = tiles[c + 1]; }
        a.N = N; a.K = K; a.group = 64;
        a.x = (const bf16_t*)x; a.norm_w = (const bf16_t*)nw; a.eps = 1e-6f; a.resid = (const bf16_t*)resid;
        a.out = (bf16_t*)out; a.argmax_slot = (unsigned long long*)slot;
        return launch_qgemv(a, bits, pro, epi, s);
    };
    OMX_HIP_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < n_copies + 2; ++i)
        if (run(i)) return 1;
    OMX_HIP_CHECK(hipStreamSynchronize(s));
    OMX_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        if (run(i)) return 1;
    OMX_HIP_CHECK(hipEventRecord(e1, s));
    OMX_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    OMX_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    for (size_t i = 0; i < w.size(); ++i) { (void)hipFree(w[i]); (void)hipFree(sc[i]); (void)hipFree(bi[i]); if (tiles[i]) (void)hipFree(tiles[i]); }
    (void)hipFree(x); (void)hipFree(nw); (void)hipFree(out); (void)hipFree(resid); (void)hipFree(slot);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    return 0;
}

/* test hook (tests/test_gpu_quant.py): ONE fused packed-GEMV launch of the decode step on caller-owned device tensors -- the
 * prologue / epilogue forms the engine uses (pro: 0 none, 1 RMSNorm; epi: gemv.hpp EPI_*), which the raw omx_quantized_matmul never
 * reaches.  w1 / s1 / b1: the up matrix of a SwiGLU pair, or null; n0: rows of the first member when w1 stacks below it (q | k). */
extern "C" int omx_debug_qgemv(void* out, float* out_f32, unsigned long long* argmax_slot, const void* x, const void* norm_w, const void* resid,
                               const void* w0, const void* s0, const void* b0, const void* w1, const void* s1, const void* b1, int n0,
                               int N, int K, int group, int bits, int pro, int epi, float eps, int single_round, void* stream) {
    using namespace omx;
    QGemvArgs a = {};
    if (epi == EPI_SWIGLU) {
        a.m[0] = QMat{(const uint32_t*)w0, (const bf16_t*)s0, (const bf16_t*)b0, N};
        a.m[1] = QMat{(const uint32_t*)w1, (const bf16_t*)s1, (const bf16_t*)b1, N};
    } else {
        a.m[0] = QMat{(const uint32_t*)w0, (const bf16_t*)s0, (const bf16_t*)b0, w1 ? n0 : N};
        if (w1) a.m[1] = QMat{(const uint32_t*)w1, (const bf16_t*)s1, (const bf16_t*)b1, N - n0};
    }
    a.N = N; a.K = K; a.group = group;
    a.x = (const bf16_t*)x; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.resid = (const bf16_t*)resid;
    a.out = (bf16_t*)out; a.out_f32 = out_f32; a.argmax_slot = argmax_slot; a.swiglu_single_round = single_round;
    uint32_t* tiles[2] = {nullptr, nullptr};
    if (qgemv4m_shape_ok(K, group, bits))
        for (int i = 0; i < 2; ++i)
            if (a.m[i].w) {
                OMX_HIP_CHECK(hipMalloc((void**)&tiles[i], qgemv4m_tile_words(a.m[i].n, K) * 4));
                if (launch_qgemv4m_repack(tiles[i], a.m[i].w, a.m[i].scales, a.m[i].biases, a.m[i].n, K, (hipStream_t)stream)) return 1;
                a.m[i].tiles = tiles[i];
            }
    const int rc = launch_qgemv(a, bits, pro, epi, (hipStream_t)stream);
    OMX_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < 2; ++i)
        if (tiles[i]) (void)hipFree(tiles[i]);
    return rc;
}
extern "C" int omx_debug_qgemv_grid(int N) { return omx::qgemv_grid(N); }

// ---- what does a COLD weight matrix cost a streaming GEMV at its start?  (tools/tlb_probe.py)
//      mode 0: rotate through n_copies matrices (cold, as in a decode step)
//      mode 1: before each GEMV, a tiny kernel reads ONE line every `stride` bytes of that matrix from every XCD
//              (address translations warm, data cold)
//      mode 2: before each GEMV, a kernel streams the whole matrix (translations and Infinity Cache warm)
//      Per-kernel times come from rocprofv3 --kernel-trace --stats (different kernel names).
namespace {
__global__ void page_touch_kernel(const unsigned char* base, size_t bytes, size_t stride, unsigned* sink) {
    // blockIdx % 8 is the XCD (round-robin dispatch): every XCD sweeps ALL pages, 1/(blocks per XCD) each
    const size_t per_xcd = gridDim.x / 8, j = blockIdx.x / 8;
    unsigned acc = 0;
    for (size_t off = (j * blockDim.x + threadIdx.x) * stride; off < bytes; off += per_xcd * blockDim.x * stride)
        acc += *reinterpret_cast<const volatile unsigned*>(base + off);
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void stream_read_kernel(const uint4* base, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = base[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
}  // namespace

extern "C" int omx_bench_gemv_warm(int N, int K, int pro, int epi, int n_copies, int iters, int mode, long stride,
                                   int one_allocation, float* avg_ms) {
    using namespace omx;
    OMX_REQUIRE(avg_ms && n_copies > 0 && iters > 0, "omx_bench_gemv_warm: bad arguments");
    const int mats = (epi == EPI_SWIGLU) ? 2 : 1;
    const size_t wbytes = (((size_t)N * K * 2 * mats) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    std::vector<void*> w(n_copies, nullptr);
    void* big = nullptr;
    if (one_allocation) {
        OMX_HIP_CHECK(hipMalloc(&big, wbytes * n_copies));
        for (int i = 0; i < n_copies; ++i) w[i] = (char*)big + wbytes * i;
    } else {
        for (auto& p : w) OMX_HIP_CHECK(hipMalloc(&p, wbytes));
    }
    for (int i = 0; i < n_copies; ++i)
        if (omx_fill_uniform(w[i], (size_t)N * K * mats, 17u + i, 0.03f, 0.f, OMX_BFLOAT16, nullptr)) return 1;
    void *x = nullptr, *nw = nullptr, *out = nullptr, *resid = nullptr, *slot = nullptr;
    OMX_HIP_CHECK(hipMalloc(&x, (size_t)K * 2));
    OMX_HIP_CHECK(hipMalloc(&nw, (size_t)K * 2));
    OMX_HIP_CHECK(hipMalloc(&out, (size_t)N * 4));
    OMX_HIP_CHECK(hipMalloc(&resid, (size_t)N * 2));
    OMX_HIP_CHECK(hipMalloc(&slot, 8 * 65536));
    omx_fill_uniform(x, K, 3, 1.0f, 0.f, OMX_BFLOAT16, nullptr);
    omx_fill_uniform(nw, K, 4, 0.1f, 1.f, OMX_BFLOAT16, nullptr);
    omx_fill_uniform(resid, N, 5, 1.0f, 0.f, OMX_BFLOAT16, nullptr);
    OMX_HIP_CHECK(hipMemset(slot, 0, 8 * 65536));
    hipStream_t s;
    OMX_HIP_CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    OMX_HIP_CHECK(hipEventCreate(&e0));
    OMX_HIP_CHECK(hipEventCreate(&e1));
    auto run = [&](int i) {
        const bf16_t* base = (const bf16_t*)w[i % n_copies];
        if (mode == 1) page_touch_kernel<<<64, 256, 0, s>>>((const unsigned char*)base, (size_t)N * K * 2 * mats, (size_t)stride, (unsigned*)slot);
        if (mode == 2) stream_read_kernel<<<2048, 256, 0, s>>>((const uint4*)base, (size_t)N * K * 2 * mats / 16, (unsigned*)slot);
        GemvArgs a = {};
        a.w0 = base;
        a.w1 = base + (size_t)N * K;
        a.n0 = N; a.N = N; a.K = K;
        a.x = (const bf16_t*)x;
        a.norm_w = (const bf16_t*)nw;
        a.eps = 1e-6f;
        a.resid = (const bf16_t*)resid;
        a.out = out;
        a.argmax_slot = (unsigned long long*)slot;
        return launch_gemv(a, pro, epi, s);
    };
    for (int i = 0; i < n_copies + 2; ++i)
        if (run(i)) return 1;
    OMX_HIP_CHECK(hipStreamSynchronize(s));
    OMX_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        if (run(i)) return 1;
    OMX_HIP_CHECK(hipEventRecord(e1, s));
    OMX_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    OMX_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    if (big) (void)hipFree(big);
    else for (auto p : w) (void)hipFree(p);
    (void)hipFree(x); (void)hipFree(nw); (void)hipFree(out); (void)hipFree(resid); (void)hipFree(slot);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    return 0;
}
