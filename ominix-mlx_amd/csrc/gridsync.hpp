// Device-wide barrier and coherent accessors for persistent (all blocks co-resident) kernels on gfx950.
//
// The decode step is a chain of dependent phases (QKV -> attention -> O -> gate/up -> down, x layers).
// As separate launches every phase pays launch + HBM ramp + drain (~2.5-4 us of an 8-35 us kernel,
// profiles/r01_b_kernel_stats_ctx2048.csv).  Inside ONE persistent kernel a block can put the next
// phase's first weight rows in flight BEFORE it waits here, so HBM keeps streaming through the barrier.
//
// What was measured on MI355X (tools/barrier_probe.py, 512 blocks x 256 threads, one exchange + barrier):
//   * agent-scope release/acquire FENCES are unusable: every wave's `buffer_wbl2 sc1` / `buffer_inv sc1`
//     walks the XCD's L2 and they serialise -- 55 us per barrier;
//   * one atomic counter without fences: 8.4 us (same-address atomics serialise at ~16 ns each);
//   * per-block arrive FLAGS gathered by block 0 + one release word: 2.4 us.  <- this file
// So there are no fences and no read-modify-write atomics here.  Everything blocks exchange inside the
// launch goes through the *_coh accessors below: relaxed agent-scope atomic loads/stores, which hipcc
// lowers to `global_load/store ... sc1` (write-through / read-past the non-coherent per-XCD L2).  A block
// waits for its own stores (`s_waitcnt vmcnt(0)`) before it raises its flag.
//
// A lost block must not hang the GPU (a hung box is a strike on the test pool): spins are bounded; on
// timeout the abort word is set, every later barrier falls through and the host reports the error.
#pragma once
#include "common.hpp"

namespace omx {

// ---- coherent accessors (data written by one block and read by another inside the same launch) ----
__device__ __forceinline__ uint32_t ld_coh32(const void* p) {
    return __hip_atomic_load(reinterpret_cast<uint32_t*>(const_cast<void*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_coh64(const void* p) {
    return __hip_atomic_load(reinterpret_cast<uint64_t*>(const_cast<void*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32x4 ld_coh128(const void* p) {
    const uint64_t lo = ld_coh64(p), hi = ld_coh64(reinterpret_cast<const char*>(p) + 8);
    return u32x4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
}
__device__ __forceinline__ float ld_coh_f32(const float* p) { return __uint_as_float(ld_coh32(p)); }
__device__ __forceinline__ float ld_coh_bf16(const bf16_t* p) {
    return bf16_to_f32(__hip_atomic_load(const_cast<bf16_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_coh32(void* p, uint32_t v) {
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_coh64(void* p, uint64_t v) {
    __hip_atomic_store(reinterpret_cast<uint64_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_coh_f32(float* p, float v) { st_coh32(p, __float_as_uint(v)); }
__device__ __forceinline__ void st_coh_bf16(bf16_t* p, bf16_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the barrier ----
// words[0] = release word, words[16] = abort word, words[64 + 16*b] = arrive flag of block b (64 B apart).
// Flags carry a monotonically increasing epoch (signed-difference compare: wrap-around is harmless); the
// host advances the launch's first epoch by the number of barriers each launch executes.
struct GridSync {
    unsigned* words;
    unsigned epoch;      // the value the NEXT barrier publishes
    unsigned nblocks;
    bool dead;           // this thread has seen the abort word: stop spinning
};
constexpr size_t grid_sync_words(int nblocks) { return 64 + 16 * (size_t)nblocks; }
constexpr unsigned kGridSpinLimit = 1u << 19;   // polls (>= ~1 us each) before a waiter gives up

template <int SLEEP>
__device__ __forceinline__ bool grid_spin(GridSync& g, unsigned* word) {
    if (g.dead) return false;
    for (unsigned it = 1;; ++it) {
        if ((int)(ld_coh32(word) - g.epoch) >= 0) return true;
        __builtin_amdgcn_s_sleep(SLEEP);
        if ((it & 1023u) == 0 && (ld_coh32(g.words + 16) != 0 || it >= kGridSpinLimit)) {
            st_coh32(g.words + 16, 1u);
            g.dead = true;
            return false;
        }
    }
}

// every thread of every block calls arrive then wait; loads issued between the two (weight prefetch for
// the next phase) stay in flight across the barrier
// drain = false: the block has stored nothing since its last arrive (do not stall on loads in flight)
__device__ __forceinline__ void grid_arrive(GridSync& g, bool drain = true) {
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's coherent stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) st_coh32(g.words + 64 + 16 * blockIdx.x, g.epoch);
}
__device__ __forceinline__ void grid_wait(GridSync& g) {
    if (blockIdx.x == 0) {
        for (unsigned b = threadIdx.x; b < g.nblocks; b += blockDim.x) grid_spin<1>(g, g.words + 64 + 16 * b);
        asm volatile("s_barrier" ::: "memory");
        if (threadIdx.x == 0) st_coh32(g.words, g.epoch);
    } else if (threadIdx.x == 0) {
        grid_spin<2>(g, g.words);
    }
    // raw s_barrier: no LDS data crosses it, and __syncthreads()'s waitcnt would stall on the prefetch
    asm volatile("s_barrier" ::: "memory");
    g.epoch += 1;
}
__device__ __forceinline__ void grid_sync(GridSync& g) {
    grid_arrive(g);
    grid_wait(g);
}

}  // namespace omx
