// Device side of the one-hop xGMI peer-store reduction (peer_allreduce.hip explains the protocol): the table every rank keeps in
// device memory and the three steps a kernel makes -- tag of this call, tagged stores of one payload word into every inbox, bounded
// poll + rank-ordered sum of one word from the own inbox -- plus the hand-over of the sequence number by the kernel's last block.
// Used by the standalone all-reduce kernel and by the GEMV epilogue that reduces its own output rows (gemv.hip, EPI_F32 + peer).
#pragma once
#include "common.hpp"

namespace omx {

constexpr int kPeerMaxWorld = 8, kPeerMaxWords = 8192;
constexpr unsigned kPeerSpinLimit = 1u << 23;   // polls (~1 us each: a peer may legitimately be seconds late, e.g. re-capturing its graph) before a rank gives up

struct PeerDev {
    uint64_t* peers[kPeerMaxWorld];   // every rank's inbox [2][world][kPeerMaxWords] granules as mapped on this rank
    uint64_t* inbox;                  // == peers[rank]
    uint32_t* state;                  // [0] sequence number, [1] blocks done, [2] abort, [3..5] the large kernel's three arrival counters
    int rank, world;
    // the large (two-shot) path, peer_allreduce.hip: behind every rank's inbox in the same exported allocation
    uint64_t* flags[kPeerMaxWorld];        // [2 phases][world] tags: "rank r finished pushing phase p of call tag"
    unsigned char* stage1[kPeerMaxWorld];  // [world][slice] contributions to the slice that rank owns
    unsigned char* stage2[kPeerMaxWorld];  // [world][slice] the reduced slices, pushed by their owners
    size_t stage_bytes;                    // 0: the large path is off
    int sys_scope;                         // large path: 1 = system-scope release / acquire around the stage hand-offs (ranks on different GPUs),
                                           // 0 = agent-scope fences + relaxed flags (all ranks on ONE GPU; unproven across xGMI)
};

typedef __attribute__((address_space(1))) unsigned long long peer_gu64;

// all blocks of a kernel read the number before its LAST block advances it (peer_block_done)
__device__ __forceinline__ unsigned peer_tag(const PeerDev* p) {
    return __hip_atomic_load(p->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
}
__device__ __forceinline__ size_t peer_plane(const PeerDev* p, unsigned tag) { return (size_t)(tag & 1u) * p->world * kPeerMaxWords; }

__device__ __forceinline__ void peer_store_word(const PeerDev* p, unsigned tag, int i, unsigned v) {
    const size_t slot = peer_plane(p, tag) + (size_t)p->rank * kPeerMaxWords + i;
    const unsigned long long g = ((unsigned long long)tag << 32) | v;
#pragma unroll
    for (int r = 0; r < kPeerMaxWorld; ++r)
        if (r < p->world) __hip_atomic_store((peer_gu64*)(p->peers[r] + slot), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// word i of every rank, in rank order; false (and the abort word raised) when a peer never arrived
__device__ __forceinline__ bool peer_poll_words(const PeerDev* p, unsigned tag, int i, unsigned (&w)[kPeerMaxWorld]) {
    const uint64_t* base = p->inbox + peer_plane(p, tag) + i;
    unsigned long long g[kPeerMaxWorld];
#pragma unroll
    for (int r = 0; r < kPeerMaxWorld; ++r) g[r] = 0;
    bool ok = false;
    for (unsigned spins = 0;; ++spins) {
        ok = true;
#pragma unroll
        for (int r = 0; r < kPeerMaxWorld; ++r)
            if (r < p->world && (unsigned)(g[r] >> 32) != tag) {
                g[r] = __hip_atomic_load((peer_gu64*)(base + (size_t)r * kPeerMaxWords), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = ok && (unsigned)(g[r] >> 32) == tag;
            }
        if (ok) break;
        if (spins >= kPeerSpinLimit) {   // void result, loud flag, no hang
            __hip_atomic_store(p->state + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int r = 0; r < kPeerMaxWorld; ++r) w[r] = (unsigned)g[r];
    return ok;
}

__device__ __forceinline__ float peer_poll_sum_f32(const PeerDev* p, unsigned tag, int i) {
    unsigned w[kPeerMaxWorld];
    const bool ok = peer_poll_words(p, tag, i, w);
    float s = __uint_as_float(w[0]);
#pragma unroll
    for (int r = 1; r < kPeerMaxWorld; ++r)
        if (r < p->world) s += __uint_as_float(w[r]);
    return ok ? s : 0.f;
}

// thread 0 of every block, after the block's last poll: the last of `nblocks` hands the sequence number on
__device__ __forceinline__ void peer_block_done(const PeerDev* p, unsigned tag, unsigned nblocks) {
    const unsigned done = __hip_atomic_fetch_add(p->state + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == nblocks - 1) {
        __hip_atomic_store(p->state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p->state, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace omx
