// ds_read_b128 bank behaviour on gfx950 (run on the GPU box): cycles per instruction for the lane -> address patterns the GEMM tiles use.
//   hipcc --offload-arch=gfx950 -O3 -o lds_b128_probe lds_b128_probe.hip && ./lds_b128_probe
// One wave (and four waves, one per SIMD) issue 256 back-to-back ds_read_b128 at a fixed per-lane address; s_memtime around them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void probe(uint64_t* out, int pattern) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, l32 = lane & 31, hi = lane >> 5;
    unsigned addr = 0;
    switch (pattern) {
        case 0: addr = lane * 16; break;                                                   // linear: the ideal
        case 1: addr = l32 * 128 + ((hi ^ ((l32 >> 1) & 7)) * 16); break;                   // 32x32 fragment, chunk ^ (row >> 1) & 7   (round 2..5 kernels)
        case 2: addr = l32 * 128 + ((hi ^ (l32 & 7)) * 16); break;                          // 32x32 fragment, chunk ^ row & 7
        case 3: addr = (lane & 15) * 1040 + (lane >> 4) * 16; break;                        // 16x16 fragment, rows padded to 1040 B (vendor)
        case 4: addr = (lane & 15) * 128 + (((lane >> 4) ^ ((lane & 15) >> 1)) & 7) * 16; break;   // 16x16 fragment, chunk ^ (row >> 1) & 7
        case 5: addr = (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) & 7) * 16; break;    // 16x16 fragment, chunk ^ row & 7
        case 6: addr = l32 * 128 + (((2 + hi) ^ ((l32 >> 1) & 7)) * 16); break;             // pattern 1 at k sub-step 1
        case 7: addr = l32 * 144 + hi * 16; break;                                          // 32x32 fragment, rows padded by 16 B
        case 8: addr = l32 * 128 + hi * 16; break;                                          // 32x32 fragment, no swizzle: the worst case
    }
    addr += (unsigned)(uintptr_t)smem + (threadIdx.x >> 6) * 8192;
    uint64_t t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
#define R4 "ds_read_b128 v[40:43], %1\n\tds_read_b128 v[44:47], %1 offset:4096\n\tds_read_b128 v[48:51], %1\n\tds_read_b128 v[52:55], %1 offset:4096\n\t"
#define R16 R4 R4 R4 R4
#define R64 R16 R16 R16 R16
    asm volatile(R64 R64 R64 R64 "s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(t1) : "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "memory");
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    uint64_t* d;
    hipMalloc(&d, 64);
    const char* names[] = {"linear lane * 16", "32-row frag, chunk ^ (row >> 1) & 7", "32-row frag, chunk ^ row & 7", "16-row frag, rows of 1040 B",
                           "16-row frag, chunk ^ (row >> 1) & 7", "16-row frag, chunk ^ row & 7", "pattern 1 at k sub-step 1", "32-row frag, rows of 144 B", "32-row frag, no swizzle"};
    for (int waves = 1; waves <= 4; waves *= 4)
        for (int p = 0; p < 9; ++p) {
            uint64_t h[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 3; ++rep) {
                probe<<<1, 64 * waves, 65536>>>(d, p);
                hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
            }
            printf("%d wave(s)  %-40s %6.1f memtime ticks per ds_read_b128 (wave 0)\n", waves, names[p], (double)h[0] / 256.0);
        }
    return 0;
}
