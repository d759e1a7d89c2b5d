// Probe kernel: a GEMV-like weight stream (each wave owns rows, 16 B / lane loads, dot with x from LDS) -- shape of the engine's
// QKV / O GEMVs -- for tools/aql_probe/probe.cpp.  Built device-only: hipcc --genco --offload-arch=gfx950 -mcode-object-version=4
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct Args { const uint16_t* w; const uint16_t* x; float* out; int N; int K; int rows_per_wave; int pad; };
__device__ inline float lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ inline float hi(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }
extern "C" __global__ __launch_bounds__(256) void stream_gemv(Args a) {
    __shared__ u32x4 xs[512];                                   // K = 4096
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * a.rows_per_wave;
    u32x4 w[2][8];
    for (int r = 0; r < 2; ++r) {
        const u32x4* p = reinterpret_cast<const u32x4*>(a.w + (size_t)min(row0 + r, a.N - 1) * a.K);
        for (int j = 0; j < 8; ++j) w[r][j] = __builtin_nontemporal_load(p + j * 64 + lane);
    }
    for (int v = threadIdx.x; v < 512; v += 256) xs[v] = reinterpret_cast<const u32x4*>(a.x)[v];
    __syncthreads();
    for (int rb = 0; rb < a.rows_per_wave; rb += 2) {
        float acc[2] = {0.f, 0.f};
        for (int j = 0; j < 8; ++j) {
            const u32x4 xp = xs[j * 64 + lane];
            for (int r = 0; r < 2; ++r)
                for (int q = 0; q < 4; ++q) acc[r] += lo(w[r][j][q]) * lo(xp[q]) + hi(w[r][j][q]) * hi(xp[q]);
        }
        if (rb + 2 < a.rows_per_wave)
            for (int r = 0; r < 2; ++r) {
                const u32x4* p = reinterpret_cast<const u32x4*>(a.w + (size_t)min(row0 + rb + 2 + r, a.N - 1) * a.K);
                for (int j = 0; j < 8; ++j) w[r][j] = __builtin_nontemporal_load(p + j * 64 + lane);
            }
        for (int r = 0; r < 2; ++r) {
            float v = acc[r];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0 && row0 + rb + r < a.N) a.out[row0 + rb + r] = v;
        }
    }
}
