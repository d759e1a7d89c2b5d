// Does v_dot2c_f32_bf16 (D = A.lo*B.lo + A.hi*B.hi + C) round like two chained fmaf (lo first, then hi)?  Counts mismatches over
// random bf16 pairs and accumulators.  build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off dot2_probe.hip -o dot2_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ uint32_t rng(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__global__ void probe(unsigned long long* out, int iters, int mode) {
    uint32_t s = 0x9E3779B9u * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long bad_lohi = 0, bad_hilo = 0, bad_chain = 0;
    float chain_d = 0.f, chain_f = 0.f;
    for (int i = 0; i < iters; ++i) {
        // bf16 values with moderate exponents (weights x activations), packed in a dword
        auto mk = [&]() { uint32_t r = rng(s); uint32_t e = 118 + (r >> 8) % 12; return (uint16_t)(((r & 1) << 15) | (e << 7) | ((r >> 16) & 0x7F)); };
        const uint32_t a = mk() | ((uint32_t)mk() << 16), b = mk() | ((uint32_t)mk() << 16);
        float c;
        if (mode == 0) { uint32_t r = rng(s); uint32_t e = 110 + (r >> 8) % 20; c = __uint_as_float(((r & 1u) << 31) | (e << 23) | (rng(s) & 0x7FFFFF)); }
        else c = chain_f;
        const float alo = __uint_as_float(a << 16), ahi = __uint_as_float(a & 0xFFFF0000u);
        const float blo = __uint_as_float(b << 16), bhi = __uint_as_float(b & 0xFFFF0000u);
        const float d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), mode == 0 ? c : chain_d, false);
        const float f1 = fmaf(ahi, bhi, fmaf(alo, blo, c));
        const float f2 = fmaf(alo, blo, fmaf(ahi, bhi, c));
        if (mode == 0) {
            bad_lohi += __float_as_uint(d) != __float_as_uint(f1);
            bad_hilo += __float_as_uint(d) != __float_as_uint(f2);
        } else {
            chain_d = d; chain_f = f1;
            if ((i & 63) == 63) { bad_chain += __float_as_uint(chain_d) != __float_as_uint(chain_f); chain_d = chain_f = 0.f; }
        }
    }
    atomicAdd(&out[0], bad_lohi); atomicAdd(&out[1], bad_hilo); atomicAdd(&out[2], bad_chain);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 32); 
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(d, 0, 32);
        probe<<<256, 256>>>(d, 4096, mode);
        unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        const double n = 256.0 * 256 * 4096;
        if (mode == 0) printf("single step: %.0f trials, mismatches vs fma(hi,fma(lo,c)) %llu, vs fma(lo,fma(hi,c)) %llu\n", n, h[0], h[1]);
        else printf("64-step chains: %.0f chains, mismatching chains %llu\n", n / 64, h[2]);
    }
    return 0;
}
