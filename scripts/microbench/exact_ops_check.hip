// Device check of csrc/exact_ops.h: exact_sqrt_unscaled(x) against sqrtf(x) for every float x outside
// (0, 2^-96) and the negative subnormals (all other bit patterns); exact_reciprocal(d) against 1.0f / d for every significand
// of d at several exponents, and exact_quotient / exact_quotient_fixup against a / d for every
// significand of d with pseudo-random numerators (and, for a few denominators, every numerator
// significand).  Prints the number of differences; exit status 1 if any.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "../../path-tracer_amd/csrc/exact_ops.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// counts[0]: reciprocal differs; [1]: quotient differs; [2]: quotient with fixup differs
__global__ void sweep_denominators(int exp_d, int numerators, unsigned long long* counts)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;   // 2^23 threads
    const float d = __uint_as_float(((uint32_t)(127 + exp_d) << 23) | m);
    const float y = ptk_exact::exact_reciprocal(d);
    unsigned bad_r = __float_as_uint(y) != __float_as_uint(1.0f / d), bad_q = 0, bad_f = 0;
    for (int i = 0; i < numerators; i++) {
        const uint64_t h = mix(((uint64_t)m << 20) + (uint64_t)i * 0x9E3779B97F4A7C15ull + (uint64_t)(exp_d + 64));
        // numerator: random significand and sign, exponent within +-20 of the denominator's
        const uint32_t ea = (uint32_t)(127 + exp_d - 20 + (int)((h >> 40) % 41));
        const float a = __uint_as_float(((uint32_t)(h >> 63) << 31) | (ea << 23) | (uint32_t)(h & 0x7fffffu));
        const float want = a / d;
        bad_q += __float_as_uint(ptk_exact::exact_quotient(a, y, d)) != __float_as_uint(want);
        bad_f += __float_as_uint(ptk_exact::exact_quotient_fixup(a, y, d)) != __float_as_uint(want);
    }
    if (bad_r) atomicAdd(&counts[0], 1ull);
    if (bad_q) atomicAdd(&counts[1], (unsigned long long)bad_q);
    if (bad_f) atomicAdd(&counts[2], (unsigned long long)bad_f);
}

__global__ void sweep_numerators(float d, unsigned long long* counts)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    const float y = ptk_exact::exact_reciprocal(d);
    const float a = __uint_as_float(0x3f800000u | m);
    if (__float_as_uint(ptk_exact::exact_quotient(a, y, d)) != __float_as_uint(a / d)) atomicAdd(&counts[1], 1ull);
}

// zero, signed zero, infinities, NaN numerators through the fixup form
__global__ void special_numerators(unsigned long long* counts)
{
    const float ds[4] = {3.0f, 49.0f, 128.0f, -7.0f};
    const uint32_t as[6] = {0x00000000u, 0x80000000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x00000001u};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 6; j++) {
            const float a = __uint_as_float(as[j]), d = ds[i];
            const float got = ptk_exact::exact_quotient_fixup(a, ptk_exact::exact_reciprocal(d), d), want = a / d;
            const bool same = (got != got && want != want) || __float_as_uint(got) == __float_as_uint(want);
            if (!same && j != 5) atomicAdd(&counts[2], 1ull);   // a subnormal numerator is outside the contract
        }
}

// counts[3]: roots that differ, over all 2^32 bit patterns outside (0, 2^-96); NaN equals NaN
__global__ void sweep_sqrt(unsigned long long* counts)
{
    const uint32_t hi = blockIdx.x * blockDim.x + threadIdx.x;   // 2^24 threads, 256 patterns each
    unsigned bad = 0;
    for (uint32_t lo = 0; lo < 256; lo++) {
        const uint32_t bits = (hi << 8) | lo;
        const float x = __uint_as_float(bits);
        if (x > 0.0f && x < 1.2621774e-29f) continue;            // (0, 2^-96): outside the contract
        if (x < 0.0f && x > -1.17549435e-38f) continue;           // negative subnormals: -0 here, NaN there
        float xv = x;
        asm volatile("" : "+v"(xv));                              // keep the two roots apart
        const float got = ptk_exact::exact_sqrt_unscaled(x), want = __builtin_sqrtf(xv);
        const bool same = (got != got && want != want) || __float_as_uint(got) == __float_as_uint(want);
        bad += !same;
    }
    if (bad) atomicAdd(&counts[3], (unsigned long long)bad);
}

int main()
{
    unsigned long long* counts; unsigned long long h[4] = {0, 0, 0, 0};
    CHECK(hipMalloc(&counts, sizeof h));
    CHECK(hipMemset(counts, 0, sizeof h));
    const int exps[] = {-47, -30, -8, -1, 0, 1, 7, 23, 30, 47};
    for (int e : exps) hipLaunchKernelGGL(sweep_denominators, dim3((1u << 23) / 256), dim3(256), 0, 0, e, 48, counts);
    const float few[] = {3.0f, 9.0f, 49.0f, 128.0f, 384.0f, 1.9999999f, 1.0000001f, 0.7071068f, 5592405.0f * 3.0f};
    for (float d : few) hipLaunchKernelGGL(sweep_numerators, dim3((1u << 23) / 256), dim3(256), 0, 0, d, counts);
    hipLaunchKernelGGL(special_numerators, dim3(1), dim3(1), 0, 0, counts);
    hipLaunchKernelGGL(sweep_sqrt, dim3((1u << 24) / 256), dim3(256), 0, 0, counts);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, counts, sizeof h, hipMemcpyDeviceToHost));
    printf("exact_reciprocal vs 1.0f / d: %llu of %d x 2^23 denominators differ\n", h[0], (int)(sizeof exps / sizeof exps[0]));
    printf("exact_quotient vs a / d: %llu of %.3g quotients differ\n", h[1],
           (double)(sizeof exps / sizeof exps[0]) * 8388608.0 * 48 + (double)(sizeof few / sizeof few[0]) * 8388608.0);
    printf("exact_quotient_fixup vs a / d (zeros, infinities, NaN included): %llu differ\n", h[2]);
    printf("exact_sqrt_unscaled vs sqrtf, every float but (0, 2^-96) and negative subnormals: %llu differ\n", h[3]);
    return h[0] || h[1] || h[2] || h[3] ? 1 : 0;
}
