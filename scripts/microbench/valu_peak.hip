// VALU issue-rate probe for gfx950: independent fp32 chains at 1..8 waves per SIMD.
// Prints lane-operations per second for v_fma_f32, v_mul_f32+v_add_f32 (uncontracted),
// v_add_u32 and a 64-bit integer add, so kernel instruction counts can be priced.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 2048, CHAINS = 8;

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, float b, float c)
{
    float a[CHAINS];
    unsigned u[CHAINS];
    unsigned long long q[CHAINS];
    for (int i = 0; i < CHAINS; i++) {
        a[i] = threadIdx.x * 1e-3f + i;
        u[i] = threadIdx.x + i;
        q[i] = threadIdx.x * 77ull + i;
    }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (KIND == 0) a[i] = __builtin_fmaf(a[i], b, c);
            if (KIND == 1) a[i] = __fadd_rn(__fmul_rn(a[i], b), c);
            if (KIND == 2) u[i] = u[i] * 3u + (unsigned)it;          // mul_lo + add (or mad)
            if (KIND == 3) q[i] = q[i] + (q[i] >> 7);                // 64-bit shift + add
            if (KIND == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (KIND == 5) a[i] = __builtin_sqrtf(a[i] + c);
        }
    }
    float s = 0;
    for (int i = 0; i < CHAINS; i++) s += a[i] + u[i] + (float)q[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, double ops_per_iter_chain)
{
    float* out;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 256 * sizeof(float) / 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 1; w <= 8; w *= 2) {
        int blocks = 256 * w;   // one 256-thread block = one wave per SIMD of a CU
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-6f);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-6f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double ops = (double)blocks * 256 * ITERS * CHAINS * ops_per_iter_chain;
        printf("%-28s waves/SIMD %d  %.3f ms  %.2f T lane-ops/s\n", name, w, ms, ops / ms / 1e9);
    }
    CHECK(hipFree(out));
}

int main()
{
    run<0>("v_fma_f32", 1);
    run<1>("v_mul_f32 + v_add_f32", 2);
    run<2>("u32 mul+add", 2);
    run<3>("u64 shift+add", 2);
    run<4>("v_rcp_f32", 1);
    run<5>("v_add + sqrtf (IEEE)", 2);
    return 0;
}
