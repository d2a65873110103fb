// VALU issue-rate probe for gfx950: lane-operations per second of common fp32 / integer
// instruction kinds at 1..8 waves per SIMD, with 8 independent chains per lane or one
// dependent chain, so kernel instruction counts can be priced.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 32768;

template <int KIND, int CHAINS>
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
            if (KIND == 2) u[i] = u[i] * 3u + (unsigned)it;
            if (KIND == 3) q[i] = q[i] + (q[i] >> 7);
            if (KIND == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (KIND == 5) a[i] = __builtin_sqrtf(a[i] + c);
            if (KIND == 6) a[i] = a[i] > b ? a[i] - c : a[i] + c;      // cmp + 2 alu + cndmask
            if (KIND == 7) a[i] = __builtin_fmaxf(a[i] - c, b);        // sub + max
            if (KIND == 8) a[i] = (float)((double)a[i] * (double)b + (double)c);  // cvt, f64 fma, cvt
        }
    }
    float s = 0;
    for (int i = 0; i < CHAINS; i++) s += a[i] + u[i] + (float)q[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int CHAINS>
void run(const char* name, double ops_per_iter_chain)
{
    float* out;
    CHECK(hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-34s chains %d:", name, CHAINS);
    for (int w = 1; w <= 8; w *= 2) {
        int blocks = 256 * w;   // one 256-thread block = one wave per SIMD of a CU
        hipLaunchKernelGGL((probe<KIND, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-6f);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<KIND, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-6f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double ops = (double)blocks * 256 * ITERS * CHAINS * ops_per_iter_chain;
        printf("  w%d %6.2f", w, ops / ms / 1e9);
    }
    printf("   T lane-ops/s\n");
    CHECK(hipFree(out));
}

int main()
{
    run<0, 8>("v_fma_f32", 1);
    run<0, 1>("v_fma_f32", 1);
    run<1, 8>("v_mul_f32 + v_add_f32", 2);
    run<1, 1>("v_mul_f32 + v_add_f32", 2);
    run<6, 8>("cmp + sub + add + cndmask", 4);
    run<6, 1>("cmp + sub + add + cndmask", 4);
    run<7, 8>("v_sub_f32 + v_max_f32", 2);
    run<2, 8>("u32 mul + add", 2);
    run<3, 8>("u64 shift + add", 2);
    run<4, 8>("v_rcp_f32", 1);
    run<5, 8>("v_add + sqrtf (IEEE)", 2);
    run<8, 8>("cvt + v_fma_f64 + cvt", 3);
    return 0;
}
