// Packed fp32 probe for gfx950: lane-operations per second of v_pk_fma_f32 / v_pk_add_f32 /
// v_pk_mul_f32 against v_fma_f32 at 1..8 waves per SIMD, eight independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 32768;
template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, float b, float c)
{
    f2 a[8];
    float s[8];
    double dd[8];
    const double db = b, dc = c;
    for (int i = 0; i < 8; i++) { a[i] = {threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i}; s[i] = threadIdx.x * 1e-3f + i; dd[i] = threadIdx.x * 1e-3 + i; }
    const f2 bb = {b, b * 1.0001f}, cc = {c, c * 1.5f};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) s[i] = __builtin_fmaf(s[i], b, c);
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(bb), "v"(cc));
            if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(cc));
            if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(bb));
            if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(c));
            if (KIND == 5) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b), "s"(0x5555555555555555ull));
            if (KIND == 6) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s[i]));
            if (KIND == 7) asm volatile("v_sub_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(s[i]) : "v"(c));
            if (KIND == 8) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(c));
            if (KIND == 9) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(b));
            if (KIND == 10) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(c));
            if (KIND == 11) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(b) : "vcc");
            if (KIND == 12) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(s[i]), "v"(b) : "vcc");
            if (KIND == 13) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" : : "v"(s[i]), "v"(b) : "s20", "s21");
            if (KIND == 14) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(s[i]) : "v"(c));
            if (KIND == 15) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(c));
            if (KIND == 16) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(c));
            if (KIND == 17) asm volatile("v_fmaak_f32 %0, %0, %1, 0x358637bd" : "+v"(s[i]) : "v"(c));
            if (KIND == 18) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(s[i]) : "s"(c));
            if (KIND == 19) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "s"(b), "v"(c));
            if (KIND == 20) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(s[i]));
            if (KIND == 21) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(s[i]) : "v"(c));
            if (KIND == 22) asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(dd[i]) : "v"(s[i]));
            if (KIND == 23) asm volatile("v_cvt_f32_f64_e32 %0, %1" : "=v"(s[i]) : "v"(dd[i]));
            if (KIND == 24) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(db));
            if (KIND == 25) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dd[i]) : "v"(db), "v"(dc));
            if (KIND == 26) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(c));
            if (KIND == 27) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(s[i]));
            if (KIND == 28) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(s[i]));
            if (KIND == 29) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(s[i]));
            if (KIND == 30) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(c));
            if (KIND == 31) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(dc));
        }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i].x + a[i].y + s[i] + (float)dd[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int KIND>
void run(const char* name, double ops)
{
    float* out;
    CHECK(hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-28s", name);
    for (int w = 1; w <= 8; w *= 2) {
        int blocks = 256 * w;
        hipLaunchKernelGGL((probe<KIND>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-6f);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<KIND>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-6f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double n = (double)blocks * 256 * ITERS * 8;
        printf("  w%d %6.2f", w, n * ops / ms / 1e9);
    }
    printf("   T lane-ops/s\n");
    CHECK(hipFree(out));
}
int main()
{
    run<0>("fmaf (compiler)", 1);
    run<4>("v_fma_f32", 1);
    run<1>("v_pk_fma_f32", 2);
    run<2>("v_pk_add_f32", 2);
    run<3>("v_pk_mul_f32", 2);
    run<5>("v_cndmask_b32_e64 (sgpr mask)", 1);
    run<6>("v_mov_b32_dpp wave_shr", 1);
    run<7>("v_sub_f32_dpp wave_shr", 1);
    run<8>("v_add_f32_e32", 1);
    run<9>("v_mul_f32_e32", 1);
    run<10>("v_fmac_f32_e32", 1);
    run<11>("v_cndmask_b32_e32 vcc", 1);
    run<12>("v_cmp_gt_f32_e32", 1);
    run<13>("v_cmp_gt_f32_e64 sgpr", 1);
    run<14>("v_add_f32_e64", 1);
    run<15>("v_add_u32_e32", 1);
    run<16>("v_max_f32_e32", 1);
    run<17>("v_fmaak_f32", 1);
    run<18>("v_add_f32_e32 sgpr src", 1);
    run<19>("v_fma_f32 sgpr src", 1);
    run<20>("v_lshlrev_b32", 1);
    run<21>("v_mov_b32", 1);
    run<22>("v_cvt_f64_f32", 1);
    run<23>("v_cvt_f32_f64", 1);
    run<24>("v_mul_f64", 1);
    run<25>("v_fma_f64", 1);
    run<31>("v_add_f64", 1);
    run<26>("v_div_fixup_f32", 1);
    run<27>("v_rcp_f32", 1);
    run<28>("v_sqrt_f32", 1);
    run<29>("v_exp_f32", 1);
    run<30>("v_max3_f32", 1);
}
