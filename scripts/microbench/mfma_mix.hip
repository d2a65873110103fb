// v_mfma_f32_4x4x1_16b_f32 next to ordinary vector work on gfx950: operand layout check, issue
// cost alone, and how much of it hides behind VALU instructions of the same and of other waves.
// Also the rates of a few more VALU forms (bitwise, med3, literal operands) and of a broadcast
// ds_read_b128 issued between VALU instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 16384;

__global__ void layout_probe(float* out)
{
    const int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    // a: "node" value held by lane 4b+i; b: the lane's own value
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l % 4 + 1) + 0.25f * (l / 4), (float)(100 + l), c, 0, 0, 0);
    for (int i = 0; i < 4; i++) out[l * 4 + i] = c[i];
}

// KIND 0: 8 v_fmac chains; 1: 3 mfma chains; 2: both in one wave; 3: 8 fmac with sgpr source;
// 4: 8 fmac (vgpr) + one broadcast ds_read_b128 per 4; 5..: single-instruction rates
template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, float b, float c)
{
    __shared__ f4 lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = {b, c, b, c};
    __syncthreads();
    float s[8];
    f4 acc[3];
    for (int i = 0; i < 8; i++) s[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 3; i++) acc[i] = {0, 0, 0, 0};
    float wv = b + (threadIdx.x & 3);
    f4 ld = {0, 0, 0, 0};
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0 || KIND == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(c));
        }
        if (KIND == 1 || KIND == 2) {
#pragma unroll
            for (int i = 0; i < 3; i++)
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(wv), "v"(s[i]));
        }
        if (KIND == 20) {   // interleaved: mfma, 3 independent fmac, ...
#pragma unroll
            for (int i = 0; i < 3; i++) {
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(wv), "v"(c));
                asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[2 * i]) : "v"(b), "v"(c));
                asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[2 * i + 1]) : "v"(b), "v"(c));
            }
            asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[6]) : "v"(b), "v"(c));
            asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[7]) : "v"(b), "v"(c));
        }
        if (KIND == 21) {   // B operand written by the VALU instruction just before each mfma
#pragma unroll
            for (int i = 0; i < 3; i++) {
                asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(c));
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(wv), "v"(s[i]));
            }
#pragma unroll
            for (int i = 3; i < 8; i++) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(c));
        }
        if (KIND == 22) {   // one accumulator chain only (every mfma waits for the one before)
#pragma unroll
            for (int i = 0; i < 3; i++)
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(wv), "v"(c));
        }
        if (KIND == 23) {   // compiler-scheduled builtin, operands from LDS and VALU
            f4 wrow = lds[(threadIdx.x & 3) + 4 * (it & 7)];
#pragma unroll
            for (int i = 0; i < 3; i++) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(wrow[i], s[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = __builtin_fmaf(s[i], b, c);
        }
        if (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "s"(b), "v"(c));
        }
        if (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if ((i & 3) == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld) : "v"(0), "n"(16 * (i / 4)));
                asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(c));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 5) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(c));
            if (KIND == 6) asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(c));
            if (KIND == 7) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(s[i]) : "v"(b), "v"(c));
            if (KIND == 8) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(c));
            if (KIND == 9) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(c));
            if (KIND == 10) asm volatile("v_add_f32_e32 %0, 0x3f99999a, %0" : "+v"(s[i]));
            if (KIND == 11) asm volatile("v_mul_f32_e32 %0, 2.0, %0" : "+v"(s[i]));
            if (KIND == 12) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(c));
            if (KIND == 13) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(c));
            if (KIND == 14) asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(c));
            if (KIND == 15) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(s[i]) : "v"(b));
            if (KIND == 16) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(c));
            if (KIND == 17) asm volatile("v_mul_f32_e64 %0, |%0|, %1" : "+v"(s[i]) : "v"(b));
            if (KIND == 18) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(b));
            if (KIND == 19) asm volatile("v_max_u32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(b));
        }
    }
    float r = ld[0] + ld[3];
    for (int i = 0; i < 8; i++) r += s[i];
    for (int i = 0; i < 3; i++) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
void run(const char* name, double instrs_per_iter)
{
    float* out;
    CHECK(hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-34s", name);
    for (int w = 1; w <= 8; w *= 2) {
        int blocks = 256 * w;
        hipLaunchKernelGGL((probe<KIND>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-6f);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<KIND>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-6f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        // cycles per loop iteration per SIMD at 2.4 GHz: w waves share a SIMD
        double cyc = ms * 1e-3 * 2.4e9 / ITERS / w;
        printf("  w%d %6.1f", w, cyc);
    }
    printf("   cycles / iteration / wave (%g instructions)\n", instrs_per_iter);
    CHECK(hipFree(out));
}

int main()
{
    float* out; float host[256];
    CHECK(hipMalloc(&out, sizeof host));
    hipLaunchKernelGGL(layout_probe, dim3(1), dim3(64), 0, 0, out);
    CHECK(hipMemcpy(host, out, sizeof host, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 4; i++) {
            float want = ((float)(i + 1) + 0.25f * (l / 4)) * (float)(100 + l);
            if (host[l * 4 + i] != want) bad++;
        }
    printf("4x4x1 layout: D[i] of lane l = a(lane 4*(l/4)+i) * b(lane l): %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad) for (int l = 0; l < 8; l++) printf("  lane %d: %g %g %g %g\n", l, host[4*l], host[4*l+1], host[4*l+2], host[4*l+3]);
    run<0>("8 v_fmac (vgpr)", 8);
    run<1>("3 mfma 4x4x1", 3);
    run<2>("8 v_fmac + 3 mfma", 11);
    run<20>("8 v_fmac + 3 mfma interleaved", 11);
    run<21>("8 v_fmac + 3 mfma, B just written", 11);
    run<22>("3 mfma, one chain", 3);
    run<23>("8 fma + 3 mfma + ds_read (compiler)", 12);
    run<3>("8 v_fmac (sgpr source)", 8);
    run<4>("8 v_fmac + 2 ds_read_b128 bcast", 10);
    run<5>("8 v_and_b32", 8);
    run<6>("8 v_or_b32", 8);
    run<13>("8 v_xor_b32", 8);
    run<7>("8 v_bfi_b32", 8);
    run<8>("8 v_med3_f32", 8);
    run<9>("8 v_min_f32", 8);
    run<19>("8 v_max_u32", 8);
    run<10>("8 v_add_f32 literal", 8);
    run<11>("8 v_mul_f32 inline const", 8);
    run<12>("8 v_sub_f32", 8);
    run<14>("8 v_sub_u32", 8);
    run<15>("8 v_fma_f32 vgpr, vgpr, 1.0", 8);
    run<16>("8 v_fma_f32 -v, v, v", 8);
    run<17>("8 v_mul_f32_e64 |v|, v", 8);
    run<18>("8 v_cndmask_b32_e32 vcc", 8);
}
