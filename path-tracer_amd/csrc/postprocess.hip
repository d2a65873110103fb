// postprocess.hip -- the display transform of a frame (CUDAPostProcess,
// CUDARender.cu:959-1022) and the 8-bit quantisation of PT::SaveImage
// (PT.cpp:484-501), on the planes already resident in HBM.
//
// A streaming kernel: 12 B read and 12 B (+3 B) written per pixel, one pixel per
// lane, plane-major so every access of a wave is one contiguous 256-byte run.
// pow(v, 1/g) is exp2(log2(v) / g) on the hardware's log2 / exp2 (1 ulp each): for
// display values in [1e-3, 10] the result is within 4e-7 relative of powf (the error grows
// with |log2 v|: 4e-6 at 1e30; results below the smallest normal number come out as zero), and
// the special cases are powf's (see power()).  The library
// powf, let alone a double-precision pow, made this streaming kernel
// compute-bound (42 us at 1080p against 12 us for the bytes).
#include <hip/hip_runtime.h>

#include "ptk_dev.h"

namespace {

struct Px { float c[3]; };

// powf(v, y) for the exponent of a launch (uniform): exp2(y * log2 |v|), with powf's rules for
// what that leaves out -- a negative base gives NaN unless y is an integer (the sign then
// follows y's parity: g = 1 passes signed planes through), -inf and -0 keep their sign for an
// odd y and lose it otherwise, and y = 0 gives 1 for every v.
__device__ __forceinline__ float power(float v, float y)
{
    if (y == 0.0f) return 1.0f;
    const float av = __builtin_fabsf(v);
    // the hardware logarithm reads a subnormal number as zero: scale those into range first
    const bool tiny = av < 1.17549435e-38f;
    const float l = __builtin_amdgcn_logf(tiny ? av * 4294967296.0f : av) - (tiny ? 32.0f : 0.0f);
    float r = __builtin_amdgcn_exp2f(y * l);
    if (__float_as_int(v) < 0 && v == v) {                // sign bit set, not a NaN
        const bool y_int = y == __builtin_truncf(y);
        const bool y_odd = y_int && __builtin_fabsf(y) < 16777216.0f && ((int)y & 1);
        const bool keeps = y_int || av == 0.0f || av == __builtin_inff();
        r = y_odd ? -r : (keeps ? r : __builtin_nanf(""));
    }
    return r;
}

__device__ __forceinline__ float source(const PostArgs& a, float n, int ch, float raw)
{
    float v;
    switch (a.metric) {   // uniform
    case 0: case 3: case 4: case 5: v = raw / n; break;
    case 1: case 7: v = raw; break;
    case 2: v = (raw / n + 1.0f) / 2.0f; break;
    default: v = 1.0f / (raw / n); break;
    }
    v = v / a.exposure;
    return power(v, a.inv_gamma);
}

__device__ __forceinline__ const float* source_plane(const PostArgs& a, int ch)
{
    switch (a.metric) {
    case 0: return a.planes + (size_t)(PTK_PL_COL + ch) * a.pixels;
    case 1: return a.denoised + (size_t)ch * a.pixels;
    case 7: return a.target + (size_t)ch * a.pixels;
    case 2: return a.planes + (size_t)(PTK_PL_NORMAL + ch) * a.pixels;
    case 3: return a.planes + (size_t)(PTK_PL_ALB1 + ch) * a.pixels;
    case 4: return a.planes + (size_t)(PTK_PL_ALB2 + ch) * a.pixels;
    case 5: return a.planes + (size_t)PTK_PL_DL * a.pixels;
    default: return a.planes + (size_t)(PTK_PL_WPOS + ch) * a.pixels;
    }
}

__device__ __forceinline__ void quantise(const PostArgs& a, size_t i, const float (&c)[3])
{
    // top row first: image row j lands in output row height - 1 - j
    const size_t row = i / (size_t)a.width, col = i - row * (size_t)a.width;
    uint8_t* o = a.rgb8 + 3 * (((size_t)a.height - 1 - row) * (size_t)a.width + col);
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float v = c[ch];
        double q = 255.999 * (double)v;
        int b = (v == v) ? (q >= 256.0 ? 255 : (q <= 0.0 ? 0 : (int)q)) : 0;
        o[ch] = (uint8_t)b;
    }
}

// VEC = 4: four consecutive pixels per lane through 16-byte loads and stores
// (needs pixels % 4 == 0 so every plane starts 16-byte aligned); VEC = 1 otherwise.
template <int VEC>
__global__ __launch_bounds__(256) void post_process_kernel(PostArgs a)
{
    const size_t P = a.pixels;
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (i0 >= P) return;
    const float n = (float)a.sample_count;
    float out[VEC][3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float* src = source_plane(a, ch);
        float raw[VEC];
        if (VEC == 4) {
            const float4 r = *reinterpret_cast<const float4*>(src + i0);
            raw[0] = r.x; raw[1 % VEC] = r.y; raw[2 % VEC] = r.z; raw[3 % VEC] = r.w;
        } else {
            raw[0] = src[i0];
        }
#pragma unroll
        for (int k = 0; k < VEC; k++) out[k][ch] = source(a, n, ch, raw[k]);
        float* dst = a.post + (size_t)ch * P + i0;
        if (VEC == 4)
            *reinterpret_cast<float4*>(dst) =
                make_float4(out[0][ch], out[1 % VEC][ch], out[2 % VEC][ch], out[3 % VEC][ch]);
        else
            dst[0] = out[0][ch];
    }
    if (a.rgb8) {
#pragma unroll
        for (int k = 0; k < VEC; k++) quantise(a, i0 + k, out[k]);
    }
}

}  // namespace

hipError_t ptk_launch_post_process(const PostArgs& a, hipStream_t stream)
{
    if (a.pixels % 4 == 0) {
        const unsigned grid = (unsigned)((a.pixels / 4 + 255) / 256);
        hipLaunchKernelGGL(post_process_kernel<4>, dim3(grid), dim3(256), 0, stream, a);
    } else {
        const unsigned grid = (unsigned)((a.pixels + 255) / 256);
        hipLaunchKernelGGL(post_process_kernel<1>, dim3(grid), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}
