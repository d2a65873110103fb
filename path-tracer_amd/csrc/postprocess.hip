// postprocess.hip -- the display transform of a frame (CUDAPostProcess,
// CUDARender.cu:959-1022) and the 8-bit quantisation of PT::SaveImage
// (PT.cpp:484-501), on the planes already resident in HBM.
//
// A streaming kernel: 12 B read and 12 B (+3 B) written per pixel, one pixel per
// lane, plane-major so every access of a wave is one contiguous 256-byte run.
// pow is evaluated in double and rounded once (the reference calls powf on the
// device; the CPU restatement calls libm's, which is correctly rounded in all
// but rare cases, as this is).
#include <hip/hip_runtime.h>

#include "ptk_dev.h"

namespace {

__global__ __launch_bounds__(256) void post_process_kernel(PostArgs a)
{
    const size_t P = a.pixels;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float n = (float)a.sample_count;
    float c[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float v;
        switch (a.metric) {   // uniform
        case 0: v = a.planes[(size_t)(PTK_PL_COL + ch) * P + i] / n; break;
        case 1: v = a.denoised[(size_t)ch * P + i]; break;
        case 2: v = (a.planes[(size_t)(PTK_PL_NORMAL + ch) * P + i] / n + 1.0f) / 2.0f; break;
        case 3: v = a.planes[(size_t)(PTK_PL_ALB1 + ch) * P + i] / n; break;
        case 4: v = a.planes[(size_t)(PTK_PL_ALB2 + ch) * P + i] / n; break;
        case 5: v = a.planes[(size_t)PTK_PL_DL * P + i] / n; break;
        default: v = 1.0f / (a.planes[(size_t)(PTK_PL_WPOS + ch) * P + i] / n); break;
        }
        v = v / a.exposure;
        c[ch] = (float)__ocml_pow_f64((double)v, (double)a.inv_gamma);
        a.post[(size_t)ch * P + i] = c[ch];
    }
    if (a.rgb8) {
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
}

}  // namespace

hipError_t ptk_launch_post_process(const PostArgs& a, hipStream_t stream)
{
    const unsigned grid = (unsigned)((a.pixels + 255) / 256);
    hipLaunchKernelGGL(post_process_kernel, dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}
