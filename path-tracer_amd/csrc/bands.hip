// bands.hip -- device half of the multi-GPU accumulation exchange for the row-band denoiser.
//
// Every GPU renders its share of the samples of every pixel; before the denoiser, GPU d needs
// the accumulators of ITS row band (plus the halo rows the stencils read) merged over all
// GPUs.  The exchange is one all-to-all owned by the caller (RCCL, torch.distributed, MPI):
//   ptk_pack_bands      lays this GPU's 16 sum and 16 squared-deviation planes out as
//                       [destination][plane][row of the destination's span][x]
//   (all-to-all)        block d of the send buffer goes to GPU d
//   ptk_merge_received  one kernel reads the receive buffer [source][plane][row][x] once and
//                       writes the merged planes of this GPU's span in place
// The reference renders on one GPU and has no counterpart of either (Renderers.cpp:845-888 is
// the single-stream accumulation whose statistic the merge rule below extends).
//
// Merge rule, per pixel, ranks taken in rank order so that a row two GPUs both hold (a halo
// row) gets the same bits on both:
//   sum_k   = ((x_0k + x_1k) + x_2k) + ...                            16 sums
//   mean_k  = sum_k / n,  n = n_0 + n_1 + ...
//   sdv_k   = sum over ranks r, in order, of  v_rk + n_r * (x_rk / n_r - mean_k)^2
//             (the parallel-variance identity; a rank with n_r = 0 adds v_rk alone)
//   stdDev  = (sdv_3j + sdv_3j+1 + sdv_3j+2) / n  (j < 5),  sdv_15 / n
// all in IEEE fp32 with no contraction.
#include <hip/hip_runtime.h>

#include "ptk_dev.h"

namespace {

#pragma clang fp contract(off)

// One thread per (pixel of the span, feature group): group g < 5 owns sum / deviation planes
// 3g .. 3g+2 and stdDev plane g; group 5 owns plane 15 and stdDev plane 5.
__global__ __launch_bounds__(256) void merge_received_kernel(BandMergeArgs a)
{
    const size_t span_pixels = (size_t)(a.row_hi - a.row_lo) * (size_t)a.width;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int g = (int)blockIdx.y;
    if (i >= span_pixels) return;
    const size_t P = (size_t)a.width * (size_t)a.height;
    const size_t plane_stride = (size_t)a.rows_max * (size_t)a.width;   // within a source's block
    const size_t src_stride = 32 * plane_stride;
    const size_t out = (size_t)a.row_lo * (size_t)a.width + i;
    const float n = a.n_total;
    const int j0 = 3 * g, nj = g < 5 ? 3 : 1;
    float sd = 0.0f;
    for (int jj = 0; jj < nj; jj++) {
        const int j = j0 + jj;
        const float* x = a.recv + (size_t)j * plane_stride + i;
        const float* v = a.recv + (size_t)(16 + j) * plane_stride + i;
        float sum = x[0];
        for (int r = 1; r < a.world; r++) sum += x[(size_t)r * src_stride];
        const float mean = sum / n;
        float sdv = 0.0f;
        for (int r = 0; r < a.world; r++) {
            float part = v[(size_t)r * src_stride];
            const float nr = a.n_rank[r];
            if (nr > 0.0f) {
                const float d = x[(size_t)r * src_stride] / nr - mean;
                part += nr * (d * d);
            }
            sdv = r == 0 ? part : sdv + part;
        }
        a.planes[(size_t)j * P + out] = sum;
        a.planes[(size_t)(PTK_PL_SDV + j) * P + out] = sdv;
        sd = jj == 0 ? sdv : sd + sdv;
    }
    a.planes[(size_t)(PTK_PL_SD + g) * P + out] = sd / n;
}

}  // namespace

hipError_t ptk_launch_merge_received(const BandMergeArgs& a, hipStream_t stream)
{
    const size_t span_pixels = (size_t)(a.row_hi - a.row_lo) * (size_t)a.width;
    if (span_pixels == 0) return hipSuccess;
    dim3 grid((unsigned)((span_pixels + 255) / 256), 6);
    hipLaunchKernelGGL(merge_received_kernel, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}
