// train.hip -- the reductions of one denoiser training step: the relative MSE
// of the filtered image against a target (DenoiserNN::GenRelMSE,
// DenoiserNN.cpp:1461-1474) and the weight update of CPUBackProp with its
// CPUFilterDerivative (DenoiserNN.cpp:959-1103).
//
// The reference's update adds, for every pixel and each of the seven filter
// parameters, learningRate * dot(errorOverColour, colourOverParam) * input to
// every weight, where `input` depends only on the weight's index modulo its
// layer's fan-in.  Per pixel that is one scalar G (the sum of the seven dot
// products) times 36 + 10 + 10 inputs, so the whole step is 56 sums of
// G * input over the pixels plus the relMSE sum: 57 reductions, accumulated
// here per thread over a grid-stride run of pixels, then across the wave and
// into sharded double atomics.  The reference accumulates sequentially in
// fp32; results are compared within a tolerance.
//
// CPUFilterDerivative is followed as written, including what looks unintended:
// its flat index of pixel (j, i) is j * yRes + i, neighbours clamp rows at
// xRes - 1 and columns at yRes - 1, the spatial term uses the absolute j and i,
// and the colour difference is the raw colour SUM minus the filtered colour.
// Its reads are therefore not a stencil around the pixel and go through the
// caches rather than an LDS tile.
#include <hip/hip_runtime.h>

#include "ptk_dev.h"

namespace {

#define DEV __device__ __forceinline__

DEV double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;   // lane 0 holds the sum
}

// G of pixel (j, i): sum over var of dot(errorOverColour, CPUFilterDerivative(j, i, var)).
DEV float pixel_gradient(const TrainArgs& a, int j, int i, size_t P, float& rel)
{
    const int xRes = a.width, yRes = a.height, N = a.n;
    const float n = (float)a.sample_count;
    const size_t pixel = (size_t)j * xRes + i;
    float eoc[3];
    rel = 0.0f;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float t = a.target[(size_t)ch * P + pixel];
        const float d = a.denoised[(size_t)ch * P + pixel];
        eoc[ch] = n * (d - t) / (t * t + 0.0001f);
        const float diff = d - t;
        rel += diff * diff / (t * t + 0.00001f);
    }

    const int tW = yRes, tH = xRes;
    const size_t iIndex = (size_t)j * tW + i;
    float var[7], sd_i[6], ci[16], den_i[3];
#pragma unroll
    for (int k = 0; k < 7; k++) var[k] = a.variances[(size_t)k * P + iIndex];
#pragma unroll
    for (int k = 0; k < 6; k++) sd_i[k] = a.planes[(size_t)(PTK_PL_SD + k) * P + iIndex];
#pragma unroll
    for (int k = 0; k < 16; k++) ci[k] = a.planes[(size_t)k * P + iIndex] / n;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) den_i[ch] = a.denoised[(size_t)ch * P + iIndex];
    const float wc = a.wc_sum[iIndex];
    float inv_cube[7];
#pragma unroll
    for (int k = 0; k < 7; k++) inv_cube[k] = 1.0f / (var[k] * var[k] * var[k]);
    const float pd0 = (float)j * (float)j + (float)i * (float)i;   // exact below 2^24
    const float dv0 = pd0 / (2.0f * var[0] + 0.000001f);

    float cop[7][3];
#pragma unroll
    for (int v = 0; v < 7; v++) cop[v][0] = cop[v][1] = cop[v][2] = 0.0f;

    for (int j1 = -N; j1 <= N; j1++) {
        for (int i1 = -N; i1 <= N; i1++) {
            size_t jIndex = (size_t)(j + j1 < 0 ? 0 : (j + j1 >= tH ? tH - 1 : j + j1));
            jIndex *= (size_t)tW;
            jIndex += (size_t)(i + i1 < 0 ? 0 : (i + i1 >= tW ? tW - 1 : i + i1));
            float raw[16];
#pragma unroll
            for (int k = 0; k < 16; k++) raw[k] = a.planes[(size_t)k * P + jIndex];
            const float sd0_j = a.planes[(size_t)PTK_PL_SD * P + jIndex];

            float pd[7], dv[7];
            pd[0] = pd0;
            dv[0] = dv0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                float s2 = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const float d = ci[3 * k + ch] - raw[3 * k + ch] / n;
                    s2 += d * d;
                }
                pd[1 + k] = s2;
            }
            {
                const float d = ci[15] - raw[15] / n;
                pd[6] = d * d;
            }
            dv[1] = pd[1] / (2.0f * var[1] * (sd_i[0] + sd0_j) + 0.000001f);
#pragma unroll
            for (int k = 2; k < 7; k++) dv[k] = pd[k] / (2.0f * var[k] * sd_i[k - 1] + 0.000001f);
            float mult = 1.0f;
#pragma unroll
            for (int k = 0; k < 7; k++) mult *= __expf(-dv[k]) + 0.000001f;
            float vec[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) vec[ch] = (raw[ch] - den_i[ch]) / wc;
#pragma unroll
            for (int v = 0; v < 7; v++) {
                const float wop = mult * pd[v] * inv_cube[v];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) cop[v][ch] += vec[ch] * wop;
            }
        }
    }
    float G = 0.0f;
#pragma unroll
    for (int v = 0; v < 7; v++) G += eoc[0] * cop[v][0] + eoc[1] * cop[v][1] + eoc[2] * cop[v][2];
    return G;
}

__global__ __launch_bounds__(256) void train_kernel(TrainArgs a)
{
    const size_t P = (size_t)a.width * (size_t)a.height;
    float acc[PTK_TRAIN_TERMS];
#pragma unroll
    for (int k = 0; k < PTK_TRAIN_TERMS; k++) acc[k] = 0.0f;

    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += stride) {
        const int j = (int)(p / (size_t)a.width), i = (int)(p - (size_t)j * a.width);
        float rel;
        const float G = pixel_gradient(a, j, i, P, rel);
        acc[0] += rel;
#pragma unroll
        for (int k = 0; k < 36; k++) acc[1 + k] += G * a.features[(size_t)k * P + p];
#pragma unroll
        for (int k = 0; k < 20; k++) acc[37 + k] += G * a.layers[(size_t)k * P + p];
    }

    // wave sums in double, then one atomic per wave and term on this workgroup's shard
    const int lane = threadIdx.x & 63;
    double* shard = a.sums + (size_t)(blockIdx.x % PTK_TRAIN_SHARDS) * PTK_TRAIN_TERMS;
#pragma unroll
    for (int k = 0; k < PTK_TRAIN_TERMS; k++) {
        const double s = wave_sum((double)acc[k]);
        if (lane == 0) atomicAdd(shard + k, s);
    }
}

__global__ __launch_bounds__(256) void save_target_kernel(const float* __restrict__ planes,
                                                          float* __restrict__ target, size_t P,
                                                          float n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * P) return;
    target[i] = planes[i] / n;   // colour planes are the first three
}

}  // namespace

hipError_t ptk_launch_train(const TrainArgs& a, hipStream_t stream)
{
    const size_t P = (size_t)a.width * (size_t)a.height;
    size_t blocks = (P + 255) / 256;
    if (blocks > 2048) blocks = 2048;   // eight resident workgroups per CU; the rest by stride
    hipLaunchKernelGGL(train_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t ptk_launch_save_target(const float* planes, float* target, size_t pixels,
                                  int sample_count, hipStream_t stream)
{
    const unsigned grid = (unsigned)((3 * pixels + 255) / 256);
    hipLaunchKernelGGL(save_target_kernel, dim3(grid), dim3(256), 0, stream, planes, target, pixels,
                       (float)sample_count);
    return hipGetLastError();
}
