// exposure.hip -- luminance map + sum over the colour planes
// (Renderers.cpp:1151-1159 and 1174-1193), and the stdDev refresh used after
// a cross-GPU reduction.
//
// HBM-bound: 12 bytes read per pixel, one double atomic per workgroup.  The
// atomics are spread over PTK_EXPOSURE_SHARDS partial sums, each on a cache
// line of its own (same-address atomics retire one at a time, ~12 ns each, which
// would cost more than the whole read); the host adds the shards in index order.  Each
// lane reads float4 slices, evaluates the reference's fp32 per-pixel term and
// sums in double; the wave reduces with DPP row operations.
#include <hip/hip_runtime.h>

#include <mutex>

#include "ptk_dev.h"

#pragma clang fp contract(off)

namespace {

template <int CTRL>
__device__ __forceinline__ double dpp_add(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}

// Sum over the 64 lanes of a wave; the result is valid in every lane.
__device__ __forceinline__ double wave_sum(double v)
{
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror: every lane of a 16-lane row holds the row sum
    int lo = __double2loint(v), hi = __double2hiint(v);
    double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

template <int FORMULA>
__device__ __forceinline__ float luminance(float r, float g, float b)
{
    if (FORMULA == 1)  // SkePUExposureFuncCalc: two of the products are double
        return (float)((double)(0.2125f * r) + 0.7154 * (double)g + 0.0721 * (double)b);
    return 0.2125f * r + 0.7154f * g + 0.0721f * b;
}

template <int FORMULA>
__global__ __launch_bounds__(256) void exposure_kernel(const float* __restrict__ planes, size_t P,
                                                       double* __restrict__ sum)
{
    const float* pr = planes;
    const float* pg = planes + P;
    const float* pb = planes + 2 * P;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    double part = 0.0;
    // 16-byte loads need all three planes 16-byte aligned: a pixel count that is a multiple of
    // four and an aligned base; anything else (a caller's sub-view at a 4-byte offset) takes
    // the one-pixel loop
    if ((P & 3) == 0 && ((unsigned long long)planes & 15ull) == 0) {
        const size_t n4 = P >> 2;
        const float4* r4 = (const float4*)pr;
        const float4* g4 = (const float4*)pg;
        const float4* b4 = (const float4*)pb;
        for (size_t i = tid; i < n4; i += nthreads) {
            float4 r = r4[i], g = g4[i], b = b4[i];
            // every per-pixel term is the reference's fp32 value; the sum is double
            part += ((double)luminance<FORMULA>(r.x, g.x, b.x) +
                     (double)luminance<FORMULA>(r.y, g.y, b.y)) +
                    ((double)luminance<FORMULA>(r.z, g.z, b.z) +
                     (double)luminance<FORMULA>(r.w, g.w, b.w));
        }
    } else {
        for (size_t i = tid; i < P; i += nthreads)
            part += (double)luminance<FORMULA>(pr[i], pg[i], pb[i]);
    }
    part = wave_sum(part);
    __shared__ double wave_part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_part[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(sum + (blockIdx.x % PTK_EXPOSURE_SHARDS) * PTK_EXPOSURE_SHARD_STRIDE,
                  (wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]));
}

__global__ __launch_bounds__(256) void refresh_std_dev_kernel(float* __restrict__ planes, size_t P,
                                                              float n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    for (int k = 0; k < 5; k++) {
        float v0 = planes[(size_t)(PTK_PL_SDV + 3 * k + 0) * P + i];
        float v1 = planes[(size_t)(PTK_PL_SDV + 3 * k + 1) * P + i];
        float v2 = planes[(size_t)(PTK_PL_SDV + 3 * k + 2) * P + i];
        planes[(size_t)(PTK_PL_SD + k) * P + i] = (v0 + v1 + v2) / n;
    }
    planes[(size_t)(PTK_PL_SD + 5) * P + i] = planes[(size_t)(PTK_PL_SDV + 15) * P + i] / n;
}

}  // namespace

hipError_t ptk_launch_exposure(const float* planes, size_t pixels, int formula, double* sum_d,
                               hipStream_t stream)
{
    size_t work = (pixels & 3) == 0 ? pixels / 4 : pixels;
    size_t blocks = (work + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (formula == 1)
        hipLaunchKernelGGL(exposure_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, planes,
                           pixels, sum_d);
    else
        hipLaunchKernelGGL(exposure_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, planes,
                           pixels, sum_d);
    return hipGetLastError();
}

// ---- the exposure reduce for callers that cannot allocate.
//
// The shard sums live in a file-scope device array instead of a hipMalloc'd
// buffer, and the result is finished on the device (a one-thread kernel adds the
// shards in index order and applies the formula exactly as the host code of
// ptk_auto_exposure does) and stored through device pointers, so the call
// enqueues a memset and two kernels and nothing else.  Every GPU holds its own
// copy of the array at its own address; the address is looked up once per
// device, with that device current, and kept.  All calls on one GPU share the
// array: it is cleared at the start of each call, and calls that overlap on one
// GPU (different streams, or different threads) race on it.
namespace {

static __device__ __align__(256) double scratch[PTK_EXPOSURE_SUM_DOUBLES];   // 4 KiB per GPU

constexpr int MAX_DEVICES = 64;
double* g_scratch_ptr[MAX_DEVICES];     // device address of `scratch` on each GPU, once known
std::mutex g_scratch_mutex;

__global__ void exposure_finish_kernel(const double* __restrict__ shards, int formula, int width,
                                       int height, unsigned long long pixels, int sample_count,
                                       float* __restrict__ exposure, double* __restrict__ raw_sum)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double sum = 0.0;
    for (int i = 0; i < PTK_EXPOSURE_SHARDS; i++) sum += shards[i * PTK_EXPOSURE_SHARD_STRIDE];
    if (raw_sum) *raw_sum = sum;
    if (exposure) {
        const float n = (float)sample_count;
        float e = (float)sum;
        if (formula == PTK_EXPOSURE_CPU)
            e *= 9.6f / (float)width * (float)height * n;   // as written, left to right
        else if (formula == PTK_EXPOSURE_SKEPU)
            e = 9.6f * e / (float)(int)pixels * n;
        else
            e = (float)(9.6 * sum / ((double)pixels * (double)sample_count));
        *exposure = e;
    }
}

}  // namespace

hipError_t ptk_exposure_static_setup(double** ptr_out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        double* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[dev] = p;
    }
    if (ptr_out) *ptr_out = g_scratch_ptr[dev];
    return hipSuccess;
}

hipError_t ptk_launch_exposure_static(const float* planes, size_t pixels, int width, int height,
                                      int sample_count, int formula, float* exposure_d,
                                      double* raw_sum_d, hipStream_t stream)
{
    double* shards = nullptr;
    hipError_t e = ptk_exposure_static_setup(&shards);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(shards, 0, PTK_EXPOSURE_SUM_DOUBLES * sizeof(double), stream);
    if (e != hipSuccess) return e;
    e = ptk_launch_exposure(planes, pixels, formula == PTK_EXPOSURE_SKEPU ? 1 : 0, shards, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(exposure_finish_kernel, dim3(1), dim3(64), 0, stream, shards, formula, width,
                       height, (unsigned long long)pixels, sample_count, exposure_d, raw_sum_d);
    return hipGetLastError();
}

hipError_t ptk_launch_refresh_std_dev(float* planes, size_t pixels, int sample_count,
                                      hipStream_t stream)
{
    unsigned blocks = (unsigned)((pixels + 255) / 256);
    hipLaunchKernelGGL(refresh_std_dev_kernel, dim3(blocks), dim3(256), 0, stream, planes, pixels,
                       (float)sample_count);
    return hipGetLastError();
}
