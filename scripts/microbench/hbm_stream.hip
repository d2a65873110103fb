// hbm_stream.hip -- what HBM sustains on this board for read-only streams, at sizes well
// beyond the 256 MB Infinity Cache: 16-byte loads, grid-stride, several loads in flight per
// lane.  Build: hipcc -O3 --offload-arch=gfx950 -o hbm_stream hbm_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ p, size_t n, float* out)
{
    float acc = 0.0f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n; i += stride) acc += p[i].x;
    if (acc == 12345.678f) *out = acc;   // never true: keeps the loads
}

// each workgroup streams a contiguous chunk (what a tile- or row-owning kernel does)
template <int UNROLL>
__global__ __launch_bounds__(256) void read_chunk_kernel(const float4* __restrict__ p, size_t n, float* out)
{
    float acc = 0.0f;
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b = (size_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    size_t i = b + threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < e; i += UNROLL * 256) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < e; i += 256) acc += p[i].x;
    if (acc == 12345.678f) *out = acc;
}

// 4-byte loads, each wave streaming its own contiguous chunk 256 bytes at a time, optionally
// starting 4 bytes off a 256-byte boundary (what a 62-column strip with a halo column does)
template <int UNROLL>
__global__ __launch_bounds__(256) void read_dword_kernel(const float* __restrict__ p, size_t n, int skew, float* out)
{
    float acc = 0.0f;
    const size_t waves = (size_t)gridDim.x * 4, wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t per = (n / waves) & ~(size_t)63;
    const size_t b = wave * per + (size_t)skew, e = wave * per + per;
    size_t i = b + (threadIdx.x & 63);
    for (; i + (UNROLL - 1) * 64 < e; i += UNROLL * 64) {
        float v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = p[i + u * 64];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    if (acc == 12345.678f) *out = acc;
}

// the filter's pattern: PLANES planes of `rows` x `width` floats; a wave owns a 62-column strip
// and a run of rows, and per row loads 64 floats (one halo column either side) from each plane
template <int PLANES, bool NT = false>
__global__ __launch_bounds__(256, 3) void read_planes_kernel(const float* __restrict__ p, int width, int rows,
                                                             int rows_per_wave, float* out, size_t plane_pad = 0)
{
    const int strips = (width + 61) / 62, groups = (strips + 3) / 4;
    const int seg = blockIdx.x / groups, grp = blockIdx.x - seg * groups;
    const int x0 = (grp * 4 + (threadIdx.x >> 6)) * 62;
    if (x0 >= width) return;
    int x = x0 - 1 + (threadIdx.x & 63);
    x = x < 0 ? 0 : (x >= width ? width - 1 : x);
    const size_t plane = (size_t)width * rows + plane_pad;
    float acc = 0.0f;
    const int y0 = seg * rows_per_wave, y1 = y0 + rows_per_wave < rows ? y0 + rows_per_wave : rows;
    for (int y = y0; y < y1; y++) {
        float v[PLANES];
#pragma unroll
        for (int k = 0; k < PLANES; k++)
            v[k] = NT ? __builtin_nontemporal_load(p + (size_t)k * plane + (size_t)y * width + x)
                      : p[(size_t)k * plane + (size_t)y * width + x];
#pragma unroll
        for (int k = 0; k < PLANES; k++) acc += v[k];
    }
    if (acc == 12345.678f) *out = acc;
}

// the same walk with COLS adjacent columns per lane (one 4*COLS-byte load per plane and row): a
// wave owns 64*COLS - 2 columns, so the pieces it streams are 256*COLS bytes long
template <int PLANES, int COLS>
__global__ __launch_bounds__(256, 2) void read_planes_wide_kernel(const float* __restrict__ p, int width, int rows,
                                                                  int rows_per_wave, float* out)
{
    constexpr int OWN = 64 * COLS - 2;
    const int strips = (width + OWN - 1) / OWN, groups = (strips + 3) / 4;
    const int seg = blockIdx.x / groups, grp = blockIdx.x - seg * groups;
    const int x0 = (grp * 4 + (threadIdx.x >> 6)) * OWN;
    if (x0 >= width) return;
    int x = x0 - 1 + (threadIdx.x & 63) * COLS;
    x = x < 0 ? 0 : (x + COLS > width ? width - COLS : x);
    const size_t plane = (size_t)width * rows;
    float acc = 0.0f;
    const int y0 = seg * rows_per_wave, y1 = y0 + rows_per_wave < rows ? y0 + rows_per_wave : rows;
    struct __attribute__((packed, aligned(4))) Piece { float v[COLS]; };
    for (int y = y0; y < y1; y++) {
        Piece v[PLANES];
#pragma unroll
        for (int k = 0; k < PLANES; k++) v[k] = *(const Piece*)(p + (size_t)k * plane + (size_t)y * width + x);
#pragma unroll
        for (int k = 0; k < PLANES; k++)
#pragma unroll
            for (int c = 0; c < COLS; c++) acc += v[k].v[c];
    }
    if (acc == 12345.678f) *out = acc;
}

template <class F>
double time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; r++) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    for (size_t mb : {240, 1024, 4096}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        float4* p;
        float* out;
        if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
        hipMemset(p, 0, bytes);
        for (int grid : {1024, 2048, 4096, 8192, 16384}) {
            double t4 = time_ms([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, 0, p, n, out); }, 10);
            double t8 = time_ms([&] { hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, p, n, out); }, 10);
            double c8 = time_ms([&] { hipLaunchKernelGGL(read_chunk_kernel<8>, dim3(grid), dim3(256), 0, 0, p, n, out); }, 10);
            printf("%5zu MB grid %5d: strided x4 %.0f GB/s, strided x8 %.0f GB/s, chunked x8 %.0f GB/s\n", mb, grid,
                   bytes / t4 / 1e6, bytes / t8 / 1e6, bytes / c8 / 1e6);
        }
        for (int skew : {0, 1}) {
            double d8 = time_ms([&] { hipLaunchKernelGGL(read_dword_kernel<8>, dim3(768), dim3(256), 0, 0, (const float*)p, n * 4, skew, out); }, 10);
            double d32 = time_ms([&] { hipLaunchKernelGGL(read_dword_kernel<32>, dim3(768), dim3(256), 0, 0, (const float*)p, n * 4, skew, out); }, 10);
            double d32b = time_ms([&] { hipLaunchKernelGGL(read_dword_kernel<32>, dim3(2048), dim3(256), 0, 0, (const float*)p, n * 4, skew, out); }, 10);
            printf("%5zu MB 4-byte loads, skew %d: 768 workgroups x8 %.0f GB/s, x32 %.0f GB/s; 2048 workgroups x32 %.0f GB/s\n",
                   mb, skew, bytes / d8 / 1e6, bytes / d32 / 1e6, bytes / d32b / 1e6);
        }
        hipFree(p);
        hipFree(out);
    }
    {   // 29 planes of 3840 x 2160 (962 MB), the filter's walk: 16 column groups x 48 segments of 45 rows
        const int W = 3840, H = 2160;
        const size_t bytes = (size_t)29 * W * H * 4;
        float *p, *out;
        if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
        hipMemset(p, 0, bytes);
        for (int rpw : {45, 30, 23, 15}) {
            const int segs = (H + rpw - 1) / rpw, groups = 16;
            double t = time_ms([&] { hipLaunchKernelGGL((read_planes_kernel<29, false>), dim3(segs * groups), dim3(256), 0, 0, p, W, H, rpw, out, (size_t)0); }, 10);
            printf("29 planes 3840x2160, strips of 62 columns, %d rows per wave (%d workgroups): %.0f GB/s\n", rpw, segs * groups, bytes / t / 1e6);
        }
        for (int rpw : {45, 15}) {   // the same with non-temporal loads (the planes are read once)
            const int segs = (H + rpw - 1) / rpw, groups = 16;
            double t = time_ms([&] { hipLaunchKernelGGL((read_planes_kernel<29, true>), dim3(segs * groups), dim3(256), 0, 0, p, W, H, rpw, out, (size_t)0); }, 10);
            printf("29 planes 3840x2160, strips of 62 columns, %d rows per wave, non-temporal loads: %.0f GB/s\n", rpw, bytes / t / 1e6);
        }
        for (int rpw : {45, 23, 11}) {
            const int segs = (H + rpw - 1) / rpw;
            double t2 = time_ms([&] { hipLaunchKernelGGL((read_planes_wide_kernel<29, 2>), dim3(segs * 8), dim3(256), 0, 0, p, W, H, rpw, out); }, 10);
            double t4 = time_ms([&] { hipLaunchKernelGGL((read_planes_wide_kernel<29, 4>), dim3(segs * 4), dim3(256), 0, 0, p, W, H, rpw, out); }, 10);
            printf("29 planes 3840x2160, %d rows per wave: two columns per lane (strips of 126, %d workgroups) %.0f GB/s, "
                   "four (strips of 254, %d workgroups) %.0f GB/s\n", rpw, segs * 8, bytes / t2 / 1e6, segs * 4, bytes / t4 / 1e6);
        }
        hipFree(p);
        hipFree(out);
    }
    {   // the same walk with the planes a little further apart than width x rows floats: do 29
        // streams that start at the same offset within every 16 KB collide on a channel?
        const int W = 3840, H = 2160;
        for (size_t pad : {(size_t)0, (size_t)64, (size_t)1024 + 64, (size_t)4096 + 192, (size_t)16384 + 448}) {
            const size_t bytes = (size_t)29 * ((size_t)W * H + pad) * 4;
            float *p, *out;
            if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
            hipMemset(p, 0, bytes);
            const int rpw = 45, segs = (H + rpw - 1) / rpw, groups = 16;
            double t = time_ms([&] { hipLaunchKernelGGL((read_planes_kernel<29, false>), dim3(segs * groups), dim3(256), 0, 0, p, W, H, rpw, out, pad); }, 10);
            printf("29 planes 3840x2160, 45 rows per wave, planes %zu floats further apart: %.0f GB/s\n", pad,
                   (double)29 * W * H * 4 / t / 1e6);
            hipFree(p);
            hipFree(out);
        }
    }
    return 0;
}
