// denoise.hip -- denoiser inference: neighbourhood feature statistics + MLP
// (DenoiserNN.cpp:12-301) and the cross-bilateral filter (Denoiser.cpp:222-355).
//
// Both passes are LDS-tiled stencils with clamp-to-edge halos.  A block of 256
// threads produces a 32x8 tile of pixels; a wave covers two 32-pixel rows, so
// the 32 lanes of a half-wave always read 32 consecutive LDS words (one bank
// each) whatever the tap offset.  Discrete parts (Sobel order, the "median
// getter" procedure, tie-breaking) consume bit-exact primary features; the
// smooth parts (7x7 variance, mean deviation, MLP, Gaussian weights) are
// evaluated in fp32 with FMA and the hardware exp/log, and are compared with
// the reference within a tolerance.
#include <hip/hip_runtime.h>

#include "ptk_dev.h"
#include "exact_ops.h"

#define DEV __device__ __forceinline__

namespace {

// waves per SIMD the register allocation is capped for, per kernel
#ifndef PTK_DN_WAVES
#define PTK_DN_WAVES 3
#endif
// measurement builds: bit 0 drops the "median getter", bit 1 the 7x7 statistics; in the tolerance-held
// form 64 drops the 7x7 statistics, 128 nine tenths of layers two and three, 256 five sixths of layer one
#ifndef PTK_DN_ABLATE
#define PTK_DN_ABLATE 0
#endif
#ifndef PTK_DN_WAVES_FILTER
#define PTK_DN_WAVES_FILTER 3
#endif
constexpr int TW = 32, TH = 8;         // output tile
constexpr int FH = 3;                  // feature halo (7x7 window)
constexpr int FW = TW + 2 * FH;        // 38
constexpr int FR = TH + 2 * FH;        // 14
constexpr int FPLANE = FW * FR;        // 532 words per feature plane
constexpr int PTILE_FLOATS = 10 * FPLANE;   // primaries and primaries / 49
constexpr int CW = FW;                 // columns of a column-statistics plane
constexpr int CPLANE = CW * TH;        // 304 words: one entry per tile column and output row
#ifndef PTK_DN_LDS_PAD
#define PTK_DN_LDS_PAD 0               // measurement builds: unused LDS words, to lower occupancy
#endif

DEV int clampi(int v, int hi) { return v < 0 ? 0 : (v >= hi ? hi - 1 : v); }

DEV float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
DEV float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// x / d for many x and one d, bit-identical to the IEEE quotient (exact_ops.h).
// Denominators in this file are sample counts, 3*count, 49 and 9, all far inside
// the range where the compiler's division scales nothing.
struct Recip { float r, d; };
DEV Recip make_recip(float d) { return {ptk_exact::exact_reciprocal(d), d}; }
DEV float div_by(float x, const Recip& R) { return ptk_exact::exact_quotient_fixup(x, R.r, R.d); }

// The reference's "median getter" procedure (DenoiserNN.cpp:190-256).  It is
// not a median: each pass pushes the nine values through a five-slot array
// with the rules below, and slot 0 is the answer.  The reference scans
// k = 1..4 and stops at the first k with !(v > m[k]) (the comparison with m[0]
// has no effect in either pass).  Written here as selects on five registers:
// the comparisons all use the array as it was before the insertion, exactly as
// the scan does, because the scan only writes at the step where it stops.
//   pass 1: stop at k -> m[k-1] = v;           never stopped -> m[0..3] = m[4], m[4] = v
//   pass 2: stop at k -> m[0..k-1] = m[k], m[k] = v;  never stopped -> the same as k = 4
#pragma clang fp contract(off)
struct Slots { float m0, m1, m2, m3, m4; };
DEV void mg_insert_pass1(float v, Slots& s)
{
    const bool g1 = v > s.m1, g2 = v > s.m2, g3 = v > s.m3, g4 = v > s.m4;
    const bool a2 = g1 && g2, a3 = a2 && g3, a4 = a3 && g4;
    const float m4 = s.m4;
    s.m0 = a4 ? m4 : (g1 ? s.m0 : v);
    s.m1 = a4 ? m4 : ((g1 && !g2) ? v : s.m1);
    s.m2 = a4 ? m4 : ((a2 && !g3) ? v : s.m2);
    s.m3 = a4 ? m4 : ((a3 && !g4) ? v : s.m3);
    s.m4 = a4 ? v : m4;
}
DEV void mg_insert_pass2(float v, Slots& s)
{
    const bool g1 = v > s.m1, g2 = v > s.m2, g3 = v > s.m3;
    const bool c1 = !g1, c2 = g1 && !g2, a2 = g1 && g2, c3 = a2 && !g3, c4 = a2 && g3;
    const float fill = c1 ? s.m1 : (c2 ? s.m2 : (c3 ? s.m3 : s.m4));
    const float m2 = s.m2, m3 = s.m3;
    s.m0 = fill;
    s.m1 = c1 ? v : fill;
    s.m2 = c1 ? m2 : (c2 ? v : fill);
    s.m3 = (c1 || c2) ? m3 : (c3 ? v : fill);
    s.m4 = c4 ? v : s.m4;
}
// The same procedure as a network in which every select is one v_cndmask on a wave mask
// held in scalar registers (mad_network.h).  Written with explicit instructions because the
// compiler turns the select form above into a branch per insertion (all four upper slots
// share one condition), which costs more than the selects and keeps the five channels'
// chains from being interleaved.
#define PTK_MAD_FN __device__ __forceinline__
#define PTK_MAD_UNROLL _Pragma("unroll")
#include "mad_network.h"
struct WaveOps {
    using Mask = unsigned long long;
    static DEV Mask gt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 2 /* ordered > */); }
    static DEV Mask land(Mask a, Mask b) { return a & b; }
    static DEV Mask lor(Mask a, Mask b) { return a | b; }
    static DEV Mask andn(Mask a, Mask b) { return a & ~b; }
    static DEV Mask lnot(Mask a) { return ~a; }
    static DEV float sel(Mask c, float x, float y)
    {
        float r;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(y), "v"(x), "s"(c));
        return r;
    }
    static DEV float fabs(float x) { return __builtin_fabsf(x); }
};
#ifndef PTK_DN_MAD_NETWORK
#define PTK_DN_MAD_NETWORK 1
#endif
#ifndef PTK_DN_FLAT_WINDOWS
#define PTK_DN_FLAT_WINDOWS 1   // the wave-wide shortcut for flat windows in features_and_mlp
#endif

DEV float mad_procedure(float (&v)[9])
{
#if PTK_DN_MAD_NETWORK
    return ptk_mad::Network<WaveOps>::run(v);
#endif
    Slots s = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 9; i++) mg_insert_pass1(v[i], s);
    const float m0 = s.m0;
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] = __builtin_fabsf(v[i] - m0);
    s = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 9; i++) mg_insert_pass2(v[i], s);
    return s.m0;
}

// The float at byte offset `byte` of a plane whose base is the same for the whole wave: the
// base stays in scalar registers and the lane's part of the address is one 32-bit offset
// (a frame has at most 2^28 pixels), where base + 64-bit index costs every load a 64-bit
// multiply-add in vector registers.
typedef __attribute__((address_space(1))) char GlobalByte;
DEV float plane_at(const float* base, unsigned byte)
{
    GlobalByte* b = (GlobalByte*)(unsigned long long)base;
    asm("" : "+s"(b));   // stays in scalar registers: the lane offset is not folded in
    return *(const __attribute__((address_space(1))) float*)(b + byte);
}
DEV void plane_put(float* base, unsigned byte, float v)
{
    GlobalByte* b = (GlobalByte*)(unsigned long long)base;
    asm("" : "+s"(b));
    *(__attribute__((address_space(1))) float*)(b + byte) = v;
}

// Primary feature of one pixel (DenoiserNN.cpp:16-36): three separate
// quotients summed in order, bit-exact.
DEV float primary3(const float* __restrict__ p, size_t P, size_t i, const Recip& three_n)
{
    float v = 0.0f;
    v += div_by(p[i], three_n);
    v += div_by(p[P + i], three_n);
    v += div_by(p[2 * P + i], three_n);
    return v;
}
#pragma clang fp contract(fast)

// Two multiply-adds in one v_pk_fma_f32: the weights of two consecutive nodes are one
// scalar register pair, the input is the same for both halves.  Each half is an IEEE fma,
// so the sums have the bits the one-at-a-time form gives.  An fma whose weight sits in a
// scalar register issues at the slow rate; the packed form moves two of them at about the
// cost of one.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) f2* W2;
#ifndef PTK_DN_PK_FMA
#define PTK_DN_PK_FMA 1
#endif
DEV f2 pk_fma(float x, f2 w, f2 a)
{
    const f2 xx = {x, x};
    return __builtin_elementwise_fma(xx, w, a);
}

DEV float sigmoid(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
DEV float softplus(float x) { return fast_log(1.0f + fast_exp(x)); }

// Device weight layout (built by the host from the reference's file order):
//   [0, 350)   first layer, regrouped [channel k][group g][node]: w12[36*node + 5*g + k]
//   [350, 360) first layer, input 35 (1/spp) per node
//   [360, 460) second layer, transposed: w23[10*node + input] at 360 + 10*input + node
//   [460, 540) third layer, transposed:  w34[10*node + input] at 460 + 8*input + node
//              (seven nodes, rows padded to eight)
// so the weights of two consecutive nodes for one input are an aligned pair everywhere.
//
// 36 features + MLP for the pixel at tile coordinates (ly, lx) of an LDS tile
// `t` of 10 planes [FR][FW]: the primaries q, then q/49 (the terms the reference
// adds up for its 7x7 mean, each rounded once; that mean is reproduced bit for bit.
// The 3x3 mean only feeds the mean absolute deviation, which is compared within a
// tolerance, and is formed as sum / 9).
//
// The five feature channels are processed one after another in a rolled loop
// and each channel's seven features are folded into the ten first-layer sums
// as soon as they exist, so the live state is ten accumulators rather than 36
// features (the first layer's summation order differs from the reference's
// index order; the MLP is compared within a tolerance either way).
// Feature index of group g (0 value, 1 sd, 2 mean7, 3 sd7, 4 gradient,
// 5 mean deviation, 6 "MAD") and channel k is 5*g + k; index 35 is 1/spp.
//
// FAST is the tolerance-held form (ptk_denoise variant 2): the discrete parts (primaries,
// "median getter") are the same instructions on the same values, the smooth statistics are
// formed a cheaper way -- the 7x7 mean and deviation from per-column statistics shared
// through LDS (`cs`, see column_stats), the Sobel magnitude from symmetric partial sums, roots
// from the hardware instruction -- and agree with the other form to a few ulp of the
// channel's magnitude.  `t` then holds the five primaries only.
template <bool FAST>
DEV void features_and_mlp(const float* t, const float* cs, int ly, int lx, const float (&sd5)[5],
                          float inv_spp, const __attribute__((address_space(4))) float* w,
                          float* features_out, float* layers_out, size_t P,
                          unsigned idb /* pixel's byte offset */, float (&var)[7])
{
    const W2 w2 = (W2)w;               // node pairs: w2[i] = {w[2i], w[2i+1]}
    f2 a1[5];                          // first-layer sums of nodes (2p, 2p+1)
#pragma unroll
    for (int p = 0; p < 5; p++) a1[p] = f2{inv_spp, inv_spp} * w2[175 + p];

#pragma unroll 1
    for (int k = 0; k < 5; k++) {
        const float* tk = t + k * FPLANE + ly * FW + lx;   // window origin (top-left of 7x7)
        float fk[7];
        fk[0] = tk[FH * FW + FH];
        fk[1] = k == 0 ? sd5[0] : k == 1 ? sd5[1] : k == 2 ? sd5[2] : k == 3 ? sd5[3] : sd5[4];

        float mean7 = 0.0f;
        float ss = 0.0f;
        if (FAST && (PTK_DN_ABLATE & 64)) {   // measurement build: no 7x7 statistics
            mean7 = tk[0];
            fk[3] = tk[1];
        } else if (FAST) {
            // The window's seven columns each bring their mean c_i and their squared deviations
            // about it, S_i (column_stats).  Window mean = mean of the c_i; squared deviations
            // about it = sum of S_i + 7 * sum of (c_i - mean)^2, an identity in which every
            // square is of a centred quantity, so nothing cancels.
            const float* ck = cs + (2 * k) * CPLANE + ly * CW + lx;
            float c[7];
#pragma unroll
            for (int i = 0; i < 7; i++) c[i] = ck[i];
            mean7 = (((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + c[6])) * (1.0f / 7.0f);
            const float* sk = ck + CPLANE;
            float within = ((sk[0] + sk[1]) + (sk[2] + sk[3])) + ((sk[4] + sk[5]) + sk[6]);
            float between = 0.0f;
#pragma unroll
            for (int i = 0; i < 7; i++) {
                float d = c[i] - mean7;
                between = __builtin_fmaf(d, d, between);
            }
            ss = __builtin_fmaf(7.0f, between, within);
            fk[3] = __builtin_amdgcn_sqrtf(ss * (1.0f / 49.0f));
        } else {
            const float* tk49 = tk + 5 * FPLANE;
#if PTK_DN_ABLATE & 2
            mean7 = tk49[0]; ss = tk[0];             // measurement build: no 7x7 statistics
#else
            // 7x7 mean: the reference adds q/49 term by term in row-major order
#pragma unroll
            for (int j = 0; j < 7; j++)
#pragma unroll
                for (int i = 0; i < 7; i++) mean7 += tk49[j * FW + i];
            // 7x7 standard deviation about that mean
#pragma unroll
            for (int j = 0; j < 7; j++)
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    float d = tk[j * FW + i] - mean7;
                    ss = __builtin_fmaf(d, d, ss);
                }
#endif
            // the compiler's root without its rescaling of arguments below 2^-96 (exact_ops.h):
            // the same bits above that, and this feature is compared within a tolerance
            fk[3] = ptk_exact::exact_sqrt_unscaled(ss * (1.0f / 49.0f));
        }
        fk[2] = mean7;

        float win[9];
        float mean3 = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int o = (FH - 1 + j) * FW + FH - 1 + i;
                win[j * 3 + i] = tk[o];
                mean3 += tk[o];
            }
        mean3 *= 1.0f / 9.0f;
        if (FAST) {
            // each gradient as the difference of two three-tap sums formed the same way, so a
            // flat window gives exactly zero
            const float gx = (__builtin_fmaf(2.0f, win[1], win[0]) + win[2]) -
                             (__builtin_fmaf(2.0f, win[7], win[6]) + win[8]);
            const float gy = (__builtin_fmaf(2.0f, win[3], win[0]) + win[6]) -
                             (__builtin_fmaf(2.0f, win[5], win[2]) + win[8]);
            fk[4] = __builtin_amdgcn_sqrtf(__builtin_fmaf(gx, gx, gy * gy));
        } else {
#pragma clang fp contract(off)
            // Sobel sums in the reference's accumulation order; zero taps add +-0
            float gx = 0.0f, gy = 0.0f;
            gx += win[0]; gx += 2.0f * win[1]; gx += win[2];
            gx += 0.0f * win[3]; gx += 0.0f * win[4]; gx += 0.0f * win[5];
            gx += -1.0f * win[6]; gx += -2.0f * win[7]; gx += -1.0f * win[8];
            gy += win[0]; gy += 0.0f * win[1]; gy += -1.0f * win[2];
            gy += 2.0f * win[3]; gy += 0.0f * win[4]; gy += -2.0f * win[5];
            gy += win[6]; gy += 0.0f * win[7]; gy += -1.0f * win[8];
            fk[4] = __builtin_sqrtf(gx * gx + gy * gy);
        }
        float md = 0.0f;
#pragma unroll
        for (int i = 0; i < 9; i++) md += __builtin_fabsf(win[i] - mean3) * (1.0f / 9.0f);
        fk[5] = md;

        const W2 wk = w2 + 35 * k;
#if PTK_DN_ABLATE & 256   // measurement build: one feature group of six in the first layer
        fk[0] += fk[2] + fk[3] + fk[4] + fk[5];
#endif
#pragma unroll
        for (int g = 0; g < ((PTK_DN_ABLATE & 256) ? 1 : 6); g++)
#pragma unroll
            for (int p = 0; p < 5; p++) a1[p] = pk_fma(fk[g], wk[5 * g + p], a1[p]);
        if (features_out) {
#pragma unroll
            for (int g = 0; g < 6; g++) plane_put(features_out + (size_t)(5 * g + k) * P, idb, fk[g]);
        }
    }
    if (features_out) plane_put(features_out + (size_t)35 * P, idb, inv_spp);

    // The "MAD" of the five channels, unrolled side by side: each is one long
    // chain of dependent compares and selects, and five independent chains in one
    // block let the scheduler interleave them.
    {
        float mad[5];
        float win[5][9];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float* tk = t + k * FPLANE + ly * FW + lx;
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) win[k][j * 3 + i] = tk[(FH - 1 + j) * FW + FH - 1 + i];
        }
#if PTK_DN_ABLATE & 1
#pragma unroll
        for (int k = 0; k < 5; k++) mad[k] = win[k][4] + win[k][0] * win[k][8];   // measurement build: no "median getter"
#elif PTK_DN_FLAT_WINDOWS
        // Nine equal finite values give +0 (every comparison of the first pass fails or stores
        // the value itself, so every |v - m0| is +0; checked on the host for every float), and
        // on most surfaces the first hit's normal and albedo are the same over the whole
        // footprint of a wave.  One branch for the wave, with both sides in one block each.
        bool flat = true;
#pragma unroll
        for (int k = 0; k < 2; k++) {
#pragma unroll
            for (int i = 0; i < 9; i++)
                if (i != 4) flat = flat && win[k][i] == win[k][4];
            flat = flat && (win[k][4] - win[k][4]) == 0.0f;
        }
        if (__builtin_amdgcn_ballot_w64(!flat) == 0) {
            mad[0] = 0.0f;
            mad[1] = 0.0f;
#pragma unroll
            for (int k = 2; k < 5; k++) mad[k] = mad_procedure(win[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) mad[k] = mad_procedure(win[k]);
        }
#else
#pragma unroll
        for (int k = 0; k < 5; k++) mad[k] = mad_procedure(win[k]);
#endif
#pragma unroll
        for (int k = 0; k < 5; k++) {
#pragma unroll
            for (int p = 0; p < 5; p++) a1[p] = pk_fma(mad[k], w2[35 * k + 30 + p], a1[p]);
        }
        // one branch after all five: a conditional store per channel would split the
        // chains above into five blocks, one after the other
        if (features_out) {
#pragma unroll
            for (int k = 0; k < 5; k++) plane_put(features_out + (size_t)(30 + k) * P, idb, mad[k]);
        }
    }

    // layers two and three: the sums of a node pair run over the inputs in index order,
    // starting from zero, as a one-node dot product would
    float l2[10], l3[10];
#pragma unroll
    for (int p = 0; p < 5; p++) {
        l2[2 * p] = sigmoid(a1[p].x);
        l2[2 * p + 1] = sigmoid(a1[p].y);
    }
    f2 s3[5], s4[4];
#pragma unroll
    for (int p = 0; p < 5; p++) s3[p] = f2{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < ((PTK_DN_ABLATE & 128) ? 1 : 10); i++)   // (measurement build: one input of ten)
#pragma unroll
        for (int p = 0; p < 5; p++) s3[p] = pk_fma(l2[i], w2[180 + 5 * i + p], s3[p]);
#pragma unroll
    for (int p = 0; p < 5; p++) {
        l3[2 * p] = sigmoid(s3[p].x);
        l3[2 * p + 1] = sigmoid(s3[p].y);
    }
#pragma unroll
    for (int p = 0; p < 4; p++) s4[p] = f2{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < ((PTK_DN_ABLATE & 128) ? 1 : 10); i++)
#pragma unroll
        for (int p = 0; p < 4; p++) s4[p] = pk_fma(l3[i], w2[230 + 4 * i + p], s4[p]);
#pragma unroll
    for (int p = 0; p < 3; p++) {
        var[2 * p] = softplus(s4[p].x);
        var[2 * p + 1] = softplus(s4[p].y);
    }
    var[6] = softplus(s4[3].x);
    if (layers_out) {   // layerTwoValues / layerThreeValues, kept for training
#pragma unroll
        for (int nd = 0; nd < 10; nd++) {
            plane_put(layers_out + (size_t)nd * P, idb, l2[nd]);
            plane_put(layers_out + (size_t)(10 + nd) * P, idb, l3[nd]);
        }
    }
}

// Loads the (32+6)x(8+6) tile of primaries (and their /49 images) around
// the block's outputs.  All of a thread's global loads are issued before the
// first quotient is formed, so the tile costs one memory latency, not three.
template <bool WITH49>
DEV void load_primary_tile(const float* __restrict__ planes, size_t P, int W, int H, int x0, int y0,
                           float n, float* t)
{
    constexpr int EPT = (FPLANE + 255) / 256;
    const Recip three_n = make_recip(3.0f * n), one_n = make_recip(n);
    const Recip r49 = make_recip(49.0f);
    float raw[EPT][13];
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int e = threadIdx.x + 256 * i;
        if (e < FPLANE) {
            int ty = e / FW, tx = e - ty * FW;
            int gy = clampi(y0 + ty - FH, H), gx = clampi(x0 + tx - FH, W);
            const unsigned gb = ((unsigned)gy * (unsigned)W + (unsigned)gx) * 4u;
#pragma unroll
            for (int k = 0; k < 13; k++) raw[i][k] = plane_at(planes + (size_t)(PTK_PL_NORMAL + k) * P, gb);
        }
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int e = threadIdx.x + 256 * i;
        if (e < FPLANE) {
            float q[5];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // three separate quotients summed in order (DenoiserNN.cpp:16-36), bit-exact
                float v = 0.0f;
                v += div_by(raw[i][3 * k], three_n);
                v += div_by(raw[i][3 * k + 1], three_n);
                v += div_by(raw[i][3 * k + 2], three_n);
                q[k] = v;
            }
            q[4] = div_by(raw[i][12], one_n);
#pragma unroll
            for (int k = 0; k < 5; k++) {
                t[k * FPLANE + e] = q[k];
                if (WITH49) t[(5 + k) * FPLANE + e] = div_by(q[k], r49);
            }
        }
    }
}

// Tile of this workgroup.  Workgroups are dealt to the eight XCDs round-robin in
// launch order and each XCD has its own L2, so with the natural numbering the
// eight neighbours of a tile sit on eight different L2s and every halo element
// is fetched across the fabric once per tile.  Here workgroup b takes tile
// (b % 8) * ceil(tiles / 8) + b / 8: each XCD works through one contiguous run
// of tile rows, and the halo re-reads hit its own L2.  The grid is padded to a
// multiple of eight; padding workgroups own no tile.
#ifndef PTK_DN_XCD_REMAP
#define PTK_DN_XCD_REMAP 1
#endif
DEV int tiles_x(const DenoiseArgs& a) { return (a.width + TW - 1) / TW; }
DEV int tiles_y(const DenoiseArgs& a) { return (a.row_end - a.row_begin + TH - 1) / TH; }
DEV bool tile_origin(const DenoiseArgs& a, int& x0, int& y0)
{
    const int nbx = tiles_x(a), nb = nbx * tiles_y(a);
#if PTK_DN_XCD_REMAP
    const int per = (nb + 7) >> 3;
    const int t = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
#else
    const int t = blockIdx.x;
#endif
    if (t >= nb) return false;
    x0 = (t % nbx) * TW;
    y0 = a.row_begin + (t / nbx) * TH;
    return true;
}

// KEEP: the 36 features and / or the two hidden layers are written out as well (a caller's
// request, or the training step); the inference launch carries no such stores.
template <bool KEEP>
__global__ __launch_bounds__(256, PTK_DN_WAVES) void features_mlp_kernel(DenoiseArgs a)
{
    __shared__ float tile[PTILE_FLOATS + PTK_DN_LDS_PAD];
    const int W = a.width, H = a.height;
    const size_t P = (size_t)W * H;
    const float n = (float)a.sample_count;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    int x0, y0;
    if (!tile_origin(a, x0, y0)) return;   // uniform over the workgroup, before any barrier
    const int x = x0 + lx, y = y0 + ly;
    const bool inside = x < W && y < a.row_end;
    const size_t idx = inside ? (size_t)y * W + x : 0;
    const unsigned idb = inside ? ((unsigned)y * (unsigned)W + (unsigned)x) * 4u : 0u;
    // issued before the tile load so their latency overlaps it
    float sd5[5];
#pragma unroll
    for (int k = 0; k < 5; k++) sd5[k] = plane_at(a.planes + (size_t)(PTK_PL_SD + 1 + k) * P, idb);

    load_primary_tile<true>(a.planes, P, W, H, x0, y0, n, tile);
    __syncthreads();
    if (!inside) return;

    float var[7];
    const auto* w = (const __attribute__((address_space(4))) float*)a.weights;
    features_and_mlp<false>(tile, nullptr, ly, lx, sd5, 1.0f / n, w, KEEP ? a.features : nullptr,
                            KEEP ? a.layers : nullptr, P, idb, var);
#pragma unroll
    for (int k = 0; k < 7; k++) plane_put(a.variances + (size_t)k * P, idb, var[k]);
}

// Per-column statistics of the tile for the 7x7 window of the tolerance-held form: for each
// of the tile's 38 columns and each of its 8 output rows y, the mean c of the column's seven
// values in rows y-3 .. y+3 and the sum S of their squared deviations from c, for the five
// channels.  Element e = row * CW + column of a statistics plane belongs to tile element e
// of the window's top row (CW == FW), so consecutive threads read and write consecutive words.
// Layout: plane 2k holds c of channel k, plane 2k + 1 holds S.
DEV void column_stats(const float* t, float* cs)
{
    for (int e = threadIdx.x; e < CPLANE; e += 256) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float* p = t + k * FPLANE + e;
            float q[7];
#pragma unroll
            for (int j = 0; j < 7; j++) q[j] = p[j * FW];
            const float c = (((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + q[6])) * (1.0f / 7.0f);
            float s2 = 0.0f;
#pragma unroll
            for (int j = 0; j < 7; j++) {
                const float d = q[j] - c;
                s2 = __builtin_fmaf(d, d, s2);
            }
            cs[(2 * k) * CPLANE + e] = c;
            cs[(2 * k + 1) * CPLANE + e] = s2;
        }
    }
}

// The tolerance-held form of the pass (ptk_denoise variant 2): primaries and the "median
// getter" as above, bit for bit; 7x7 mean and deviation from shared column statistics.
template <bool KEEP>
__global__ __launch_bounds__(256, PTK_DN_WAVES) void features_mlp_fast_kernel(DenoiseArgs a)
{
    __shared__ float tile[5 * FPLANE];
    __shared__ float cstat[10 * CPLANE];
    const int W = a.width, H = a.height;
    const size_t P = (size_t)W * H;
    const float n = (float)a.sample_count;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    int x0, y0;
    if (!tile_origin(a, x0, y0)) return;   // uniform over the workgroup, before any barrier
    const int x = x0 + lx, y = y0 + ly;
    const bool inside = x < W && y < a.row_end;
    const unsigned idb = inside ? ((unsigned)y * (unsigned)W + (unsigned)x) * 4u : 0u;
    float sd5[5];
#pragma unroll
    for (int k = 0; k < 5; k++) sd5[k] = plane_at(a.planes + (size_t)(PTK_PL_SD + 1 + k) * P, idb);

    load_primary_tile<false>(a.planes, P, W, H, x0, y0, n, tile);
    __syncthreads();
    column_stats(tile, cstat);
    __syncthreads();
    if (!inside) return;

    float var[7];
    const auto* w = (const __attribute__((address_space(4))) float*)a.weights;
    features_and_mlp<true>(tile, cstat, ly, lx, sd5, 1.0f / n, w, KEEP ? a.features : nullptr,
                           KEEP ? a.layers : nullptr, P, idb, var);
#pragma unroll
    for (int k = 0; k < 7; k++) plane_put(a.variances + (size_t)k * P, idb, var[k]);
}

// ---- filter ---------------------------------------------------------------

constexpr int FV = 17;  // per tile element: col, normal, alb1, alb2, wpos (/n), dl/n, sd0

DEV void load_filter_tile(const float* __restrict__ planes, size_t P, int W, int H, int x0, int y0,
                          int N, float nf, float* t)
{
    const int SW = TW + 2 * N, SH = TH + 2 * N;
    const int SS = SW * SH;
    const Recip dv = make_recip(nf);
    // two elements' loads in flight per trip
    for (int e0 = threadIdx.x; e0 < SS; e0 += 512) {
        float raw[2][FV];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int e = e0 + 256 * i;
            if (e < SS) {
                int ty = e / SW, tx = e - ty * SW;
                int gy = clampi(y0 + ty - N, H), gx = clampi(x0 + tx - N, W);
                const unsigned gb = ((unsigned)gy * (unsigned)W + (unsigned)gx) * 4u;
#pragma unroll
                for (int k = 0; k < 16; k++) raw[i][k] = plane_at(planes + (size_t)k * P, gb);
                raw[i][16] = plane_at(planes + (size_t)PTK_PL_SD * P, gb);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int e = e0 + 256 * i;
            if (e < SS) {
#pragma unroll
                for (int k = 0; k < 16; k++) t[k * SS + e] = div_by(raw[i][k], dv);
                t[16 * SS + e] = raw[i][16];
            }
        }
    }
}

struct FilterCoef {
    float inv_idx, col_scale, sd0, inv_nrm, inv_a1, inv_a2, inv_wp, inv_dl;
};
DEV FilterCoef make_coef(const float* sd, const float* var)
{
    FilterCoef c;
    c.sd0 = sd[0];
    c.inv_idx = fast_rcp(2.0f * var[0]);
    c.col_scale = 2.0f * var[1];
    c.inv_nrm = fast_rcp(sd[1] * 2.0f * var[2] + 0.000001f);
    c.inv_a1 = fast_rcp(sd[2] * 2.0f * var[3] + 0.000001f);
    c.inv_a2 = fast_rcp(sd[3] * 2.0f * var[4] + 0.000001f);
    c.inv_wp = fast_rcp(2.0f * sd[4] * var[5] + 0.000001f);
    c.inv_dl = fast_rcp(sd[5] * 2.0f * var[6] + 0.000001f);
    return c;
}

// One filter tap (Denoiser.cpp:301-348): the product of the seven Gaussians is one exp of
// the summed exponents.  `nb` supplies the neighbour: diff(k) = its value k minus the centre
// pixel's, sd0_sum(s) = s + its stdDev[0], accumulate(w, wsum) adds its colour times w.
// Every fused multiply-add is written out, so that each kernel built on this function forms
// the same roundings whatever its surrounding code.
template <class Neighbour>
DEV void filter_tap_core(const Neighbour& nb, float dist_idx, const FilterCoef& c,
                         float (&wsum)[3], float& wc)
{
#pragma clang fp contract(off)
    float d2[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        float dx = nb.diff(3 * k), dy = nb.diff(3 * k + 1), dz = nb.diff(3 * k + 2);
        d2[k] = __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
    }
    float ddl = nb.diff(15);
    float arg = dist_idx * c.inv_idx;
    arg = __builtin_fmaf(d2[0], fast_rcp(__builtin_fmaf(nb.sd0_sum(c.sd0), c.col_scale, 0.000001f)), arg);
    arg = __builtin_fmaf(d2[1], c.inv_nrm, arg);
    arg = __builtin_fmaf(d2[2], c.inv_a1, arg);
    arg = __builtin_fmaf(d2[3], c.inv_a2, arg);
    arg = __builtin_fmaf(d2[4], c.inv_wp, arg);
    arg = __builtin_fmaf(ddl * ddl, c.inv_dl, arg);
    float w = fast_exp(-arg);
    nb.accumulate(w, wsum);
    wc += w;
}

// The neighbour whose 16 values and stdDev[0] are in registers.
struct NeighbourValues {
    const float (&q)[16];
    float sd0q;
    const float (&pv)[16];
    DEV float diff(int k) const { return q[k] - pv[k]; }
    DEV float sd0_sum(float s) const { return s + sd0q; }
    DEV void accumulate(float w, float (&wsum)[3]) const
    {
        wsum[0] = __builtin_fmaf(q[0], w, wsum[0]);
        wsum[1] = __builtin_fmaf(q[1], w, wsum[1]);
        wsum[2] = __builtin_fmaf(q[2], w, wsum[2]);
    }
};

// The tap whose values sit at word offset e of an LDS tile with planes SS words apart.
DEV void filter_tap(const float* t, int SS, int e, float dist_idx, const float (&pv)[16],
                    const FilterCoef& c, float (&wsum)[3], float& wc)
{
    float q[16];
#pragma unroll
    for (int k = 0; k < 16; k++) q[k] = t[k * SS + e];
    filter_tap_core(NeighbourValues{q, t[16 * SS + e], pv}, dist_idx, c, wsum, wc);
}

// All taps of the pixel at tile coordinates (ly, lx).  NFIX > 0 unrolls the
// (2N+1)^2 window at compile time.
template <int NFIX>
DEV void filter_pixel(const float* t, int SW, int SS, int ly, int lx, int N, const float* sd,
                      const float* var, float (&wsum)[3], float& wc)
{
    const int ce = (ly + N) * SW + lx + N;
    float pv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) pv[k] = t[k * SS + ce];
    const FilterCoef c = make_coef(sd, var);
    wsum[0] = wsum[1] = wsum[2] = 0.0f;
    wc = 0.0f;
    if (NFIX > 0) {
#pragma unroll
        for (int j = -NFIX; j <= NFIX; j++)
#pragma unroll
            for (int i = -NFIX; i <= NFIX; i++)
                filter_tap(t, SS, ce + j * SW + i, (float)(j * j + i * i), pv, c, wsum, wc);
    } else {
        for (int j = -N; j <= N; j++)
            for (int i = -N; i <= N; i++)
                filter_tap(t, SS, ce + j * SW + i, (float)(j * j + i * i), pv, c, wsum, wc);
    }
}

template <int NFIX>
__global__ __launch_bounds__(256, PTK_DN_WAVES_FILTER) void filter_kernel(DenoiseArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float ftile[];
    const int W = a.width, H = a.height, N = NFIX > 0 ? NFIX : a.n;
    const size_t P = (size_t)W * H;
    const int SW = TW + 2 * N, SS = SW * (TH + 2 * N);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    int x0, y0;
    if (!tile_origin(a, x0, y0)) return;   // uniform over the workgroup, before any barrier
    const int x = x0 + lx, y = y0 + ly;
    const bool inside = x < W && y < a.row_end;
    const size_t idx = inside ? (size_t)y * W + x : 0;
    float sd[6], var[7];
#pragma unroll
    for (int k = 0; k < 6; k++) sd[k] = plane_at(a.planes + (size_t)(PTK_PL_SD + k) * P, (unsigned)idx * 4u);
#pragma unroll
    for (int k = 0; k < 7; k++) var[k] = plane_at(a.variances + (size_t)k * P, (unsigned)idx * 4u);

    load_filter_tile(a.planes, P, W, H, x0, y0, N, (float)a.sample_count, ftile);
    __syncthreads();
    if (!inside) return;

    float wsum[3], wc;
    filter_pixel<NFIX>(ftile, SW, SS, ly, lx, N, sd, var, wsum, wc);
    const float inv = 1.0f / wc;
    a.denoised[idx] = wsum[0] * inv;
    a.denoised[P + idx] = wsum[1] * inv;
    a.denoised[2 * P + idx] = wsum[2] * inv;
    a.wc_sum[idx] = wc;
}

// ---- filter, software-pipelined over a strip of tiles.
//
// A workgroup of the one-tile kernel above loads, waits, computes and stores in
// turn, and three quarters of its waves' lifetime is spent parked on the tile
// load.  Here a workgroup walks a sequence of tiles and holds the NEXT tile's
// raw values in registers (global loads issued before the current tile's
// taps are evaluated), so the loads of tile i+1 fly under the arithmetic of
// tile i.  Workgroup b of XCD b % 8 takes tiles first + b / 8, + groups, ...
// of that XCD's contiguous run of tiles: at any moment the resident workgroups
// of an XCD cover a compact band of the image and share halos through its L2.
#ifndef PTK_DN_WAVES_STRIP
#define PTK_DN_WAVES_STRIP 3   // N = 1; wider windows stage more registers and take 2
#endif
struct StripSchedule {
    int t, t_end, step, nbx;
};
DEV StripSchedule strip_schedule(const DenoiseArgs& a)
{
    StripSchedule s;
    s.nbx = tiles_x(a);
    const int nb = s.nbx * tiles_y(a), per = (nb + 7) >> 3;
    const int xcd = (int)(blockIdx.x & 7u);
    s.step = (int)(gridDim.x >> 3);
    s.t = xcd * per + (int)(blockIdx.x >> 3);
    s.t_end = (xcd + 1) * per < nb ? (xcd + 1) * per : nb;
    return s;
}

template <int N>
__global__ __launch_bounds__(256, N == 1 ? PTK_DN_WAVES_STRIP : 2)
void filter_strip_kernel(DenoiseArgs a)
{
    constexpr int SW = TW + 2 * N, SH = TH + 2 * N, SS = SW * SH;
    constexpr int EPT = (SS + 255) / 256;   // tile elements staged per thread
    extern __shared__ __attribute__((aligned(16))) float ftile[];
    const int W = a.width, H = a.height;
    const size_t P = (size_t)W * H;
    const int tid = threadIdx.x, lx = tid & 31, ly = tid >> 5;
    const Recip dv = make_recip((float)a.sample_count);
    StripSchedule sc = strip_schedule(a);

    float stage[EPT][FV];   // raw values of the next tile's elements
    float sdv_next[13];     // the next tile's centre-pixel sd[6] and variances[7]
    size_t idx_next = 0;
    bool inside_next = false;
    auto issue = [&](int t) {
#if PTK_DN_ABLATE & 8   // measurement build: every tile reads the first tile's (cached) values
        const int x0 = 0, y0 = a.row_begin;
#else
        const int x0 = (t % sc.nbx) * TW, y0 = a.row_begin + (t / sc.nbx) * TH;
#endif
        const int x = x0 + lx, y = y0 + ly;
        inside_next = x < W && y < a.row_end;
        idx_next = inside_next ? (size_t)y * W + x : 0;
#pragma unroll
        for (int k = 0; k < 6; k++) sdv_next[k] = plane_at(a.planes + (size_t)(PTK_PL_SD + k) * P, (unsigned)idx_next * 4u);
#pragma unroll
        for (int k = 0; k < 7; k++) sdv_next[6 + k] = plane_at(a.variances + (size_t)k * P, (unsigned)idx_next * 4u);
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            const int e = tid + 256 * i;
            if (e < SS) {
                const int ty = e / SW, tx = e - ty * SW;
                const int gy = clampi(y0 + ty - N, H), gx = clampi(x0 + tx - N, W);
                const size_t gi = (size_t)gy * W + gx;
#pragma unroll
                for (int k = 0; k < 16; k++) stage[i][k] = a.planes[(size_t)k * P + gi];
                stage[i][16] = a.planes[(size_t)PTK_PL_SD * P + gi];
            }
        }
    };

    if (sc.t < sc.t_end) issue(sc.t);
    for (int t = sc.t; t < sc.t_end; t += sc.step) {
        // the staged tile becomes the current one
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            const int e = tid + 256 * i;
            if (e < SS) {
#pragma unroll
                for (int k = 0; k < 16; k++) ftile[k * SS + e] = div_by(stage[i][k], dv);
                ftile[16 * SS + e] = stage[i][16];
            }
        }
        float sd[6], var[7];
#pragma unroll
        for (int k = 0; k < 6; k++) sd[k] = sdv_next[k];
#pragma unroll
        for (int k = 0; k < 7; k++) var[k] = sdv_next[6 + k];
        const size_t idx = idx_next;
        const bool inside = inside_next;
        __syncthreads();

        if (t + sc.step < sc.t_end) issue(t + sc.step);   // in flight during the taps below

        if (inside) {
            float wsum[3], wc;
#if PTK_DN_ABLATE & 4   // measurement build: no taps
            wsum[0] = ftile[(ly + N) * SW + lx + N] + sd[0] + sd[5] + var[0] + var[6];
            wsum[1] = ftile[7 * SS + (ly + N) * SW + lx + N] + sd[1] + sd[2] + sd[3] + var[1] + var[2];
            wsum[2] = ftile[16 * SS + (ly + N) * SW + lx + N] + sd[4] + var[3] + var[4] + var[5];
            wc = 1.0f;
#else
            filter_pixel<N>(ftile, SW, SS, ly, lx, N, sd, var, wsum, wc);
#endif
            const float inv = 1.0f / wc;
            a.denoised[idx] = wsum[0] * inv;
            a.denoised[P + idx] = wsum[1] * inv;
            a.denoised[2 * P + idx] = wsum[2] * inv;
            a.wc_sum[idx] = wc;
        }
        __syncthreads();   // the tile is overwritten at the top of the next iteration
    }
}

// ---- filter, N = 1, on a window of rows held in registers.
//
// A wave owns a strip of 62 pixel columns and walks down a run of rows.  Lane l holds, for
// column x0 - 1 + l (clamped to the image), the 16 averaged values and stdDev[0] of three
// consecutive rows; the left and right neighbours of a value are the same register one lane
// down or up, read through a DPP wave shift, so the 3x3 window needs no LDS, no barrier and no
// second copy of the data: lanes 1..62 produce outputs, lanes 0 and 63 only carry the halo
// columns.  Each step converts the row that has just arrived (sums / sample count, the exact
// quotient), issues the loads of the row after it, and evaluates the nine taps of the row in
// the middle while those loads fly.  Waves never wait for one another, so a compute unit
// keeps (rows in flight) x (waves) x 7.4 KB of loads outstanding all the time.  The taps and
// their order are those of filter_pixel<1>, operand for operand.
#ifndef PTK_DN_WAVES_ROWS
#define PTK_DN_WAVES_ROWS 3
#endif
#ifndef PTK_DN_ROWS_PER_WAVE
#define PTK_DN_ROWS_PER_WAVE 0   // measurement builds: fixed run length instead of the host's choice
#endif
constexpr int ROWS_COLS = 62;     // output columns per wave

struct RowsGrid {
    int col_groups, segments;
};

// Arithmetic whose first operand is taken from the lane below (DX = -1: lane l reads lane
// l - 1) or above (DX = +1) through the DPP wave shift of the instruction itself, so that a
// neighbour's value costs no instruction of its own.  Lanes with no such neighbour (lane 0,
// lane 63) read zero; their results are never used.  The hardware needs two wait states
// between a vector write of a register and a DPP read of it, and the compiler cannot see
// inside these statements: the only registers read this way are a row's values, and
// rows_settle() below stands between their conversion and the taps.
#define PTK_DPP_BELOW " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define PTK_DPP_ABOVE " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
DEV void rows_settle()
{
    __builtin_amdgcn_sched_barrier(0);   // nothing is moved across: conversions above, taps below
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}
template <int DX>
DEV float shifted_sub(float q, float p)     // q(lane + DX) - p
{
    float d;
    if (DX < 0) asm("v_sub_f32_dpp %0, %1, %2" PTK_DPP_BELOW : "=v"(d) : "v"(q), "v"(p));
    else asm("v_sub_f32_dpp %0, %1, %2" PTK_DPP_ABOVE : "=v"(d) : "v"(q), "v"(p));
    return d;
}
template <int DX>
DEV float shifted_add(float q, float p)     // q(lane + DX) + p
{
    float d;
    if (DX < 0) asm("v_add_f32_dpp %0, %1, %2" PTK_DPP_BELOW : "=v"(d) : "v"(q), "v"(p));
    else asm("v_add_f32_dpp %0, %1, %2" PTK_DPP_ABOVE : "=v"(d) : "v"(q), "v"(p));
    return d;
}
// acc[i] = fma(q[i](lane + DX), w, acc[i]), i = 0..2.  w comes straight from v_exp_f32, and a
// transcendental result needs a wait state before an ordinary vector instruction reads it
// (inserted by the compiler for its own instructions only).
template <int DX>
DEV void shifted_fmac3(float (&acc)[3], const float (&q)[16], float w)
{
    if (DX < 0)
        asm("s_nop 1\n\tv_fmac_f32_dpp %0, %3, %6" PTK_DPP_BELOW "\n\tv_fmac_f32_dpp %1, %4, %6" PTK_DPP_BELOW
            "\n\tv_fmac_f32_dpp %2, %5, %6" PTK_DPP_BELOW
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]) : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(w));
    else
        asm("s_nop 1\n\tv_fmac_f32_dpp %0, %3, %6" PTK_DPP_ABOVE "\n\tv_fmac_f32_dpp %1, %4, %6" PTK_DPP_ABOVE
            "\n\tv_fmac_f32_dpp %2, %5, %6" PTK_DPP_ABOVE
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]) : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(w));
}

struct RowRegs {
    float q[16];
    float sd0;
};
struct RawRow {
    float v[FV];
};
struct CentreRegs {   // stdDev[1..5] and the seven bandwidths of the pixel being filtered
    float sd[5], var[7];
};

// Loads of one image row: every address is a wave-uniform row base (scalar registers) plus
// the lane's byte offset in the row, which never changes, so the 29 streams of a wave cost one
// address register between them.  The row base is pinned to scalar registers: otherwise the
// compiler folds the lane offset into it and carries one 64-bit vector address per stream
// through the row loop.
DEV GlobalByte* rows_base(const void* plane0, int W, int H, int y)
{
    // the clamp compiles to a vector min/max; back to a scalar register before the multiply
    const int yc = __builtin_amdgcn_readfirstlane(clampi(y, H));
    const unsigned long long v =
        (unsigned long long)plane0 + (unsigned long long)yc * (unsigned long long)W * sizeof(float);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (GlobalByte*)(((unsigned long long)hi << 32) | lo);
}
DEV float rows_load(GlobalByte* row, size_t plane_bytes, int k, unsigned lane_bytes)
{
    GlobalByte* plane_row = row + (size_t)k * plane_bytes;
    asm("" : "+s"(plane_row));   // stays in scalar registers: the lane offset is not folded in
    return *(const __attribute__((address_space(1))) float*)(plane_row + lane_bytes);
}
DEV void rows_store(GlobalByte* row, unsigned lane_bytes, float v)
{
    asm("" : "+s"(row));
    *(__attribute__((address_space(1))) float*)(row + lane_bytes) = v;
}
DEV void rows_issue(const DenoiseArgs& a, size_t PB, unsigned lane_bytes, int y, RawRow& raw)
{
    GlobalByte* row = rows_base(a.planes, a.width, a.height, y);
#pragma unroll
    for (int k = 0; k < 16; k++) raw.v[k] = rows_load(row, PB, k, lane_bytes);
    raw.v[16] = rows_load(row, PB, PTK_PL_SD, lane_bytes);
}
DEV void rows_issue_centre(const DenoiseArgs& a, size_t PB, unsigned lane_bytes, int y, CentreRegs& c)
{
    GlobalByte* row = rows_base(a.planes, a.width, a.height, y);
    GlobalByte* vrow = rows_base(a.variances, a.width, a.height, y);
#pragma unroll
    for (int k = 0; k < 5; k++) c.sd[k] = rows_load(row, PB, PTK_PL_SD + 1 + k, lane_bytes);
#pragma unroll
    for (int k = 0; k < 7; k++) c.var[k] = rows_load(vrow, PB, k, lane_bytes);
}
DEV void rows_convert(const RawRow& raw, const Recip& dv, RowRegs& r)
{
#pragma unroll
    for (int k = 0; k < 16; k++) r.q[k] = div_by(raw.v[k], dv);
    r.sd0 = raw.v[16];
}

// The neighbour DX lanes away in row r.
template <int DX>
struct NeighbourLane {
    const RowRegs& r;
    const float (&pv)[16];
    DEV float diff(int k) const { return DX == 0 ? r.q[k] - pv[k] : shifted_sub<DX>(r.q[k], pv[k]); }
    DEV float sd0_sum(float s) const { return DX == 0 ? s + r.sd0 : shifted_add<DX>(r.sd0, s); }
    DEV void accumulate(float w, float (&wsum)[3]) const
    {
        if (DX == 0) {
            wsum[0] = __builtin_fmaf(r.q[0], w, wsum[0]);
            wsum[1] = __builtin_fmaf(r.q[1], w, wsum[1]);
            wsum[2] = __builtin_fmaf(r.q[2], w, wsum[2]);
        } else {
            shifted_fmac3<DX>(wsum, r.q, w);
        }
    }
};
template <int DX>
DEV void rows_tap(const RowRegs& r, float dist_idx, const float (&pv)[16], const FilterCoef& c,
                  float (&wsum)[3], float& wc)
{
    filter_tap_core(NeighbourLane<DX>{r, pv}, dist_idx, c, wsum, wc);
}

// One row of outputs: `up`, `mid`, `down` are rows y-1, y, y+1.
DEV void rows_output(const DenoiseArgs& a, size_t PB, const RowRegs& up, const RowRegs& mid,
                     const RowRegs& down, const CentreRegs& cr, bool store, int y,
                     unsigned lane_bytes)
{
    float sd[6], wsum[3] = {0.0f, 0.0f, 0.0f}, wc = 0.0f;
    sd[0] = mid.sd0;
#pragma unroll
    for (int k = 0; k < 5; k++) sd[1 + k] = cr.sd[k];
    const FilterCoef c = make_coef(sd, cr.var);
#if PTK_DN_ABLATE & 4   // measurement build: no taps
    wsum[0] = up.q[0] + up.q[5] + up.q[9] + up.q[15] + c.inv_idx + c.inv_nrm + c.inv_a1 + up.q[1] + up.q[6] + up.q[10] + up.sd0;
    wsum[1] = mid.q[1] + mid.q[4] + mid.q[8] + mid.q[14] + c.inv_a2 + c.inv_wp + c.inv_dl + up.q[2] + up.q[7] + up.q[11];
    wsum[2] = down.q[2] + down.q[3] + down.q[7] + down.q[13] + c.col_scale + c.sd0 + up.q[3] + up.q[8] + up.q[12] + up.q[4] + up.q[13] + up.q[14];
    wc = 1.0f;
#elif PTK_DN_ABLATE & 16   // measurement build: three taps of nine
    rows_tap<-1>(up, 2.0f, mid.q, c, wsum, wc);
    rows_tap<0>(mid, 0.0f, mid.q, c, wsum, wc);
    rows_tap<1>(down, 2.0f, mid.q, c, wsum, wc);
#else
    rows_tap<-1>(up, 2.0f, mid.q, c, wsum, wc);
    rows_tap<0>(up, 1.0f, mid.q, c, wsum, wc);
    rows_tap<1>(up, 2.0f, mid.q, c, wsum, wc);
    rows_tap<-1>(mid, 1.0f, mid.q, c, wsum, wc);
    rows_tap<0>(mid, 0.0f, mid.q, c, wsum, wc);
    rows_tap<1>(mid, 1.0f, mid.q, c, wsum, wc);
    rows_tap<-1>(down, 2.0f, mid.q, c, wsum, wc);
    rows_tap<0>(down, 1.0f, mid.q, c, wsum, wc);
    rows_tap<1>(down, 2.0f, mid.q, c, wsum, wc);
#endif
    if (store) {
        const float inv = 1.0f / wc;
        GlobalByte* out = rows_base(a.denoised, a.width, a.height, y);
        GlobalByte* wout = rows_base(a.wc_sum, a.width, a.height, y);
        rows_store(out, lane_bytes, wsum[0] * inv);
        rows_store(out + PB, lane_bytes, wsum[1] * inv);
        rows_store(out + 2 * PB, lane_bytes, wsum[2] * inv);
        rows_store(wout, lane_bytes, wc);
    }
}

// `count` rows starting at row y0 and moving by DIR (+1 down the image, -1 up).  r0, r1, r2
// rotate through the roles (row behind, row being filtered, row ahead).
template <int DIR>
DEV void rows_walk(const DenoiseArgs& a, size_t PB, unsigned gx, bool writer, const Recip& dv, int y0,
                   int count)
{
    RowRegs r0, r1, r2;
    RawRow raw;
    CentreRegs cnext;
    {
        RawRow first;
        rows_issue(a, PB, gx, y0 - DIR, first);
        rows_issue(a, PB, gx, y0, raw);
        rows_convert(first, dv, r0);
        rows_convert(raw, dv, r1);
        rows_settle();
    }
    rows_issue(a, PB, gx, y0 + DIR, raw);
    rows_issue_centre(a, PB, gx, y0, cnext);

    // step: rows (behind, mid) are converted, `raw` holds the row ahead and cnext the centre
    // values of row y, both in flight; the taps always run from the row above to the row below
#define PTK_ROWS_STEP(behind, mid, ahead, y)                                               \
    {                                                                                      \
        rows_convert(raw, dv, ahead);                                                      \
        rows_settle();                                                                     \
        const CentreRegs cr = cnext;                                                       \
        rows_issue(a, PB, gx, (y) + 2 * DIR, raw);                                         \
        rows_issue_centre(a, PB, gx, (y) + DIR, cnext);                                    \
        if (DIR > 0) rows_output(a, PB, behind, mid, ahead, cr, writer, (y), gx);          \
        else rows_output(a, PB, ahead, mid, behind, cr, writer, (y), gx);                  \
    }
    for (int i = 0, y = y0; i < count; i += 3, y += 3 * DIR) {
        PTK_ROWS_STEP(r0, r1, r2, y)
        if (i + 1 >= count) break;
        PTK_ROWS_STEP(r1, r2, r0, y + DIR)
        if (i + 2 >= count) break;
        PTK_ROWS_STEP(r2, r0, r1, y + 2 * DIR)
    }
#undef PTK_ROWS_STEP
}

__global__ __launch_bounds__(256, PTK_DN_WAVES_ROWS) void filter_rows_kernel(DenoiseArgs a, RowsGrid g)
{
    const int W = a.width;
    const size_t PB = (size_t)W * a.height * sizeof(float);   // bytes per plane
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: rows and strips are wave-uniform
    // workgroup -> (row segment, group of four strips); each XCD takes a contiguous run
    const int nwg = g.col_groups * g.segments, per = (nwg + 7) >> 3;
    const int wg = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || wg >= nwg) return;
    const int seg = wg / g.col_groups, x0 = ((wg - seg * g.col_groups) * 4 + wave) * ROWS_COLS;
    if (x0 >= W) return;
    // the band's rows are dealt to the segments as evenly as integers allow
    const int rows = a.row_end - a.row_begin;
    const int y_begin = a.row_begin + (int)((long long)seg * rows / g.segments);
    const int y_end = a.row_begin + (int)((long long)(seg + 1) * rows / g.segments);
    const int xl = x0 - 1 + lane;
    const unsigned gx = (unsigned)clampi(xl, W) * (unsigned)sizeof(float);   // byte offset in a row
    const bool writer = lane >= 1 && lane <= ROWS_COLS && xl < W;
    const Recip dv = make_recip((float)a.sample_count);

    // Even segments walk down, odd ones up: a segment and its neighbour then start at
    // adjacent rows and meet at adjacent rows, so the halo row each reads from the other is
    // in the XCD's L2 when it is wanted instead of being fetched from HBM a second time.
    if (seg & 1)
        rows_walk<-1>(a, PB, gx, writer, dv, y_end - 1, y_end - y_begin);
    else
        rows_walk<1>(a, PB, gx, writer, dv, y_begin, y_end - y_begin);
}

size_t filter_lds_bytes(int N)
{
    size_t sw = TW + 2 * (size_t)N, sh = TH + 2 * (size_t)N;
    return sw * sh * FV * sizeof(float);
}

unsigned denoise_grid(const DenoiseArgs& a)
{
    const unsigned nb = (unsigned)((a.width + TW - 1) / TW) *
                        (unsigned)((a.row_end - a.row_begin + TH - 1) / TH);
    return (nb + 7u) / 8u * 8u;
}

template <class K>
hipError_t launch_dyn(K kernel, const DenoiseArgs& a, size_t lds, hipStream_t stream)
{
    dim3 grid(denoise_grid(a));
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

}  // namespace

// Compute units of the current device (0 on failure).
static int device_cus()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return 0;
        cus = prop.multiProcessorCount;
    }
    return cus;
}

// Grid of a strip kernel: `per_cu` workgroups for each of the device's compute
// units (a multiple of eight, one share per XCD), never more than one per tile.
unsigned strip_grid(const DenoiseArgs& a, int per_cu)
{
    const int cus = device_cus();
    if (cus == 0) return 0;
    const unsigned tiles = (unsigned)((a.width + TW - 1) / TW) *
                           (unsigned)((a.row_end - a.row_begin + TH - 1) / TH);
    unsigned groups = (unsigned)cus * (unsigned)per_cu;
    const unsigned per_xcd_tiles = (tiles + 7u) / 8u;
    unsigned per_xcd = (groups + 7u) / 8u;
    if (per_xcd > per_xcd_tiles) per_xcd = per_xcd_tiles;
    return per_xcd * 8u;
}

template <class K>
hipError_t launch_strip(K kernel, const DenoiseArgs& a, size_t lds, int per_cu, hipStream_t stream)
{
    const unsigned grid = strip_grid(a, per_cu);
    if (grid == 0) return hipErrorInvalidDevice;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, stream, a);
    return hipGetLastError();
}

#ifndef PTK_DN_STRIP
#define PTK_DN_STRIP 1
#endif
#ifndef PTK_DN_STRIP_PER_CU
#define PTK_DN_STRIP_PER_CU 3
#endif
hipError_t ptk_launch_features_mlp(const DenoiseArgs& a, hipStream_t stream, bool fast)
{
    dim3 grid(denoise_grid(a));
    const bool keep = a.features || a.layers;
    if (fast && keep)
        hipLaunchKernelGGL(features_mlp_fast_kernel<true>, grid, dim3(256), 0, stream, a);
    else if (fast)
        hipLaunchKernelGGL(features_mlp_fast_kernel<false>, grid, dim3(256), 0, stream, a);
    else if (keep)
        hipLaunchKernelGGL(features_mlp_kernel<true>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(features_mlp_kernel<false>, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

// Run length of the register-window kernel: as many waves as the device holds at once (or
// a whole multiple of that), so that every wave slot streams for the whole launch.
hipError_t launch_rows(const DenoiseArgs& a, hipStream_t stream)
{
    const int cus = device_cus();
    if (cus == 0) return hipErrorInvalidDevice;
    const int rows = a.row_end - a.row_begin;
    RowsGrid g;
    const int strips = (a.width + ROWS_COLS - 1) / ROWS_COLS;
    g.col_groups = (strips + 3) / 4;
    const int slots = cus * PTK_DN_WAVES_ROWS;          // workgroups resident at once
    int segs = slots / g.col_groups;                    // one resident round
#if PTK_DN_ROWS_PER_WAVE
    segs = (rows + PTK_DN_ROWS_PER_WAVE - 1) / PTK_DN_ROWS_PER_WAVE;
#endif
    if (segs > rows / 8) segs = rows / 8;               // runs of at least eight rows
    if (segs < 1) segs = 1;
    g.segments = segs;
    const unsigned nwg = (unsigned)(g.col_groups * g.segments);
    hipLaunchKernelGGL(filter_rows_kernel, dim3((nwg + 7u) / 8u * 8u), dim3(256), 0, stream, a, g);
    return hipGetLastError();
}

#ifndef PTK_DN_ROWS
#define PTK_DN_ROWS 1
#endif
hipError_t ptk_launch_filter(const DenoiseArgs& a, hipStream_t stream, bool lds_tiles)
{
    size_t lds = filter_lds_bytes(a.n);
#if PTK_DN_ROWS
    if (a.n == 1 && !lds_tiles) return launch_rows(a, stream);
#endif
#if PTK_DN_STRIP
    if (a.n == 1) return launch_strip(filter_strip_kernel<1>, a, lds, PTK_DN_STRIP_PER_CU, stream);
    if (a.n == 2) return launch_strip(filter_strip_kernel<2>, a, lds, 2, stream);
    if (a.n == 3) return launch_strip(filter_strip_kernel<3>, a, lds, 2, stream);
#endif
    if (a.n == 1) return launch_dyn(filter_kernel<1>, a, lds, stream);
    return launch_dyn(filter_kernel<0>, a, lds, stream);
}
