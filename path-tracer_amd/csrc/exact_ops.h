// exact_ops.h -- correctly rounded fp32 quotients of many numerators by one denominator,
// and the correctly rounded square root without the parts its callers cannot need.
//
// hipcc expands a correctly rounded a / d into: scale both operands (v_div_scale, the
// identity unless an exponent is extreme), r = rcp(d), one Newton step on r, q = a*r, two
// residual corrections of q, v_div_fmas (a plain fma when nothing was scaled) and
// v_div_fixup for zeros / infinities / NaNs: 11 VALU operations.  Quotients that share a
// denominator can share most of that.  exact_reciprocal runs the same sequence on the
// numerator 1, which gives y = RN(1 / d), the correctly rounded reciprocal (7 operations),
// and with that y ONE residual correction is enough for any numerator (3 operations):
//     q = RN(a*y);  t = a - d*q (exact in an fma);  q' = RN(q + t*y) = RN(a / d).
// Why (Markstein's theorem; d in [1, 2) without loss, ulp = that of a / d): d*y = 1 + e
// with |e| <= d * 2^-25, so a*y is within one ulp of a / d and q is one of the two floats
// around a / d, or -- only when a / d is at least 2^-23 ulp clear of the rounding midpoint
// -- the next one out.  Then q + t*y = a/d + (a/d - q) * e, i.e. a / d moved by less than
// d * 2^-26 ulp when q is a neighbour, while a / d, a ratio of 24-bit numbers, stays at
// least 2^-24 / d ulp away from every midpoint; d*d < 4, so the last rounding lands on
// RN(a / d).  scripts/check_exact_quotient.c compares the three operations with `/` on
// the host for every numerator significand of many denominators, and
// scripts/microbench/exact_ops_check.hip does the same on the device, reciprocal included.
//
// Valid while d is an ordinary number (exponent well inside the range: callers check,
// or their denominators are sample counts and small constants) and nothing underflows;
// zero, infinite and NaN numerators come out as `/` gives them only through
// exact_quotient_fixup (v_div_fixup restores the sign of a zero and the infinity).
#pragma once

namespace ptk_exact {

__device__ __forceinline__ float exact_reciprocal(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    // the compiler's two residual corrections, on q = 1 * r1
    e = __builtin_fmaf(-d, r1, 1.0f);
    const float y = __builtin_fmaf(e, r1, r1);
    e = __builtin_fmaf(-d, y, 1.0f);
    return __builtin_fmaf(e, r1, y);
}

// a / d given y = exact_reciprocal(d); finite non-zero a (see the header comment)
__device__ __forceinline__ float exact_quotient(float a, float y, float d)
{
    const float q = a * y;
    const float t = __builtin_fmaf(-d, q, a);
    return __builtin_fmaf(t, y, q);
}

// the same for any a
__device__ __forceinline__ float exact_quotient_fixup(float a, float y, float d)
{
    return __builtin_amdgcn_div_fixupf(exact_quotient(a, y, d), d, a);
}

// sqrt(x), bit-identical to the compiler's correctly rounded expansion for every x that is
// NaN, a negative normal number, a zero of either sign, +infinity, or at least 2^-96 (a
// negative subnormal gives -0 where the expansion gives NaN).  That expansion is:
// multiply x by 2^32 when it is below 2^-96 (the hardware root is not accurate on the
// smallest inputs), s = v_sqrt_f32 (within one unit in the last place), try the floats below
// and above s -- the residuals x - (s -+ ulp)*s, each exact enough in one fma, say on which
// side of s the root lies -- undo the scaling, and return x itself for zeros and infinity.
// Here the scaling and the class test are left out: 9 instructions instead of 16, five of the
// seven dropped being compares and selects.  Zeros and infinity still come out right
// (the neighbours of 0 and of infinity give residuals that are NaN or zero, which select
// nothing), and a negative x gives the NaN the full expansion gives.  Callers either know
// their argument cannot lie in (0, 2^-96) -- a difference 1 - y is a multiple of 2^-24, a
// uniform variate a multiple of 2^-64, a value that is discarded below a threshold -- or check
// the result's exponent and redo the work with the compiler's sqrt (vnormalize).
__device__ __forceinline__ float exact_sqrt_unscaled(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __int_as_float(__float_as_int(s) - 1);
    const float above = __int_as_float(__float_as_int(s) + 1);
    const float r_below = __builtin_fmaf(-below, s, x);
    const float r_above = __builtin_fmaf(-above, s, x);
    float r = (r_below <= 0.0f) ? below : s;
    r = (r_above > 0.0f) ? above : r;
    return r;
}

}  // namespace ptk_exact
