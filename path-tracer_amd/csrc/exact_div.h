// exact_div.h -- correctly rounded fp32 quotients of many numerators by one denominator.
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
// scripts/microbench/exact_div_check.hip does the same on the device, reciprocal included.
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

}  // namespace ptk_exact
