/* The quotient the kernels use for many numerators over one denominator,
 *     y = RN(1/d);  q = RN(a*y);  t = a - d*q (exact in an fma);  q' = RN(q + t*y),
 * against the IEEE quotient a / d: for each of N denominators (random significands, plus the
 * all-zeros, all-ones and neighbouring ones) every one of the 2^23 numerator significands.
 * Exponents do not matter while nothing over- or underflows: scaling by two is exact.
 * Prints the number of differing quotients and of first quotients q more than one float away
 * from RN(a/d); exit status 1 if either is not zero.  Usage: check_exact_quotient [N] */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float as_float(uint32_t b) { float x; memcpy(&x, &b, 4); return x; }
static inline uint32_t as_bits(float x) { uint32_t b; memcpy(&b, &x, 4); return b; }

int main(int argc, char** argv)
{
    int nd = argc > 1 ? atoi(argv[1]) : 64;
    long differ = 0, far = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : differ, far)
    for (int k = 0; k < nd; k++) {
        uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1);
        s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
        uint32_t md = (uint32_t)(s & 0x7fffffu);
        if (k == 0) md = 0x7fffff;
        if (k == 1) md = 0;
        if (k == 2) md = 1;
        if (k == 3) md = 0x7ffffe;
        if (k == 4) md = 0x440000;   /* 49 */
        if (k == 5) md = 0x100000;   /* 9 */
        const float d = as_float(0x3f800000u | md), y = 1.0f / d;
        for (uint32_t m = 0; m < (1u << 23); m++) {
            const float a = as_float(0x3f800000u | m);
            const float q = a * y;
            const float t = fmaf(-d, q, a);
            const float q2 = fmaf(t, y, q);
            const float want = a / d;
            if (q2 != want) differ++;
            const int32_t step = (int32_t)as_bits(q) - (int32_t)as_bits(want);
            if (step > 1 || step < -1) far++;
        }
    }
    printf("%d denominators x 2^23 numerators: %ld quotients differ, %ld first quotients more than one float off\n",
           nd, differ, far);
    return differ || far ? 1 : 0;
}
