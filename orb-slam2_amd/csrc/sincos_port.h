// cosf / sinf with the arithmetic of glibc 2.35's float routines (sysdeps/ieee754/flt-32
// s_sinf.c, s_cosf.c, sincosf.h: double-precision range reduction by pi/2 and two short
// polynomials), so that the device rotates the BRIEF pattern with the same a = cos, b = sin the
// host libm gives the reference (src/ORBextractor.cc:159-160).  Valid for |x| < 120; the
// descriptor angle is within [0, 2*pi].  The constants are the entries of glibc's
// __sincosf_table.  ORB_SINCOS_FMA selects fused evaluation of the a*b+c steps (what glibc's
// FMA-dispatched x86-64 build does); the two variants are compared against the host libm over
// every float of the working range by the host unit test.
#ifndef ORBHIP_SINCOS_PORT_H
#define ORBHIP_SINCOS_PORT_H
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "hd.h"

namespace orbtrig {

ORB_HD inline uint32_t bits_of(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

template <bool FMA>
ORB_HD inline double madd(double a, double b, double c) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    if (FMA) return fma(a, b, c);
    double p = a * b;
    return p + c;
}

// n even: sine polynomial of x; n odd: cosine polynomial.  alt selects the negated-cosine table.
template <bool FMA>
ORB_HD inline float poly(double x, double x2, bool alt, int n) {
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = madd<FMA>(x2, S3, S2);
        double x7 = x3 * x2;
        double s = madd<FMA>(x3, S1, x);
        return (float)madd<FMA>(x7, s1, s);
    }
    const double sg = alt ? -1.0 : 1.0;
    const double C0 = sg * 0x1p0, C1 = sg * -0x1.ffffffd0c621cp-2, C2 = sg * 0x1.55553e1068f19p-5,
                 C3 = sg * -0x1.6c087e89a359dp-10, C4 = sg * 0x1.99343027bf8c3p-16;
    double x4 = x2 * x2;
    double c2 = madd<FMA>(x2, C4, C3);
    double c1 = madd<FMA>(x2, C1, C0);
    double x6 = x4 * x2;
    double c = madd<FMA>(x4, C2, c1);
    return (float)madd<FMA>(x6, c2, c);
}

template <bool FMA>
ORB_HD inline double reduce(double x, int* np) {
    const double HPI_INV_2P24 = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    double r = x * HPI_INV_2P24;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return madd<FMA>(-(double)n, HPI, x);
}

template <bool FMA>
ORB_HD inline float sin_f(float y) {
    double x = y;
    uint32_t top = (bits_of(y) >> 20) & 0x7ff;
    if (top < 0x3f4) {  // |y| < 0.75 by the top 12 bits, as glibc tests against pi/4
        if (top < 0x398) return y;  // |y| < 2^-12
        return poly<FMA>(x, x * x, false, 0);
    }
    int n;
    x = reduce<FMA>(x, &n);
    double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return poly<FMA>(x * s, x * x, (n & 2) != 0, n);
}

template <bool FMA>
ORB_HD inline float cos_f(float y) {
    double x = y;
    uint32_t top = (bits_of(y) >> 20) & 0x7ff;
    if (top < 0x3f4) {
        if (top < 0x398) return 1.0f;
        return poly<FMA>(x, x * x, false, 1);
    }
    int n;
    x = reduce<FMA>(x, &n);
    double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return poly<FMA>(x * s, x * x, (n & 2) != 0, n ^ 1);
}

}  // namespace orbtrig
#endif
