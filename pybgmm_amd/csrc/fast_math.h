// Short double-precision log / exp / division for the one-workgroup kernels whose cost is the length
// of a dependent instruction chain (a visit of the mover-dense path evaluates two logarithms and one
// exponential between two barriers).  The library routines reach their last-bit accuracy through
// double-double arithmetic (~130 instructions each on gfx950); these are the classic table-free
// argument-reduction + minimax-polynomial forms (error below 1 ulp for log and exp, W. Kahan / K. C. Ng,
// as published in FreeBSD msun e_log.c / e_exp.c: restated here, ~35 instructions each).
// Arguments: log -- finite, positive, normal; exp -- any finite (underflows to 0 below -700).
// Host-compilable (tests/test_host_cpu.py::test_short_log_exp_against_libm checks them against libm on the CPU).
//
// The constants and the reduction scheme of fm_log / fm_exp are those of fdlibm's e_log.c / e_exp.c, whose
// notice asks to be kept:
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//   Developed at SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define FM_HD __host__ __device__ __forceinline__
#else
#define FM_HD inline
#endif

// a / b for operands far from the overflow / underflow thresholds: reciprocal, two Newton steps, one
// residual correction (the scaling and fix-up steps of the IEEE division sequence left out)
FM_HD double fm_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
#else
    return a / b;
#endif
}

FM_HD double fm_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
#if defined(__HIP_DEVICE_COMPILE__)
    double m = __builtin_amdgcn_frexp_mant(x);      // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
#else
    int e;
    double m = std::frexp(x, &e);
#endif
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;                             // [sqrt(1/2), sqrt(2))
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double s = fm_div(f, 2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * ln2_hi - ((hfsq - __builtin_fma(s, hfsq + R, dk * ln2_lo)) - f);
}

// log(1 + f) for -0.28 < f < 0.28 (1 + f inside [sqrt(1/2), sqrt(2)]: the k = 0 branch of the same
// algorithm, without forming 1 + f -- what the Student-t predictive of a point near a large component asks for)
FM_HD double fm_log1p_small(double f) {
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    const double s = fm_div(f, 2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return f - (hfsq - s * (hfsq + R));
}

// 1 / sqrt(x), x positive and normal: hardware estimate + two Newton steps
FM_HD double fm_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    y = __builtin_fma(y, 0.5 * __builtin_fma(-(x * y), y, 1.0), y);
    y = __builtin_fma(y, 0.5 * __builtin_fma(-(x * y), y, 1.0), y);
    return y;
#else
    return 1.0 / std::sqrt(x);
#endif
}

FM_HD double fm_exp(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (!(x > -700.0)) return x == x ? 0.0 : x;      // (underflow; NaN stays NaN)
    if (x > 709.0) return INFINITY;
    const double kf = std::rint(x * invln2);
    const int k = (int)kf;
    const double hi = __builtin_fma(-kf, ln2_hi, x), lo = kf * ln2_lo;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, P5, P4), P3), P2), P1);
    const double y = 1.0 - ((lo - fm_div(r * c, 2.0 - c)) - hi);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ldexp(y, k);
#else
    return std::ldexp(y, k);
#endif
}
