// lean_sincos.h — sin and cos of a double in [0, 6.5] for the rBRIEF rotation (ORBFinder.cpp:95-98: a = (float)cos(angle), b = (float)sin(angle)).
// One rounding to the nearest multiple of pi/2, the remainder with a two-term pi/2 (two fma), then fdlibm's __kernel_sin / __kernel_cos
// polynomials evaluated with fma: 30 double-precision instructions instead of the ~100 of a general-purpose sincos with its large-argument
// path.  Every operation is an IEEE-754 double operation (fma, mul, add, rint), so the host and the device produce the same bits, and
// tests/test_sincos_exhaustive.py checks on the host that, ROUNDED TO FLOAT, the results equal libm's (float)sin / (float)cos for EVERY
// float argument in [0, 6.5] (1.09e9 values; the rotation angle is a float below 2*pi).  Included by kernels_describe.hip and by that test.
#pragma once
#include <math.h>
#ifndef HS_HD
#define HS_HD
#endif
HS_HD static inline void hs_lean_sincos(double x, double* s, double* c)
{
    const double two_over_pi = 6.36619772367581382433e-01, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double k = rint(x * two_over_pi);
    double r = fma(-k, pio2_hi, x);
    r = fma(-k, pio2_lo, r);
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ps = fma(z, S6, S5); ps = fma(z, ps, S4); ps = fma(z, ps, S3); ps = fma(z, ps, S2); ps = fma(z, ps, S1);
    const double sr = fma(z * r, ps, r);
    double pc = fma(z, C6, C5); pc = fma(z, pc, C4); pc = fma(z, pc, C3); pc = fma(z, pc, C2); pc = fma(z, pc, C1);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + z * z * pc);
    const int q = (int)k & 3;
    const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}
