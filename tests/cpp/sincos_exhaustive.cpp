// Exhaustive host check of hyslam_amd/csrc/lean_sincos.h against libm: every float theta with bit pattern in [argv[1], argv[2]).
// Build: g++ -O2 -mfma -ffp-contract=off (fma() must be a single rounding; without -mfma libm's fma is used: slower, same result).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../hyslam_amd/csrc/lean_sincos.h"
int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const uint32_t lo = (uint32_t)strtoul(argv[1], nullptr, 0), hi = (uint32_t)strtoul(argv[2], nullptr, 0);
    unsigned long long bad = 0, n = 0;
    for (uint32_t b = lo; b < hi; b++) {
        float t; memcpy(&t, &b, 4);
        double s, c; hs_lean_sincos((double)t, &s, &c);
        const float fs = (float)s, fc = (float)c, rs = (float)sin((double)t), rc = (float)cos((double)t);
        if (fs != rs || fc != rc) { if (bad < 5) printf("theta %a: sin %a vs %a, cos %a vs %a\n", t, fs, rs, fc, rc); bad++; }
        n++;
    }
    printf("%llu values, %llu mismatches\n", n, bad);
    return bad != 0;
}
