// lasgun_amd/csrc/trig.h -- portable f64 sin / cos / atan2 / acos for the HIP kernels.
//
// The reference calls libm inside Sphere::intersect (/root/reference/src/shape/sphere.rs:99-114);
// ROCm's device libm is not glibc, so the product owns one algorithm made only of IEEE-754
// + - * / sqrt floor (tools/gen_trig.py explains it and generates trig_tables.h).  Max error
// measured against 50-digit references: sin/cos 1.4 ulp, atan2 2.4 ulp, acos 3.2 ulp.
#pragma once
#include "vecmath.h"
#include "trig_tables.h"

namespace lg {

LG_HD double poly_S(double z) {
    return LGT_S0 + z * (LGT_S1 + z * (LGT_S2 + z * (LGT_S3 + z * (LGT_S4 + z * (LGT_S5 + z * (LGT_S6 + z * LGT_S7))))));
}
LG_HD double poly_C(double z) {
    return LGT_C0 + z * (LGT_C1 + z * (LGT_C2 + z * (LGT_C3 + z * (LGT_C4 + z * (LGT_C5 + z * (LGT_C6 + z * LGT_C7))))));
}
LG_HD double poly_A(double z) {
    double p = LGT_A15;
    p = LGT_A14 + z * p; p = LGT_A13 + z * p; p = LGT_A12 + z * p; p = LGT_A11 + z * p;
    p = LGT_A10 + z * p; p = LGT_A9 + z * p;  p = LGT_A8 + z * p;  p = LGT_A7 + z * p;
    p = LGT_A6 + z * p;  p = LGT_A5 + z * p;  p = LGT_A4 + z * p;  p = LGT_A3 + z * p;
    p = LGT_A2 + z * p;  p = LGT_A1 + z * p;  p = LGT_A0 + z * p;
    return p;
}
LG_HD double ksin(double r) {
    double z = r * r;
    return r + r * (z * poly_S(z));
}
LG_HD double kcos(double r) {
    double z = r * r;
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * z) * poly_C(z));
}
// x = k*(pi/2) + r, |x| < 2^20 * pi/2
LG_HD double trig_reduce(double x, int &quadrant) {
    double k = floor(x * LGT_INVPIO2 + 0.5);
    double r = ((x - k * LGT_P1) - k * LGT_P2) - k * LGT_P3;
    quadrant = (int)((long long)k & 3);
    return r;
}
LG_HD double p_sin(double x) {
    if (!(x == x) || x - x != 0.0) return x - x;
    if (!(fabs(x) < 1647099.0)) return x - x;
    int q;
    double r = trig_reduce(x, q);
    double s = ksin(r), c = kcos(r);
    double v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
LG_HD double p_cos(double x) {
    if (!(x == x) || x - x != 0.0) return x - x;
    if (!(fabs(x) < 1647099.0)) return x - x;
    int q;
    double r = trig_reduce(x, q);
    double s = ksin(r), c = kcos(r);
    double v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
}
// p_sin(x) and p_cos(x) from ONE argument reduction (bit-identical to the two separate calls)
LG_HD void p_sincos(double x, double &sn, double &cs) {
    if (!(x == x) || x - x != 0.0 || !(fabs(x) < 1647099.0)) { sn = x - x; cs = x - x; return; }
    int q;
    double r = trig_reduce(x, q);
    double s = ksin(r), c = kcos(r);
    double vs = (q & 1) ? c : s, vc = (q & 1) ? s : c;
    sn = (q & 2) ? -vs : vs;
    cs = ((q + 1) & 2) ? -vc : vc;
}
LG_HD double atan01(double t) { // t in [0, 1]
    if (t > LGT_TAN_PIO8) {
        double u = (t - 1.0) / (t + 1.0);
        double z = u * u;
        double a = u + u * (z * poly_A(z));
        return LGT_PIO4_HI + (a + LGT_PIO4_LO);
    }
    double z = t * t;
    return t + t * (z * poly_A(z));
}
LG_HD double p_atan2(double y, double x) {
    if (!(x == x) || !(y == y)) return x + y;
    double ax = fabs(x), ay = fabs(y);
    bool xneg = __builtin_signbit(x) != 0, yneg = __builtin_signbit(y) != 0;
    double r;
    if (ay == 0.0) {
        r = xneg ? (LGT_PI_HI + LGT_PI_LO) : 0.0;
    } else if (ax == 0.0) {
        r = LGT_PIO2_HI + LGT_PIO2_LO;
    } else if (ax - ax != 0.0 || ay - ay != 0.0) {
        if (ax - ax != 0.0 && ay - ay != 0.0) r = xneg ? 3.0 * (LGT_PIO4_HI + LGT_PIO4_LO) : (LGT_PIO4_HI + LGT_PIO4_LO);
        else if (ax - ax != 0.0) r = xneg ? (LGT_PI_HI + LGT_PI_LO) : 0.0;
        else r = LGT_PIO2_HI + LGT_PIO2_LO;
    } else {
        if (ay > ax) {
            double a = atan01(ax / ay);
            r = LGT_PIO2_HI - (a - LGT_PIO2_LO);
        } else {
            r = atan01(ay / ax);
        }
        if (xneg) r = LGT_PI_HI - (r - LGT_PI_LO);
    }
    return yneg ? -r : r;
}
LG_HD double p_acos(double x) {
    if (!(x == x) || x > 1.0 || x < -1.0) return (x - x) / (x - x);
    return 2.0 * p_atan2(sqrt(1.0 - x), sqrt(1.0 + x));
}

} // namespace lg
