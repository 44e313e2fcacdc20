/* oracle/lg_trig.h -- TEST INFRASTRUCTURE (CPU oracle). Not part of the product.
 *
 * Portable IEEE-754 f64 sin / cos / atan2 / acos built only from + - * / sqrt
 * floor, used by the oracle's "portable" trig mode.  The oracle's default mode
 * calls glibc libm, which is what the reference's Rust f64::atan2/acos/sin/cos
 * resolve to (/root/reference/src/shape/sphere.rs:99-114).  The portable mode
 * exists so that GPU radiance can be compared BIT-FOR-BIT with a CPU run that
 * uses the same published algorithm (tools/gen_trig.py docstring); the libm
 * mode shows the RGBA8 bytes do not depend on which of the two is used.
 *
 * Compile with -ffp-contract=off.
 */
#ifndef ORACLE_LG_TRIG_H
#define ORACLE_LG_TRIG_H

#include "lg_trig_tables.h"

static inline double orc_poly_S(double z) {
    return LGT_S0 + z * (LGT_S1 + z * (LGT_S2 + z * (LGT_S3 + z * (LGT_S4 + z * (LGT_S5 + z * (LGT_S6 + z * LGT_S7))))));
}
static inline double orc_poly_C(double z) {
    return LGT_C0 + z * (LGT_C1 + z * (LGT_C2 + z * (LGT_C3 + z * (LGT_C4 + z * (LGT_C5 + z * (LGT_C6 + z * LGT_C7))))));
}
static inline double orc_poly_A(double z) {
    double p = LGT_A15;
    p = LGT_A14 + z * p; p = LGT_A13 + z * p; p = LGT_A12 + z * p; p = LGT_A11 + z * p;
    p = LGT_A10 + z * p; p = LGT_A9 + z * p;  p = LGT_A8 + z * p;  p = LGT_A7 + z * p;
    p = LGT_A6 + z * p;  p = LGT_A5 + z * p;  p = LGT_A4 + z * p;  p = LGT_A3 + z * p;
    p = LGT_A2 + z * p;  p = LGT_A1 + z * p;  p = LGT_A0 + z * p;
    return p;
}

/* kernels on |r| <= pi/4 */
static inline double orc_ksin(double r) {
    double z = r * r;
    return r + r * (z * orc_poly_S(z));
}
static inline double orc_kcos(double r) {
    double z = r * r;
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * z) * orc_poly_C(z));
}

/* Argument reduction: x = k*(pi/2) + r.  Valid for |x| < 2^20 * pi/2; the hot
 * path only passes phi in [0, 2pi] and theta in [0, pi]. */
static inline double orc_reduce(double x, int *quadrant) {
    double k = __builtin_floor(x * LGT_INVPIO2 + 0.5);
    double r = ((x - k * LGT_P1) - k * LGT_P2) - k * LGT_P3;
    *quadrant = (int)((long long)k & 3);
    return r;
}

static inline double orc_sin(double x) {
    if (!(x == x) || x - x != 0.0) return x - x;      /* NaN or inf -> NaN */
    if (!(__builtin_fabs(x) < 1647099.0)) return x - x; /* outside supported range: NaN (never on the hot path) */
    int q; double r = orc_reduce(x, &q);
    switch (q) {
        case 0: return orc_ksin(r);
        case 1: return orc_kcos(r);
        case 2: return -orc_ksin(r);
        default: return -orc_kcos(r);
    }
}
static inline double orc_cos(double x) {
    if (!(x == x) || x - x != 0.0) return x - x;
    if (!(__builtin_fabs(x) < 1647099.0)) return x - x;
    int q; double r = orc_reduce(x, &q);
    switch (q) {
        case 0: return orc_kcos(r);
        case 1: return -orc_ksin(r);
        case 2: return -orc_kcos(r);
        default: return orc_ksin(r);
    }
}

/* atan(t) for t in [0, 1] */
static inline double orc_atan01(double t) {
    if (t > LGT_TAN_PIO8) {
        double u = (t - 1.0) / (t + 1.0);
        double z = u * u;
        double a = u + u * (z * orc_poly_A(z));
        return LGT_PIO4_HI + (a + LGT_PIO4_LO);
    } else {
        double z = t * t;
        return t + t * (z * orc_poly_A(z));
    }
}

static inline double orc_atan2(double y, double x) {
    if (!(x == x) || !(y == y)) return x + y;
    double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    int xneg = __builtin_signbit(x) != 0, yneg = __builtin_signbit(y) != 0;
    double r;
    if (ay == 0.0) {
        r = xneg ? (LGT_PI_HI + LGT_PI_LO) : 0.0;
    } else if (ax == 0.0) {
        r = LGT_PIO2_HI + LGT_PIO2_LO;
    } else if (ax - ax != 0.0 || ay - ay != 0.0) { /* an infinity */
        if (ax - ax != 0.0 && ay - ay != 0.0) r = xneg ? 3.0 * (LGT_PIO4_HI + LGT_PIO4_LO) : (LGT_PIO4_HI + LGT_PIO4_LO);
        else if (ax - ax != 0.0) r = xneg ? (LGT_PI_HI + LGT_PI_LO) : 0.0;
        else r = LGT_PIO2_HI + LGT_PIO2_LO;
    } else {
        if (ay > ax) {
            double a = orc_atan01(ax / ay);
            r = LGT_PIO2_HI - (a - LGT_PIO2_LO);
        } else {
            r = orc_atan01(ay / ax);
        }
        if (xneg) r = LGT_PI_HI - (r - LGT_PI_LO);
    }
    return yneg ? -r : r;
}

/* acos(x) = 2*atan2(sqrt(1-x), sqrt(1+x)), |x| <= 1 */
static inline double orc_acos(double x) {
    if (!(x == x) || x > 1.0 || x < -1.0) return (x - x) / (x - x);
    return 2.0 * orc_atan2(__builtin_sqrt(1.0 - x), __builtin_sqrt(1.0 + x));
}

#endif
