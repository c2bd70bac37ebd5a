// mwb_glibc_trig.h - float64 sin / cos that reproduce, bit for bit, what `math.sin` / `math.cos` return in the
// reference's process (entity.py:72-90 dir_vec / right_vec, math.py:16-17 gen_rot_matrix): CPython calls libm,
// i.e. glibc 2.35 (the version this image pins) sysdeps/ieee754/dbl-64/s_sin.c - the IBM Accurate Mathematical
// Library routines __sin / __cos with do_sin, do_cos, TAYLOR_SIN, reduce_sincos and the 440-word table
// __sincostab - in the variant every FMA-capable x86-64 host selects through its ifunc (__sin_fma / __cos_fma,
// sysdeps/x86_64/fpu/multiarch/s_sin-fma.c: the same C source built with -mfma -mavx2, so that GCC contracted
// a * b + c into fused multiply-adds).  Which products are fused decides the last bit, so the contraction pattern
// below is the one read from the disassembly of this image's libm.so.6 (sin: 0x789b0, cos: 0x791c0); every fma()
// here is one vfmadd/vfnmadd/vfmsub there and every other operation is a separately rounded one.
// Build with -ffp-contract=off (as the rest of the library) so that nothing else fuses.
//
// Covered: |x| < 105414350 (everything an agent heading can reach: it grows by <= 0.35 rad per step).  Beyond
// that glibc switches to __branred (a 1200-bit table of 2/pi); callers fall back to the platform's sin / cos
// there - not bit-pinned, never reached.
//
// tests/test_glibc_trig.py compiles this header for the host and compares it with libm on > 10^7 arguments.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MWB_TRIG_FN __host__ __device__ static inline
#define MWB_TRIG_TABLE __device__ __constant__
#else
#define MWB_TRIG_FN static inline
#define MWB_TRIG_TABLE static const
#endif

// sin(k/128) hi, lo, cos(k/128) hi, lo for k = 0..109 (scripts/gen_sincos_table.py)
MWB_TRIG_TABLE double mwb_sincostab[440] = {
#include "mwb_sincos_table.inc"
};
#if defined(__HIPCC__)
static const double mwb_sincostab_host[440] = {
#include "mwb_sincos_table.inc"
};
#endif

namespace mwb_trig {

MWB_TRIG_FN const double *table() {
#if defined(__HIP_DEVICE_COMPILE__)
    return mwb_sincostab;
#elif defined(__HIPCC__)
    return mwb_sincostab_host;
#else
    return mwb_sincostab;
#endif
}

MWB_TRIG_FN uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }

// usncs.h / trigo.h constants of s_sin.c (values as stored in libm.so.6's .rodata)
#define MWB_S1 (-0x1.5555555555555p-3)
#define MWB_S2 (0x1.1111111110ecep-7)
#define MWB_S3 (-0x1.a01a019db08b8p-13)
#define MWB_S4 (0x1.71de27b9a7ed9p-19)
#define MWB_S5 (-0x1.addffc2fcdf59p-26)
#define MWB_SN3 (-0x1.5555555555515p-3)
#define MWB_SN5 (0x1.11110e829872fp-7)
#define MWB_CS2 (0.5)
#define MWB_CS4 (-0x1.5555555555535p-5)
#define MWB_CS6 (0x1.6c16bedd9e239p-10)
#define MWB_BIG (0x1.8p+45)
#define MWB_TOINT (0x1.8p+52)
#define MWB_HPINV (0x1.45f306dc9c883p-1)
#define MWB_HP0 (0x1.921fb54442d18p+0)
#define MWB_HP1 (0x1.1a62633145c07p-54)
#define MWB_MP1 (0x1.921fb58000000p+0)
#define MWB_MP2 (-0x1.dde973c000000p-27)
#define MWB_PP3 (-0x1.cb3b398000000p-55)
#define MWB_PP4 (-0x1.d747f23e32ed7p-83)

// TAYLOR_SIN (xx, a, da): a + ((POLYNOMIAL (xx) * a - 0.5 * da) * xx + da)
MWB_TRIG_FN double taylor_sin(double a, double da) {
    const double xx = a * a;
    double p = fma(xx, MWB_S5, MWB_S4);
    p = fma(xx, p, MWB_S3);
    p = fma(xx, p, MWB_S2);
    p = fma(xx, p, MWB_S1);
    const double h = da * 0.5;
    const double t = fma(fma(p, a, -h), xx, da);
    return a + t;
}

// SINCOS_TABLE_LOOKUP: u = big + |x| rounds |x| to a multiple of 1/128; its low word is the table index
MWB_TRIG_FN int table_index(double u) { return (int)(uint32_t)(bits(u) & 0xffffffffu) * 4; }

// do_sin (x, dx)
MWB_TRIG_FN double do_sin(double x, double dx) {
    const double ax = fabs(x);
    if (ax < 0.126) return taylor_sin(x, dx);
    if (x <= 0) dx = -dx;
    const double u = MWB_BIG + ax;
    const double r = ax - (u - MWB_BIG);
    const double *T = table() + table_index(u);
    const double sn = T[0], ssn = T[1], cs = T[2], ccs = T[3];
    const double xx = r * r;
    const double s = r + fma(r * xx, fma(xx, MWB_SN5, MWB_SN3), dx);
    double q = fma(xx, MWB_CS6, MWB_CS4);
    q = fma(xx, q, MWB_CS2);
    const double c = fma(r, dx, xx * q);
    const double cor = fma(s, cs, fma(-c, sn, fma(s, ccs, ssn)));
    return copysign(sn + cor, x);
}

// do_cos (x, dx)
MWB_TRIG_FN double do_cos(double x, double dx) {
    if (x < 0) dx = -dx;
    const double ax = fabs(x);
    const double u = MWB_BIG + ax;
    const double r = (ax - (u - MWB_BIG)) + dx;
    const double *T = table() + table_index(u);
    const double sn = T[0], ssn = T[1], cs = T[2], ccs = T[3];
    const double xx = r * r;
    const double s = fma(r * xx, fma(xx, MWB_SN5, MWB_SN3), r);
    double q = fma(xx, MWB_CS6, MWB_CS4);
    q = fma(xx, q, MWB_CS2);
    const double c = xx * q;
    const double cor = fma(-s, sn, fma(-c, cs, fma(-s, ssn, ccs)));
    return cs + cor;
}

// reduce_sincos (x, &a, &da): quadrant n and x - n pi/2 as a + da, for |x| < 105414350
MWB_TRIG_FN int reduce_sincos(double x, double &a, double &da) {
    const double t = fma(x, MWB_HPINV, MWB_TOINT);
    const double xn = t - MWB_TOINT;
    const int n = (int)(bits(t) & 3u);
    const double y = fma(-xn, MWB_MP2, fma(-xn, MWB_MP1, x));
    const double t2 = fma(-xn, MWB_PP3, y);
    double db = fma(-xn, MWB_PP3, y - t2);
    const double b = fma(-xn, MWB_PP4, t2);
    db = db + fma(-xn, MWB_PP4, t2 - b);
    a = b; da = db;
    return n;
}

MWB_TRIG_FN double do_sincos(double a, double da, int n) {
    const double r = (n & 1) ? do_cos(a, da) : do_sin(a, da);
    return (n & 2) ? -r : r;
}

// true where the restatement applies (finite |x| < 105414350: high word below 0x419921FB)
MWB_TRIG_FN bool in_range(double x) { return (uint32_t)((bits(x) >> 32) & 0x7fffffffu) < 0x419921FBu; }

// __sin
MWB_TRIG_FN double sin_glibc(double x) {
    const uint32_t k = (uint32_t)((bits(x) >> 32) & 0x7fffffffu);
    if (k < 0x3e500000u) return x;                               // |x| < 2^-26
    if (k < 0x3feb6000u) return do_sin(x, 0.0);                  // |x| < 0.855469
    if (k < 0x400368fdu) return copysign(do_cos(MWB_HP0 - fabs(x), MWB_HP1), x);   // |x| < 2.426265
    double a, da;
    const int n = reduce_sincos(x, a, da);
    return do_sincos(a, da, n);
}

// __cos
MWB_TRIG_FN double cos_glibc(double x) {
    const uint32_t k = (uint32_t)((bits(x) >> 32) & 0x7fffffffu);
    if (k < 0x3e400000u) return 1.0;                             // |x| < 2^-27
    if (k < 0x3feb6000u) return do_cos(x, 0.0);
    if (k < 0x400368fdu) {
        const double y = MWB_HP0 - fabs(x);
        const double a = y + MWB_HP1;
        const double da = (y - a) + MWB_HP1;
        return do_sin(a, da);
    }
    double a, da;
    const int n = reduce_sincos(x, a, da);
    return do_sincos(a, da, n + 1);
}

}   // namespace mwb_trig
