// Host build of gym_miniworld_amd/csrc/mwb_glibc_trig.h for tests/test_glibc_trig.py: the restatement next to
// the libm it restates (the same libm.so.6 CPython's math.sin / math.cos call).  Test infrastructure.
#include "../gym_miniworld_amd/csrc/mwb_glibc_trig.h"

extern "C" {
// out[0] = number of sin mismatches, out[1] = cos mismatches, out[2] = index of the first mismatch or -1
void trig_compare(const double *x, long n, long *out) {
    long bad_s = 0, bad_c = 0, first = -1;
    for (long i = 0; i < n; i++) {
        if (!mwb_trig::in_range(x[i])) continue;
        const double s = mwb_trig::sin_glibc(x[i]), c = mwb_trig::cos_glibc(x[i]);
        const double rs = sin(x[i]), rc = cos(x[i]);
        const bool ms = mwb_trig::bits(s) != mwb_trig::bits(rs), mc = mwb_trig::bits(c) != mwb_trig::bits(rc);
        bad_s += ms; bad_c += mc;
        if ((ms || mc) && first < 0) first = i;
    }
    out[0] = bad_s; out[1] = bad_c; out[2] = first;
}
void trig_eval(const double *x, long n, double *s, double *c) {
    for (long i = 0; i < n; i++) { s[i] = mwb_trig::sin_glibc(x[i]); c[i] = mwb_trig::cos_glibc(x[i]); }
}
// the walk of an agent heading: dir += turn * (pi / 180), compared at every step (miniworld.py:635-656)
void trig_compare_walk(double dir0, const double *turn_deg, long n, long *out) {
    long bad = 0;
    double d = dir0;
    for (long i = 0; i < n; i++) {
        d += turn_deg[i] * (3.141592653589793 / 180);
        const double h = d / 2.0;   // gen_rot_matrix takes the half angle (math.py:16-17)
        bad += mwb_trig::bits(mwb_trig::sin_glibc(d)) != mwb_trig::bits(sin(d));
        bad += mwb_trig::bits(mwb_trig::cos_glibc(d)) != mwb_trig::bits(cos(d));
        bad += mwb_trig::bits(mwb_trig::sin_glibc(h)) != mwb_trig::bits(sin(h));
        bad += mwb_trig::bits(mwb_trig::cos_glibc(h)) != mwb_trig::bits(cos(h));
    }
    out[0] = bad;
}
}
