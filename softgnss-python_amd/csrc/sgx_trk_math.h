// Short-dependency-chain fp64 routines for the tracking kernel's loop-filter waves (sgx_trk2.hip).
//
// The per-block chain  sums -> discriminators -> NCO -> next block's parameters  runs on one wave; a dependent
// fp64 operation costs ~11 cycles there and libm's atan / sqrt / IEEE division cost 240 / 145 / 105
// (tools/ubench_chain.hip).  These replacements trade correct rounding for depth: every result is within a few
// ulp, which is far inside what the discriminators need (the loop inputs are continuous in them),
// while everything that feeds an integer rounding (block length, chip indices) keeps the reference's exact
// arithmetic elsewhere.  The functions compile for the host as well (the hardware seeds are replaced by
// float-precision ones, i.e. WORSE seeds), so tests/test_cabi_and_host.py checks the ulp bounds on the CPU.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SGX_HD __host__ __device__ __forceinline__
#else
#define SGX_HD static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SGX_RCP_SEED(x) __builtin_amdgcn_rcp(x)
#define SGX_RSQ_SEED(x) __builtin_amdgcn_rsq(x)
#else
#define SGX_RCP_SEED(x) ((double)(1.0f / (float)(x)))
#define SGX_RSQ_SEED(x) ((double)(1.0f / sqrtf((float)(x))))
#endif

// 1 / x, |error| <= 1 ulp for normal x (two Newton steps on a >= 20-bit seed)
SGX_HD double sgx_fast_rcp(double x) {
    double y = SGX_RCP_SEED(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}

// a / b given y ~ 1/b: one residual correction, |error| <= 1 ulp
SGX_HD double sgx_div_with_rcp(double a, double b, double y) {
    const double q = a * y;
    const double r = __builtin_fma(-q, b, a);
    return __builtin_fma(r, y, q);
}

SGX_HD double sgx_fast_div(double a, double b) { return sgx_div_with_rcp(a, b, sgx_fast_rcp(b)); }

// sqrt(x) for x >= 0 (0 -> 0), |error| <= 1 ulp: coupled Newton iteration on (sqrt, 1/(2 sqrt)) + one residual step
SGX_HD double sgx_fast_sqrt(double x) {
    const double y = SGX_RSQ_SEED(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x > 0.0 ? g : 0.0;
}

// a / b with ONE Newton step on the reciprocal seed and one residual correction of the quotient: the correction
// squares the reciprocal's error (<= 2^-40 even from a 20-bit seed), so the quotient is within 1 ulp - two
// instructions shorter than sgx_fast_div on the per-block chain
SGX_HD double sgx_div1(double a, double b) {
    double y = SGX_RCP_SEED(b);
    y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
    const double q = a * y;
    return __builtin_fma(__builtin_fma(-q, b, a), y, q);
}

// sqrt(x) for x >= 0 (0 -> 0), |error| <= 1 ulp: ONE coupled Newton iteration on (sqrt, 1/(2 sqrt)) + one residual
// step (which squares the remaining error, as in sgx_div1)
SGX_HD double sgx_sqrt1(double x) {
    const double y = SGX_RSQ_SEED(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x > 0.0 ? g : 0.0;
}

// the same without the x > 0 select: x = 0 gives NaN (0 * inf), which is what the DLL discriminator makes of two zero
// envelopes anyway ((0 - 0) / (0 + 0), tracking.py:238-244)
SGX_HD double sgx_sqrt1_pos(double x) {
    const double y = SGX_RSQ_SEED(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

#define SGX_ATAN_SHORT_MAX 0.25
// atan(z) for |z| <= 0.25: z + z u Q(u), u = z^2, Q of degree 8 (tools/fit_atan.py: < 1 ulp), evaluated
// Estrin-style (depth 6 after z instead of 11)
SGX_HD double sgx_atan_short(double z) {
    const double u = z * z;
    const double u2 = u * u;
    const double zu = z * u;
    const double u4 = u2 * u2;
    const double p01 = __builtin_fma(1.99999999999989325e-01, u, -3.33333333333333315e-01);
    const double p23 = __builtin_fma(1.11111110359868467e-01, u, -1.42857142852585106e-01);
    const double p45 = __builtin_fma(7.69201787723738512e-02, u, -9.09090287496941568e-02);
    const double p67 = __builtin_fma(5.75299439047802855e-02, u, -6.65868741469774345e-02);
    const double q0 = __builtin_fma(p23, u2, p01);
    const double q1 = __builtin_fma(p67, u2, p45);
    const double r0 = __builtin_fma(q1, u4, q0);
    const double r = __builtin_fma(-4.10342669854194958e-02 * u4, u4, r0);
    return __builtin_fma(zu, r, z);
}

// the same with the nine coefficients handed in (the caller keeps them in registers across its loop: a constant the
// compiler materialises in front of every use costs an instruction on the chain)
struct SgxAtanCoef {
    double c0, c1, c2, c3, c4, c5, c6, c7, c8;
};
SGX_HD SgxAtanCoef sgx_atan_coef() {
    SgxAtanCoef k;
    k.c0 = -3.33333333333333315e-01; k.c1 = 1.99999999999989325e-01; k.c2 = -1.42857142852585106e-01;
    k.c3 = 1.11111110359868467e-01; k.c4 = -9.09090287496941568e-02; k.c5 = 7.69201787723738512e-02;
    k.c6 = -6.65868741469774345e-02; k.c7 = 5.75299439047802855e-02; k.c8 = -4.10342669854194958e-02;
    return k;
}
SGX_HD double sgx_atan_short_k(double z, const SgxAtanCoef& k) {
    const double u = z * z;
    const double u2 = u * u;
    const double zu = z * u;
    const double u4 = u2 * u2;
    const double p01 = __builtin_fma(k.c1, u, k.c0);
    const double p23 = __builtin_fma(k.c3, u, k.c2);
    const double p45 = __builtin_fma(k.c5, u, k.c4);
    const double p67 = __builtin_fma(k.c7, u, k.c6);
    const double q0 = __builtin_fma(p23, u2, p01);
    const double q1 = __builtin_fma(p67, u2, p45);
    const double r0 = __builtin_fma(q1, u4, q0);
    const double r = __builtin_fma(k.c8 * u4, u4, r0);
    return __builtin_fma(zu, r, z);
}

// atan(q / i) with the quotient by sgx_div1 and the coefficients in registers
SGX_HD double sgx_atan_ratio_k(double q, double i, const SgxAtanCoef& k) {
    const double z = sgx_div1(q, i);
    // (the short polynomial unconditionally, libm behind ONE cold branch: the usual path then runs straight through)
    double r = sgx_atan_short_k(z, k);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(r));   // (the polynomial BEFORE the branch: the compare's result is long there when the branch asks for it)
#endif
    if (__builtin_expect(!(fabs(z) <= SGX_ATAN_SHORT_MAX), 0)) r = atan(q / i);
    return r;
}

// atan(q / i): the PLL discriminator's argument (tracking.py:223).  Short path while |q / i| <= 0.25 (a locked
// channel is there ~95 % of the time), libm otherwise (also +-inf and NaN, IEEE semantics as in numpy).
SGX_HD double sgx_atan_ratio(double q, double i) {
    const double y = sgx_fast_rcp(i);
    const double z = sgx_div_with_rcp(q, i, y);
    if (fabs(z) <= SGX_ATAN_SHORT_MAX) return sgx_atan_short(z);
    return atan(q / i);
}

// sin and cos of 2 pi u for u in [0, 2): quarter-turn reduction (exact), Taylor polynomials on |theta| <= pi/4
// evaluated Estrin-style (depth 5 after theta^2 instead of 9).  ~1 ulp.
SGX_HD void sgx_sincos_turns_short(double u, double& sn, double& cs) {
    const double q = rint(u * 4.0);
    const double f = __builtin_fma(q, -0.25, u);          // exact, |f| <= 1/8
    const int qi = (int)q & 3;
    const double th = f * 6.283185307179586476925287;
    const double t2 = th * th;
    const double t4 = t2 * t2;
    const double t8 = t4 * t4;
    // sin(th) = th - th^3 (S0 + S1 t2 + ... + S7 t2^7),  S_k = (-1)^k / (2k+3)!
    const double s01 = __builtin_fma(-8.3333333333333332e-03, t2, 1.6666666666666666e-01);
    const double s23 = __builtin_fma(-2.7557319223985893e-06, t2, 1.9841269841269841e-04);
    const double s45 = __builtin_fma(-1.6059043836821613e-10, t2, 2.5052108385441720e-08);
    const double s67 = __builtin_fma(-2.8114572543455206e-15, t2, 7.6471637318198164e-13);
    const double sa = __builtin_fma(s23, t4, s01);
    const double sb = __builtin_fma(s67, t4, s45);
    const double ps = __builtin_fma(sb, t8, sa);
    // cos(th) = 1 + t2 (C0 + C1 t2 + ... + C7 t2^7),  C_k = (-1)^(k+1) / (2k+2)!
    const double c01 = __builtin_fma(4.1666666666666664e-02, t2, -0.5);
    const double c23 = __builtin_fma(2.4801587301587302e-05, t2, -1.3888888888888889e-03);
    const double c45 = __builtin_fma(2.0876756987868100e-09, t2, -2.7557319223985888e-07);
    const double c67 = __builtin_fma(4.7794773323873853e-14, t2, -1.1470745597729725e-11);
    const double ca = __builtin_fma(c23, t4, c01);
    const double cb = __builtin_fma(c67, t4, c45);
    const double pc = __builtin_fma(cb, t8, ca);
    const double s0 = __builtin_fma(-(ps * t2), th, th);
    const double c0 = __builtin_fma(pc, t2, 1.0);
    sn = (qi == 0) ? s0 : (qi == 1) ? c0 : (qi == 2) ? -s0 : -c0;
    cs = (qi == 0) ? c0 : (qi == 1) ? -s0 : (qi == 2) ? -c0 : s0;
}

// sin and cos of a small angle, |ph| <= 0.34 rad: Taylor to ph^13 / ph^12 (next terms 6e-20 / 3e-18), Estrin; the
// twelve coefficients are handed in like the atan's
struct SgxRotCoef {
    double s0, s1, s2, s3, s4, s5, c0, c1, c2, c3, c4, c5;
};
SGX_HD SgxRotCoef sgx_rot_coef() {
    SgxRotCoef k;
    k.s0 = -1.6666666666666666e-01; k.s1 = 8.3333333333333332e-03; k.s2 = -1.9841269841269841e-04;
    k.s3 = 2.7557319223985893e-06; k.s4 = -2.5052108385441720e-08; k.s5 = 1.6059043836821613e-10;
    k.c0 = -0.5; k.c1 = 4.1666666666666664e-02; k.c2 = -1.3888888888888889e-03;
    k.c3 = 2.4801587301587302e-05; k.c4 = -2.7557319223985888e-07; k.c5 = 2.0876756987868100e-09;
    return k;
}
#define SGX_ROT_MAX 0.34
SGX_HD void sgx_rot_small(double ph, const SgxRotCoef& k, double& sn, double& cs) {
    const double t = ph * ph;
    const double t2 = t * t;
    const double t4 = t2 * t2;
    const double pt = ph * t;
    const double s01 = __builtin_fma(k.s1, t, k.s0);
    const double s23 = __builtin_fma(k.s3, t, k.s2);
    const double s45 = __builtin_fma(k.s5, t, k.s4);
    const double c01 = __builtin_fma(k.c1, t, k.c0);
    const double c23 = __builtin_fma(k.c3, t, k.c2);
    const double c45 = __builtin_fma(k.c5, t, k.c4);
    const double sa = __builtin_fma(s23, t2, s01);
    const double ca = __builtin_fma(c23, t2, c01);
    const double ps = __builtin_fma(s45, t4, sa);
    const double pc = __builtin_fma(c45, t4, ca);
    sn = __builtin_fma(pt, ps, ph);
    cs = __builtin_fma(t, pc, 1.0);
}

// a / b correctly rounded, given y = RN(1/b): reciprocal multiply plus two FMA corrections (Markstein).  Bit-identical
// to IEEE division for the divisors used here (pi, fs, block lengths), checked against exact rational arithmetic in
// tests/test_cabi_and_host.py
SGX_HD double sgx_div_rn(double a, double b, double y) {
    const double q0 = a * y;
    const double r0 = __builtin_fma(-q0, b, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-q1, b, a);
    return __builtin_fma(r1, y, q1);
}

// Block length of T1 (tracking.py:148-151), ceil(a / step) with a = 1023 - remCodePhase and step = RN(codeFreq / fs),
// without a division on the chain.  step_a = codeFreq * RN(1/fs) is within 3 ulp of that step, the corrected reciprocal
// quotient within 1 ulp of a / step_a, so the quotient is within 6 ulp of the reference's and ceil() of it is the
// reference's block length unless it lies that close to an integer; then (probability ~1e-10 per block, and block 0,
// whose quotient IS an integer) the exact arithmetic decides.  Also returns step_a and ~1 / step_a (2^-40).
SGX_HD int sgx_block_length(double a, double codeFreq, double fs, double inv_fs, double& step_a, double& inv_step) {
    step_a = codeFreq * inv_fs;
    double y = SGX_RCP_SEED(step_a);
    y = __builtin_fma(y, __builtin_fma(-step_a, y, 1.0), y);
    const double q0 = a * y;
    const double q = __builtin_fma(__builtin_fma(-q0, step_a, a), y, q0);
    inv_step = y;
    const double c = ceil(q);
    const double lo = q - c + 1.0;                         // distance above the integer below (exact near it)
    const double tol = q * 1.4e-15;
    if (__builtin_expect(c - q < tol || lo < tol, 0)) return (int)ceil(a / sgx_div_rn(codeFreq, fs, inv_fs));
    return (int)c;
}

// ceil(a / b) for a, b > 0 without an IEEE division: the quotient from the reciprocal is within 2 ulp of a / b, so
// ceil() of it equals ceil() of the correctly rounded quotient unless that lies within 4 ulp of an integer; then (and
// only then) the true division decides.  Equal to (int)ceil(a / b) always.
SGX_HD int sgx_ceil_div(double a, double b) {
    const double q = sgx_fast_div(a, b);
    const double c = ceil(q);
    const double lo = q - c + 1.0;                         // distance above the integer below (exact near it)
    const double tol = q * 8.9e-16;
    if (__builtin_expect(c - q < tol || lo < tol, 0)) return (int)ceil(a / b);
    return (int)c;
}
