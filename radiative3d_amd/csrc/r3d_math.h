// r3d_math.h -- fp64 vector / complex helpers for the traversal kernels.
//
// Directions are carried as unit vectors (the reference carries (theta, phi)
// pairs and converts to xyz on every use, geom_r3.cpp:36-40 -- 9 % of its CPU
// time per SURVEY.md); the spherical unit vectors theta^, phi^ that the
// reference gets from trigonometry (geom_r3.cpp:85-126, :212-233) are formed
// here algebraically from the direction cosines.
#ifndef R3D_MATH_H_
#define R3D_MATH_H_

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define R3D_HD __host__ __device__ __forceinline__
#else
#define R3D_HD inline
#endif

// A point the instruction scheduler may not move code across (device builds).  The long
// straight-line solves are otherwise interleaved for instruction-level parallelism until their
// temporaries no longer fit the register budget of three waves per SIMD.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(R3D_NO_SCHED_FENCE)
#define R3D_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define R3D_SCHED_FENCE() ((void)0)
#endif

namespace r3d {

constexpr double kPi = 3.14159265358979323846;
constexpr double kPi90 = kPi * 0.5;
constexpr double kPi360 = kPi * 2.0;

R3D_HD double pos_inf() { return __builtin_inf(); }
#if !defined(__HIPCC__)
inline double rsqrt(double x) { return 1.0 / sqrt(x); }   // device builds use the HIP intrinsic
#endif

// Square root and reciprocal square root for the hot path.  The device library's versions
// spend half of their ~20 instructions on re-scaling for subnormal and huge arguments, which
// the quantities here (squared lengths of O(1) vectors, 1 - cos^2) never are: v_rsq_f64 plus
// Newton steps gives the same result to an ulp in 8 resp. 12 instructions, and the traversal
// kernel is bound by instruction issue.  0 -> 0 resp. inf (NaN through the Newton step, as
// every caller selects around a zero argument); negative -> NaN.  On the host: libm.
R3D_HD double frsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = __builtin_fma(y, __builtin_fma(-h * y, y, 0.5), y);
  y = __builtin_fma(y, __builtin_fma(-h * y, y, 0.5), y);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}
R3D_HD double fsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  y = __builtin_fma(y, __builtin_fma(-0.5 * x * y, y, 0.5), y);   // 1 / sqrt(x) to ~2^-50
  double g = x * y;
  const double d = __builtin_fma(-g, g, x);                       // residual of g^2 against x
  g = __builtin_fma(d, 0.5 * y, g);
  return (x == 0.0) ? 0.0 : g;
#else
  return sqrt(x);
#endif
}

// 1 / x for finite, non-zero x of moderate size (no scaling, no special cases: 0 and inf give NaN):
// v_rcp_f64 plus two Newton steps, 5 instructions against the ~13 of an IEEE division; result within
// an ulp or two.  Only where the operand is known to be a plain number (a velocity, a radius, a
// denominator near 1); quotients whose zero / infinite cases mean something keep the division.
R3D_HD double frcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
  y = __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
  y = __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
  return y;
#else
  return 1.0 / x;
#endif
}
// The same two with ONE Newton step (relative error ~2^-50 instead of an ulp): for quantities that feed a
// decision held with a margin of 1e-8, or a length that is good to 1e-13 anyway (the tetra move's search).
R3D_HD double frsqrt1(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  return __builtin_fma(y, __builtin_fma(-0.5 * x * y, y, 0.5), y);
#else
  return 1.0 / sqrt(x);
#endif
}
R3D_HD double frcp1(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
  return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
#else
  return 1.0 / x;
#endif
}
// True if the condition holds in every active lane of the wave (on the host: in this one).  For
// choosing a shorter series when ALL lanes qualify -- a per-lane branch would run both.
//
// Consequence: WHICH form evaluates a history's argument depends on the histories that share its
// wave, and the forms differ in the last bit or two -- so a history's results are defined to
// rounding (~1e-16 relative per call), not to the bit, and which histories share a wave is decided
// by the pool's scheduling.  Building with -DR3D_REPRODUCIBLE (make repro -> libr3d_hip_repro.so,
// loaded under R3D_REPRODUCIBLE=1; that build also turns floating-point contraction off, so that the
// diagnostic and the production kernels round alike) makes every vote fail, i.e. the general form
// serves every lane always: a history's result is then a function of (model, seed, id) alone, bit for bit,
// whatever the pool size, launch boundaries or batch-mates (tests/test_gpu_parity.py
// test_reproducible_build_*), at the cost stated in DESIGN.md section 4.
R3D_HD bool all_lanes(bool c) {
#if defined(R3D_FLOOR_TIERS)   // (tools/microbench/phase_floor.hip: instruction counts of the short tiers; never run)
#if !defined(R3D_DEV_BUILD)
#error "R3D_FLOOR_TIERS is a counting-only switch of tools/microbench/phase_floor.hip"
#endif
  return true || c;
#elif defined(__HIP_DEVICE_COMPILE__) && defined(R3D_REPRODUCIBLE)
  return false && c;
#elif defined(__HIP_DEVICE_COMPILE__)
  return __all(c) != 0;
#else
  return c;
#endif
}

// True if the condition holds in ANY active lane (on the host: in this one).  Only ever used to skip code that no
// lane needs: what a lane computes does not depend on it, so the reproducible build keeps it.
R3D_HD bool any_lanes(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(c) != 0ull;
#else
  return c;
#endif
}

// A per-lane truth value held as the wave's 64-bit mask in scalar registers (on the host: a bool).
// Conditions that are only ever combined with one another and then used to select -- the tetra face
// search forms a hundred of them per move -- cost one scalar instruction per AND / OR in this form; as
// C++ bools the compiler turned the mixed selects into exec-mask branches, fifteen scalar instructions
// per test (and scalar issue is not free: a hundred gratuitous s_add per move cost the NSCP launch 4 %).
// lm(c): the mask of the lanes where c holds; lm_lane(m): this lane's bit, as a select condition
// (v_cndmask takes the scalar pair directly); m must come from lm() under the same exec mask.
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned long long LaneMask;
R3D_HD LaneMask lm(bool c) { return __builtin_amdgcn_ballot_w64(c); }
R3D_HD bool lm_lane(LaneMask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
R3D_HD LaneMask lm_andnot(LaneMask a, LaneMask b) { return a & ~b; }
R3D_HD LaneMask lm_all() { return ~0ull; }
#else
typedef bool LaneMask;
R3D_HD LaneMask lm(bool c) { return c; }
R3D_HD bool lm_lane(LaneMask m) { return m; }
R3D_HD LaneMask lm_andnot(LaneMask a, LaneMask b) { return a && !b; }
R3D_HD LaneMask lm_all() { return true; }
#endif

// arcsin for |x| <= 0.5: x + x t P(t) / Q(t), t = x^2 -- the classical rational
// approximation (fdlibm e_asin.c, error below one ulp on this interval).  Used where a small
// angle is known by its sine and the sign of its cosine (the arc length of a tetra leg):
// a third of the instructions of the general atan2.
R3D_HD double asin_small(double x) {
  const double t = x * x;
  if (all_lanes(t <= 0.00390625)) {   // |x| <= 1/16 (a leg of a fraction of a degree, the usual case in a
    // tetrahedral grid): the Maclaurin series through x^13 (next term 0.014 x^14: 4e-19 x)
    double s = 231.0 / 13312.0;
    s = __builtin_fma(s, t, 63.0 / 2816.0);
    s = __builtin_fma(s, t, 35.0 / 1152.0);
    s = __builtin_fma(s, t, 5.0 / 112.0);
    s = __builtin_fma(s, t, 3.0 / 40.0);
    s = __builtin_fma(s, t, 1.0 / 6.0);
    return __builtin_fma(x * t, s, x);
  }
  const double p = t * (1.66666666666666657415e-01 + t * (-3.25565818622400915405e-01 + t * (2.01212532134862925881e-01 +
                   t * (-4.00555345006794114027e-02 + t * (7.91534994289814532176e-04 + t * 3.47933107596021167570e-05)))));
  const double q = 1.0 + t * (-2.40339491173441421878e+00 + t * (2.02094576023350569471e+00 +
                   t * (-6.88283971605453293030e-01 + t * 7.70381505559019352791e-02)));
  return x + x * (p * frcp(q));   // (q within [0.6, 1])
}

// ---- lean elementary functions for the hot path ------------------------------------------------
// The device library's exp / log / sincos carry argument screening, table look-ups and special-case
// repair that the traversal never needs (arguments are finite, of moderate size, and a NaN may
// simply propagate); with the loop's constants re-materialised on every iteration they cost 45-120
// instructions a call, and the kernel is bound by instruction issue.  These versions are plain
// argument reduction + one polynomial; errors stay below 1-2 ulp, far inside the 1e-9 the parity
// tests allow.
//
// exp(x), any finite x of moderate size (the attenuation exponent -pi f t / Q): x = k ln2 + r,
// |r| <= ln2 / 2, e^r by its Taylor polynomial of degree 13 (|r|^14 / 14! < 5e-18), scaled by 2^k.
R3D_HD double exp_lean(double x) {
  const double k = rint(x * 1.4426950408889634074);
  double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);   // ln2 in two parts
  r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;   // 1/13!
  p = __builtin_fma(p, r, 1.0 / 479001600.0);
  p = __builtin_fma(p, r, 1.0 / 39916800.0);
  p = __builtin_fma(p, r, 1.0 / 3628800.0);
  p = __builtin_fma(p, r, 1.0 / 362880.0);
  p = __builtin_fma(p, r, 1.0 / 40320.0);
  p = __builtin_fma(p, r, 1.0 / 5040.0);
  p = __builtin_fma(p, r, 1.0 / 720.0);
  p = __builtin_fma(p, r, 1.0 / 120.0);
  p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return ldexp(p, (int)k);
}
// log(x) for finite x > 0 (no zero / negative / subnormal handling: the callers' arguments are
// uniforms in (0, 1] and ratios near 1): x = 2^e m, m in [sqrt(1/2), sqrt(2)), then the classical
// kernel  log m = f - (hfsq - s (hfsq + R)),  f = m - 1, s = f / (2 + f)  (fdlibm e_log.c
// coefficients, error below one ulp).
R3D_HD double log_lean(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const long long bits = __double_as_longlong(x);
#else
  long long bits;
  __builtin_memcpy(&bits, &x, 8);
#endif
  int e = (int)((bits >> 52) & 0x7FF) - 1023;
  long long mant = bits & 0x000FFFFFFFFFFFFFll;
  const bool upper = mant >= 0x6A09E667F3BCDll;          // m >= sqrt(2): take it as m / 2, e + 1
  e += upper ? 1 : 0;
  mant |= upper ? 0x3FE0000000000000ll : 0x3FF0000000000000ll;
#if defined(__HIP_DEVICE_COMPILE__)
  const double m = __longlong_as_double(mant);
#else
  double m;
  __builtin_memcpy(&m, &mant, 8);
#endif
  const double f = m - 1.0;
  const double s_ = f * frcp(2.0 + f);   // (2 + f lies in [1.7, 2.42])
  const double z = s_ * s_;
  double r = 1.479819860511658591e-01;
  r = __builtin_fma(r, z, 1.531383769920937332e-01);
  r = __builtin_fma(r, z, 1.818357216161805012e-01);
  r = __builtin_fma(r, z, 2.222219843214978396e-01);
  r = __builtin_fma(r, z, 2.857142874366239149e-01);
  r = __builtin_fma(r, z, 3.999999999940941908e-01);
  r = __builtin_fma(r, z, 6.666666666666735130e-01);
  r *= z;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  return dk * 6.93147180369123816490e-01 - ((hfsq - (s_ * (hfsq + r) + dk * 1.90821492927058770002e-10)) - f);
}
// atanh(y) for |y| < 1 -- the travel time along an arc is a difference of two, folded into one
// (media.cpp:487-488, :877-957).  By the odd series y + y^3/3 + ..., as long as every lane of the wave
// allows: through y^13 for |y| <= 1/16 (legs of a fraction of a degree: a tetrahedral grid), through
// y^27 for |y| <= 0.268; lanes between 1/4 and 1/2 (long legs in thick shells) first halve the
// argument, atanh y = 2 atanh(y / (1 + sqrt(1 - y^2))) -- with selects, so that a wave holding both
// kinds runs ONE series; only beyond 1/2 (rare) the logarithm.  Errors below 1e-17 relative.
R3D_HD double atanh_lean(double y) {
  double t = y * y;
  if (all_lanes(t <= 0.00390625)) {
    double q = 1.0 / 13.0;
    q = __builtin_fma(q, t, 1.0 / 11.0);
    q = __builtin_fma(q, t, 1.0 / 9.0);
    q = __builtin_fma(q, t, 1.0 / 7.0);
    q = __builtin_fma(q, t, 1.0 / 5.0);
    q = __builtin_fma(q, t, 1.0 / 3.0);
    return __builtin_fma(y * t, q, y);
  }
  double z = y, scale = 1.0;
  if (!all_lanes(t <= 0.0625)) {
    const bool big = t > 0.0625;
    const double zz = y * frcp(1.0 + fsqrt(1.0 - t));
    z = big ? zz : y, scale = big ? 2.0 : 1.0;
    t = z * z;
  }
  double p = 1.0 / 27.0;
  p = __builtin_fma(p, t, 1.0 / 25.0);
  p = __builtin_fma(p, t, 1.0 / 23.0);
  p = __builtin_fma(p, t, 1.0 / 21.0);
  p = __builtin_fma(p, t, 1.0 / 19.0);
  p = __builtin_fma(p, t, 1.0 / 17.0);
  p = __builtin_fma(p, t, 1.0 / 15.0);
  p = __builtin_fma(p, t, 1.0 / 13.0);
  p = __builtin_fma(p, t, 1.0 / 11.0);
  p = __builtin_fma(p, t, 1.0 / 9.0);
  p = __builtin_fma(p, t, 1.0 / 7.0);
  p = __builtin_fma(p, t, 1.0 / 5.0);
  p = __builtin_fma(p, t, 1.0 / 3.0);
  double r = scale * __builtin_fma(z * t, p, z);
  if (!(fabs(y) <= 0.5))   // (log_lean works on the bits: a NaN, or an argument outside (-1, 1), is passed on as such)
    r = (fabs(y) < 1.0) ? 0.5 * log_lean((1.0 + y) / (1.0 - y)) : y * pos_inf();
  return r;
}
// sin and cos for |x| <= pi/4: the classical kernels (fdlibm k_sin.c / k_cos.c coefficients, error
// below one ulp on this interval); no argument reduction.
R3D_HD void sincos_small(double x, double* s, double* c) {
  const double z = x * x;
  double ps = 1.58969099521155010221e-10;
  ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
  ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
  ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
  ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
  ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
  *s = __builtin_fma(x * z, ps, x);
  double pc = -1.13596475577881948265e-11;
  pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
  pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
  pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
  pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
  pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
  *c = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
}

// sin and cos of a rotation angle up to pi either way (a scatter leg's angle on its arc): the small-
// argument kernels on the angle itself, on its half or on its quarter, doubled back once or twice --
// chosen per lane with selects, so that a wave holding short and long legs evaluates ONE pair of
// polynomials.  Beyond pi (never seen): the library's.
R3D_HD void rotation(double x, double* s, double* c) {
  const double ax = fabs(x);
  if (all_lanes(ax <= 0.78539816339744830962)) {
    sincos_small(x, s, c);
    return;
  }
  const bool two = ax > 0.78539816339744830962, four = ax > 1.57079632679489661923;
  double sh, ch;
  sincos_small(x * (four ? 0.25 : two ? 0.5 : 1.0), &sh, &ch);
  double s2 = 2.0 * sh * ch, c2 = 1.0 - 2.0 * sh * sh;
  sh = two ? s2 : sh, ch = two ? c2 : ch;
  s2 = 2.0 * sh * ch, c2 = 1.0 - 2.0 * sh * sh;
  sh = four ? s2 : sh, ch = four ? c2 : ch;
  *s = sh, *c = ch;
  if (!(ax <= 3.14159265358979323846)) sincos(x, s, c);
}

// The angle in [-pi, pi] whose sine and cosine are s, c (unit to rounding): atan2(s, c) without its
// ~190 instructions.  Where every lane of the wave is within 30 degrees of zero (a leg rarely spans
// more) the small arcsine; else the direction is referred to the nearest of 0, +-60, +-120, 180
// degrees -- residual within +-30 degrees, its sine by the subtraction formula -- and the same arcsine
// serves all lanes at once.
R3D_HD double angle_from_sincos(double s, double c) {
  if (all_lanes(c > 0 && fabs(s) <= 0.5)) return asin_small(s);
  const double k866 = 0.86602540378443864676;
  const bool near0 = c >= k866, front = c >= 0.0, back = c >= -k866;
  const double cf = near0 ? 1.0 : front ? 0.5 : back ? -0.5 : -1.0;
  const double sf = copysign((near0 || !back) ? 0.0 : k866, s);
  const double phi = copysign(near0 ? 0.0 : front ? 1.04719755119659774615 : back ? 2.09439510239319549231
                                                                             : 3.14159265358979323846, s);
  return phi + asin_small(s * cf - c * sf);
}

struct V3 {
  double x, y, z;
};
R3D_HD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
R3D_HD V3 v3(const double* a) { return V3{a[0], a[1], a[2]}; }
R3D_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
R3D_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
R3D_HD V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
R3D_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
R3D_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
R3D_HD V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
R3D_HD double mag2(V3 a) { return dot(a, a); }
R3D_HD double mag(V3 a) { return fsqrt(mag2(a)); }
R3D_HD bool is_zero(V3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
R3D_HD V3 unit(V3 a) {
  double s = frsqrt(mag2(a));
  return s * a;
}
R3D_HD V3 unit_else(V3 a, V3 fallback) {  // reference geom_r3.hpp:127-132
  const double m2 = mag2(a);
  if (m2 == 0) return fallback;
  return frsqrt(m2) * a;
}

// The reference stores a direction as theta = acos(z), phi = atan2(y, x) and
// regenerates (sin th cos ph, sin th sin ph, cos th) from them.  This is that
// round trip without the trigonometry: z is kept, the horizontal part is
// rescaled to sqrt(1 - z^2).  `v` must already be (very nearly) unit length.
R3D_HD V3 through_angles(V3 v) {
  double h2 = v.x * v.x + v.y * v.y;
  double st = fsqrt(fmax(0.0, 1.0 - v.z * v.z));
  if (h2 == 0) return v3(st, 0.0, v.z);  // atan2(0,0) = 0
  double s = st * frsqrt(h2);
  return v3(s * v.x, s * v.y, v.z);
}

// theta^ and phi^ at unit direction d (reference OrthoAxes E1, E2,
// geom_r3.cpp:222-224).
R3D_HD void sph_basis(V3 d, V3& th_hat, V3& ph_hat) {
  const double h2 = d.x * d.x + d.y * d.y;
  double ih = frsqrt(h2);
  double cp = d.x * ih;
  // (a ray along the pole: phi = atan2(0, 0) = 0.  Rare -- served under a vote of the wave, so that the lanes of every
  //  other batch do not pay the selects; the values of the other lanes are the same either way)
  if (any_lanes(h2 == 0)) {
    if (h2 == 0) ih = 0.0, cp = 1.0;
  }
  const double st = h2 * ih;     // sqrt(h2)
  const double sp = d.y * ih;
  th_hat = v3(d.z * cp, d.z * sp, -st);
  ph_hat = v3(-sp, cp, 0.0);
}

// Unit vector perpendicular to `self`, in the plane of self and other, on
// other's side (reference geom_r3.cpp:146-171).
R3D_HD V3 in_plane_unit_perp(V3 self, V3 other) {
  V3 mp = cross(self, other);
  if (is_zero(mp)) {
    mp = cross(self, v3(1, 0, 0));
    if (is_zero(mp)) mp = cross(self, v3(0, 1, 0));
  }
  mp = unit(mp);
  return cross(mp, self);   // (mp is unit and normal to the unit `self`: unit already)
}

// ---- complex numbers (the reference's Complex = std::complex<Real>,
//      complex.hpp:39-56) ---------------------------------------------------
struct Cx {
  double re, im;
};
R3D_HD Cx cx(double re, double im = 0.0) { return Cx{re, im}; }
R3D_HD Cx operator+(Cx a, Cx b) { return cx(a.re + b.re, a.im + b.im); }
R3D_HD Cx operator-(Cx a, Cx b) { return cx(a.re - b.re, a.im - b.im); }
R3D_HD Cx operator-(Cx a) { return cx(-a.re, -a.im); }
R3D_HD Cx operator*(Cx a, Cx b) { return cx(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
R3D_HD Cx operator*(double s, Cx a) { return cx(s * a.re, s * a.im); }
R3D_HD Cx operator*(Cx a, double s) { return cx(s * a.re, s * a.im); }
R3D_HD Cx operator/(Cx a, double s) { return cx(a.re / s, a.im / s); }
R3D_HD Cx operator+(double s, Cx a) { return cx(s + a.re, a.im); }
R3D_HD Cx operator-(double s, Cx a) { return cx(s - a.re, -a.im); }
R3D_HD Cx operator/(Cx a, Cx b) {
  double den = b.re * b.re + b.im * b.im;
  return cx((a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den);
}
R3D_HD double norm(Cx a) { return a.re * a.re + a.im * a.im; }  // squared modulus
// sqrt of the real number s as a complex number (principal branch):
// post-critical cosines come out purely imaginary (rtcoef.cpp:312-318).
R3D_HD Cx sqrt_real(double s) {
  const double r = fsqrt(fabs(s));   // one root, then placed (not one root per branch)
  return s >= 0 ? cx(r, 0.0) : cx(0.0, r);
}

}  // namespace r3d
#endif
