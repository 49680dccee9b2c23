// r3d_physics.h -- per-history physics of the traversal kernel, one phonon per
// work-item.  Everything here is straight-line scalar fp64 code on registers
// plus table reads; wave-level concerns (refill, counters, LDS staging) live
// in r3d_engine.hip.  Functions are __host__ __device__ so that a test-only
// host build (tests/emul) can single-step the same code on a CPU.
//
// What is computed follows the reference function by function (cited); how it
// is computed is chosen for the GPU:
//   * direction = unit vector, polarisation = angle about it; theta^/phi^ by
//     algebra, not trigonometry (r3d_math.h);
//   * a boundary search returns only (arc length, face); the single "advance
//     by length" that follows serves both the boundary and the scatter branch,
//     so all lanes of a wave run it together (the reference advances inside
//     GetPathToBoundary and again on scatter, media.cpp:562-565, phonons.cpp:608);
//   * tetra arcs: the position on the arc is carried as (sin a, cos a), the
//     travel time uses  ln|tan(a/2+pi/4)| = atanh(sin a)  so one log replaces
//     two log(tan()) pairs (media.cpp:487-488);
//   * plane faces are stored as (n, n.p).
#ifndef R3D_PHYSICS_H_
#define R3D_PHYSICS_H_

#include "r3d_math.h"
#include "r3d_rng.h"
#include "r3d_tables.h"

namespace r3d {

enum { RAY_P = 0, RAY_S = 1 };
enum { CELL_CYL = 0, CELL_TET = 1, CELL_SPH = 2 };
enum { FATE_ALIVE = 0, FATE_LOST = 1, FATE_TIMEOUT = 2, FATE_INVALID = 3 };

struct Phonon {            // reference phonons.hpp:69-126
  double t, path, recent;
  double lamp;             // ln(amplitude): the attenuation exponents -pi f t / Q of the legs, summed -- the
                           // reference multiplies the amplitude by exp() of each (phonons.cpp:62-70); the
                           // exponential is taken where the amplitude is used (a catch, a report line)
  V3 loc, dir;
  double pc, ps;           // cos, sin of the polarisation angle mPol (the angle itself is never needed)
  int32_t type, cell;
  uint32_t moves;
};

struct Exit {              // where the current ray leaves the current cell
  double len;
  int face;
};

R3D_HD double amplitude(const Phonon& p) { return exp_lean(p.lamp); }
R3D_HD double amplitude2(const Phonon& p) { return exp_lean(2.0 * p.lamp); }   // its square: the energy weight

R3D_HD uint32_t face_flags(uint32_t packed, int f) { return (packed >> (8 * f)) & 0xFFu; }

// Particle-motion direction (reference Phonon::DirectionOfMotion,
// phonons.cpp:201-211): along the ray for P, else the S1 axis of the
// (theta, phi, pol) frame.
R3D_HD V3 direction_of_motion(const Phonon& p) {
  if (p.type == RAY_P) return p.dir;
  V3 th, ph;
  sph_basis(p.dir, th, ph);
  return p.pc * th + p.ps * ph;
}
// Polarisation of particle motion `pdom` about direction d: the reference
// stores atan2(pdom.phi^, pdom.theta^) (phonons.cpp:389-391, :462-465) and
// later takes its cosine and sine; this yields those two directly.
R3D_HD void set_pol(Phonon& p, V3 pdom, V3 d) {
  V3 th, ph;
  sph_basis(d, th, ph);
  const double x = dot(pdom, th), y = dot(pdom, ph);
  const double h2 = x * x + y * y;
  const double ih = frsqrt(h2);
  p.pc = x * ih, p.ps = y * ih;
  if (any_lanes(h2 == 0)) {   // atan2(0, 0) = 0 (rare: under a vote of the wave, as in sph_basis)
    if (h2 == 0) p.pc = 1.0, p.ps = 0.0;
  }
}

// ===================================================================== CYL ==
R3D_HD double plane_exit(const double n[3], double d, V3 loc, V3 dir) {
  // reference PlaneFace::LinearRayDistToExit, media_cellface.cpp:262-324
  double d_sh = d - (n[0] * loc.x + n[1] * loc.y + n[2] * loc.z);
  double d_fact = n[0] * dir.x + n[1] * dir.y + n[2] * dir.z;
  if (d_fact < 0) return pos_inf();
  if (d_fact == 0) return d_sh < 0 ? -pos_inf() : pos_inf();
  return d_sh * frcp(d_fact);   // (0 < d_fact <= 1)
}
R3D_HD double cylwall_exit(double rad2, V3 loc, V3 dir) {
  // reference CylinderFace::LinearRayDistToExit, media_cellface.cpp:531-562
  double A = dir.x * dir.x + dir.y * dir.y;
  double C = loc.x * loc.x + loc.y * loc.y - rad2;
  if (A == 0) return C <= 0 ? pos_inf() : -pos_inf();
  double B = 2 * (loc.x * dir.x + loc.y * dir.y);
  double urad = B * B - 4 * A * C;
  if (urad < 0) return -pos_inf();
  return (fsqrt(urad) - B) * frcp(2 * A);   // (A > 0)
}
// reference RCUCylinder::GetPathToBoundary, media.cpp:236-330.  Face ids:
// 0 top, 1 bottom, 2 lateral wall (phonon is lost there).
R3D_HD Exit cyl_exit(const CellCyl& c, double wall_rad2, const Phonon& p) {
  double dl = cylwall_exit(wall_rad2, p.loc, p.dir);
  double dt = plane_exit(c.n[0], c.d[0], p.loc, p.dir);
  double db = plane_exit(c.n[1], c.d[1], p.loc, p.dir);
  if (dl < 0) dl = 0;
  if (dt < 0) dt = 0;
  if (db < 0) db = 0;
  Exit e{dl, 2};
  if (dt < e.len) e.len = dt, e.face = 0;
  if (db < e.len) e.len = db, e.face = 1;
  return e;
}
// reference RCUCylinder::AdvanceLength + Phonon::Move, media.cpp:208-222,
// phonons.cpp:62-70
R3D_HD void cyl_advance(const CellCyl& c, Phonon& p, double len) {
  double time = len * frcp(c.v);
  p.path += len, p.t += time, p.recent += time;
  p.loc = p.loc + len * p.dir;
  p.lamp += c.att * time;
  p.moves += 1;
}

// ===================================================================== TET ==
// Frame in which the ray through a linear-velocity cell is a circle about the
// origin of the (x,z) plane (reference CoordinateTransformation,
// media.hpp:560-587): rows v1,v2,v3; `trans` is the arc centre in the rotated
// frame; the phonon sits at angle a0 measured from +z towards +x, always in
// [-pi/2, pi/2] (cos a0 = t.v1 >= 0); velocity is |g| R cos a along the arc.
// Only the two in-plane axes and the arc centre in model coordinates are kept:
// a face's trace in the arc plane is n.v1, n.v3 and its distance (n.p - n.centre).
//
// ANGLES ARE NEVER FORMED.  Every angle the reference compares
// (media_cellface.cpp:333-426, :767-794, media.cpp:542-559) is either +-inf or
// lies in (-pi/2, pi/2), where sin is monotone; so each is represented by its
// sine (with +-inf kept as +-inf) and all orderings carry over.  Only the arc
// length of the chosen exit needs an angle (one atan2 of the sine / cosine of
// the difference), and the end point of a boundary leg is the exit point
// itself, so its sine / cosine are already known.
struct TetArc {
  V3 v1, v3;          // in-plane axes: v1 = component of the ray direction normal to grad v, v3 = unit grad v
  V3 center;          // arc centre (model coordinates); it lies on the plane where v = 0
  double R, s0, c0;   // radius, sine / cosine of the start angle
};
R3D_HD TetArc tet_arc(const CellTet& c, const Phonon& p) {
  TetArc A;
  V3 g = v3(c.g);          // (the record of this cell for the phonon's ray type)
  double vel = dot(p.loc, g) + c.v0;
  // w2 = g x dir, w1 = w2 x g (the part of dir normal to g, times |g|^2): |w1| = |w2| |g|, so with
  // iw = 1 / |w2| everything follows from ONE reciprocal square root and no division:
  //   v1 = w1 / (|w2| |g|),   t.v1 = |w2| / |g|,   R = v / (|g| t.v1) = v / |w2|   (media.hpp:568-569)
  V3 w2 = cross(g, p.dir), w1 = cross(w2, g);
  const double m2 = mag2(w2);
  const double iw = frsqrt(m2);
  A.v1 = (iw * c.inv_gmag) * w1, A.v3 = c.inv_gmag * g;
  const double txp = (m2 * iw) * c.inv_gmag, tzp = dot(p.dir, A.v3);
  A.R = vel * iw;
  // In the rotated frame the phonon sits at R (-tz', 0, tx') from the centre
  // (media.hpp:574-580), i.e. centre = loc + R tz' v1 - R tx' v3; the direction is a unit vector in
  // the (v1, v3) plane, so (-tz', tx') ARE the sine and cosine of its angle on the circle (the
  // velocity, hence R, is positive).
  A.center = p.loc + ((A.R * tzp) * A.v1 + (-A.R * txp) * A.v3);
  A.s0 = -tzp, A.c0 = txp;
  return A;
}
// One plane against the arc circle, in sine space (reference
// PlaneFace::GetCircArcDistToFace, media_cellface.cpp:333-426): the arc is
// outside the face between the exit angle bis - q and the entry angle bis + q,
// bis = direction of the in-plane normal, cos q = (centre-to-trace distance)/R.
constexpr double kTetBig = 2.0;   // stands for an infinite angle among sines (see tet_face_arc)
struct Gcad {
  double entry_lo;            // sine of (entry - 1e-10 rad), or +-inf: the Inside() test's lower bound
  double exit, half;          // sines of the exit and bisector angles, or +-inf
  double exit_cos;            // cosine of the exit angle where that is finite
  LaneMask continuous;        // the inside region is ONE interval [entry, exit] (else: two pieces)
};
R3D_HD Gcad tet_face_arc(const double n[3], double dplane, const TetArc& A, double inv_R) {
  // (written with selects and lane masks: this runs four times per iteration for every lane, and
  //  branches here cost more than the arithmetic)
  // (the reference's +-infinity angles are carried as sines of +-2 here: every real sine lies in
  //  [-1, 1] to rounding, so all orderings carry over, and +-2 is an inline constant of the vector
  //  unit where an infinity is a literal that costs two moves each time it is selected)
  const double inf = kTetBig;
  V3 nn = v3(n);
  const double rx = dot(nn, A.v1), rz = dot(nn, A.v3);      // in-plane components of the face normal
  const double ir = frsqrt(rx * rx + rz * rz);
  const double sb = rx * ir, cb = rz * ir;                  // sin, cos of the bisector angle
  const double ratio = ((dplane - dot(nn, A.center)) * ir) * inv_R;   // cos q
  const LaneMask FRONT = lm(cb > 0);            // bisector within (-pi/2, pi/2)
  const LaneMask CROSS = lm(ratio < 1) & lm(ratio > -1);
  const LaneMask ALL_IN = lm(ratio >= 1), ALL_OUT = lm(ratio <= -1);
  const bool front = lm_lane(FRONT), crosses = lm_lane(CROSS);
  const double sq = fsqrt(fmax(0.0, 1.0 - ratio * ratio));   // sin q > 0 when the circle crosses the plane
  const double se = sb * ratio + cb * sq, ce = cb * ratio - sb * sq;   // entry = bis + q
  const double sx = sb * ratio - cb * sq, cx = cb * ratio + sb * sq;   // exit  = bis - q
  // an angle is inside (-pi/2, pi/2) iff its cosine is positive; which infinity
  // replaces it otherwise depends on the side the bisector is on
  double entry = (ce > 0) ? se : (front ? inf : -inf);
  double exit = (cx > 0) ? sx : (front ? -inf : inf);
  const bool bis_nan = !(cb == cb);     // NaN bisector (the reference exit(1)s): propagate
  entry = bis_nan ? cb : entry, exit = bis_nan ? cb : exit;
  // ratio outside (-1, 1) or NaN: the reference's defaults, then its two overrides
  entry = crosses ? entry : 0.0, exit = crosses ? exit : 0.0;
  // (the slack below applies to real entry angles only: where the entry is a sentinel, ce <= 0)
  double entry_cos = crosses ? fmax(ce, 0.0) : 1.0;
  Gcad g;
  g.exit_cos = crosses ? cx : 1.0;
  // continuous = crosses ? !front : true, then false where all_out (which excludes crosses)
  g.continuous = lm_andnot(lm_andnot(lm_all(), FRONT & CROSS), ALL_OUT);
  g.half = front ? sb : inf;
  const bool all_in = lm_lane(ALL_IN), all_out = lm_lane(ALL_OUT);
  entry = all_in ? -inf : (all_out ? inf : entry);
  exit = all_in ? inf : (all_out ? -inf : exit);
  g.half = all_out ? -inf : g.half;
  g.exit = exit;
  // the reference's 1e-10 rad of slack on the entry side becomes 1e-10 cos(entry) in sine space
  g.entry_lo = entry - 0.0000000001 * entry_cos;
  return g;
}
// reference Tetra::GetPathToBoundary, media.cpp:518-567 (search part).
// Result: face, and the exit point on the circle as (sin, cos); len is filled
// in by tet_exit_length().
struct TetExit {
  double s, c;      // sine / cosine of the exit angle (s may be +-inf)
  int face;
};
R3D_HD TetExit tet_exit(const CellTet& c, const TetArc& A) {
  const double inv_R = frcp(A.R);
  Gcad rv[4];
#pragma unroll
  for (int i = 0; i < 4; i++) rv[i] = tet_face_arc(c.n[i], c.d[i], A, inv_R);
  TetExit e{kTetBig, 1.0, 0};
  const double ninf = -kTetBig;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    // Is face i's exit angle x inside the forward region of each of the other three faces
    // (GCAD_RetVal::Inside, media_cellface.cpp:767-782)?  With A = (x >= entry_lo), B = (x <= exit):
    //   one interval:  A & B;     two pieces:  ((x > -inf) & B) | (A & (x < inf)).
    // For finite x the second is A | B; for x = -inf it is A, and so is A & B (B holds); for x = +inf
    // the exit is never taken (x < e.s fails below), so the test's value is immaterial.  Hence
    //   inside = (A & B) | (two_piece & (x > -inf) & (A | B)),
    // five scalar instructions on lane masks per pair of faces.
    const double x = rv[i].exit;
    const LaneMask GT = lm(x > ninf);
    LaneMask proper = lm_all();
#pragma unroll
    for (int jj = 1; jj < 4; jj++) {
      const Gcad& o = rv[(i + jj) & 3];
      const LaneMask Am = lm(x >= o.entry_lo), Bm = lm(x <= o.exit);
      proper = proper & ((Am & Bm) | (lm_andnot(GT, o.continuous) & (Am | Bm)));
    }
    // an exit behind the phonon (negative arc) is dismissed only beyond the face's bisector
    const LaneMask dismissed = lm(x < A.s0) & lm(A.s0 > rv[i].half);
    const bool take = lm_lane(lm_andnot(proper, dismissed) & lm(x < e.s));
    e.s = take ? x : e.s, e.c = take ? rv[i].exit_cos : e.c, e.face = take ? i : e.face;
  }
  return e;
}
// Arc length to the exit: R (a_exit - a0), the difference taken from its sine
// and cosine (both angles lie in [-pi/2, pi/2]).
R3D_HD double tet_exit_length(const TetArc& A, const TetExit& e) {
  if (!(e.s > -kTetBig && e.s < kTetBig))   // the sentinels back to the reference's +-inf (NaN propagates)
    return e.s >= kTetBig ? pos_inf() : e.s <= -kTetBig ? -pos_inf() : e.s;
  const double sd = e.s * A.c0 - e.c * A.s0, cd = e.c * A.c0 + e.s * A.s0;   // sine, cosine of the arc angle
  return A.R * angle_from_sincos(sd, cd);
}
// reference Tetra::AdvanceLength (media.cpp:442-499) + Phonon::Move.  (s1, c1)
// are the sine / cosine of the end angle: the exit's own for a boundary leg,
// a0 + len/R for a scatter leg.
R3D_HD void tet_advance(const CellTet& c, const TetArc& A, Phonon& p, double len, double s1, double c1) {
  V3 nl = A.center + ((A.R * s1) * A.v1 + (A.R * c1) * A.v3);   // point of the circle at the end angle
  V3 nd = c1 * A.v1 + (-s1) * A.v3;                             // tangent (cos a, 0, -sin a)
  // time = (ln|tan(a1/2+pi/4)| - ln|tan(a0/2+pi/4)|) / |g|,  ln|tan(a/2+pi/4)| = atanh(sin a), and
  // atanh(s1) - atanh(s0) = atanh(y), y = (s1 - s0) / (1 - s0 s1): a leg spans a few degrees, so y
  // is small and the series does (one division, no logarithm)
  const double y = (s1 - A.s0) * frcp(1.0 - A.s0 * s1);
  double time = c.inv_gmag * atanh_lean(y);
  p.path += len, p.t += time, p.recent += time;
  p.loc = nl;
  // (nd = c1 v1 - s1 v3 with v1, v3 orthonormal is unit to rounding; the reference's
  //  renormalisation + (theta, phi) round trip changes it by ~1e-16 and is skipped --
  //  the next leg rebuilds v1 from scratch, so nothing accumulates)
  p.dir = nd;
  p.lamp += c.att * time;
  p.moves += 1;
}

// ------------------------------------------------------------------------------------------------
// THE SAME MOVE IN LOCAL FORM (round 5).  The search above is the reference's own construction: every
// face's entry / exit / bisector angle on the ray circle, measured from the circle's CENTRE (a thousand
// cell sizes away in a gently graded model), then twelve interval tests with the reference's slack and
// dismissal rules -- about 410 vector instructions of a tetra move's ~1000.  What it computes, whenever
// the phonon is inside its cell (or on a face of it, moving in), is the first point at which the arc
// leaves the cell.  That has a closed form in quantities measured FROM THE PHONON:
//
//   with u = unit vector from the arc's centre to the phonon (u = w / |w|, w = g - (g.d) d: the part of
//   grad v normal to the direction d), R = v / |w|, a point of the arc at arc angle th ahead is
//       x(th) = loc + R sin(th) d - R (1 - cos th) u,        tangent  d(th) = cos(th) d - sin(th) u,
//   and for a face (n, n.p) with  h = n.p - n.loc  (distance inside),  P = R n.d,  M = R n.u  the arc is
//   on the face where  P sin(th) - M (1 - cos th) = h,  i.e. with t = tan(th / 2):
//       (2 M + h) t^2 - 2 P t + h = 0 ,      disc = P^2 - h (2 M + h) .
//   Of its two roots the one where the arc goes from inside to outside is ALWAYS
//       t_exit = (P - sqrt(disc)) / (2 M + h) = h / (P + sqrt(disc))
//   (the derivative of the left side there is -2 sqrt(disc)); the first form is stable for P < 0, the
//   second for P >= 0.  The first exit of the cell is the smallest positive t_exit of the four faces.
//
// No centre, no rotated frame, no angles, no sentinels: ~40 instructions a face, and nothing cancels but
// h itself.  This is not an approximation of the search above -- both are exact formulations, good to
// rounding -- but the two can DECIDE differently where the reference's rules are not the geometry's:
// a start outside the cell beyond the 1e-10 rad slack (media_cellface.cpp:768), exits behind the phonon
// that are dismissed or not by the bisector rule (media.cpp:550-552), angles clipped at the +-90 degree
// window, ties between faces that meet in an edge, tangent faces.  So the local form CERTIFIES its
// answer: every quantity a rule could hinge on must be away from its threshold by a margin (1e-8 in
// tan(th/2) -- rounding is 1e-15) or the lane takes the reference's construction above, unchanged.
// tests/test_face_filter.py runs both on 1e7 random and 1e5 adversarial starts (edges, vertices,
// shallow angles, retrograde micro-steps): wherever the local form certifies, the two agree.
//
// Certified means (why each suffices is argued in DESIGN.md section 4, "the local tetra move"):
//   (1) every face: inside (h >= 0), or outside by so little while moving in (h >= 1e-11 P, P = R n.d < 0)
//       that its entry lies within 1e-11 rad ahead -- a phonon that has just come through that face sits
//       within 1e-16 R of it, either side -- which is inside the reference's slack;
//   (2) every face: |disc| >= 1e-10 (R^2 + |loc|^2).  sqrt(disc) / R is the sine of the angle at which the ray CIRCLE meets
//       the face's plane (negative disc: it does not) times the length of the face normal's part in the arc's
//       plane, and the reference's entry and exit angles are good to ~1e-16 over that product (its plane
//       offset n.p - n.centre and its in-plane normal are each good to 1e-16 absolute; measured,
//       tests/test_face_filter.py; of |loc| + R where the cell lies further from the origin than the arc's centre
//       from the cell): held to 1e-5, its angles are good to 1e-11 -- a thousandth of the margins
//       below -- and "does the circle cross this plane at all" is not a matter of rounding.  (For a phonon
//       ON a face this also says it is not grazing: there disc = P^2, so |n.d| >= 1e-5.)
//   (3) every face that is crossed: |t_exit| >= 1e-8 (no exit within 2e-8 rad either side of the phonon);
//   (4) the smallest positive t_exit is <= 1 and the next one is at least 1e-8 larger (no tie);
//   (5) the arc's start and its exit point have cosines >= 1e-3 in the reference's frame (inside the
//       window where velocity is positive, and where its comparisons of sines are well conditioned).
// (tests/emul tallies which condition sent a lane to the reference's construction; nothing in a device build)
#ifndef R3D_LOC_REASON
#define R3D_LOC_REASON(k, mask) ((void)0)
#endif
constexpr double kLocTau = 1e-11, kLocEpsT = 1e-8, kLocEpsC = 1e-10, kLocKappa = 1e-3;
constexpr double kLocNone = 4.0;   // stands for "no exit ahead" among the t_exit (every real one is <= 1)
struct TetLocal {
  V3 U, w;               // loc - centre = R u;  w = g - (g.d) d
  double R, iw, m2, gd;  // radius, 1 / |w|, |w|^2, g.d
};
struct TetFast {
  double t, sn, cs, omc;   // tan(th/2), sin th, cos th, 1 - cos th of the exit
  int face;
  bool ok;                 // certified: else the lane takes tet_arc / tet_exit
};
R3D_HD TetFast tet_fast_exit(const CellTet& c, const Phonon& p, TetLocal& L) {
  const V3 g = v3(c.g);
  const double vel = dot(p.loc, g) + c.v0;
  L.gd = dot(g, p.dir);
  L.w = g - L.gd * p.dir;
  L.m2 = mag2(L.w);
  L.iw = frsqrt(L.m2);
  L.R = vel * L.iw;
  L.U = (L.R * L.iw) * L.w;
  // (R^2 + |loc|^2: the reference's plane offset n.p - n.centre is good to 1e-16 of |centre| <= |loc| + R)
  const double eR2 = kLocEpsC * (L.R * L.R + mag2(p.loc));
  LaneMask good = lm(L.R > 0.0);   // (a velocity <= 0 or a NaN: not this routine's business)
  double tq[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const V3 n = v3(c.n[i]);
    const double h = c.d[i] - dot(n, p.loc);
    const double r = dot(n, p.dir);
    const double M = dot(n, L.U);
    const double P = L.R * r;
    const double k = 2.0 * M + h;
    const double P2 = P * P;
    const double disc = P2 - h * k;
    const LaneMask Dpos = lm(disc > eR2), Dneg = lm(disc < -eR2);
    const double S = disc * frsqrt1(disc);   // (garbage where disc <= 0: never looked at there)
    // t_exit = h / (P + S) for P >= 0, (P - S) / k for P < 0   (q = P + sign(P) S by a copysign: no fewer instructions)
    const bool fwd = P >= 0.0;
    const double q = fwd ? P + S : P - S;
    const double num = fwd ? h : q, den = fwd ? q : k;
    const double t = num * frcp1(den);
    const LaneMask Tpos = lm(t >= kLocEpsT), Tneg = lm(t <= -kLocEpsT);
    const LaneMask inside = lm(h >= 0.0) | lm(h >= kLocTau * P);
    R3D_LOC_REASON(0, lm_andnot(lm_all(), Dpos | Dneg));
    R3D_LOC_REASON(1, lm_andnot(Dpos, Tpos | Tneg));
    R3D_LOC_REASON(2, lm_andnot(lm_all(), inside));
    good = good & (Dneg | (Dpos & (Tpos | Tneg))) & inside;
    tq[i] = lm_lane(Dpos & Tpos) ? t : kLocNone;
  }
  TetFast F;
  const double t01 = fmin(tq[0], tq[1]), t23 = fmin(tq[2], tq[3]);
  const int f01 = tq[1] < tq[0] ? 1 : 0, f23 = tq[3] < tq[2] ? 3 : 2;
  F.t = fmin(t01, t23);
  F.face = t23 < t01 ? f23 : f01;
  const double lim = F.t + kLocEpsT;
  const LaneMask n0 = lm(tq[0] < lim), n1 = lm(tq[1] < lim), n2 = lm(tq[2] < lim), n3 = lm(tq[3] < lim);
  const LaneMask tie = (n0 & n1) | (n2 & n3) | ((n0 | n1) & (n2 | n3));
  R3D_LOC_REASON(3, tie);
  R3D_LOC_REASON(4, lm_andnot(lm_all(), lm(F.t <= 1.0)));
  const double inv = frcp(1.0 + F.t * F.t);
  F.sn = (2.0 * F.t) * inv, F.omc = F.t * F.sn, F.cs = 1.0 - F.omc;
  // the window of the reference's frame: cos a0 = |w| / |g|, sin a0 = -g.d / |g|
  const double c0 = (L.m2 * L.iw) * c.inv_gmag, s0 = -L.gd * c.inv_gmag;
  const double c1 = c0 * F.cs - s0 * F.sn;
  R3D_LOC_REASON(5, lm_andnot(lm_all(), lm(c0 >= kLocKappa) & lm(c1 >= kLocKappa)));
  good = lm_andnot(good, tie) & lm(F.t <= 1.0) & lm(c0 >= kLocKappa) & lm(c1 >= kLocKappa);
  F.ok = lm_lane(good);
  return F;
}
#if defined(__HIP_DEVICE_COMPILE__)
// The same search by FOUR LANES PER HISTORY (the drain of a launch: r3d_pool.h kQuad).  What a drain waits for is the
// serial chain of its longest histories, served three or four lanes to a wave, and alone on a SIMD a wave issues an
// instruction every ~8 cycles whatever its lanes do: the chain is the move's instruction count.  So the idle lanes
// take a face each: the four lanes of a quad hold the SAME history (position, direction, cell record), lane f of the
// quad evaluates face f -- the per-face arithmetic of tet_fast_exit, to the letter --, the four exits are exchanged
// within the quad (data-parallel quad permutes: no LDS), and every lane finishes the search on all four as
// tet_fast_exit does: same values in, same exit out, in all four lanes.  The certificate holds for the quad if it
// holds for each of its faces.
__device__ __forceinline__ double quad_lane_value(double v, int which) {   // lane `which` (0..3) of this lane's quad
  const int lo = __double2loint(v), hi = __double2hiint(v);
  int rl, rh;
  switch (which) {   // (the permutation is part of the instruction)
    case 0: rl = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xf, 0xf, true), rh = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xf, 0xf, true); break;
    case 1: rl = __builtin_amdgcn_mov_dpp(lo, 0x55, 0xf, 0xf, true), rh = __builtin_amdgcn_mov_dpp(hi, 0x55, 0xf, 0xf, true); break;
    case 2: rl = __builtin_amdgcn_mov_dpp(lo, 0xAA, 0xf, 0xf, true), rh = __builtin_amdgcn_mov_dpp(hi, 0xAA, 0xf, 0xf, true); break;
    default: rl = __builtin_amdgcn_mov_dpp(lo, 0xFF, 0xf, 0xf, true), rh = __builtin_amdgcn_mov_dpp(hi, 0xFF, 0xf, 0xf, true); break;
  }
  return __hiloint2double(rh, rl);
}
__device__ __forceinline__ TetFast tet_fast_exit_quad(const CellTet& c, const V3 n, const double dpl, const Phonon& p, TetLocal& L) {
  // (n, dpl: this lane's face of the record, read by the caller with the record -- a lane-indexed read of memory; picked
  //  out of the register copy it becomes a table in scratch)
  const V3 g = v3(c.g);
  const double vel = dot(p.loc, g) + c.v0;
  L.gd = dot(g, p.dir);
  L.w = g - L.gd * p.dir;
  L.m2 = mag2(L.w);
  L.iw = frsqrt(L.m2);
  L.R = vel * L.iw;
  L.U = (L.R * L.iw) * L.w;
  const double eR2 = kLocEpsC * (L.R * L.R + mag2(p.loc));
  double my_tq;
  LaneMask face_good;
  {
    const double h = dpl - dot(n, p.loc);
    const double r = dot(n, p.dir);
    const double M = dot(n, L.U);
    const double P = L.R * r;
    const double k = 2.0 * M + h;
    const double P2 = P * P;
    const double disc = P2 - h * k;
    const LaneMask Dpos = lm(disc > eR2), Dneg = lm(disc < -eR2);
    const double S = disc * frsqrt1(disc);
    const bool fwd = P >= 0.0;
    const double q = fwd ? P + S : P - S;
    const double num = fwd ? h : q, den = fwd ? q : k;
    const double t = num * frcp1(den);
    const LaneMask Tpos = lm(t >= kLocEpsT), Tneg = lm(t <= -kLocEpsT);
    const LaneMask inside = lm(h >= 0.0) | lm(h >= kLocTau * P);
    face_good = (Dneg | (Dpos & (Tpos | Tneg))) & inside;
    my_tq = lm_lane(Dpos & Tpos) ? t : kLocNone;
  }
  // a quad is good if its four faces are: bit 4j <- AND of bits 4j .. 4j+3, then spread back over the quad
  LaneMask gq = face_good & (face_good >> 1);
  gq = gq & (gq >> 2) & 0x1111111111111111ull;
  gq = gq | (gq << 1);
  gq = gq | (gq << 2);
  LaneMask good = lm(L.R > 0.0) & gq;
  double tq[4];
#pragma unroll
  for (int i = 0; i < 4; i++) tq[i] = quad_lane_value(my_tq, i);
  TetFast F;
  const double t01 = fmin(tq[0], tq[1]), t23 = fmin(tq[2], tq[3]);
  const int f01 = tq[1] < tq[0] ? 1 : 0, f23 = tq[3] < tq[2] ? 3 : 2;
  F.t = fmin(t01, t23);
  F.face = t23 < t01 ? f23 : f01;
  const double lim = F.t + kLocEpsT;
  const LaneMask n0 = lm(tq[0] < lim), n1 = lm(tq[1] < lim), n2 = lm(tq[2] < lim), n3 = lm(tq[3] < lim);
  const LaneMask tie = (n0 & n1) | (n2 & n3) | ((n0 | n1) & (n2 | n3));
  const double inv = frcp(1.0 + F.t * F.t);
  F.sn = (2.0 * F.t) * inv, F.omc = F.t * F.sn, F.cs = 1.0 - F.omc;
  const double c0 = (L.m2 * L.iw) * c.inv_gmag, s0 = -L.gd * c.inv_gmag;
  const double c1 = c0 * F.cs - s0 * F.sn;
  good = lm_andnot(good, tie) & lm(F.t <= 1.0) & lm(c0 >= kLocKappa) & lm(c1 >= kLocKappa);
  F.ok = lm_lane(good);
  return F;
}
#endif
// th = 2 atan(t) for the exit's t = tan(th / 2) in (0, 1]; (sn, cs) its sine and cosine.  A tetra leg spans a
// fraction of a degree in a gently graded model: the Maclaurin series through t^13 when every lane has
// t <= 1/16 (next term t^14 / 15: 1e-18 relative); else the general routine on (sn, cs).
R3D_HD double two_atan(double t, double sn, double cs) {
  if (all_lanes(t <= 0.0625)) {
    const double z = t * t;
    double s = 1.0 / 13.0;
    s = __builtin_fma(s, z, -1.0 / 11.0);
    s = __builtin_fma(s, z, 1.0 / 9.0);
    s = __builtin_fma(s, z, -1.0 / 7.0);
    s = __builtin_fma(s, z, 1.0 / 5.0);
    s = __builtin_fma(s, z, -1.0 / 3.0);
    const double a = __builtin_fma(t * z, s, t);
    return a + a;
  }
  return angle_from_sincos(sn, cs);
}
// reference Tetra::AdvanceLength (media.cpp:442-499) + Phonon::Move, in the local form: the leg ends at arc
// angle th ahead, given by (sn, cs, omc) = (sin th, cos th, 1 - cos th).  Travel time as in tet_advance:
// atanh(s1) - atanh(s0) = atanh((s1 - s0) / (1 - s0 s1)) with s = sin(a) = -g.d / |g|, and
// s1 - s0 = (g.d (1 - cos th) + |w| sin th) / |g| straight from the rotation (no difference of nearby sines).
R3D_HD void tet_advance_local(const CellTet& c, const TetLocal& L, Phonon& p, double len, double sn, double cs,
                              double omc) {
  const V3 nl = p.loc + ((L.R * sn) * p.dir + (-omc) * L.U);
  const V3 nd = cs * p.dir + (-sn * L.iw) * L.w;
  const double s0 = -L.gd * c.inv_gmag;
  const double ds = (L.gd * omc + (L.m2 * L.iw) * sn) * c.inv_gmag;
  const double y = ds * frcp(1.0 - s0 * (s0 + ds));
  const double time = c.inv_gmag * atanh_lean(y);
  p.path += len, p.t += time, p.recent += time;
  p.loc = nl;
  p.dir = nd;
  p.lamp += c.att * time;
  p.moves += 1;
}

// The whole move by the reference's construction, for the lanes whose local form did not certify: search,
// free path, advance (what step_move did for every lane until round 5).  A function of its own, CALLED: at 168
// registers the kernel holds this code with nothing to spare, and inlined beside the local form it put
// twenty of the kernel's long-lived values in scratch memory in every phase; called, it has its own
// registers, and the caller's are saved around the call -- on a path one move in ten thousand takes.
struct TetSlowOut {
  Phonon p;
  int32_t face, scatters, fate;
};
R3D_HD TetSlowOut tet_move_reference_inline(const CellTet* cp, Phonon p, double u_free, double mfp) {
  const CellTet c = *cp;
  TetSlowOut o;
  o.fate = FATE_ALIVE, o.scatters = 0;
  const TetArc tarc = tet_arc(c, p);
  const TetExit texit = tet_exit(c, tarc);
  o.face = texit.face;
  const double elen = tet_exit_length(tarc, texit);
  if (elen == pos_inf()) {  // phonons.cpp:595-598
    o.fate = FATE_TIMEOUT, o.p = p;
    return o;
  }
  double scatlen = pos_inf();
  if (!((1.0 - u_free) * mfp >= elen)) scatlen = -log_lean(u_free) * mfp;
  const bool scatters = scatlen < elen;
  const double len = scatters ? scatlen : elen;
  double s1 = texit.s, c1 = texit.c;   // a boundary leg ends at the exit point itself
  if (scatters || !(elen > -pos_inf())) {
    double sd, cd;                      // scatter leg: rotate the start angle by len / R
    rotation(len * frcp(tarc.R), &sd, &cd);
    s1 = tarc.s0 * cd + tarc.c0 * sd, c1 = tarc.c0 * cd - tarc.s0 * sd;
  }
  tet_advance(c, tarc, p, len, s1, c1);
  o.p = p, o.scatters = scatters ? 1 : 0;
  return o;
}

// ===================================================================== SPH ==
R3D_HD double sph_linear_exit(double radius, V3 loc, V3 dir) {
  // reference SphereFace::LinearRayDistToExit, media_cellface.cpp:664-684
  bool outward = radius > 0;
  double midpt = -dot(loc, dir);
  double urad = radius * radius + midpt * midpt - mag2(loc);
  if (urad <= 0) return outward ? -pos_inf() : pos_inf();
  double sq = fsqrt(urad);
  if (outward) return midpt + sq;
  if (midpt <= 0) return pos_inf();
  return midpt - sq;
}
// Ray arc in a v = a r^2 + c shell (reference RayArcAttributes +
// cache_RD2_precompute, raypath.hpp:31-113; SphereShell::GetRayArc_RD2,
// media.cpp:795-861).
//
// ANGLES ARE NOT FORMED where the reference forms them: a position on the arc is carried as the sine
// and cosine of its angle from the arc bottom (the reference takes atan2 for the start, acos for
// each face and sin / cos of the end angle, media_cellface.cpp:717-748, media.cpp:877-957).  Which
// face the arc leaves through follows from the faces' cosines and the sign of the start angle alone
// (see sph_exit); the only inverse function left is the one arc length that is needed, from the
// sine and cosine of the angle DIFFERENCE; the end point of a boundary leg is the exit point itself;
// and the two atanh of the travel time collapse into one.
struct SphArc {
  double radius, rad2;
  V3 center, u1, u3;
  double S2, inv_TwoSQ, CotZetaBy2, timeCoef;   // (1 / TwoSQ of raypath.hpp: it only ever divides)
  double s0, c0;   // sine / cosine of the current location's angle from the arc bottom
  bool straight;   // a == 0: straight rays
};
R3D_HD V3 down_at(V3 ec, V3 loc) {  // ECS.GetDown, ecs.cpp:147-167
  return -unit_else(loc - ec, v3(0, 1, 0));
}
R3D_HD SphArc sph_arc(const CellSph& c, V3 ec, const Phonon& p) {
  SphArc A;
  A.straight = (c.a == 0);
  V3 w3 = down_at(ec, p.loc);
  V3 w2 = unit_else(cross(w3, p.dir), v3(0, 0, 0));
  V3 w1 = cross(w2, w3);
  double sini = dot(w1, p.dir);
  if (sini > 1.0) sini = 1.0;
  double cosi = dot(w3, p.dir);
  double r2 = mag2(p.loc);
  // (the quotients below whose operands are plain positive numbers use frcp; the one whose zero
  //  divisor means "vertical ray, infinite radius" stays a division)
  const double G = sini * fsqrt(r2) * frcp(c.c + c.a * r2);
  const double TwoGA = 2. * G * c.a;
  const double urad = 1. - (2. * TwoGA * G * c.c);
  double bottom = (urad > 1) ? (1. - fsqrt(urad)) * frcp(TwoGA) : 0;
  // (bottom == 0: the vertical ray, whose radius is infinite -- what the division gave)
  A.radius = (bottom == 0) ? c.zero_rad2 * pos_inf() : 0.5 * (c.zero_rad2 * frcp(bottom) - bottom);
  A.rad2 = A.radius * A.radius;
  A.center = p.loc + ((A.radius * cosi) * w1 + (-A.radius * sini) * w3);
  A.u3 = down_at(ec, A.center);
  A.u1 = cross(w2, A.u3);
  if (urad <= 1) {  // straight up or down
    A.center = v3(0, 0, 0);
    A.u3 = v3(0, 0, 0);
    A.u1 = p.dir;
  }
  A.S2 = mag2(A.center);
  double S = fsqrt(A.S2);
  A.inv_TwoSQ = frcp(2 * S * A.radius);
  double cz = (A.S2 + A.radius * A.radius - c.zero_rad2) * A.inv_TwoSQ;
  double isz = frsqrt(1 - cz * cz);     // 1 / sin zeta
  A.CotZetaBy2 = (1 + cz) * isz;
  A.timeCoef = -isz * frcp(c.a * S);
  V3 cl = p.loc - A.center;
  const double y = dot(A.u1, cl), x = dot(A.u3, cl);   // the reference's a0 = atan2(y, x)
  const double h2 = x * x + y * y;
  const double ih = frsqrt(h2);
  A.s0 = (h2 == 0) ? 0.0 : y * ih;   // atan2(0, 0) = 0
  A.c0 = (h2 == 0) ? 1.0 : x * ih;
  // (a shell of uniform velocity: straight rays.  Set here, over whatever the formulas above made of
  //  a == 0, rather than returned early: with two ways out the compiler kept part of the result in
  //  scratch memory, a store and a load through the vector-memory path in every move)
  if (A.straight) {
    A.radius = pos_inf(), A.rad2 = pos_inf(), A.S2 = 0, A.inv_TwoSQ = 0, A.CotZetaBy2 = 0;
    A.timeCoef = 0, A.s0 = 0, A.c0 = 1, A.center = v3(0, 0, 0), A.u1 = p.dir, A.u3 = v3(0, 0, 0);
  }
  return A;
}
// reference SphereShell::GetPathToBoundary, media.cpp:668-757 (search part), with
// SphereFace::CircularArcDistToExit, media_cellface.cpp:717-748, for both faces.
//
// With b = acos(cosq) in [0, pi] the reference has, for the start angle a0 in (-pi, pi]:
//   top    (outward): cosq > 1 ? -inf : (b_top - a0) R
//   bottom (inward) : cosq > 1 ? +inf : a0 >= 0 ? +inf : (-b_bot - a0) R
// and takes the top only if its distance is strictly smaller.  A finite bottom distance is never
// larger than a finite top one (-b_bot <= 0 <= b_top), so: top at -inf wins; else the bottom wins
// whenever it is reachable (its cosine <= 1 and the start angle negative, i.e. s0 < 0); else the top.
// Negative lengths are squashed to 0.  (sx, cx): sine / cosine of the exit angle when `on_arc`.
struct SphExit {
  double len, sx, cx;
  int face;
  bool on_arc;   // the leg ends at the exit point of the arc (not squashed, not a straight ray)
};
R3D_HD SphExit sph_exit(const CellSph& c, const SphArc& A, const Phonon& p) {
  SphExit e;
  e.sx = 0, e.cx = 1, e.on_arc = false;
  if (A.straight || A.S2 == 0) {
    const double dt = sph_linear_exit(c.radius[0], p.loc, p.dir);
    const double db = sph_linear_exit(c.radius[1], p.loc, p.dir);
    e.face = (dt < db) ? 0 : 1;
    e.len = e.face == 0 ? dt : db;
    if (e.len < 0) e.len = 0;
    return e;
  }
  const double base = A.S2 + A.rad2;
  const double cq_t = (base - c.radius[0] * c.radius[0]) * A.inv_TwoSQ;
  const double cq_b = (base - c.radius[1] * c.radius[1]) * A.inv_TwoSQ;
  const bool bottom_reach = !(cq_b > 1.0) && (A.s0 < 0);
  if (cq_t > 1.0) {          // top at -inf: taken, squashed to a zero-length leg
    e.face = 0, e.len = 0;
    return e;
  }
  if (!bottom_reach && !(cq_t >= -1.0)) {   // acos of the top's cosine is NaN: (NaN < +inf) is false
    e.face = 1, e.len = pos_inf();
    return e;
  }
  e.face = bottom_reach ? 1 : 0;
  e.cx = bottom_reach ? cq_b : cq_t;
  const double sq = fsqrt(1.0 - e.cx * e.cx);   // sin acos; NaN beyond [-1, 1], as acos is
  e.sx = bottom_reach ? -sq : sq;
  // arc angle = exit angle - start angle, from its sine and cosine; a top exit reached from a
  // negative start angle lies in (0, 2 pi)
  const double sd = e.sx * A.c0 - e.cx * A.s0, cd = e.cx * A.c0 + e.sx * A.s0;
  double d = angle_from_sincos(sd, cd);
  if (!bottom_reach && A.s0 < 0 && d < 0) d += kPi360;
  e.len = d * A.radius;
  e.on_arc = !(e.len < 0);
  if (e.len < 0) e.len = 0;
  return e;
}
// reference SphereShell::AdvanceLength_* (media.cpp:877-957) + Phonon::Move.  (s1, c1): sine /
// cosine of the end angle on the arc (ignored for straight rays).
R3D_HD void sph_advance(const CellSph& c, const SphArc& A, Phonon& p, double len, double s1, double c1) {
  double time, att_time;
  if (A.straight || A.radius == pos_inf()) {
    V3 nl = p.loc + len * p.dir;
    time = att_time = len / c.c;
    if (!A.straight) {  // vertical ray in a graded shell: analytic time, straight-line attenuation
      double r0 = mag(p.loc), r1 = mag(nl);
      double sqnac = sqrt(-c.a * c.c), sqnaoc = sqrt(-c.a / c.c);
      time = fabs((atanh(sqnaoc * r1) - atanh(sqnaoc * r0)) / sqnac);
    }
    p.loc = nl;
  } else {
    p.loc = A.center + ((A.radius * s1) * A.u1 + (A.radius * c1) * A.u3);
    V3 nd = c1 * A.u1 + (-s1) * A.u3;
    // time = timeCoef (atanh(x1) - atanh(x0)), x = CotZetaBy2 tan(a/2), tan(a/2) = sin a / (1 + cos a);
    // atanh(x1) - atanh(x0) = atanh((x1 - x0) / (1 - x0 x1))
    // with tan(a/2) = s / (1 + c) put over the common denominator:
    //   y = K (s1 (1 + c0) - s0 (1 + c1)) / ((1 + c0)(1 + c1) - K^2 s0 s1),  K = CotZetaBy2 -- one quotient
    const double h0 = 1.0 + A.c0, h1 = 1.0 + c1;
    const double y = (A.CotZetaBy2 * (s1 * h0 - A.s0 * h1)) * frcp(h0 * h1 - (A.CotZetaBy2 * A.CotZetaBy2) * (A.s0 * s1));
    time = att_time = A.timeCoef * atanh_lean(y);
    // (nd = c1 u1 - s1 u3 with u1, u3 orthonormal is unit to rounding; the reference's
    //  renormalisation + (theta, phi) round trip changes it by ~1e-16 and is skipped, as in tet_advance)
    p.dir = nd;
  }
  p.path += len, p.t += time, p.recent += time;
  p.lamp += c.att * att_time;
  p.moves += 1;
}

// ------------------------------------------------------------------------------------------------
// THE SHELL MOVE IN LOCAL FORM (round 5), as the tetra move above.  In a shell with v = a r^2 + c the ray
// is a circle too, and everything the tetra's local form uses carries over with the LOCAL gradient
// g = grad v = 2 a X (X = loc - Earth's centre):  w = g - (g.d) d,  u = w / |w|,  R = v / |w|,
//     x(th) = loc + R sin(th) d - R (1 - cos th) u .
// The faces are spheres about the Earth's centre, and |x(th)|^2 = r^2 + 2 R [sin(th) X.d + (1 - cos th)(R - X.u)]
// is again linear in sin th and 1 - cos th: with (all times 2 R, so that nothing is divided)
//     top    (inside while |x| <= rho_t):  h = rho_t^2 - r^2,   P = 2 R X.d,   M = 2 (X.U - R^2)      (U = R u)
//     bottom (inside while |x| >= rho_b):  h = r^2 - rho_b^2,   P, M with the other sign
// the face is met where  P sin th - M (1 - cos th) = h -- the tetra's equation, and its exit root
//     t_exit = tan(th / 2) = h / (P + S)  for P >= 0,   (P - S) / (2 M + h)  for P < 0,   S = sqrt(P^2 - h (2 M + h)).
// The reference (SphereShell::GetPathToBoundary, media.cpp:668-757; SphereFace::CircularArcDistToExit,
// media_cellface.cpp:717-748) measures both faces' angles from the arc's BOTTOM and takes the top unless the
// bottom's distance is smaller; for a phonon inside its shell that is the first exit ahead.  Where its
// special cases could apply -- a start outside the shell (distances squashed to zero), an arc that never
// reaches the top (taken at minus infinity), an arc tangent to a face, an exit at the phonon's feet, more
// than half a circle ahead, straight and vertical rays -- the lane takes the reference's construction.
// Travel time: v along the arc is v0 + 2 a R [X.d sin th + (R - X.u)(1 - cos th)], and with t = tan(th / 2)
//     time = Int R dth / v = (1 / sqrt(-a c)) atanh( 2 R sqrt(-a c) t / (v0 + 2 a R (X.d) t) )
// (the discriminant of the quadratic under the integral is -4 a c R^2 whatever the ray: every ray circle
// is orthogonal to the sphere v = 0).
struct SphFast {
  double t, sn, cs, omc;   // tan(th/2), sin th, cos th, 1 - cos th of the exit
  double aP;               // a * 2 R X.d: the travel time's denominator is v0 + aP t
  double vel;
  int face;
  bool ok;
};
constexpr double kLocEpsCSph = 1e-10;
constexpr double kLocTmaxSph = 50.0;   // exits up to 177.7 degrees ahead (t = tan(th / 2)); beyond: the reference's construction
R3D_HD SphFast sph_fast_exit(const CellSph& c, V3 ec, const Phonon& p, TetLocal& L) {
  SphFast F;
  const V3 X = p.loc - ec;
  const double r2 = mag2(X);
  F.vel = c.c + c.a * r2;
  const V3 g = (2.0 * c.a) * X;
  L.gd = dot(g, p.dir);
  L.w = g - L.gd * p.dir;
  L.m2 = mag2(L.w);
  L.iw = frsqrt(L.m2);
  L.R = F.vel * L.iw;
  L.U = (L.R * L.iw) * L.w;
  const double twoR = 2.0 * L.R;
  const double Pt = twoR * dot(X, p.dir);
  const double XU = dot(X, L.U), R2 = L.R * L.R;
  const double Mt = 2.0 * (XU - R2);
  F.aP = c.a * Pt;
  // The margins of the certificate, in the reference's own terms (measured: tests/test_face_filter.py).  It places
  // a face's crossing by the cosine (S^2 + R^2 - rho^2) / (2 S R) of its angle from the arc's bottom, S = the
  // distance of the arc's centre from the Earth's: the numerator is good to 1e-16 of its largest term, so the
  // angle is good to 1e-16 (S^2 + R^2 + rho^2) / (2 S R) over its SINE, and 1 - cosine^2 = disc / (2 R S)^2 here
  // (whether the arc reaches the face at all is the same quantity's sign): with |disc| >= 1e-10 (S^2 + R^2 + r^2)^2
  // the reference's angles are good to ~1e-10 (several roundings go into that cosine), a hundredth of the margins
  // on t.  And its arc radius comes from 1 - sqrt(1 + 4 G^2 |a| c), which loses digits as the ray nears the
  // vertical: the lost share is R^2 / zero_rad2, held to 1e6 (rays within ~1e-3 of the vertical take the
  // reference's construction, as the exactly vertical ones must).
  const double S2 = (r2 - 2.0 * XU) + R2;
  const double sc = (S2 + R2) + r2;
  const double eR2 = kLocEpsCSph * (sc * sc);
  // (a < 0: graded shells only -- uniform ones have straight rays, a > 0 is refused when the model is built)
  LaneMask good = lm(L.R > 0.0) & lm(c.a < 0.0) & lm(R2 <= 1e6 * c.zero_rad2);
  double tq[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const double rad = c.radius[i];                       // signed: +top, -bottom
    const double h = i == 0 ? rad * rad - r2 : r2 - rad * rad;
    const double P = i == 0 ? Pt : -Pt, M = i == 0 ? Mt : -Mt;
    const double k = 2.0 * M + h;
    const double disc = P * P - h * k;
    const LaneMask Dpos = lm(disc > eR2), Dneg = lm(disc < -eR2);
    const double S = disc * frsqrt1(disc);
    const bool fwd = P >= 0.0;
    const double q = fwd ? P + S : P - S;
    const double num = fwd ? h : q, den = fwd ? q : k;
    const double t = num * frcp1(den);
    const LaneMask Tpos = lm(t >= kLocEpsT), Tneg = lm(t <= -kLocEpsT);
    const LaneMask inside = lm(h >= 0.0) | lm(h >= kLocTau * P);
    // (the top: an arc that never reaches it is the reference's "minus infinity" case -- not ours)
    good = good & (i == 0 ? Dpos : (Dneg | Dpos)) & (Dneg | (Tpos | Tneg)) & inside;
    tq[i] = lm_lane(Dpos & Tpos) ? t : 2.0 * kLocTmaxSph;
  }
  F.t = fmin(tq[0], tq[1]);
  F.face = tq[1] < tq[0] ? 1 : 0;
  const double other = fmax(tq[0], tq[1]);
  good = good & lm(F.t <= kLocTmaxSph) & lm(other - F.t >= kLocEpsT);
  const double inv = frcp(1.0 + F.t * F.t);
  F.sn = (2.0 * F.t) * inv, F.omc = F.t * F.sn, F.cs = 1.0 - F.omc;
  F.ok = lm_lane(good);
  return F;
}
// reference SphereShell::AdvanceLength_Variant_RD2 (media.cpp:913-957) + Phonon::Move in the local form: the leg
// ends at arc angle th ahead, (sn, cs, omc, t) = (sin th, cos th, 1 - cos th, tan(th / 2)).
R3D_HD void sph_advance_local(const CellSph& c, const TetLocal& L, const SphFast& F, Phonon& p, double len, double sn,
                              double cs, double omc, double t) {
  const V3 nl = p.loc + ((L.R * sn) * p.dir + (-omc) * L.U);
  const V3 nd = cs * p.dir + (-sn * L.iw) * L.w;
  const double nac = -c.a * c.c;
  const double isq = frsqrt(nac);                       // 1 / sqrt(-a c)
  const double y = ((2.0 * L.R) * (nac * isq) * t) * frcp(F.vel + F.aP * t);
  const double time = isq * atanh_lean(y);
  p.path += len, p.t += time, p.recent += time;
  p.loc = nl;
  p.dir = nd;
  p.lamp += c.att * time;
  p.moves += 1;
}

// The whole shell move by the reference's construction, for the lanes whose local form did not certify (what
// step_move did for every lane until round 5): arc, search, free path, advance.
R3D_HD TetSlowOut sph_move_reference_inline(const CellSph* cp, V3 ec, Phonon p, double u_free, double mfp) {
  const CellSph c = *cp;
  TetSlowOut o;
  o.fate = FATE_ALIVE, o.scatters = 0;
  const SphArc sarc = sph_arc(c, ec, p);
  const SphExit sexit = sph_exit(c, sarc, p);
  o.face = sexit.face;
  if (sexit.len == pos_inf()) {  // phonons.cpp:595-598
    o.fate = FATE_TIMEOUT, o.p = p;
    return o;
  }
  double scatlen = pos_inf();
  if (!((1.0 - u_free) * mfp >= sexit.len)) scatlen = -log_lean(u_free) * mfp;
  const bool scatters = scatlen < sexit.len;
  const double len = scatters ? scatlen : sexit.len;
  double s1 = sexit.sx, c1 = sexit.cx;  // a boundary leg ends at the exit point itself
  if (scatters || !sexit.on_arc) {
    double sd, cd;                      // scatter leg (or a squashed one): rotate by len / R
    rotation(len * frcp(sarc.radius), &sd, &cd);
    s1 = sarc.s0 * cd + sarc.c0 * sd, c1 = sarc.c0 * cd - sarc.s0 * sd;
  }
  sph_advance(c, sarc, p, len, s1, c1);
  o.p = p, o.scatters = scatters ? 1 : 0;
  return o;
}

// ============================================================== interfaces ==
// Elastic properties either side of an interface, at the crossing point.
struct Iface {
  V3 normal;            // outward from the current cell
  double vR[2], vT[2];  // reflection side (current cell), transmission side
  double rhoR, rhoT;
  bool has_neighbor;
};

enum { R_P, R_SV, R_SH, T_P, T_SV, T_SH, RT_NUM };

// Reflection / transmission at a first-order discontinuity or the free
// surface: Aki & Richards (1980) eq. 5.40 amplitudes -> energy-flux weights ->
// random outcome -> new type, direction, polarisation.  Reference
// Phonon::Refraction_FullRT (phonons.cpp:429-476), CellFace::GetRTBasis
// (media_cellface.cpp:122-149), RTCoef (rtcoef.cpp:30-588).
//
// The outcome is chosen from weights w_k = rho_k v_k Re(cos_k) |A_k|^2.  Every
// amplitude is a numerator over the same determinant (D for P-SV, a+b for SH),
// and the chooser (rtcoef.cpp:436-475) only compares u * sum(w) with partial
// sums, so the common 1/|D|^2 is dropped: the weights here are |D|^2 times the
// reference's, which selects the same outcome.  A vanishing or non-finite
// determinant gives the reference a NaN total and hence its default choice;
// that case is tested explicitly.
//
// THE SOLVE IN SLOWNESS FORM (round 6).  The reference forms sin_k = v_k p, cos_k = sqrt(1 - sin_k^2)
// (complex beyond the critical angle) and then only ever uses cos_k / v_k; that quotient is the ray's
// VERTICAL SLOWNESS sqrt(1 / v_k^2 - p^2), purely real or purely imaginary, taken here in one step from
// p^2 = sin^2(i) / v_in^2 -- neither the incidence sine nor p itself is ever needed, only their squares
// (every numerator that carries p enters through its squared modulus), and the incident ray's own slowness is
// |cos i| / v_in with cos i = n.d at hand from the geometry: three roots where the reference takes four of
// them and a root for the sine.  The two incidence types of the P-SV system are ONE set of formulas in
// (own type, other type) velocities -- exchanging the P and the S slownesses exchanges E with F and G with
// H and leaves D alone (rtcoef.cpp:107-198 read side by side) -- so the velocities are selected once, ahead of
// the arithmetic, and the four weights put in the reference's order afterwards; the SH system (rtcoef.cpp:
// 207-278) works from the same two S slownesses and is evaluated beside it, so a batch that mixes P, SV and
// SH lanes runs one instruction stream.  The Aki-Richards parameters in the form a = (rho2 - rho1) - d p^2,
// b = rho2 - d p^2, c = rho1 + d p^2, d = 2 (mu2 - mu1) (rtcoef.cpp:336-343 multiplied out).
//
// What comes out beside the weights: the real part of each outgoing ray's vertical slowness and its
// velocity, so that the chosen ray's direction cosine v Re(slowness) needs no second root.
struct RtWeights {
  double w[4];     // chooser order with the entries that are zero by construction left out:
                   //   P / SV incidence: R_P, R_SV, T_P, T_SV;  SH incidence: R_SH, 0, T_SH, 0
  double det2;     // |determinant|^2 the weights are scaled by
  // Re(vertical slowness) and velocity of the four outgoing rays: [0] reflected / [1] transmitted, own type (the
  // incident ray's) and other type
  double zr_own[2], zr_oth[2], v_own[2], v_oth[2];
  double iv_in;    // 1 / the incident ray's velocity
};
// m2 = sin^2(i), acn = |cos i|; intype 0 P, 1 SH, 2 SV.
R3D_HD RtWeights rt_weights_slowness(const Iface& f, double m2, double acn, int intype) {
  const double rho1 = f.rhoR, rho2 = f.rhoT;
  const double a1 = f.vR[0], a2 = f.vT[0], b1 = f.vR[1], b2 = f.vT[1];
  const bool in_p = (intype == 0), sh = (intype == 1), sv = (intype == 2);
  // own: the incident ray's type, oth: the other one (for SH: own = S; what the P-SV block makes of it is not looked at)
  const double own1 = in_p ? a1 : b1, oth1 = in_p ? b1 : a1, own2 = in_p ? a2 : b2, oth2 = in_p ? b2 : a2;
  // (an S velocity is 0 in a fluid: its reciprocal is not a number then, nor is anything below, and the
  //  chooser falls back to the default outcome -- as the reference does with its division by zero)
  const double iv = frcp(own1);
  const double psq = m2 * (iv * iv);        // horizontal slowness squared
  const double x1 = acn * iv;               // the incident ray's own vertical slowness (real)
  const double io1 = frcp(oth1), ix2 = frcp(own2), io2 = frcp(oth2);
  // (sqrt_real, placed with one select: the other component is what is left)
  auto slowness = [&](double iv_k) {
    const double q = iv_k * iv_k - psq, r = fsqrt(fabs(q)), re = q >= 0 ? r : 0.0;
    return cx(re, r - re);
  };
  const Cx y1 = slowness(io1), x2 = slowness(ix2), y2 = slowness(io2);
  R3D_SCHED_FENCE();
  const double mu1 = rho1 * (b1 * b1), mu2 = rho2 * (b2 * b2);
  const double d = 2.0 * (mu2 - mu1), dp = d * psq;
  const double a = (rho2 - rho1) - dp, b = rho2 - dp, c = rho1 + dp;
  // E = b x1 + c x2, F = b y1 + c y2, G = a - d x1 y2, H = a - d x2 y1, D = E F + G H p^2
  const double bx1 = b * x1;
  const Cx cx2 = c * x2;
  const Cx E = cx(bx1 + cx2.re, cx2.im), T1 = cx(bx1 - cx2.re, -cx2.im);
  const Cx F = b * y1 + c * y2;
  const Cx dxy = (d * x1) * y2;
  const Cx G = cx(a - dxy.re, -dxy.im), T2 = cx(a + dxy.re, dxy.im);
  const Cx H = a - d * (x2 * y1);
  const Cx D = E * F + (G * H) * psq;
  const Cx same = T1 * F - (T2 * H) * psq;            // same-type reflection (numerator)
  const Cx conv = (a * b) + (c * d) * (x2 * y2);      // converted reflection, without its factor -2 p v_in x1
  const double four_cn2 = 4.0 * (acn * acn);
  const double k_rs = (rho1 * own1) * acn;            // rho v Re(cos) of the incident ray's own type
  const double k_t = ((rho1 * rho1) * rho2) * four_cn2;
  const double w_rs = k_rs * norm(same);
  // (an SH ray couples to neither converted ray: their weights through a zero factor)
  const double psq_c = sh ? 0.0 : psq;
  const double w_rc = ((rho1 * y1.re) * (four_cn2 * psq_c)) * norm(conv);
  const double w_ts = (k_t * x2.re) * norm(F);
  const double w_tc = ((k_t * y2.re) * psq_c) * norm(H);
  // SH: a = mu1 x1, b = mu2 x2; R = (a - b) / (a + b), T = 2 a / (a + b)
  const double ash = mu1 * x1;
  const Cx bsh = mu2 * x2;
  const double w_rsh = k_rs * norm(cx(ash - bsh.re, bsh.im));
  const double w_tsh = (mu2 * x2.re) * (4.0 * (ash * ash));
  RtWeights o;
  o.det2 = sh ? norm(cx(ash + bsh.re, bsh.im)) : norm(D);
  // (SH lanes: the P-SV block's numbers are finite -- the "other" velocities of an S ray are P velocities -- and left out)
  const double s_rs = sh ? w_rsh : w_rs, s_ts = sh ? w_tsh : w_ts;
  o.w[0] = sv ? w_rc : s_rs, o.w[1] = sv ? s_rs : w_rc, o.w[2] = sv ? w_tc : s_ts, o.w[3] = sv ? s_ts : w_tc;
  o.zr_own[0] = x1, o.zr_own[1] = x2.re, o.zr_oth[0] = y1.re, o.zr_oth[1] = y2.re;
  o.v_own[0] = own1, o.v_own[1] = own2, o.v_oth[0] = oth1, o.v_oth[1] = oth2;
  o.iv_in = iv;
  return o;
}
// The same as the six weights of the reference's table (rtcoef.hpp:79-87: R_P, R_SV, R_SH, T_P, T_SV, T_SH), from the
// incidence sine: w[k] = |det|^2 x the reference's mProb[k].  (The --rtcoef-test mission and the tests; the kernel
// goes through rt_event.)
R3D_HD void rt_weights(const Iface& f, double sini, int intype, double w[RT_NUM], double& det2) {
  const double m2 = sini * sini;
  const RtWeights o = rt_weights_slowness(f, m2, fsqrt(fmax(0.0, 1.0 - m2)), intype);
#pragma unroll
  for (int i = 0; i < RT_NUM; i++) w[i] = 0;
  if (intype == 1) w[R_SH] = o.w[0], w[T_SH] = o.w[2];
  else w[R_P] = o.w[0], w[R_SV] = o.w[1], w[T_P] = o.w[2], w[T_SV] = o.w[3];
  det2 = o.det2;
}

// The event's uniforms (the S-polarisation draw only for S phonons, then the outcome draw: order and
// count of draws are the reference's, rtcoef.cpp:414, :447), PINNED where this is called: they depend
// on nothing that has to be fetched, so a caller draws them while the interface's records are on
// their way -- left to itself the optimiser sinks each generator call down to its use.
R3D_HD void rt_draws(const Phonon& p, Rng& rng, RngKey key, double& u_pol, double& u_out) {
  rng_draw_pair(rng, key, u_pol, u_out);   // (u_pol: the S polarisation kind; not looked at for a P ray)
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(u_pol), "+v"(u_out));
#endif
}

// The event in two halves, so that a caller short of registers can let go of everything but four words between
// them: rt_choose() decides the incident S ray's polarisation kind (ChooseSPolType, rtcoef.cpp:406-422), forms the
// weights and draws the outcome (the chooser of rtcoef.cpp:436-475); rt_apply() turns the chosen ray into the new
// type, direction and polarisation from the SAME phonon and face normal (GetChosenRayDirection /
// GetChosenParticleDOM, rtcoef.cpp:529-588; the tangential part of the direction and the normal of the plane of
// incidence are simply formed again: same inputs, same arithmetic, same values).
//
// The reference forms unit axes fpara (in the plane of incidence, along the face) and fparash (normal to that
// plane) and writes out = sin(o) fpara + (+-cos(o)) fnorm with sin(o) = v_out p.  None of the axes needs
// normalising: sin(i) fpara IS the tangential part of the direction, dt = d - (n.d) n, so out = (v_out / v_in) dt
// +- cos(o) n; and w = n x d = sin(i) fparash serves for the SH fraction -- (pdom.fparash)^2 >= u is
// (pdom.w)^2 >= u |w|^2 -- and for the outgoing particle motion, whose scale the projection onto (theta^, phi^)
// divides away.  Normal incidence (w = 0) takes the reference's substitute axis (geom_r3.cpp:146-171).
struct RtChoice {
  int code;      // bit 0: the outgoing ray is an S ray; bit 1: reflected; bit 2: the incident ray counted as SH
  double kt;     // sin(o) / sin(i) = v_out / v_in
  double cz;     // Re(cos(o)) >= 0: 0 beyond the critical angle
};
// n x d, or the reference's substitute axis where that vanishes; w2 = its squared length (m2 = sin^2(i) unless substituted)
R3D_HD V3 rt_plane_normal(V3 n, V3 d, double m2, double& w2) {
  V3 w = cross(n, d);
  w2 = m2;                                         // |n x d|^2 = |d - (n.d) n|^2
  if (any_lanes(is_zero(w))) {
    if (is_zero(w)) {
      w = cross(n, v3(1, 0, 0));
      if (is_zero(w)) w = cross(n, v3(0, 1, 0));
      w2 = mag2(w);
    }
  }
  return w;
}
// FLAT_FACES (layered models): where the face is horizontal -- its normal (0, 0, +-1) exactly -- the plane of incidence is
// the vertical plane through the ray, so the axis normal to it IS phi^ and the SV axis IS theta^: the SH fraction of an S
// ray's particle motion is sin^2 of its polarisation angle as it stands, and the outgoing polarisation is one of four
// constants (rt_apply) -- no axis, no projection.  Lanes on other faces take the general form, under a vote of the wave.
R3D_HD bool face_is_flat(V3 n) { return n.x == 0.0 && n.y == 0.0; }
template <bool FLAT_FACES = false>
R3D_HD RtChoice rt_choose(const Phonon& p, Iface f, double u_pol, double u_out) {
  const double cn = dot(f.normal, p.dir);
  const double m2 = mag2(p.dir - cn * f.normal);   // sin^2(i)
  bool no_transmit = false;
  if (!f.has_neighbor) {  // free surface: vanishing medium on the far side
    f.rhoT = 0.0, f.vT[0] = f.vT[1] = 1e-12;
    no_transmit = true;
  }
  int intype = 0;  // 0 P, 1 SH, 2 SV
  const bool flat = FLAT_FACES && face_is_flat(f.normal);
  if (p.type == RAY_S) {
    if (FLAT_FACES) intype = (u_pol <= p.ps * p.ps) ? 1 : 2;   // (pdom . phi^)^2
    if (!FLAT_FACES || any_lanes(!flat)) {
      if (!flat) {
        double w2;
        const V3 w = rt_plane_normal(f.normal, p.dir, m2, w2);
        const double sh = dot(direction_of_motion(p), w);
        intype = (u_pol * w2 <= sh * sh) ? 1 : 2;
      }
    }
  }
  R3D_SCHED_FENCE();
  const RtWeights o = rt_weights_slowness(f, m2, fabs(cn), intype);
  R3D_SCHED_FENCE();
  // Choose, rtcoef.cpp:436-475 (the partial sums of the six-entry table with its zero entries left out: the same values)
  const double c0 = o.w[0], c1 = c0 + o.w[1], c2 = c1 + o.w[2], total = c2 + o.w[3];   // (c1 either way round: the same bits)
  const double ran = u_out * total;
  int idx = 3;
  if (ran <= c2) idx = 2;
  if (ran <= c1) idx = 1;
  if (ran <= c0) idx = 0;   // ends on the FIRST entry with ran <= its partial sum
  // GetCoefs' default (rtcoef.cpp:76-97): reflected, same type
  if (total == 0 || (total - total) != 0 || !(o.det2 > 0) || (o.det2 - o.det2) != 0) idx = (intype == 2) ? 1 : 0;
  if (no_transmit && idx >= 2) idx -= 2;  // T_x -> R_x
  const bool reflected = idx < 2;
  // the chosen ray among the four: entries 1 / 3 are the converted ray for P and SH incidence, entries 0 / 2 for SV
  const bool conv = ((idx & 1) != 0) != (intype == 2);
  const double zr = reflected ? (conv ? o.zr_oth[0] : o.zr_own[0]) : (conv ? o.zr_oth[1] : o.zr_own[1]);
  const double vo = reflected ? (conv ? o.v_oth[0] : o.v_own[0]) : (conv ? o.v_oth[1] : o.v_own[1]);
  // outgoing type: entry 0 / 2 is a P ray for P and SV incidence, entry 1 / 3 an S ray; SH stays S
  const bool out_s = (intype == 1) || ((idx & 1) != 0);
  RtChoice ch;
  ch.code = (out_s ? 1 : 0) | (reflected ? 2 : 0) | (intype == 1 ? 4 : 0);
  ch.kt = vo * o.iv_in;
  ch.cz = vo * zr;
  // the reference's clamp of the outgoing sine at 1 (rtcoef.cpp:541): the tangential part then has unit length
  // (only a default choice can pick such a ray)
  if (any_lanes((ch.kt * ch.kt) * m2 > 1.0)) {
    if ((ch.kt * ch.kt) * m2 > 1.0) ch.kt = frsqrt(m2), ch.cz = 0.0;
  }
  return ch;
}
// Returns true if the phonon crossed into the neighbour.
template <bool FLAT_FACES = false>
R3D_HD bool rt_apply(Phonon& p, V3 n, RtChoice ch) {
  const bool out_s = (ch.code & 1) != 0, reflected = (ch.code & 2) != 0, sh = (ch.code & 4) != 0;
  const double cn = dot(n, p.dir);
  const V3 dt = p.dir - cn * n;
  const V3 out = ch.kt * dt + (reflected ? -ch.cz : ch.cz) * n;
  p.type = out_s ? RAY_S : RAY_P;
  if (out_s) {
    const bool flat = FLAT_FACES && face_is_flat(n);
    if (FLAT_FACES) {
      // a horizontal face, normal (0, 0, s): the particle motion of the outgoing S ray is s phi^ (SH), -s theta^ (SV
      // reflected), s theta^ (SV transmitted) of the outgoing ray's own axes -- what the general form below comes to
      p.pc = sh ? 0.0 : (reflected ? -n.z : n.z);
      p.ps = sh ? n.z : 0.0;
    }
    if (!FLAT_FACES || any_lanes(!flat)) {
      if (!flat) {
        double w2;
        const V3 w = rt_plane_normal(n, p.dir, 0.0, w2);
        // SH stays SH: w; R_SV: out x w; T_SV: w x out = -(out x w)
        const V3 c = cross(out, w);
        const double sg = reflected ? 1.0 : -1.0;
        const V3 dopm = sh ? w : v3(sg * c.x, sg * c.y, sg * c.z);
        set_pol(p, dopm, out);
      }
    }
  }
  p.dir = out;
  return !reflected;
}
template <bool FLAT_FACES = false>
R3D_HD bool rt_event(Phonon& p, const Iface& f, double u_pol, double u_out) {
  const RtChoice ch = rt_choose<FLAT_FACES>(p, f, u_pol, u_out);
  return rt_apply<FLAT_FACES>(p, f.normal, ch);
}

// Snell bending without mode conversion across a weak velocity step
// (reference Phonon::Refraction_Bend, phonons.cpp:311-405).  Returns true if
// transmitted.
template <bool FLAT_FACES = false>
R3D_HD bool bend(Phonon& p, V3 fnorm, double veli, double velo) {
  // The reference builds unit axes fpara (in the plane of incidence, along the face), fparash (normal
  // to that plane) and the SV axes of the incoming and outgoing ray, and expresses the particle
  // motion in them.  None of them needs normalising:
  //   * sin(i) fpara is the part of the direction tangential to the face, dt = d - (n.d) n, and
  //     Snell's outgoing direction is (velo / veli) dt + cos(o) n -- no division by sin(i);
  //   * w = n x d = sin(i) fparash, and with w in place of fparash the outgoing particle motion
  //     comes out scaled by sin^2(i) > 0, which the projection onto (theta^, phi^) divides away.
  // Normal incidence (w = 0) takes the reference's substitute axis (geom_r3.cpp:146-171).
  const double cn = dot(fnorm, p.dir);
  const V3 dt = p.dir - cn * fnorm;
  const double ratio = velo * frcp(veli);   // (the incident ray's own velocity: a plain number)
  const double so2 = (ratio * ratio) * mag2(dt);   // sin^2 of the outgoing angle
  const bool transfer = !(so2 >= 1.0);
  // (total reflection: the same tangential part, the normal part reversed)
  const double kt = transfer ? ratio : 1.0;
  const double kn = transfer ? fsqrt(1.0 - so2) : -cn;
  const V3 out = kt * dt + kn * fnorm;
  if (p.type != RAY_P) {
    // FLAT_FACES, a horizontal face: the outgoing ray keeps the incoming ray's azimuth, SH is phi^ and SV theta^ on both
    // sides of the bend, so the polarisation ANGLE is what it was -- the general form below comes to exactly that.
    const bool flat = FLAT_FACES && face_is_flat(fnorm);
    if (!FLAT_FACES || any_lanes(!flat)) {
      if (!flat) {
        V3 w = cross(fnorm, p.dir);
        if (is_zero(w)) {
          w = cross(fnorm, v3(1, 0, 0));
          if (is_zero(w)) w = cross(fnorm, v3(0, 1, 0));
        }
        const V3 pdomi = direction_of_motion(p);
        const V3 pdomo = dot(pdomi, w) * w + dot(pdomi, cross(w, p.dir)) * cross(w, out);
        set_pol(p, pdomo, out);
      }
    }
  } else {
    p.pc = 1.0, p.ps = 0.0;                  // polout = 0
  }
  p.dir = out;
  return transfer;
}

// ================================================================= scatter ==
// Inverse-CDF draw: smallest k with r <= cdf[k], r = total * U
// (reference ProbDist::GetRandomIndex, probability.cpp:104-128).
R3D_HD uint64_t sample_cdf(const double* __restrict__ cdf, uint64_t n, double u) {
  uint64_t k1 = 0, k2 = n - 1;
  const double r = cdf[k2] * u;
  while (k1 != k2) {
    uint64_t k = (k1 + k2) >> 1;
    if (r <= cdf[k]) k2 = k;
    else k1 = k + 1;
  }
  return k2;
}
// The same draw with a search guide (r3d_tables.h GuideCell, r3d_pack.h build_guide_cells):
// identical result.  The guide cell of the draw holds the ends [k1, k2] of a bracket of a few
// entries (about five on average) AND, for a bracket of up to seven, its entries: the answer --
// k1 + (how many of them lie below r): the table is non-decreasing, so that is the smallest k with
// r <= cdf[k], what the bisection finds -- follows from ONE 64-byte fetch.  A longer bracket's cell
// holds seven pivots instead; they leave an eighth of the bracket, whose first eight entries are
// fetched at once: two dependent round trips for brackets of up to 64 entries, where a bisection
// of the whole table takes twenty and one of the bracket three to six.
R3D_HD uint64_t sample_cdf_guided(const double* __restrict__ cdf_, const GuideCell* __restrict__ guide_,
                                  uint32_t bits, double total, double u) {
#if defined(__HIP_DEVICE_COMPILE__)
  // the tables live in HBM: say so, or pointers that were themselves loaded from memory become
  // FLAT loads (which also count against the LDS counter and wait with it)
  typedef __attribute__((address_space(1))) const double gdouble;
  typedef __attribute__((address_space(1))) const GuideCell gcell;
  gdouble* cdf = (gdouble*)cdf_;
  gcell* guide = (gcell*)guide_;
#else
  const double* cdf = cdf_;
  const GuideCell* guide = guide_;
#endif
  const double r = total * u;
  uint32_t j = (uint32_t)(u * (double)(1u << bits));
  if (j > (1u << bits) - 1u) j = (1u << bits) - 1u;
  const GuideCell g = guide[j];
  uint64_t k1 = g.k1;
  const uint64_t k2 = g.k2;
  const bool direct = k2 - k1 <= (uint64_t)kGuideVals;
  // how many of the cell's values lie below r (entries, or pivots).  (A short bracket's cell is filled up
  // with cdf[k2], and r <= cdf[k2] -- the answer lies in the bracket --: the filling never counts, so
  // no lane asks how long its bracket is.)
  uint32_t below = 0;
#pragma unroll
  for (int i = 0; i < kGuideVals; i++) below += !(r <= g.c[i]) ? 1u : 0u;
  if (direct) return k1 + below;
  // a long bracket: the pivots leave an eighth of it, [lo, hi]; its first eight entries at once
  uint64_t lo = below ? guide_pivot(k1, k2, (int)below - 1) + 1 : k1;
  const uint64_t hi = below < (uint32_t)kGuideVals ? guide_pivot(k1, k2, (int)below) : k2;
  constexpr int kAtOnce = 8;
  double c[kAtOnce];
#pragma unroll
  for (int i = 0; i < kAtOnce; i++) c[i] = cdf[(lo + i < hi) ? lo + i : hi];
  uint32_t more = 0;
#pragma unroll
  for (int i = 0; i < kAtOnce; i++) more += !(r <= c[i]) ? 1u : 0u;   // (beyond hi: cdf[hi] again, never below r)
  lo += more;
  if (more == kAtOnce) {   // (brackets beyond 64 entries: bisect what is left)
    uint64_t top = hi;
    while (lo != top) {
      const uint64_t k = (lo + top) >> 1;
      if (r <= cdf[k]) top = k;
      else lo = k + 1;
    }
  }
  return lo;
}
// One word of the guide cell a draw will read (sample_cdf_guided): the fetch brings the cell's 64-byte
// sector near without holding sixteen registers for it.  For callers that start several draws and want
// their cells on the way before they look at the first (the pool's refill).
R3D_HD uint32_t guide_touch(const GuideCell* __restrict__ guide_, uint32_t bits, double u) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(1))) const GuideCell gcell;
  gcell* guide = (gcell*)guide_;
#else
  const GuideCell* guide = guide_;
#endif
  uint32_t j = (uint32_t)(u * (double)(1u << bits));
  if (j > (1u << bits) - 1u) j = (1u << bits) - 1u;
  return guide[j].k1;
}
R3D_HD int sample_small(const double* cdf, int n, double u) {
  const double r = cdf[n - 1] * u;
  int k = n - 1;
  for (int i = n - 2; i >= 0; i--)
    if (r <= cdf[i]) k = i;
  return k;
}
// Rotate (dir, pol) by a deflection given in the phonon's own frame
// (reference Phonon::Transform, phonons.cpp:116-170; OrthoAxes,
// geom_r3.cpp:212-286).  rel is the unit deflection vector, (rc, rs) the cosine
// and sine of the relative polarisation angle.
// The deflection comes as {cos theta, cos phi, sin phi, sin theta} (r3d_tables.h KArgs::toa_dir): the
// vector and its theta^, phi^ axes without a square root.
R3D_HD void scatter_transform(Phonon& p, const double* dirn, double rc, double rs, int new_type) {
  V3 e1, e2;
  sph_basis(p.dir, e1, e2);
  const V3 s1 = p.pc * e1 + p.ps * e2, s2 = (-p.ps) * e1 + p.pc * e2, e3 = p.dir;
  const double ct = dirn[0], cp = dirn[1], sp = dirn[2], st = dirn[3];
  const V3 rel = v3(st * cp, st * sp, ct);
  const V3 b1 = v3(ct * cp, ct * sp, -st), b2 = v3(-sp, cp, 0.0);
  const V3 bs1 = rc * b1 + rs * b2;              // S1 axis of the deflection frame
  V3 nd = rel.x * s1 + rel.y * s2 + rel.z * e3;  // AA.Express(BB.E3)
  V3 ns1 = bs1.x * s1 + bs1.y * s2 + bs1.z * e3;
  // (nd is unit to rounding -- unit rel in an orthonormal frame; no round trip through theta, phi)
  set_pol(p, ns1, nd);
  p.dir = nd;
  p.type = new_type;
}

}  // namespace r3d
#endif
