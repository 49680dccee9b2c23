// r3d_oracle.cpp -- CPU restatement of the reference's hot path
// (ShearDislocation::GenerateEventPhonon + Phonon::Propagate and callees),
// operating on the flat tables of include/r3d.h.
//
// ***  TEST INFRASTRUCTURE ONLY.  ***  Nothing in the product may include,
// link or call this file; only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py use it, as the checker / reported baseline.
//
// PARITY STATUS: "parity unpinned" leg-by-leg.  The reference cannot be
// built under this project's rules (typedefs.hpp:25 includes the
// Makefile-generated config/opt-fptype.hpp, Makefile:56-82) and ships no
// tests or golden vectors.  This restatement is pinned only by (i) the
// reference outputs recorded in SURVEY.md/BASELINE.md (scatterer MFPs and
// dipoles, cell/scatterer counts, per-history event mixes, loss counters)
// and (ii) analytic known answers (energy conservation of the R/T solve,
// straight-ray travel times, closed-form arc travel times).  See DESIGN.md.
//
// Style: scalar, one history at a time, direction kept as (theta, phi) and
// polarisation as an angle exactly like the reference's Phonon
// (phonons.hpp:69-126); every function cites the reference lines it follows.
// The only deliberate departure is the random stream: libc rand()
// (model.cpp:235) is replaced by Philox draws keyed by history id
// (oracle/philox.h), with every draw mapped to (0,1].
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <limits>

#include "../include/r3d.h"
#include "philox.h"

namespace {

const double PI = 3.14159265358979323846;
const double PI45 = PI * 0.25, PI90 = PI * 0.5, PI180 = PI, PI270 = PI * 1.5, PI360 = PI * 2.0;
const double INF = std::numeric_limits<double>::infinity();

// ---------------------------------------------------------------- vectors --
struct V {
  double x, y, z;
};
inline V mk(double x, double y, double z) { return V{x, y, z}; }
inline V mk(const double a[3]) { return V{a[0], a[1], a[2]}; }
inline V operator+(V a, V b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V operator-(V a, V b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V operator*(double s, V a) { return mk(s * a.x, s * a.y, s * a.z); }
inline double dot(V a, V b) { return b.x * a.x + b.y * a.y + b.z * a.z; }
inline V cross(V a, V b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline double mag2(V a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline double mag(V a) { return std::sqrt(mag2(a)); }
inline bool is_zero(V a) { return a.x == 0 && a.y == 0 && a.z == 0; }
inline V unit(V a) {  // geom_r3.hpp:122-125
  double s = 1.0 / mag(a);
  return mk(a.x * s, a.y * s, a.z * s);
}
inline V unit_else(V a, V fb) {  // geom_r3.hpp:127-132
  double m = mag(a);
  if (m == 0) return fb;
  double s = 1.0 / m;
  return mk(a.x * s, a.y * s, a.z * s);
}
inline V neg(V a) { return mk(-a.x, -a.y, -a.z); }

// R3::XYZ(const S2::ThetaPhi&), geom_r3.cpp:36-40
inline V from_angles(double th, double ph) {
  return mk(std::sin(th) * std::cos(ph), std::sin(th) * std::sin(ph), std::cos(th));
}
// XYZ::Theta / XYZ::Phi, geom_r3.hpp:98-106
inline double theta_of(V a) { return mag2(a) == 0 ? 0.0 : std::acos(a.z / mag(a)); }
inline double phi_of(V a) { return std::atan2(a.y, a.x); }

// R3::XYZ::ThetaHat, geom_r3.cpp:85-108
V theta_hat(V a) {
  double th = theta_of(a), ph = phi_of(a), rth, rph;
  if (th < PI90) {
    rth = PI90 + th, rph = ph;
  } else {
    rth = PI270 - th;
    rph = (ph < PI180) ? ph + PI180 : ph - PI180;
  }
  return mk(std::sin(rth) * std::cos(rph), std::sin(rth) * std::sin(rph), std::cos(rth));
}
// R3::XYZ::PhiHat, geom_r3.cpp:118-126
V phi_hat(V a) {
  double rph = phi_of(a) + PI90;
  return mk(std::cos(rph), std::sin(rph), 0);
}
// S2::ThetaPhi::ThetaHat / PhiHat (geom_s2.cpp:157-171, geom_s2.hpp:118-123)
// converted through R3::XYZ(const ThetaPhi&): what `mDir.ThetaHat()` means
// when mDir is the phonon's (theta, phi) pair (phonons.cpp:463-464).
V theta_hat_s2(double th, double ph) {
  double rth, rph;
  if (th < PI90) {
    rth = PI90 + th, rph = ph;
  } else {
    rth = PI270 - th;
    rph = (ph < PI180) ? ph + PI180 : ph - PI180;
  }
  return from_angles(rth, rph);
}
V phi_hat_s2(double ph) { return from_angles(PI90, (ph < PI270) ? ph + PI90 : ph - PI270); }
// R3::XYZ::GetInPlaneUnitPerpendicular, geom_r3.cpp:146-171
V in_plane_unit_perp(V self, V other) {
  V mp = cross(self, other);
  if (is_zero(mp)) {
    mp = cross(self, mk(1, 0, 0));
    if (is_zero(mp)) mp = cross(self, mk(0, 1, 0));
  }
  mp = unit(mp);
  return unit(cross(mp, self));
}
// S2::ThetaPhi(const S2::Node&) after Node's normalising constructor,
// geom_s2.hpp:62-66,202-205 and geom_s2.cpp:321-331
void angles_from_node(V n, double& th, double& ph) {
  if (!(n.x == 0 && n.y == 0 && n.z == 0)) {
    double norm = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
    n.x /= norm, n.y /= norm, n.z /= norm;
  }
  th = std::acos(n.z);
  ph = std::atan2(n.y, n.x);
}

// R3::OrthoAxes, geom_r3.cpp:212-233 (constructor) and :241-286 (Express)
struct Axes {
  double theta, phi, rot;
  V e1, e2, e3, s1, s2;
};
Axes make_axes(double the, double phi, double rot) {
  Axes a;
  a.theta = the, a.phi = phi, a.rot = rot;
  double ct = std::cos(the), st = std::sin(the), cp = std::cos(phi), sp = std::sin(phi);
  double cr = std::cos(rot), sr = std::sin(rot);
  a.e3 = mk(st * cp, st * sp, ct);
  a.e1 = mk(ct * cp, ct * sp, -st);
  a.e2 = mk(-sp, cp, 0);
  a.s1 = mk(cr * ct * cp - sr * sp, cr * ct * sp + sr * cp, -cr * st);
  a.s2 = mk(-sr * ct * cp - cr * sp, -sr * ct * sp + cr * cp, sr * st);
  return a;
}
inline V express(const Axes& a, V v) {  // geom_r3.hpp:560-566
  return mk(v.x * a.s1.x + v.y * a.s2.x + v.z * a.e3.x, v.x * a.s1.y + v.y * a.s2.y + v.z * a.e3.y,
            v.x * a.s1.z + v.y * a.s2.z + v.z * a.e3.z);
}
Axes express(const Axes& a, const Axes& b) {
  Axes r;
  r.s1 = express(a, b.s1), r.s2 = express(a, b.s2), r.e3 = express(a, b.e3);
  double ct = r.e3.z, the = std::acos(ct), phi = std::atan2(r.e3.y, r.e3.x);
  double st = std::sin(the), cp = std::cos(phi), sp = std::sin(phi);
  r.theta = the, r.phi = phi;
  r.e1 = mk(ct * cp, ct * sp, -st);
  r.e2 = mk(-sp, cp, 0);
  r.rot = std::atan2(dot(r.s1, r.e2), dot(r.s1, r.e1));
  return r;
}

// ----------------------------------------------------------------- state ---
struct Phonon {  // phonons.hpp:69-126
  double t, path, recent, amp;
  V loc;
  double theta, phi, pol;
  int type;  // R3D_RAY_P / R3D_RAY_S
  int cell;
  unsigned moves;
};

struct TravelRec {  // media.hpp:94-119
  double len, time;
  V loc;
  double theta, phi;
  double atten;
  int face;
};

// Volumetric scatter-event grid (include/r3d.h r3d_volume_desc): what the reference's
// video scripts histogram from the SCT / REF report lines (dataout.cpp:570-577,
// vis/scattervid/preprocess.sh:17-29, scattervid_above.m:111), in model coordinates.
const r3d_volume_desc* g_vol_desc = nullptr;
uint32_t* g_vol = nullptr;
uint64_t g_vol_outside = 0;   // events that fell outside the attached grid (r3d_result.events[R3D_EV_VOLUME_OUT])
void volume_count(double t, V loc, int type) {
  if (!g_vol) return;
  const r3d_volume_desc& v = *g_vol_desc;
  double f = t * (1.0 / v.frame_dt);
  double x = (loc.x - v.origin[0]) * (1.0 / v.cell_size[0]);
  double y = (loc.y - v.origin[1]) * (1.0 / v.cell_size[1]);
  double z = (loc.z - v.origin[2]) * (1.0 / v.cell_size[2]);
  if (!(f >= 0 && x >= 0 && y >= 0 && z >= 0 && f < v.n_frames && x < v.dims[0] && y < v.dims[1] &&
        z < v.dims[2])) {
    g_vol_outside++;
    return;
  }
  size_t idx = ((((size_t)type * v.n_frames + (size_t)f) * v.dims[2] + (size_t)z) * v.dims[1] + (size_t)y) *
                   v.dims[0] + (size_t)x;
  g_vol[idx] += 1;
}

// Per-event report stream (include/r3d.h r3d_event; reference dataout.cpp:484-617): the
// record the reference prints as one text line, appended in program order.
r3d_event* g_evlog = nullptr;
uint64_t g_evlog_cap = 0, g_evlog_count = 0;
uint32_t g_evlog_mask = 0;

struct Ctx {
  uint64_t id;   // history id (mSID)
  const r3d_model_desc* m;
  oracle_rng rng;
  r3d_result* out;
  unsigned n_catch;
};

inline double draw(Ctx& c) { return oracle_rng_draw(&c.rng); }
inline void draw_pair(Ctx& c, double& u0, double& u1) { oracle_rng_draw_pair(&c.rng, &u0, &u1); }

// Phonon::nudge_if_singular, phonons.hpp:335-344
inline void nudge(const r3d_params& p, double& theta) {
  if (theta < p.min_theta) theta = p.min_theta;
  if (theta > p.max_theta) theta = p.max_theta;
}

// MediumCell::HelperUniformAttenuation, media.cpp:98-100
inline double attenuation(double cycles, double Q) { return std::exp((-1 * PI * cycles) / Q); }

// Get{Veloc,Density}AtPoint for the three cell types: media.cpp:185-196
// (cylinder), :410-421 (tetra), :645-656 (sphere shell)
double velocity_at(const r3d_model_desc& m, const r3d_cell& c, V p, int type) {
  switch (m.cell_kind) {
    case R3D_CELL_CYLINDER: return c.vel_c[type];
    case R3D_CELL_TETRA: return dot(p, mk(c.vel_grad[type])) + c.vel_c[type];
    default: return c.vel_c[type] + c.vel_a[type] * mag2(p);
  }
}
double density_at(const r3d_model_desc& m, const r3d_cell& c, V p) {
  switch (m.cell_kind) {
    case R3D_CELL_CYLINDER: return c.rho_c;
    case R3D_CELL_TETRA: return dot(p, mk(c.rho_grad)) + c.rho_c;
    default: return c.rho_c + c.rho_a * mag2(p);
  }
}

// ---------------------------------------------------- face intersections ---
// PlaneFace::LinearRayDistToExit, media_cellface.cpp:262-324
double plane_exit(const r3d_face& f, V loc, V dir) {
  V n = mk(f.normal);
  double d_sh = dot(n, mk(f.point) - loc);
  double d_fact = dot(n, dir);
  if (d_fact < 0) return INF;
  if (d_fact == 0) return d_sh < 0 ? -INF : INF;
  return d_sh / d_fact;
}
// CylinderFace::LinearRayDistToExit, media_cellface.cpp:531-562
double cylwall_exit(const r3d_face& f, V loc, V dir) {
  double A = dir.x * dir.x + dir.y * dir.y;
  double C = loc.x * loc.x + loc.y * loc.y - f.radius * f.radius;
  if (A == 0) return C <= 0 ? INF : -INF;
  double B = 2 * (loc.x * dir.x + loc.y * dir.y);
  double urad = B * B - 4 * A * C;
  if (urad < 0) return -INF;
  return (std::sqrt(urad) - B) / (2 * A);
}
// SphereFace::LinearRayDistToExit, media_cellface.cpp:664-684
double sphere_exit(const r3d_face& f, V loc, V dir) {
  bool outward = f.radius > 0;
  double midpt = -dot(loc, dir);
  double urad = f.radius * f.radius + midpt * midpt - mag2(loc);
  if (urad <= 0) return outward ? -INF : INF;
  double sq = std::sqrt(urad);
  if (outward) return midpt + sq;
  if (midpt <= 0) return INF;
  return midpt - sq;
}
// CellFace::Normal for the three face shapes (media_cellface.hpp:286,
// media_cellface.cpp:453-456, :624-627)
V face_normal(const r3d_model_desc& m, const r3d_face& f, int face_idx, V loc) {
  if (m.cell_kind == R3D_CELL_SPHERESHELL) {
    V u = unit_else(loc, mk(0, 0, 1));
    return f.radius > 0 ? u : neg(u);
  }
  if (m.cell_kind == R3D_CELL_CYLINDER && face_idx == 2)
    return unit_else(mk(loc.x, loc.y, 0), mk(1, 0, 0));
  return mk(f.normal);
}

// ------------------------------------------------------ cylinder cells ----
// RCUCylinder::AdvanceLength, media.cpp:208-222
TravelRec cyl_advance(const r3d_model_desc& m, const r3d_cell& c, int rt, double len, V loc,
                      double th, double ph) {
  TravelRec r;
  r.len = len;
  r.time = len / c.vel_c[rt];
  r.loc = loc + len * from_angles(th, ph);
  r.theta = th, r.phi = ph;
  r.atten = attenuation(r.time * m.params.frequency, c.q[rt]);
  r.face = -1;
  return r;
}
// RCUCylinder::GetPathToBoundary, media.cpp:236-330
TravelRec cyl_path(const r3d_model_desc& m, const r3d_cell& c, int rt, V loc, double th,
                   double ph) {
  V d = from_angles(th, ph);
  double d_loss = cylwall_exit(c.faces[2], loc, d);
  double d_top = plane_exit(c.faces[0], loc, d);
  double d_bot = plane_exit(c.faces[1], loc, d);
  if (d_loss < 0) d_loss = 0;
  if (d_top < 0) d_top = 0;
  if (d_bot < 0) d_bot = 0;
  int exf = 2;
  double shortest = d_loss;
  if (d_top < shortest) exf = 0, shortest = d_top;
  if (d_bot < shortest) exf = 1, shortest = d_bot;
  TravelRec r = cyl_advance(m, c, rt, shortest, loc, th, ph);
  r.face = exf;
  return r;
}

// --------------------------------------------------------- tetra cells ----
// CoordinateTransformation, media.hpp:560-587: frame in which the ray is a
// circle of radius R about the origin of the (x,z) plane.
struct ArcFrame {
  V prime_loc, trans;
  V row[3];  // rotation matrix rows v1, v2, v3
  double R;
};
inline V rot_apply(const ArcFrame& f, V v) {
  return mk(dot(f.row[0], v), dot(f.row[1], v), dot(f.row[2], v));
}
inline V rot_apply_T(const ArcFrame& f, V v) {
  return mk(f.row[0].x * v.x + f.row[1].x * v.y + f.row[2].x * v.z,
            f.row[0].y * v.x + f.row[1].y * v.y + f.row[2].y * v.z,
            f.row[0].z * v.x + f.row[1].z * v.y + f.row[2].z * v.z);
}
ArcFrame arc_frame(double v0, V g, V loc, V t) {
  ArcFrame f;
  V v2 = cross(g, t), v1 = cross(v2, g), v3 = g;
  v1 = unit(v1), v2 = unit(v2), v3 = unit(v3);
  double txp = dot(t, v1), tzp = dot(t, v3);
  double s = txp / v0;
  double R = 1 / (s * mag(g));
  f.row[0] = v1, f.row[1] = v2, f.row[2] = v3;
  V x0rot = rot_apply(f, loc);
  V translate = mk(x0rot.x + R * tzp, x0rot.y, x0rot.z + (-1) * R * txp);
  f.prime_loc = x0rot + (-1.0) * translate;
  f.trans = translate;
  f.R = R;
  return f;
}

// Tetra::AdvanceLength, media.cpp:442-499
TravelRec tet_advance(const r3d_model_desc& m, const r3d_cell& c, int rt, double len, V loc,
                      double th, double ph) {
  V g = mk(c.vel_grad[rt]);
  ArcFrame CT = arc_frame(dot(loc, g) + c.vel_c[rt], g, loc, from_angles(th, ph));
  double theta = len / CT.R;
  V rot2d = mk(CT.R * std::sin(theta / 2), 0, CT.R * std::cos(theta / 2));
  double angle0 = std::atan2(CT.prime_loc.x, CT.prime_loc.z);
  double rot = angle0 + (theta / 2);
  rot = (rot > PI360) ? rot - PI360 : rot;
  V new2d = mk(std::cos(rot) * rot2d.x + std::sin(rot) * rot2d.z, 0,
               -std::sin(rot) * rot2d.x + std::cos(rot) * rot2d.z);
  V newloc = rot_apply_T(CT, new2d + CT.trans);
  double angle1 = std::atan2(new2d.x, new2d.z);
  V dir2d = mk(std::cos(angle1), 0, (-1) * std::sin(angle1));
  V newdir = unit(rot_apply_T(CT, dir2d));
  double time = (1 / mag(g)) * (std::log(std::fabs(std::tan(angle1 / 2 + PI45))) -
                                std::log(std::fabs(std::tan(angle0 / 2 + PI45))));
  TravelRec r;
  r.len = len;
  r.time = time;
  r.loc = newloc;
  angles_from_node(newdir, r.theta, r.phi);
  r.atten = attenuation(r.time * m.params.frequency, c.q[rt]);
  r.face = -1;
  return r;
}

// GCAD_RetVal + PlaneFace::GetCircArcDistToFace, media_cellface.cpp:333-426
struct Gcad {
  double entry, exit, half;
  bool continuous;
};
Gcad plane_arc(const r3d_face& f, const ArcFrame& CT) {
  bool continuous = true;
  V rn = rot_apply(CT, mk(f.normal));
  V x0p = rot_apply(CT, mk(f.point)) + (-1.0) * CT.trans;
  double d = (-1) * dot(rn, x0p);
  double D = -d / std::sqrt(rn.x * rn.x + rn.z * rn.z);
  V n2 = unit(mk(rn.x, 0, rn.z));
  double bis = std::atan2(n2.x, n2.z);
  double ex = 0, en = 0;
  const double R = CT.R;
  if (D / R < 1 && D / R > -1) {
    double q = std::acos(D / R);
    if (bis > (-1) * PI90 && bis < PI90) {
      en = bis + q, ex = bis - q, continuous = false;
    } else if (bis <= (-1) * PI90) {
      en = bis + q, ex = bis - q + PI360;
    } else if (bis >= PI90) {
      en = bis + q - PI360, ex = bis - q;
    } else {
      // reference prints "GCAD Bisector nan" and exit(1)s
      // (media_cellface.cpp:390-393); the oracle lets the NaN propagate so
      // the history ends as an invalid phonon instead of killing the process.
      en = ex = bis;
    }
  }
  if (bis >= PI90 || bis <= (-1) * PI90) bis = INF;
  if (en >= PI90) en = INF;
  if (en <= (-1) * PI90) en = -INF;
  if (ex >= PI90) ex = INF;
  if (ex <= (-1) * PI90) ex = -INF;
  if (D / R >= 1) en = -INF, ex = INF;
  if (D / R <= -1) en = INF, ex = -INF, bis = -INF, continuous = false;
  return Gcad{en, ex, bis, continuous};
}
// GCAD_RetVal::Inside / IsProper, media_cellface.cpp:767-794
bool gcad_inside(const Gcad& g, double theta) {
  const double slack = 0.0000000001;
  if (g.continuous) {
    if (theta <= g.exit && theta >= (g.entry - slack)) return true;
  } else {
    if ((theta >= (-1) * PI90 && theta <= g.exit) || (theta >= (g.entry - slack) && theta <= PI90))
      return true;
  }
  return false;
}
// Tetra::GetPathToBoundary, media.cpp:518-567
TravelRec tet_path(const r3d_model_desc& m, const r3d_cell& c, int rt, V loc, double th,
                   double ph) {
  V g = mk(c.vel_grad[rt]);
  ArcFrame CT = arc_frame(dot(loc, g) + c.vel_c[rt], g, loc, from_angles(th, ph));
  double angle0 = std::atan2(CT.prime_loc.x, CT.prime_loc.z);
  Gcad rv[4];
  for (int i = 0; i < 4; i++) rv[i] = plane_arc(c.faces[i], CT);
  double len = INF;
  int face = 0;
  for (int i = 0; i < 4; i++) {
    double e = rv[i].exit;
    if (gcad_inside(rv[(i + 1) % 4], e) && gcad_inside(rv[(i + 2) % 4], e) &&
        gcad_inside(rv[(i + 3) % 4], e)) {
      double nl = (e - angle0) * CT.R;
      if (nl < 0 && (angle0 > rv[i].half)) nl = len;  // dismiss exit
      if (nl < len) len = nl, face = i;
    }
  }
  TravelRec r = tet_advance(m, c, rt, len, loc, th, ph);
  r.face = face;
  return r;
}

// -------------------------------------------------- sphere-shell cells ----
// cache_RD2_precompute + RayArcAttributes, raypath.hpp:31-113, raypath.cpp:5-20
struct RayArc {
  double radius, rad2;
  V center, u3, u2, u1;
  double S, S2, TwoSQ, CosZeta, SinZeta, CotZetaBy2, timeCoef;
};
inline double arc_angle(const RayArc& a, V loc) {
  V cl = loc - a.center;
  return std::atan2(dot(a.u1, cl), dot(a.u3, cl));
}
// ECS.GetDown, ecs.cpp:147-167 + ecs.hpp GetDown
inline V down_at(const r3d_model_desc& m, V loc) {
  return neg(unit_else(loc - mk(m.params.earth_center), mk(0, 1, 0)));
}
// SphereShell::GetRayArc_RD2, media.cpp:795-861
RayArc shell_arc(const r3d_model_desc& m, const r3d_cell& c, int rt, V loc, V dir) {
  RayArc A;
  V v3 = down_at(m, loc);
  V v2 = unit_else(cross(v3, dir), mk(0, 0, 0));
  V v1 = cross(v2, v3);
  double sini = dot(v1, dir);
  if (sini > 1.0) sini = 1.0;
  double cosi = dot(v3, dir);
  const double G = sini * mag(loc) / (c.vel_c[rt] + c.vel_a[rt] * mag2(loc));
  const double TwoGA = 2. * G * c.vel_a[rt];
  const double urad = 1. - (2. * TwoGA * G * c.vel_c[rt]);
  double Bottom = (urad > 1) ? (1. - std::sqrt(urad)) / TwoGA : 0;
  A.radius = (c.zero_rad2[rt] / Bottom - Bottom) / 2.0;
  A.rad2 = A.radius * A.radius;
  V toC = (A.radius * cosi) * v1 + (-A.radius * sini) * v3;
  A.center = loc + toC;
  A.u3 = down_at(m, A.center);
  A.u2 = v2;
  A.u1 = cross(A.u2, A.u3);
  if (urad <= 1) {
    A.center = mk(0, 0, 0);
    A.u3 = A.u2 = mk(0, 0, 0);
    A.u1 = dir;
  }
  A.S2 = mag2(A.center);
  A.S = std::sqrt(A.S2);
  A.TwoSQ = 2 * A.S * A.radius;
  A.CosZeta = (A.S2 + A.radius * A.radius - c.zero_rad2[rt]) / A.TwoSQ;
  A.SinZeta = std::sqrt(1 - A.CosZeta * A.CosZeta);
  A.CotZetaBy2 = (1 + A.CosZeta) / A.SinZeta;
  A.timeCoef = -1 / (c.vel_a[rt] * A.S * A.SinZeta);
  return A;
}
// SphereFace::CircularArcDistToExit, media_cellface.cpp:717-748
double sphere_arc_exit(const r3d_face& f, V loc, V dir, const RayArc& a) {
  if (a.S2 == 0) return sphere_exit(f, loc, dir);
  bool outward = f.radius > 0;
  double cosq = (a.S2 + a.rad2 - f.radius * f.radius) / a.TwoSQ;
  if (cosq > 1.0) return outward ? -INF : INF;
  double BtoE = std::acos(cosq);
  double aloc = arc_angle(a, loc);
  if (outward) return (BtoE - aloc) * a.radius;
  if (aloc >= 0) return INF;
  return (-BtoE - aloc) * a.radius;
}
// SphereShell::AdvanceLength_Variant_RD0, media.cpp:877-889
TravelRec shell_advance_rd0(const r3d_model_desc& m, const r3d_cell& c, int rt, double len, V loc,
                            double th, double ph) {
  TravelRec r;
  r.len = len;
  r.time = len / c.vel_c[rt];
  r.loc = loc + len * from_angles(th, ph);
  r.theta = th, r.phi = ph;
  r.atten = attenuation(r.time * m.params.frequency, c.q[rt]);
  r.face = -1;
  return r;
}
// SphereShell::AdvanceLength_Variant_RD2_Impl, media.cpp:913-957 (+ :962-970)
TravelRec shell_advance_rd2(const r3d_model_desc& m, const r3d_cell& c, int rt, double len, V loc,
                            double th, double ph, const RayArc& a) {
  if (a.radius == INF) {  // vertical ray: straight line, analytic time
    TravelRec fb = shell_advance_rd0(m, c, rt, len, loc, th, ph);
    double r0 = mag(loc), r1 = mag(fb.loc);
    double sqnac = std::sqrt(-c.vel_a[rt] * c.vel_c[rt]);
    double sqnaoc = std::sqrt(-c.vel_a[rt] / c.vel_c[rt]);
    double tpm = (std::atanh(sqnaoc * r1) - std::atanh(sqnaoc * r0)) / sqnac;
    fb.time = std::fabs(tpm);
    // NB the reference leaves fallback.Attenuation as computed from the
    // straight-line time len/C (media.cpp:917-937); so does the oracle.
    return fb;
  }
  double a0 = arc_angle(a, loc);
  double a1 = a0 + len / a.radius;
  TravelRec r;
  r.len = len;
  r.loc = a.center + (a.radius * std::sin(a1)) * a.u1 + (a.radius * std::cos(a1)) * a.u3;
  V nd = std::cos(a1) * a.u1 + (-std::sin(a1)) * a.u3;
  double t0 = a.timeCoef * std::atanh(a.CotZetaBy2 * std::tan(a0 / 2));
  double t1 = a.timeCoef * std::atanh(a.CotZetaBy2 * std::tan(a1 / 2));
  r.time = t1 - t0;
  r.atten = attenuation(r.time * m.params.frequency, c.q[rt]);
  angles_from_node(nd, r.theta, r.phi);
  r.face = -1;
  return r;
}
// SphereShell::AdvanceLength, media.cpp:866-872,895-907
TravelRec shell_advance(const r3d_model_desc& m, const r3d_cell& c, int rt, double len, V loc,
                        double th, double ph) {
  if (c.vel_a[rt] == 0) return shell_advance_rd0(m, c, rt, len, loc, th, ph);
  RayArc a = shell_arc(m, c, rt, loc, from_angles(th, ph));
  return shell_advance_rd2(m, c, rt, len, loc, th, ph, a);
}
// SphereShell::GetPathToBoundary, media.cpp:668-757
TravelRec shell_path(const r3d_model_desc& m, const r3d_cell& c, int rt, V loc, double th,
                     double ph) {
  V d = from_angles(th, ph);
  if (c.vel_a[rt] < 0) {
    RayArc a = shell_arc(m, c, rt, loc, d);
    double dt = sphere_arc_exit(c.faces[0], loc, d, a);
    double db = sphere_arc_exit(c.faces[1], loc, d, a);
    int ef = (dt < db) ? 0 : 1;
    double dist = ef == 0 ? dt : db;
    if (dist < 0) dist = 0;
    TravelRec r = shell_advance_rd2(m, c, rt, dist, loc, th, ph, a);
    r.face = ef;
    return r;
  }
  // vel_a == 0: straight rays (a > 0 is rejected when the model is built)
  double dt = sphere_exit(c.faces[0], loc, d);
  double db = sphere_exit(c.faces[1], loc, d);
  int ef = (dt < db) ? 0 : 1;
  double dist = ef == 0 ? dt : db;
  if (dist < 0) dist = 0;
  TravelRec r = shell_advance_rd0(m, c, rt, dist, loc, th, ph);
  r.face = ef;
  return r;
}

TravelRec path_to_boundary(const r3d_model_desc& m, const r3d_cell& c, const Phonon& p) {
  switch (m.cell_kind) {
    case R3D_CELL_CYLINDER: return cyl_path(m, c, p.type, p.loc, p.theta, p.phi);
    case R3D_CELL_TETRA: return tet_path(m, c, p.type, p.loc, p.theta, p.phi);
    default: return shell_path(m, c, p.type, p.loc, p.theta, p.phi);
  }
}
TravelRec advance_length(const r3d_model_desc& m, const r3d_cell& c, const Phonon& p, double len) {
  switch (m.cell_kind) {
    case R3D_CELL_CYLINDER: return cyl_advance(m, c, p.type, len, p.loc, p.theta, p.phi);
    case R3D_CELL_TETRA: return tet_advance(m, c, p.type, len, p.loc, p.theta, p.phi);
    default: return shell_advance(m, c, p.type, len, p.loc, p.theta, p.phi);
  }
}

// ----------------------------------------------------------- sampling -----
// ProbDist::GetRandomIndex, probability.cpp:104-128
uint64_t sample_cdf_u(const double* cdf, uint64_t n, double u) {
  uint64_t k1 = 0, k2 = n - 1;
  double r = cdf[k2] * u;
  while (k1 != k2) {
    uint64_t k = (k1 + k2) >> 1;
    if (r <= cdf[k]) k2 = k;
    else k1 = k + 1;
  }
  return k2;
}

uint64_t sample_cdf(Ctx& c, const double* cdf, uint64_t n) { return sample_cdf_u(cdf, n, draw(c)); }

// Phonon::DirectionOfMotion, phonons.cpp:201-211
V direction_of_motion(const Phonon& p) {
  if (p.type == R3D_RAY_P) return from_angles(p.theta, p.phi);
  return make_axes(p.theta, p.phi, p.pol).s1;
}

// ---------------------------------------------------------- R/T solve -----
typedef std::complex<double> Cx;
enum { R_P, R_SV, R_SH, T_P, T_SV, T_SH, RT_NUM };

// RTCoef, rtcoef.hpp:74-327 / rtcoef.cpp:30-588
struct RT {
  double velR[2], velT[2], rhoR, rhoT;
  bool no_transmit;
  V fnorm, fpara, fparash;
  double sini;
  double sino[RT_NUM];
  Cx coso[RT_NUM];
  double prob[RT_NUM];
  int choice, defchoice;
  V outdir;
};

// rtcoef.cpp:289-393 + :107-198
void rt_coefs_psv(RT& r, bool in_p) {
  const double rho1 = r.rhoR, rho2 = r.rhoT;
  const double a1 = r.velR[0], a2 = r.velT[0], b1 = r.velR[1], b2 = r.velT[1];
  const double p = r.sini / (in_p ? a1 : b1);
  r.sino[T_P] = a2 * p, r.sino[T_SV] = b2 * p, r.sino[R_SV] = b1 * p, r.sino[R_P] = a1 * p;
  for (int k : {T_P, T_SV, R_SV, R_P}) r.coso[k] = std::sqrt(Cx(1.0 - r.sino[k] * r.sino[k]));
  const double b1s = b1 * b1, b2s = b2 * b2, psq = p * p;
  const double t1 = rho1 * (1. - 2. * b1s * psq), t2 = rho2 * (1. - 2. * b2s * psq);
  const double t3 = 2. * rho1 * b1s, t4 = 2. * rho2 * b2s;
  const double a = t2 - t1, b = t2 + t3 * psq, c = t1 + t4 * psq, d = t4 - t3;
  const Cx ci1 = r.coso[R_P] / a1, ci2 = r.coso[T_P] / a2;
  const Cx cj1 = r.coso[R_SV] / b1, cj2 = r.coso[T_SV] / b2;
  const Cx E = b * ci1 + c * ci2, F = b * cj1 + c * cj2;
  const Cx G = a - d * ci1 * cj2, H = a - d * ci2 * cj1;
  const Cx D = E * F + G * H * psq;
  const double two = 2.0;
  Cx amp[RT_NUM];
  Cx T1, T2;
  if (in_p) {
    T1 = ((b * ci1) - (c * ci2));
    T2 = ((a) + (d * ci1 * cj2));
    amp[R_P] = (T1 * F - T2 * H * psq) / D;
    T1 = (a * b + c * d * ci2 * cj2);
    amp[R_SV] = -two * ci1 * T1 * p * a1 / (b1 * D);
    T1 = two * rho1 * ci1 * a1;
    amp[T_P] = T1 * F / (a2 * D);
    amp[T_SV] = T1 * H * p / (b2 * D);
  } else {
    T1 = (a * b + c * d * ci2 * cj2);
    amp[R_P] = -two * cj1 * T1 * p * b1 / (a1 * D);
    T1 = (b * cj1 - c * cj2);
    T2 = (a + d * ci2 * cj1);
    amp[R_SV] = -(T1 * E - T2 * G * psq) / D;
    T1 = two * rho1 * cj1 * b1;
    amp[T_P] = -T1 * G * p / (a2 * D);
    amp[T_SV] = T1 * E / (b2 * D);
  }
  r.prob[R_SH] = r.prob[T_SH] = 0;
  r.prob[R_P] = rho1 * a1 * r.coso[R_P].real() * std::norm(amp[R_P]);
  r.prob[R_SV] = rho1 * b1 * r.coso[R_SV].real() * std::norm(amp[R_SV]);
  r.prob[T_P] = rho2 * a2 * r.coso[T_P].real() * std::norm(amp[T_P]);
  r.prob[T_SV] = rho2 * b2 * r.coso[T_SV].real() * std::norm(amp[T_SV]);
}
// rtcoef.cpp:207-278
void rt_coefs_sh(RT& r) {
  r.prob[R_P] = r.prob[R_SV] = r.prob[T_P] = r.prob[T_SV] = 0;
  const double rho1 = r.rhoR, rho2 = r.rhoT, b1 = r.velR[1], b2 = r.velT[1];
  r.sino[R_SH] = r.sini;
  r.sino[T_SH] = (b2 / b1) * r.sini;
  r.coso[R_SH] = std::sqrt(Cx(1.0 - r.sino[R_SH] * r.sino[R_SH]));
  r.coso[T_SH] = std::sqrt(Cx(1.0 - r.sino[T_SH] * r.sino[T_SH]));
  Cx a = rho1 * b1 * r.coso[R_SH], b = rho2 * b2 * r.coso[T_SH];
  Cx ar = (a - b) / (a + b), at = 2.0 * a / (a + b);
  r.prob[R_SH] = rho1 * b1 * r.coso[R_SH].real() * std::norm(ar);
  r.prob[T_SH] = rho2 * b2 * r.coso[T_SH].real() * std::norm(at);
}

// The event on a prepared interface (media, normal, no_transmit set): everything of Phonon::Refraction_FullRT
// (phonons.cpp:429-476) after CellFace::GetRTBasis -- the basis vectors, ChooseSPolType, GetCoefs, Choose and the chosen
// ray (rtcoef.cpp:406-588).  Returns true if the ray is transmitted; *margin (if asked for): how far the outcome draw
// was from the nearest partial sum, as a fraction of the total.
bool rt_event_core(RT& r, Phonon& p, double u_pol, double u_out, int* choice_out = nullptr, double* margin = nullptr,
                   double* pol_margin = nullptr) {
  V dir = from_angles(p.theta, p.phi);
  r.fpara = in_plane_unit_perp(r.fnorm, dir);
  r.fparash = cross(r.fnorm, r.fpara);
  r.sini = dot(r.fpara, dir);
  enum { IN_P, IN_SH, IN_SV } intype = IN_P;
  if (p.type == R3D_RAY_S) {  // ChooseSPolType, rtcoef.cpp:406-422
    double shfrac = dot(direction_of_motion(p), r.fparash);
    shfrac *= shfrac;
    intype = (u_pol <= shfrac) ? IN_SH : IN_SV;
    if (pol_margin) *pol_margin = std::fabs(u_pol - shfrac);
  }
  switch (intype) {  // GetCoefs, rtcoef.cpp:76-97
    case IN_P: r.defchoice = R_P, rt_coefs_psv(r, true); break;
    case IN_SH: r.defchoice = R_SH, rt_coefs_sh(r); break;
    default: r.defchoice = R_SV, rt_coefs_psv(r, false);
  }
  // Choose, rtcoef.cpp:436-475
  double PI_[RT_NUM];
  PI_[0] = r.prob[0];
  for (int i = 1; i < RT_NUM; i++) PI_[i] = PI_[i - 1] + r.prob[i];
  double total = PI_[RT_NUM - 1];
  double ran = u_out * total;
  int choice = RT_NUM - 1;
  for (int i = 0; i < RT_NUM - 1; i++)
    if (ran <= PI_[i]) {
      choice = i;
      break;
    }
  if (margin) {
    *margin = 1.0;
    for (int i = 0; i < RT_NUM - 1; i++)
      if (r.prob[i] != 0 || i == 0) *margin = std::min(*margin, std::fabs(ran - PI_[i]) / total);
  }
  if (total == 0 || (total - total) != 0) {
    choice = r.defchoice;
    if (margin) *margin = 1.0;
  }
  if (r.no_transmit) {
    if (choice == T_P) choice = R_P;
    if (choice == T_SV) choice = R_SV;
    if (choice == T_SH) choice = R_SH;
  }
  if (choice_out) *choice_out = choice;
  const bool reflected = (choice == R_P || choice == R_SV || choice == R_SH);
  // GetChosenRayDirection, rtcoef.cpp:529-548
  double comp_para = r.sino[choice], comp_norm = r.coso[choice].real();
  if (comp_para > 1.0) comp_para = 1.0;
  if (reflected) comp_norm *= -1;
  V outdir = comp_para * r.fpara + comp_norm * r.fnorm;
  p.type = (choice == R_P || choice == T_P) ? R3D_RAY_P : R3D_RAY_S;
  p.theta = theta_of(outdir), p.phi = phi_of(outdir);
  if (p.type == R3D_RAY_S) {  // GetChosenParticleDOM, rtcoef.cpp:559-588
    V dopm;
    if (choice == T_SH || choice == R_SH) dopm = r.fparash;
    else if (choice == R_SV) dopm = cross(outdir, r.fparash);
    else dopm = cross(r.fparash, outdir);
    p.pol = std::atan2(dot(dopm, phi_hat_s2(p.phi)), dot(dopm, theta_hat_s2(p.theta, p.phi)));
  }
  return !reflected;
}

// Phonon::Refraction_FullRT, phonons.cpp:429-476, with CellFace::GetRTBasis
// (media_cellface.cpp:122-149) and the RTCoef chooser (rtcoef.cpp:406-588).
void refraction_full_rt(Ctx& c, Phonon& p, int face_idx) {
  const r3d_model_desc& m = *c.m;
  const r3d_cell& cell = m.cells[p.cell];
  const r3d_face& f = cell.faces[face_idx];
  c.out->events[R3D_EV_RTSOLVE]++;
  double u_pol, u_out;   // the event's two uniforms (philox.h: one block)
  draw_pair(c, u_pol, u_out);
  RT r;
  r.no_transmit = false;
  r.fnorm = face_normal(m, f, face_idx, p.loc);
  r.rhoR = density_at(m, cell, p.loc);
  r.velR[0] = velocity_at(m, cell, p.loc, 0), r.velR[1] = velocity_at(m, cell, p.loc, 1);
  if (f.flags & R3D_FACE_ADJOIN) {
    const r3d_cell& o = m.cells[f.neighbor];
    r.rhoT = density_at(m, o, p.loc);
    r.velT[0] = velocity_at(m, o, p.loc, 0), r.velT[1] = velocity_at(m, o, p.loc, 1);
  } else {  // free surface emulated by a vanishing medium
    r.rhoT = 0.0, r.velT[0] = r.velT[1] = 1e-12, r.no_transmit = true;
  }
  if (rt_event_core(r, p, u_pol, u_out)) p.cell = f.neighbor;  // InsertInto, phonons.cpp:494-502
}

// Phonon::Refraction_Bend, phonons.cpp:311-405
// (the bend on a prepared face: unit normal, the ray's velocity either side; returns true if the ray crosses)
bool bend_core(Phonon& p, V fnorm, double veli, double velo) {
  V dir = from_angles(p.theta, p.phi);
  V fpara = in_plane_unit_perp(fnorm, dir);
  V fparash = cross(fnorm, fpara);
  double sini = dot(fpara, dir);
  double sino = (velo / veli) * sini;
  bool transfer;
  double coso;
  if (sino >= 1.0) {
    transfer = false, sino = sini, coso = -1.0 * dot(fnorm, dir);
  } else {
    transfer = true, coso = std::sqrt(1.0 - (sino * sino));
  }
  V outdir = sino * fpara + coso * fnorm;
  double polout = 0;
  if (p.type != R3D_RAY_P) {
    V pdomi = direction_of_motion(p);
    V svbasei = cross(fparash, dir), svbaseo = cross(fparash, outdir);
    double shcomi = dot(pdomi, fparash), svcomi = dot(pdomi, svbasei);
    V pdomo = shcomi * fparash + svcomi * svbaseo;
    polout = std::atan2(dot(pdomo, phi_hat(outdir)), dot(pdomo, theta_hat(outdir)));
  }
  p.theta = theta_of(outdir), p.phi = phi_of(outdir);
  p.pol = polout;
  return transfer;
}
void refraction_bend(Ctx& c, Phonon& p, int face_idx) {
  const r3d_model_desc& m = *c.m;
  const r3d_cell& cell = m.cells[p.cell];
  const r3d_face& f = cell.faces[face_idx];
  const V fnorm = face_normal(m, f, face_idx, p.loc);
  const double veli = velocity_at(m, cell, p.loc, p.type);
  const double velo = velocity_at(m, m.cells[f.neighbor], p.loc, p.type);
  if (bend_core(p, fnorm, veli, velo)) p.cell = f.neighbor;
}

// CellFace::VelocityJump, media_cellface.cpp:83-99
double velocity_jump(const r3d_model_desc& m, const r3d_cell& a, const r3d_cell& b, V loc) {
  double v1 = velocity_at(m, a, loc, 0), v2 = velocity_at(m, b, loc, 0);
  double dvp = std::fabs(2 * (v2 - v1) / (v2 + v1));
  v1 = velocity_at(m, a, loc, 1), v2 = velocity_at(m, b, loc, 1);
  double dvs = std::fabs(2 * (v2 - v1) / (v2 + v1));
  return dvp > dvs ? dvp : dvs;
}

// ---------------------------------------------------------- seismometers --
// DataReporter::ReportPhononCollected (dataout.cpp:545-568) +
// Seismometer::CatchPhonon (dataout.cpp:103-216).  mPassthrough is hard-wired
// true in the reference (dataout.cpp:50), so every seismometer is tested.
void collect(Ctx& c, const Phonon& p) {
  const r3d_model_desc& m = *c.m;
  const r3d_params& par = m.params;
  c.out->events[R3D_EV_COLLECT]++;
  for (int s = 0; s < m.n_seismometers; s++) {
    const r3d_seismometer& S = m.seismometers[s];
    bool within_window = true, within_radius = true;
    double arv = p.t, corr = 0;
    if (S.r_in[p.type] <= 0) {
      V to = mk(S.loc) - p.loc;
      corr = dot(to, from_angles(p.theta, p.phi));
      corr = corr / velocity_at(m, m.cells[p.cell], p.loc, p.type);
    }
    arv += corr;
    double scaled = (arv - 0.0) / par.time_per_bin;
    if (scaled < 0.0) within_window = false;
    double fl = std::floor(scaled);
    uint32_t bin = within_window ? (fl >= 4294967296.0 ? 0xFFFFFFFFu : (uint32_t)fl) : 0;
    if (bin >= par.n_bins) within_window = false;
    double dist = mag(mk(S.loc) - p.loc);
    if (dist > S.r_out[p.type]) within_radius = false;
    if (dist < S.r_in[p.type]) within_radius = false;
    if (!within_window || !within_radius) continue;
    V dopm = direction_of_motion(p);
    double xf = dot(dopm, mk(S.axes[0])), yf = dot(dopm, mk(S.axes[1])), zf = dot(dopm, mk(S.axes[2]));
    xf *= xf, yf *= yf, zf *= zf;
    double energy = p.amp * p.amp;
    energy /= par.time_per_bin;
    energy /= S.area[p.type];
    double* e = c.out->energy + ((size_t)s * par.n_bins + bin) * R3D_N_ENERGY;
    e[0] += energy * xf, e[1] += energy * yf, e[2] += energy * zf;
    e[3 + p.type] += energy;
    c.out->counts[((size_t)s * par.n_bins + bin) * R3D_N_COUNT + p.type] += 1;
    c.out->events[R3D_EV_CATCH]++;
    c.n_catch++;
  }
}

// ------------------------------------------------------------ one history --
// Phonon::Move, phonons.cpp:62-70
inline void move(Phonon& p, const TravelRec& t) {
  p.path += t.len;
  p.t += t.time;
  p.recent += t.time;
  p.loc = t.loc;
  p.theta = t.theta, p.phi = t.phi;
  p.amp *= t.atten;
  p.moves += 1;
}

void invalid(Ctx& c, int reason) {
  c.out->n_invalid++;
  c.out->invalid_reasons[reason]++;
}

// DataReporter::output_phonon_dataline's content (dataout.cpp:484-520) as a record.
// tag: 0 GEN 1 SCT 2 REF 3 COL 4 CEL 5 LST 6 TMO 7 INV.
void report(const Ctx& c, int tag, const Phonon& p) {
  if (!g_evlog || !((g_evlog_mask >> tag) & 1u)) return;
  const uint64_t at = g_evlog_count++;
  if (at >= g_evlog_cap) return;
  r3d_event& r = g_evlog[at];
  std::memset(&r, 0, sizeof r);
  r.id = c.id;
  r.time = p.t, r.path = p.path, r.amp = p.amp;
  r.loc[0] = p.loc.x, r.loc[1] = p.loc.y, r.loc[2] = p.loc.z;
  V d = from_angles(p.theta, p.phi);
  r.dir[0] = d.x, r.dir[1] = d.y, r.dir[2] = d.z;
  r.cell = (uint32_t)p.cell, r.moves = p.moves;
  r.tag = (uint8_t)tag, r.type = (uint8_t)p.type;
}

// PhononSource::GenerateRandomPhonon (sources.cpp:156-170) through
// ShearDislocation::GenerateEventPhonon (events.cpp:111-124), then
// Phonon::Propagate (phonons.cpp:540-682).  Returns the fate code.
int run_history(Ctx& c, Phonon& p) {
  const r3d_model_desc& m = *c.m;
  const r3d_params& par = m.params;
  r3d_result& out = *c.out;

  // --- spray ---
  int rt3 = (int)sample_cdf(c, m.source.whole_cdf, 3);  // 0 P, 1 SH, 2 SV
  uint64_t toa = sample_cdf(c, m.source.cdf[rt3], m.n_toa);
  p.t = p.path = p.recent = 0, p.moves = 0, p.amp = 1.0;
  p.theta = m.toa[2 * toa], p.phi = m.toa[2 * toa + 1];
  p.pol = (rt3 == 1) ? PI * 0.5 : 0.0;  // phonons.hpp:193-207
  p.type = (rt3 == 0) ? R3D_RAY_P : R3D_RAY_S;
  nudge(par, p.theta);
  p.loc = mk(m.source.loc);
  p.cell = m.source.cell;
  out.events[R3D_EV_GENERATED]++;
  report(c, 0, p);   // ReportNewEventPhonon, events.cpp:120

  // --- propagate ---
  while (true) {
    if (p.t > par.ttl) {
      out.n_timeout++;
      report(c, 6, p);   // ReportPhononTimeout, phonons.cpp:550
      return 2;
    }
    if ((p.moves % 128) == 127) {  // phonons.cpp:554-584
      if (std::isnan(p.path)) return report(c, 7, p), invalid(c, R3D_INV_PATH_NAN), 3;
      if (std::isnan(p.t)) return report(c, 7, p), invalid(c, R3D_INV_TIME_NAN), 3;
      if (p.path < 0) return report(c, 7, p), invalid(c, R3D_INV_PATH_NEGATIVE), 3;
      if ((p.t < 0) || (p.recent < 0)) return report(c, 7, p), invalid(c, R3D_INV_TIME_NEGATIVE), 3;
      if (p.recent == 0) return report(c, 7, p), invalid(c, R3D_INV_STUCK), 3;
      if (p.recent < par.slow_concern) return report(c, 7, p), invalid(c, R3D_INV_SLOW), 3;
      if (p.moves > par.loop_concern) return report(c, 7, p), invalid(c, R3D_INV_LOOP_EXCEED), 3;
      p.recent = 0;
    }
    out.events[R3D_EV_ITERATIONS]++;
    const r3d_cell& cell = m.cells[p.cell];
    TravelRec travel = path_to_boundary(m, cell, p);
    if (travel.len == INF) {
      out.n_timeout++;
      report(c, 6, p);   // phonons.cpp:596
      return 2;
    }
    const r3d_scatterer& sc = m.scatterers[cell.scatterer];
    // Scatterer::GetRandomPathLength, scatterers.cpp:297-307
    double scatlen = -std::log(draw(c)) * sc.mfp[p.type];
    if (scatlen < travel.len) {
      travel = advance_length(m, cell, p, scatlen);
      move(p, travel);
      // Scatterer::GetRandomScatteredRelativePhonon, scatterers.cpp:318-363
      double rth, rph, rpol;
      int rtype;
      if (par.no_deflect) {
        rth = 0, rph = 0, rtype = p.type, rpol = 0;
        nudge(par, rth);
      } else {
        double u_conv, u_dir;   // the event's two uniforms (philox.h: one block)
        draw_pair(c, u_conv, u_dir);
        int conv = (int)sample_cdf_u(sc.whole_cdf[p.type], 4, u_conv);  // GPP GPS GSP GSS
        rtype = (conv & 1) ? R3D_RAY_S : R3D_RAY_P;
        uint64_t k = sample_cdf_u(sc.cdf[conv], m.n_toa, u_dir);
        rpol = (conv == 3) ? sc.spol[k] : 0;
        rth = m.toa[2 * k], rph = m.toa[2 * k + 1];
        nudge(par, rth);
      }
      // Phonon::Transform, phonons.cpp:116-170
      Axes AA = make_axes(p.theta, p.phi, p.pol);
      Axes BB = make_axes(rth, rph, rpol);
      Axes SS = express(AA, BB);
      p.theta = SS.theta, p.phi = SS.phi, p.pol = SS.rot, p.type = rtype;
      out.events[R3D_EV_SCATTER]++;
      volume_count(p.t, p.loc, p.type);   // ReportScatterEvent
      report(c, 1, p);
      continue;
    }
    move(p, travel);
    const r3d_face& face = cell.faces[travel.face];
    if (face.flags & R3D_FACE_COLLECT) {
      report(c, 3, p);   // ReportPhononCollected: the incident state (phonons.cpp:630)
      collect(c, p);
    }
    if (face.flags & R3D_FACE_REFLECT) {
      refraction_full_rt(c, p, travel.face);
      out.events[R3D_EV_REFLECT]++;
      volume_count(p.t, p.loc, p.type);   // ReportReflection
      report(c, 2, p);
      continue;
    }
    if (face.flags & R3D_FACE_ADJOIN) {
      int old = p.cell;
      // Phonon::Refract, phonons.cpp:225-255
      if (face.flags & R3D_FACE_DISCON) refraction_full_rt(c, p, travel.face);
      else if (velocity_jump(m, cell, m.cells[face.neighbor], p.loc) > 0.00001)
        refraction_bend(c, p, travel.face);
      else p.cell = face.neighbor;
      out.events[p.cell == old ? R3D_EV_REFLECT : R3D_EV_TRANSFER]++;
      if (p.cell == old) volume_count(p.t, p.loc, p.type);   // ReportReflection
      report(c, p.cell == old ? 2 : 4, p);   // REF or CEL, phonons.cpp:659-661
      continue;
    }
    out.n_lost++;
    report(c, 5, p);   // ReportLostPhonon, phonons.cpp:675
    return 1;
  }
}

}  // namespace

extern "C" {

// Run histories [first_id, first_id+n) and ADD into *out (same contract as
// r3d_run in include/r3d.h).  finals may be NULL.
int r3d_oracle_run(const r3d_model_desc* model, uint64_t n, uint64_t first_id, uint64_t seed,
                   r3d_result* out, r3d_final* finals) {
  if (!model || !out || !out->energy || !out->counts) return 1;
  Ctx c;
  c.m = model;
  c.out = out;
  g_vol_outside = 0;
  for (uint64_t i = 0; i < n; i++) {
    oracle_rng_init(&c.rng, seed, first_id + i);
    c.id = first_id + i;
    c.n_catch = 0;
    Phonon p;
    int fate = run_history(c, p);
    if (finals) {
      r3d_final& f = finals[i];
      std::memset(&f, 0, sizeof f);
      f.time = p.t, f.path = p.path, f.amp = p.amp;
      f.loc[0] = p.loc.x, f.loc[1] = p.loc.y, f.loc[2] = p.loc.z;
      V d = from_angles(p.theta, p.phi);
      f.dir[0] = d.x, f.dir[1] = d.y, f.dir[2] = d.z;
      f.moves = p.moves;
      f.fate = (uint8_t)fate;
      f.type = (uint8_t)p.type;
      f.n_catch = (uint16_t)(c.n_catch > 65535 ? 65535 : c.n_catch);
    }
  }
  out->events[R3D_EV_VOLUME_OUT] += g_vol_outside;
  return 0;
}

// Attach (buf != NULL) or detach the per-event report buffer for subsequent runs (single
// threaded use); r3d_oracle_event_count() = events reported since it was attached.
void r3d_oracle_set_event_log(r3d_event* buf, uint64_t capacity, uint32_t mask) {
  g_evlog = buf, g_evlog_cap = buf ? capacity : 0, g_evlog_mask = buf ? mask : 0, g_evlog_count = 0;
}
uint64_t r3d_oracle_event_count(void) { return g_evlog_count; }

// Attach (v != NULL) or detach the volumetric scatter-event grid for subsequent runs.
void r3d_oracle_set_volume(const r3d_volume_desc* v, uint32_t* counters) {
  static r3d_volume_desc copy;
  if (v) copy = *v, g_vol_desc = &copy, g_vol = counters;
  else g_vol_desc = nullptr, g_vol = nullptr;
}

// Known-answer hook for tests: reflection/transmission probabilities of one
// interface for a given incidence sine (the quantities the reference's
// --rtcoef-test mission prints, rtcoef.cpp:687-742).  intype 0 P, 1 SH, 2 SV.
void r3d_oracle_rt_probs(double rho1, double a1, double b1, double rho2, double a2, double b2,
                         double sini, int intype, double probs[6]) {
  RT r;
  r.rhoR = rho1, r.velR[0] = a1, r.velR[1] = b1;
  r.rhoT = rho2, r.velT[0] = a2, r.velT[1] = b2;
  r.sini = sini;
  for (double& x : r.prob) x = 0;
  if (intype == 0) rt_coefs_psv(r, true);
  else if (intype == 1) rt_coefs_sh(r);
  else rt_coefs_psv(r, false);
  for (int i = 0; i < 6; i++) probs[i] = r.prob[i];
}

// Known-answer hook for tests: ONE reflection / transmission event on a bare interface -- media[6] = rho, alpha, beta of
// the reflection side, then of the transmission side (ignored without a neighbour: the free surface's vanishing medium,
// media_cellface.cpp:122-149), the face's outward unit normal, the phonon as the reference carries it (theta, phi,
// polarisation angle, type), the event's two uniforms.
//   -> out[8] = type, theta, phi, pol, transmitted (0 / 1), choice (R_P .. T_SH), margin of the outcome draw (fraction of
//      the weights' total), margin of the SH / SV draw (1 for a P ray)
void r3d_oracle_rt_event(const double media[6], int has_neighbor, const double normal[3], double theta, double phi,
                         double pol, int type, double u_pol, double u_out, double out[8]) {
  RT r;
  r.no_transmit = false;
  r.fnorm = mk(normal);
  r.rhoR = media[0], r.velR[0] = media[1], r.velR[1] = media[2];
  if (has_neighbor) r.rhoT = media[3], r.velT[0] = media[4], r.velT[1] = media[5];
  else r.rhoT = 0.0, r.velT[0] = r.velT[1] = 1e-12, r.no_transmit = true;
  Phonon p;
  p.theta = theta, p.phi = phi, p.pol = pol, p.type = type;
  int choice = 0;
  double margin = 1.0, pol_margin = 1.0;
  const bool crossed = rt_event_core(r, p, u_pol, u_out, &choice, &margin, &pol_margin);
  out[0] = p.type, out[1] = p.theta, out[2] = p.phi, out[3] = p.pol, out[4] = crossed ? 1 : 0, out[5] = choice;
  out[6] = margin, out[7] = pol_margin;
}

// Known-answer hook for tests: ONE Snell bend without conversion on a bare face (Phonon::Refraction_Bend,
// phonons.cpp:311-405): the face's outward unit normal, the phonon (theta, phi, polarisation angle, type), the ray's
// velocity on its own side and beyond the face.   -> out[4] = theta, phi, pol, crossed (0 / 1)
void r3d_oracle_bend_event(const double normal[3], double theta, double phi, double pol, int type, double veli, double velo,
                           double out[4]) {
  Phonon p;
  p.theta = theta, p.phi = phi, p.pol = pol, p.type = type;
  const bool crossed = bend_core(p, mk(normal), veli, velo);
  out[0] = p.theta, out[1] = p.phi, out[2] = p.pol, out[3] = crossed ? 1 : 0;
}

// Known-answer hook for tests: Phonon::Transform (phonons.cpp:116-170; OrthoAxes, geom_r3.cpp:212-340) -- a phonon
// (theta, phi, pol) rotated by a deflection (theta, phi, pol) given in its own frame.   -> out[3] = theta, phi, pol
void r3d_oracle_transform(double theta, double phi, double pol, double rth, double rph, double rpol, double out[3]) {
  Axes AA = make_axes(theta, phi, pol);
  Axes BB = make_axes(rth, rph, rpol);
  Axes SS = express(AA, BB);
  out[0] = SS.theta, out[1] = SS.phi, out[2] = SS.rot;
}

// Known-answer hooks for tests: one leg of ray geometry in a given cell.
//   r3d_oracle_advance: move `len` along the ray from (loc, theta, phi)
//       -> out[8] = new loc(3), new theta, new phi, travel time, attenuation, 0
//   r3d_oracle_boundary: MediumCell::GetPathToBoundary
//       -> out[8] = exit loc(3), theta, phi, time, path length, face id
void r3d_oracle_advance(const r3d_model_desc* m, int cell, int type, const double loc[3],
                        double theta, double phi, double len, double out[8]) {
  Phonon p;
  p.loc = mk(loc), p.theta = theta, p.phi = phi, p.type = type, p.cell = cell;
  TravelRec r = advance_length(*m, m->cells[cell], p, len);
  out[0] = r.loc.x, out[1] = r.loc.y, out[2] = r.loc.z, out[3] = r.theta, out[4] = r.phi;
  out[5] = r.time, out[6] = r.atten, out[7] = 0;
}
void r3d_oracle_boundary(const r3d_model_desc* m, int cell, int type, const double loc[3],
                         double theta, double phi, double out[8]) {
  Phonon p;
  p.loc = mk(loc), p.theta = theta, p.phi = phi, p.type = type, p.cell = cell;
  TravelRec r = path_to_boundary(*m, m->cells[cell], p);
  out[0] = r.loc.x, out[1] = r.loc.y, out[2] = r.loc.z, out[3] = r.theta, out[4] = r.phi;
  out[5] = r.time, out[6] = r.len, out[7] = r.face;
}

// Tetra::GetPathToBoundary (media.cpp:518-567) on ONE bare cell given as arrays -- four outward unit normals, a
// point on each face, grad v and v at the origin -- from `loc` along the unit vector `dir` (taken through
// (theta, phi), as the reference carries it).  Returns the exit face; *len: the arc length (inf: none).
// For tests/test_face_filter.py: the engine's local tetra move against this, case by case.
int r3d_oracle_tet_search(const double normals[12], const double points[12], const double g[3], double v0,
                          const double loc[3], const double dir[3], double* len) {
  r3d_cell c;
  std::memset(&c, 0, sizeof c);
  for (int f = 0; f < 4; f++)
    for (int k = 0; k < 3; k++) c.faces[f].normal[k] = normals[3 * f + k], c.faces[f].point[k] = points[3 * f + k];
  for (int k = 0; k < 3; k++) c.vel_grad[0][k] = g[k];
  c.vel_c[0] = v0;
  double th, ph;
  angles_from_node(mk(dir), th, ph);
  const V lc = mk(loc), gg = mk(c.vel_grad[0]);
  ArcFrame CT = arc_frame(dot(lc, gg) + v0, gg, lc, from_angles(th, ph));
  double angle0 = std::atan2(CT.prime_loc.x, CT.prime_loc.z);
  Gcad rv[4];
  for (int i = 0; i < 4; i++) rv[i] = plane_arc(c.faces[i], CT);
  double best = INF;
  int face = 0;
  for (int i = 0; i < 4; i++) {
    double e = rv[i].exit;
    if (gcad_inside(rv[(i + 1) % 4], e) && gcad_inside(rv[(i + 2) % 4], e) && gcad_inside(rv[(i + 3) % 4], e)) {
      double nl = (e - angle0) * CT.R;
      if (nl < 0 && (angle0 > rv[i].half)) nl = best;  // dismiss exit
      if (nl < best) best = nl, face = i;
    }
  }
  *len = best;
  return face;
}

// SphereShell::GetPathToBoundary (media.cpp:668-757) on ONE bare shell v = a r^2 + c (a < 0) between the radii
// r_top and r_bottom about the origin, from `loc` along the unit vector `dir`.  Returns the exit face (0 top,
// 1 bottom); *len: the arc length, squashed at zero as the reference does.  For tests/test_face_filter.py.
int r3d_oracle_shell_search(double a, double c0, double r_top, double r_bottom, const double loc[3], const double dir[3],
                            double* len) {
  r3d_model_desc m;
  std::memset(&m, 0, sizeof m);
  r3d_cell c;
  std::memset(&c, 0, sizeof c);
  c.vel_a[0] = a, c.vel_c[0] = c0, c.zero_rad2[0] = -c0 / a;
  c.faces[0].radius = r_top, c.faces[1].radius = -r_bottom;
  const V lc = mk(loc), d = mk(dir);
  RayArc A = shell_arc(m, c, 0, lc, d);
  double dt = sphere_arc_exit(c.faces[0], lc, d, A);
  double db = sphere_arc_exit(c.faces[1], lc, d, A);
  int ef = (dt < db) ? 0 : 1;
  double dist = ef == 0 ? dt : db;
  if (dist < 0) dist = 0;
  *len = dist;
  return ef;
}

void r3d_oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  oracle_philox4x32_10(ctr, key, out);
}
double r3d_oracle_draw(uint64_t seed, uint64_t id, uint32_t k) {
  oracle_rng g;
  oracle_rng_init(&g, seed, id);
  double v = 0;
  for (uint32_t i = 0; i <= k; i++) v = oracle_rng_draw(&g);
  return v;
}

}  // extern "C"
