// models_builtin.cpp -- compiled-in Earth-model definitions.
//
// Grid::ConstructGridManual(selection, args) is the hook through which the
// reference selects a model (reference user.cpp:59-134).  The reference keeps
// one hand-written function per model in user_*_inc.cpp; those files remain
// the user's and still compile against this library's grid.hpp.  What is
// built in here are table-driven definitions of every model the reference's
// dispatcher knows, with the same selection codes, argument lists and node
// values as the reference's functions:
//
//   40      halfspace, two layers          (user_Halfspace_inc.cpp:28-183)
//   1..4    Lop Nor layered crust+mantle   (user_LopNorCyl_inc.cpp:30-400); with 20 or more
//           arguments the variant with a Moho transition zone (user_LopNorCylMoho_inc.cpp)
//   21      Lop Nor, Moho transition, the more gradual velocity profile
//           (user_LopNorCylMoho2_inc.cpp; do-lopnor-vids.sh)
//   5..7    North Sea crust pinch, tetra   (user_NSCP_inc.cpp:13-206)
//   16      whole-Earth spherical shells   (user_SphereEarth_inc.cpp:13-86)
//   30      two-shell toy sphere           (user_ToySphere_inc.cpp; do-toysphere-vids.sh)
//   8       crust upthrust, tetra          (user_Upthrust_inc.cpp)
//   128     scattering-parameter study, 42 layers  (user.cpp:207-309)
//
// Where the reference prints to stderr and calls exit(1) on a bad argument
// count, these throw Runtime.
#include <algorithm>
#include <iostream>

#include "grid.hpp"

namespace {

using Elastic::HetSpec;
using Elastic::HSneak;
using Elastic::Q;
using Elastic::QmQk;
using Elastic::VpVs;

const Real kInf = std::numeric_limits<Real>::infinity();

[[noreturn]] void bad_arg_count() {
  throw Runtime("Error: wrong number of model args passed to compiled-in "
                "grid-building function.");
}

// One (nu, eps, a, kappa, Qs) group as it appears in --model-args.
struct Neakq {
  Real nu, eps, a, k, q;
  HetSpec hs() const { return HSneak(nu, eps, a, k); }
  Q qq() const { return QmQk(q); }
};
Neakq neakq_at(const std::vector<Real>& v, size_t o) {
  return {v.at(o), v.at(o + 1), v.at(o + 2), v.at(o + 3), v.at(o + 4)};
}

// A layered ("cylinder") grid is 3 plumb lines x nk sheets; attributes live
// on the first plumb line only (model.cpp:647-724 reads Node(0,0,k)).
void place_sheet(Grid& g, Index k, const Real xy[3][2], const Real z[3]) {
  for (Index p = 0; p < 3; p++) g.WNode(p, 0, k).SetLocation(xy[p][0], xy[p][1], z[p]);
}

// ---------------------------------------------------------------- 40 ------
void build_halfspace(Grid& g, const std::vector<Real>& a) {
  Neakq top{0.8, 0.05, 0.50, 0.5, kInf}, bot{0.8, 0.04, 1.0, 0.5, kInf};
  Real tv[4] = {6.40, 3.63, 2.83, -21.5};  // Vp, Vs, rho, z_bottom
  Real bv[4] = {6.40, 3.63, 2.83, -100.0};
  switch (a.size()) {
    case 0:
      break;
    case 18:
      for (int i = 0; i < 4; i++) tv[i] = a[10 + i], bv[i] = a[14 + i];
      [[fallthrough]];
    case 10:
      top = neakq_at(a, 0), bot = neakq_at(a, 5);
      break;
    default:
      bad_arg_count();
  }
  g.SetSize(3, 1, 3);
  g.SetIndexBase(0);
  const Real xy[3][2] = {{0, 0}, {100, 0}, {0, 100}};
  const Real depth[3] = {0.0, tv[3], bv[3]};
  for (Index k = 0; k < 3; k++) {
    const Real z[3] = {depth[k], depth[k], depth[k]};
    place_sheet(g, k, xy, z);
  }
  // The interface gets two attribute sets (=> full R/T) only when the layers
  // actually differ elastically.
  bool contrast = !(tv[0] == bv[0] && tv[1] == bv[1] && tv[2] == bv[2]);
  g.WNode(0, 0, 0).SetAttributes(VpVs(tv[0], tv[1]), tv[2], top.qq(), top.hs());
  if (contrast) g.WNode(0, 0, 1).SetAttributes(VpVs(tv[0], tv[1]), tv[2], top.qq(), top.hs());
  g.WNode(0, 0, 1).SetAttributes(VpVs(bv[0], bv[1]), bv[2], bot.qq(), bot.hs());
  g.WNode(0, 0, 2).SetAttributes(VpVs(bv[0], bv[1]), bv[2], bot.qq(), bot.hs());
}

// --------------------------------------------------------------- 1..4 -----
void build_lopnor(Grid& g, const std::vector<Real>& a) {
  Neakq sedi{0.8, 0.06, 0.25, 0.5, kInf}, crust{0.8, 0.05, 0.50, 0.5, kInf},
      mant{0.8, 0.04, 1.0, 0.5, kInf};
  Neakq* grp[3] = {&sedi, &crust, &mant};
  size_t per = 0;  // values per group: nu,eps,a[,k[,Q]]
  switch (a.size()) {
    case 0: break;
    case 9: per = 3; break;
    case 12: per = 4; break;
    case 15: per = 5; break;
    default: bad_arg_count();
  }
  for (size_t r = 0; r < 3 && per; r++) {
    const Real* v = &a[r * per];
    grp[r]->nu = v[0], grp[r]->eps = v[1], grp[r]->a = v[2];
    if (per > 3) grp[r]->k = v[3];
    if (per > 4) grp[r]->q = v[4];
  }

  g.SetSize(3, 1, 22);
  g.SetIndexBase(0);
  // Plumb lines under Lop Nor, station MAK, station WUS; the five crustal
  // sheets are tilted, the mantle sheets are level.
  const Real xy[3][2] = {{492.31, -263.65}, {-102.27, 430.84}, {-390.04, -167.18}};
  const Real crust_z[5][3] = {{1.050, 0.600, 1.457},
                              {0.563, 0.118, 0.963},
                              {-18.901, -16.743, -18.812},
                              {-38.365, -33.122, -38.587},
                              {-47.610, -43.720, -47.980}};
  const Real mantle_z[17] = {-80.0,  -120.0, -165.0, -210.0, -260.0, -310.0,
                             -360.0, -410.0, -460.0, -510.0, -560.0, -610.0,
                             -660.0, -710.0, -760.0, -809.5, -859.0};
  for (Index k = 0; k < 5; k++) place_sheet(g, k, xy, crust_z[k]);
  for (Index k = 0; k < 17; k++) {
    const Real z[3] = {mantle_z[k], mantle_z[k], mantle_z[k]};
    place_sheet(g, k + 5, xy, z);
  }

  // {sheet, region(0 sedi,1 crust,2 mantle), Vp, Vs, rho}; a sheet listed
  // twice is a first-order discontinuity (above, then below).
  struct Row { Index k; int region; Real vp, vs, rho; };
  static const Row rows[] = {
      {0, 0, 2.50, 1.20, 2.10},       {1, 1, 2.50, 1.20, 2.10},
      {1, 1, 6.13, 3.53, 2.75},       {2, 1, 6.40, 3.63, 2.83},
      {3, 1, 7.23, 4.00, 3.10},       {4, 1, 7.23, 4.00, 3.10},
      {4, 2, 8.07, 4.63, 3.35},       {5, 2, 8.040, 4.480, 3.502},
      {5, 2, 8.045, 4.490, 3.502},    {6, 2, 8.0505, 4.5000, 3.4268},
      {7, 2, 8.1750, 4.5090, 3.3711}, {8, 2, 8.3007, 4.5184, 3.3243},
      {9, 2, 8.4822, 4.6094, 3.3663}, {10, 2, 8.6650, 4.6964, 3.4110},
      {11, 2, 8.8476, 4.7832, 3.4577}, {12, 2, 9.0302, 4.8702, 3.5068},
      {12, 2, 9.3601, 5.0806, 3.9317}, {13, 2, 9.5280, 5.1864, 3.9273},
      {14, 2, 9.6962, 5.2922, 3.9233}, {15, 2, 9.8640, 5.3989, 3.9218},
      {16, 2, 10.0320, 5.5047, 3.9206}, {17, 2, 10.2000, 5.6104, 3.9201},
      {17, 2, 10.7909, 5.9607, 4.2387}, {18, 2, 10.9222, 6.0898, 4.2986},
      {19, 2, 11.0553, 6.2100, 4.3565}, {20, 2, 11.1355, 6.2424, 4.4118},
      {21, 2, 11.2228, 6.2799, 4.4650}};
  for (const Row& r : rows)
    g.WNode(0, 0, r.k).SetAttributes(VpVs(r.vp, r.vs), r.rho, grp[r.region]->qq(),
                                     grp[r.region]->hs());
}

// ------------------------------------------- 1..4 with >= 20 args, 21 -----
// Lop Nor with a Moho transition: 12 tilted sheets (sediments, crust, four "Moho
// complexity" steps, a four-step mantle ramp) over 17 level mantle sheets.  `gradual`
// selects the alternative velocity profile of model 21.
void build_lopnor_moho(Grid& g, const std::vector<Real>& a, bool gradual) {
  Neakq crust{0.8, 0.05, 0.50, 0.5, kInf};
  Neakq sedi{0.8, 0.06, 0.25, 0.5, kInf}, mant{0.8, 0.04, 1.0, 0.5, kInf}, moho = crust;
  Neakq* grp[4] = {&sedi, &crust, &mant, &moho};   // argument order
  size_t per = 0, groups = 3;  // values per group: nu,eps,a[,k[,Q]]
  switch (a.size()) {
    case 0: break;
    case 9: per = 3; break;
    case 12: per = 4; break;
    case 15: per = 5; break;
    case 20: per = 5, groups = 4; break;
    default: bad_arg_count();
  }
  for (size_t r = 0; r < groups && per; r++) {
    const Real* v = &a[r * per];
    grp[r]->nu = v[0], grp[r]->eps = v[1], grp[r]->a = v[2];
    if (per > 3) grp[r]->k = v[3];
    if (per > 4) grp[r]->q = v[4];
  }

  g.SetSize(3, 1, 29);
  g.SetIndexBase(0);
  const Real xy[3][2] = {{492.31, -263.65}, {-102.27, 430.84}, {-390.04, -167.18}};
  // the three sites' depths of the first three sheets; sheets 3..11 step down in 2 km
  // increments below each site's Moho depth
  const Real top_z[3][3] = {{1.050, 0.600, 1.457}, {0.563, 0.118, 0.963}, {-18.901, -16.743, -18.812}};
  const Real mantle_z[17] = {-80.0,  -120.0, -165.0, -210.0, -260.0, -310.0,
                             -360.0, -410.0, -460.0, -510.0, -560.0, -610.0,
                             -660.0, -710.0, -760.0, -809.5, -859.0};
  for (Index k = 0; k < 3; k++) place_sheet(g, k, xy, top_z[k]);
  for (Index k = 3; k < 12; k++) {
    static const Real lop[9] = {-38.365, -40.365, -42.365, -44.365, -46.365, -48.365, -50.365, -52.365, -54.365};
    static const Real mak[9] = {-33.122, -35.122, -37.122, -39.122, -41.122, -43.122, -45.122, -47.122, -49.122};
    static const Real wus[9] = {-38.587, -40.587, -42.587, -44.587, -46.587, -48.587, -50.587, -52.587, -54.587};
    const Real z[3] = {lop[k - 3], mak[k - 3], wus[k - 3]};
    place_sheet(g, k, xy, z);
  }
  for (Index k = 0; k < 17; k++) {
    const Real z[3] = {mantle_z[k], mantle_z[k], mantle_z[k]};
    place_sheet(g, k + 12, xy, z);
  }

  // {sheet, region (0 sedi, 1 crust, 2 mantle, 3 Moho zone), Vp, Vs, rho}; a sheet listed
  // twice is a first-order discontinuity (above, then below).
  struct Row { Index k; int region; Real vp, vs, rho; };
  static const Row head[] = {{0, 0, 2.50, 1.20, 2.10}, {1, 0, 2.50, 1.20, 2.10},
                             {1, 1, 6.13, 3.53, 2.75}, {2, 1, 6.40, 3.63, 2.83}};
  // the transition zone, sheets 3..11, in the two profiles
  static const Row steep[] = {
      {3, 1, 6.80, 3.83, 3.10},    {3, 3, 7.22, 4.01, 3.15},    {4, 3, 7.22, 4.01, 3.15},
      {4, 3, 7.13, 3.96, 3.13},    {5, 3, 7.13, 3.96, 3.13},    {5, 3, 7.43, 4.11, 3.22},
      {6, 3, 7.43, 4.11, 3.22},    {6, 3, 7.28, 4.00, 3.15},    {7, 3, 7.28, 4.00, 3.22},
      {7, 3, 8.000, 4.460, 3.502}, {8, 3, 8.000, 4.460, 3.502}, {8, 3, 8.010, 4.465, 3.502},
      {9, 3, 8.010, 4.465, 3.502}, {9, 3, 8.020, 4.470, 3.502}, {10, 3, 8.020, 4.470, 3.502},
      {10, 3, 8.030, 4.475, 3.502}, {11, 3, 8.030, 4.475, 3.502}};
  static const Row soft[] = {
      {3, 1, 6.40, 3.63, 3.10},    {3, 3, 7.081, 3.885, 3.15},  {4, 3, 7.081, 3.885, 3.15},
      {4, 3, 6.991, 3.835, 3.13},  {5, 3, 6.991, 3.835, 3.13},  {5, 3, 7.291, 3.985, 3.22},
      {6, 3, 7.291, 3.985, 3.22},  {6, 3, 7.141, 3.875, 3.15},  {7, 3, 7.141, 3.875, 3.22},
      {7, 3, 7.624, 4.183, 3.502}, {8, 3, 7.624, 4.183, 3.502}, {8, 3, 7.728, 4.257, 3.502},
      {9, 3, 7.728, 4.257, 3.502}, {9, 3, 7.832, 4.331, 3.502}, {10, 3, 7.832, 4.331, 3.502},
      {10, 3, 7.936, 4.406, 3.502}, {11, 3, 7.936, 4.406, 3.502}};
  static const Row tail[] = {
      {11, 2, 8.040, 4.480, 3.502},   {12, 2, 8.040, 4.480, 3.502},   {12, 2, 8.045, 4.490, 3.502},
      {13, 2, 8.0505, 4.5000, 3.4268}, {14, 2, 8.1750, 4.5090, 3.3711}, {15, 2, 8.3007, 4.5184, 3.3243},
      {16, 2, 8.4822, 4.6094, 3.3663}, {17, 2, 8.6650, 4.6964, 3.4110}, {18, 2, 8.8476, 4.7832, 3.4577},
      {19, 2, 9.0302, 4.8702, 3.5068}, {19, 2, 9.3601, 5.0806, 3.9317}, {20, 2, 9.5280, 5.1864, 3.9273},
      {21, 2, 9.6962, 5.2922, 3.9233}, {22, 2, 9.8640, 5.3989, 3.9218}, {23, 2, 10.0320, 5.5047, 3.9206},
      {24, 2, 10.2000, 5.6104, 3.9201}, {24, 2, 10.7909, 5.9607, 4.2387}, {25, 2, 10.9222, 6.0898, 4.2986},
      {26, 2, 11.0553, 6.2100, 4.3565}, {27, 2, 11.1355, 6.2424, 4.4118}, {28, 2, 11.2228, 6.2799, 4.4650}};
  auto put = [&](const Row& r) {
    g.WNode(0, 0, r.k).SetAttributes(VpVs(r.vp, r.vs), r.rho, grp[r.region]->qq(), grp[r.region]->hs());
  };
  for (const Row& r : head) put(r);
  if (gradual) for (const Row& r : soft) put(r);
  else for (const Row& r : steep) put(r);
  for (const Row& r : tail) put(r);
}

// ---------------------------------------------------------------- 30 ------
// Toy sphere: a "mantle" shell over a "core" ball, fixed scattering parameters.
void build_toy_sphere(Grid& g, const std::vector<Real>&) {
  const HetSpec hs = HSneak(0.8, 0.01, 4.00, 0.8);
  const Q q = QmQk(1000);
  g.SetSize(1, 1, 3);
  g.SetIndexBase(0);
  g.SetMapping(Grid::GC_RAE, Grid::GC_SPHERICAL);
  const Real depth[3] = {0, -4000.0, -6371.0};
  for (Index k = 0; k < 3; k++) g.WNode(0, 0, k).SetLocation(0, 0, depth[k]);
  struct Row { Index k; Real vp, vs, rho; };
  static const Row rows[] = {{0, 5.00, 2.60, 3.60}, {1, 9.00, 6.00, 4.00},
                             {1, 10.00, 8.00, 4.20}, {2, 14.00, 12.00, 4.90}};
  for (const Row& r : rows) g.WNode(0, 0, r.k).SetAttributes(VpVs(r.vp, r.vs), r.rho, q, hs);
}

// --------------------------------------------------------------- 5..7 -----
void build_crustpinch(Grid& g, const std::vector<Real>& a) {
  // Layer groups: sediments, crust, pinched crust, Moho transition, mantle.
  Neakq dflt{0.8, 0.01, 4.00, 0.8, 200};
  Neakq sedi = dflt, crust = dflt, pinch = dflt, moho = dflt, mant = dflt;
  Real sedi_thick = 2.0, crust_thick = 30.0, moho_thick = 10.0;
  Real pinch_frac = 0.70;  // pinched crust thickness / normal thickness
  Real depth_frac = 0.50;  // 0 top-aligned, 0.5 common midline, 1 bottom-aligned
  Real sedi_frac = 1.0, moho_frac = 1.0;
  switch (a.size()) {
    case 0:
      break;
    case 32:
      pinch_frac = a[28], depth_frac = a[29], sedi_frac = a[30], moho_frac = a[31];
      [[fallthrough]];
    case 28:
      sedi_thick = a[25], crust_thick = a[26], moho_thick = a[27];
      [[fallthrough]];
    case 25:
      sedi = neakq_at(a, 0), crust = neakq_at(a, 5), pinch = neakq_at(a, 10);
      moho = neakq_at(a, 15), mant = neakq_at(a, 20);
      break;
    default:
      bad_arg_count();
  }

  const Count nR = 14, nAz = 6, nZ = 8;
  const Real azi_far[nAz] = {45.0, 56.25, 78.75, 101.25, 123.75, 135.0};
  const Real azi_near[nAz] = {5.0, 39.00, 73.00, 107.00, 141.00, 175.0};
  const Real range[nR] = {-120, -60, 60, 120, 220, 310, 370, 420, 470, 530, 650, 770, 890, 1020};

  Real z_norm[nZ] = {0.0,
                     -sedi_thick,
                     -(sedi_thick + crust_thick),
                     -(sedi_thick + crust_thick + moho_thick),
                     -80, -120, -210, -360};
  Real pinch_thick = pinch_frac * crust_thick;
  Real pinch_top = -sedi_thick - (crust_thick - pinch_thick) * depth_frac;
  Real sedi_fill = std::max(sedi_frac * (-sedi_thick - pinch_top), Real(0));
  Real z_pinch[nZ];
  z_pinch[0] = pinch_top + (sedi_thick + sedi_fill);
  z_pinch[1] = pinch_top;
  z_pinch[2] = pinch_top - pinch_thick;
  z_pinch[3] = z_pinch[2] - moho_frac * moho_thick;
  for (Count k = 4; k < nZ; k++) z_pinch[k] = z_norm[k];

  g.SetSize(nR, nAz, nZ);
  g.SetIndexBase(0);
  g.SetMapping(Grid::GC_RAE, Grid::GC_CURVED);

  for (Index ja = 0; ja < nAz; ja++) {
    for (Index k = 0; k < nZ; k++)
      for (Index ir = 0; ir < nR; ir++) {
        // Behind the origin (negative range) the azimuth fan is mirrored;
        // the two columns nearest the origin use the wide close-range fan.
        Real az = (ir == 0)   ? azi_far[nAz - 1 - ja]
                  : (ir == 1) ? azi_near[nAz - 1 - ja]
                  : (ir == 2) ? azi_near[ja]
                              : azi_far[ja];
        bool pinched_col = (ir >= 6 && ir <= 8);
        g.WNode(ir, ja, k).SetLocation(range[ir], az, pinched_col ? z_pinch[k] : z_norm[k]);
      }
    for (Index ir = 0; ir < nR; ir++) {
      const Neakq& cr = (ir >= 6 && ir < 8) ? pinch : crust;
      struct Row { Index k; const Neakq* grp; Real vp, vs, rho; };
      const Row rows[] = {{0, &sedi, 4.50, 2.60, 2.20},  {1, &sedi, 4.52, 2.61, 2.21},
                          {1, &cr, 6.20, 3.58, 2.80},    {2, &cr, 6.24, 3.60, 2.82},
                          {2, &moho, 7.70, 4.44, 3.39},  {3, &mant, 8.00, 4.46, 3.40},
                          {4, &mant, 8.040, 4.48, 3.50}, {4, &mant, 8.045, 4.49, 3.50},
                          {5, &mant, 8.051, 4.50, 3.43}, {6, &mant, 8.301, 4.52, 3.32},
                          {7, &mant, 8.848, 4.78, 3.46}};
      for (const Row& r : rows)
        g.WNode(ir, ja, r.k).SetAttributes(VpVs(r.vp, r.vs), r.rho, r.grp->qq(), r.grp->hs());
    }
  }
}

// ----------------------------------------------------------------- 8 ------
// Crust upthrust: the NSCP fan (15 ranges x 6 azimuths) with ten sheets.  Across the six
// "active" range columns the Moho transition is thrust upwards (or downwards) by up to
// thrust_l on the near side and thrust_r on the far side, ramping as (step/2)^gamma; where
// the displaced transition overlaps the undisplaced one, two interpolated sheets carry
// the layering through.  The columns are built per active-region step m = 0..5 from an
// eight-sheet plumb line.
struct Mat {
  Real vp, vs, rho;
  const Neakq* grp;
};
struct PlumbSheet {
  Real z;
  Mat above, below;   // equal where the sheet is not a discontinuity
  bool jump;
};
struct ColumnSheet {
  Real z;
  Mat above, below;
  bool jump;
};
Mat blend(Real w_above, const Mat& a, const Mat& b, const Neakq* grp) {
  const Real w_below = 1 - w_above;
  return Mat{w_above * a.vp + w_below * b.vp, w_above * a.vs + w_below * b.vs,
             w_above * a.rho + w_below * b.rho, grp};
}
// Sheet n (0..9) of the column at active-region step m (0..5).
ColumnSheet upthrust_sheet(const PlumbSheet pl[8], Real thrust_l, Real thrust_r, Real gamma, Index m,
                           Index n) {
  auto plain = [&](Index k, Real dz) {
    return ColumnSheet{pl[k].z + dz, pl[k].above, pl[k].below, pl[k].jump};
  };
  if (n < 2 || n > 5) return plain(n > 5 ? n - 2 : n, 0);   // above and below the transition
  const bool near_side = m < 3;
  Real frac = near_side ? ((Real)m) / 2.0 : ((Real)(5 - m)) / 2.0;
  frac = std::pow(frac, gamma);
  const Real shift = near_side ? thrust_l * frac : thrust_r * frac;
  // the displaced copy of the transition's two sheets: below the originals on the near
  // side (sheets 4, 5), above them on the far side (sheets 2, 3)
  if ((n < 4) != near_side) {
    ColumnSheet c = plain(n < 4 ? n : n - 2, 0);
    c.z += shift;
    return c;
  }
  // the other two sheets interpolate between the layer above and the layer below
  const Real thick = pl[2].z - pl[3].z;
  const Real z_above = near_side ? pl[1].z : pl[3].z + shift;
  const Real z_below = near_side ? pl[2].z + shift : pl[4].z;
  const Real z_mid = (z_above + z_below) / 2;
  Real z;
  if (m == 2) z = pl[n].z + thrust_r;
  else if (m == 3) z = pl[n - 2].z + thrust_l;
  else z = z_mid + ((n == 2 || n == 4) ? 0.5 * thick : -0.5 * thick);
  const Mat& top = near_side ? pl[1].below : pl[3].below;
  const Mat& bottom = near_side ? pl[2].above : pl[4].above;
  const Real w_above = (z_below - z) / (z_below - z_above);
  const Mat mixed = blend(w_above, top, bottom, top.grp);
  return ColumnSheet{z, mixed, mixed, false};
}

void build_upthrust(Grid& g, const std::vector<Real>& a) {
  Neakq dflt{0.8, 0.01, 4.00, 0.8, 200};
  Neakq sedi = dflt, crust = dflt, active = dflt, moho = dflt, mant = dflt;
  Real sedi_thick = 2.0, crust_thick = 30.0, moho_thick = 10.0;
  Real thrust_l = -10, thrust_r = 0, gamma = 1.0, shear = 0;
  switch (a.size()) {
    case 0:
      break;
    case 32:
      thrust_l = a[28], thrust_r = a[29], gamma = a[30], shear = a[31];
      [[fallthrough]];
    case 28:
      sedi_thick = a[25], crust_thick = a[26], moho_thick = a[27];
      [[fallthrough]];
    case 25:
      sedi = neakq_at(a, 0), crust = neakq_at(a, 5), active = neakq_at(a, 10);
      moho = neakq_at(a, 15), mant = neakq_at(a, 20);
      break;
    default:
      bad_arg_count();
  }

  const Count nR = 15, nAz = 6, nZ = 10;
  const Index first_active = 5;
  const Count n_active = 6;
  const Real azi_far[nAz] = {45.0, 56.25, 78.75, 101.25, 123.75, 135.0};
  const Real azi_near[nAz] = {5.0, 39.00, 73.00, 107.00, 141.00, 175.0};
  const Real range[nR] = {-120, -60, 60, 120, 220, 310, 365, 418, 422, 475, 530, 650, 770, 890, 1020};

  auto plumb = [&](const Neakq* cr, PlumbSheet pl[8]) {
    const Mat se0{4.50, 2.60, 2.20, &sedi}, se1{4.52, 2.61, 2.21, &sedi};
    const Mat cr1{6.20, 3.58, 2.80, cr}, cr2{6.24, 3.60, 2.82, cr}, mo2{7.70, 4.44, 3.39, &moho};
    const Mat ma3{8.00, 4.46, 3.40, &mant}, ma4a{8.040, 4.48, 3.50, &mant}, ma4b{8.045, 4.49, 3.50, &mant};
    const Mat ma5{8.051, 4.50, 3.43, &mant}, ma6{8.301, 4.52, 3.32, &mant}, ma7{8.848, 4.78, 3.46, &mant};
    const PlumbSheet sheets[8] = {{0.0, se0, se0, false},
                                  {-sedi_thick, se1, cr1, true},
                                  {-(sedi_thick + crust_thick), cr2, mo2, true},
                                  {-(sedi_thick + crust_thick + moho_thick), ma3, ma3, false},
                                  {-80, ma4a, ma4b, true},
                                  {-120, ma5, ma5, false},
                                  {-210, ma6, ma6, false},
                                  {-360, ma7, ma7, false}};
    for (int k = 0; k < 8; k++) pl[k] = sheets[k];
  };
  PlumbSheet pl_plain[8], pl_active[8];
  plumb(&crust, pl_plain);
  plumb(&active, pl_active);

  g.SetSize(nR, nAz, nZ);
  g.SetIndexBase(0);
  g.SetMapping(Grid::GC_RAE, Grid::GC_CURVED);
  for (Index ir = 0; ir < nR; ir++) {
    Index m = (ir > first_active) ? ir - first_active : 0;
    m = (m < n_active) ? m : n_active - 1;
    const bool in_active = (ir >= first_active && ir < first_active + n_active - 1);
    const PlumbSheet* pl = in_active ? pl_active : pl_plain;
    Real tl = thrust_l, tr = thrust_r, sh = shear;
    if (thrust_r < thrust_l) {          // thrust from the far side: mirror the ramp
      m = (n_active - 1) - m;
      tl = thrust_r, tr = thrust_l, sh = -shear;
    } else if (thrust_r == thrust_l) {
      m = 0, sh = 0;
    }
    for (Index k = 0; k < nZ; k++) {
      const ColumnSheet c = upthrust_sheet(pl, tl, tr, gamma, m, k);
      for (Index ja = 0; ja < nAz; ja++) {
        const Real az = (ir == 0)   ? azi_far[nAz - 1 - ja]
                        : (ir == 1) ? azi_near[nAz - 1 - ja]
                        : (ir == 2) ? azi_near[ja]
                                    : azi_far[ja];
        GridNode& node = g.WNode(ir, ja, k);
        node.SetLocation(range[ir], az, c.z);
        node.SetAttributes(VpVs(c.above.vp, c.above.vs), c.above.rho, c.above.grp->qq(), c.above.grp->hs());
        if (c.jump)
          node.SetAttributes(VpVs(c.below.vp, c.below.vs), c.below.rho, c.below.grp->qq(), c.below.grp->hs());
        // shear across the thrust: the transition's sheets slide in range at the two
        // middle steps of the ramp
        if ((m == 2 || m == 3) && (k == 2 || k == 3)) node.AdjustLocation(-sh, 0, 0);
        else if ((m == 2 || m == 3) && (k == 4 || k == 5)) node.AdjustLocation(+sh, 0, 0);
      }
    }
  }
}

// --------------------------------------------------------------- 128 ------
// Scattering-parameter study: 42 layers 10 km thick, each varying one of nu, eps, a, kappa,
// the S wavenumber (through the velocity scale) or the Vp/Vs ratio about a common default.
void build_scat_params_study(Grid& g, const std::vector<Real>&) {
  const Real nu = 0.8, eps = 0.05, corr = 1.0, kap = 0.3, vs = 4.0, rho = 1.0;
  const Real vp = vs * 1.7321;
  const Q q = QmQk(kInf);
  const Count per = 7, sheets = per * 6 + 1;
  g.SetSize(3, 1, sheets);
  g.SetIndexBase(0);
  for (Index k = 0; k < sheets; k++) {
    const Real depth = -1.0 * 10 * k;
    g.WNode(0, 0, k).SetLocation(0.0, 0.0, depth);
    g.WNode(1, 0, k).SetLocation(1.0, 0.0, depth);
    g.WNode(2, 0, k).SetLocation(0.0, 1.0, depth);
  }
  const Real nus[per] = {0.3, 0.5, 0.7, 0.9, 1.1, 1.3, 1.5};
  const Real epss[per] = {0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07};
  const Real corrs[per] = {0.3, 0.6, 0.8, 1.0, 1.2, 1.5, 1.8};
  const Real kaps[per] = {0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.8};
  const Real scales[per] = {1.5, 1.3, 1.1, 1.0, 0.9, 0.7, 0.5};
  const Real ratios[per] = {1.3321, 1.5321, 1.6321, 1.7321, 1.8321, 1.9321, 2.1321};
  Index k = 0;
  for (Real v : nus) g.WNode(0, 0, k++).SetAttributes(VpVs(vp, vs), rho, q, HSneak(v, eps, corr, kap));
  for (Real v : epss) g.WNode(0, 0, k++).SetAttributes(VpVs(vp, vs), rho, q, HSneak(nu, v, corr, kap));
  for (Real v : corrs) g.WNode(0, 0, k++).SetAttributes(VpVs(vp, vs), rho, q, HSneak(nu, eps, v, kap));
  for (Real v : kaps) g.WNode(0, 0, k++).SetAttributes(VpVs(vp, vs), rho, q, HSneak(nu, eps, corr, v));
  const HetSpec hs = HSneak(nu, eps, corr, kap);
  for (Real v : scales) g.WNode(0, 0, k++).SetAttributes(VpVs(v * vp, v * vs), rho, q, hs);
  for (Real v : ratios) g.WNode(0, 0, k++).SetAttributes(VpVs(v * vs, vs), rho, q, hs);
  g.WNode(0, 0, k++).SetAttributes(VpVs(vp, vs), rho, q, hs);
}

// ---------------------------------------------------------------- 16 ------
void build_sphere_earth(Grid& g, const std::vector<Real>&) {
  const Real q = 2000;
  const HetSpec hs_std = HSneak(0.8, 0.005, 4.00, 0.8);
  const HetSpec hs_outer_core = HSneak(0.8, 0.005, 8.00, 0.8);
  const Q q_crust = QmQk(0.8 * q), q_mantle = QmQk(q), q_outer_core = QmQk(1, 20 * q),
          q_inner_core = QmQk(q);

  const Real depth[16] = {0,     -100.0,  -410.0,  -660.0,  -958.0,  -1354.0, -2047.0, -2789.0,
                          -2891.0, -3594.0, -4298.0, -4852.0, -5153.0, -5661.0, -6066.0, -6371.0};
  g.SetSize(1, 1, 16);
  g.SetIndexBase(0);
  g.SetMapping(Grid::GC_RAE, Grid::GC_SPHERICAL);
  for (Index k = 0; k < 16; k++) g.WNode(0, 0, k).SetLocation(0, 0, depth[k]);

  // region: 0 crust, 1 mantle, 2 outer core (liquid: Vs = 1e-5), 3 inner core
  struct Row { Index k; int region; Real vp, vs, rho; };
  static const Row rows[] = {
      {0, 0, 5.80, 3.20, 2.60},    {1, 0, 6.80, 3.90, 2.92},    {1, 1, 8.04, 4.48, 3.64},
      {2, 1, 9.03, 4.87, 3.51},    {2, 1, 9.36, 5.08, 3.93},    {3, 1, 10.20, 5.61, 3.92},
      {3, 1, 10.79, 5.96, 4.24},   {4, 1, 11.39, 6.35, 5.60},   {5, 1, 11.99, 6.60, 4.57},
      {6, 1, 12.85, 6.94, 5.13},   {7, 1, 13.65, 7.26, 5.72},   {8, 1, 13.66, 7.28, 5.77},
      {8, 2, 8.00, 1e-5, 9.91},    {9, 2, 9.08, 1e-5, 10.9},    {10, 2, 9.79, 1e-5, 11.6},
      {11, 2, 10.17, 1e-5, 12.0},  {12, 2, 10.29, 1e-5, 12.1},  {12, 3, 11.04, 3.50, 12.7},
      {13, 3, 11.18, 3.61, 12.9},  {14, 3, 11.25, 3.66, 13.0},  {15, 3, 11.26, 3.67, 13.0}};
  const Q* qs[4] = {&q_crust, &q_mantle, &q_outer_core, &q_inner_core};
  for (const Row& r : rows)
    g.WNode(0, 0, r.k).SetAttributes(VpVs(r.vp, r.vs), r.rho, *qs[r.region],
                                     r.region == 2 ? hs_outer_core : hs_std);
}

}  // namespace

// Weak so that a program linking the reference's own user.cpp (which defines
// this same member) overrides the built-in dispatcher.
__attribute__((weak)) void Grid::ConstructGridManual(int Selection,
                                                     const std::vector<Real>& args) {
  const char* head = "|  ConstructGridManual: ";
  if (Selection == 0) {
    Selection = 5;
    std::cout << head << "Changing selection 0 (no selection) to selection " << Selection
              << ".\n";
  }
  switch (Selection) {
    case 1: case 2: case 3: case 4:
      if (args.size() >= 20) {
        std::cout << head << "Selected Lop Nor Moho Model (Layered).\n";
        build_lopnor_moho(*this, args, false);
      } else {
        std::cout << head << "Selected Lop Nor Baseline Model (Layered).\n";
        build_lopnor(*this, args);
      }
      break;
    case 21:
      std::cout << head << "Selected Lop Nor Moho Model Alt2 (Layered).\n";
      build_lopnor_moho(*this, args, true);
      break;
    case 30:
      std::cout << head << "Selected Spherical Toy Model (Spherical).\n";
      build_toy_sphere(*this, args);
      break;
    case 8:
      std::cout << head << "Selected Crust Upthrust Model (Tetra WCG).\n";
      build_upthrust(*this, args);
      break;
    case 128:
      std::cout << head << "Selected Scatter Params Study (Layered).\n";
      build_scat_params_study(*this, args);
      break;
    case 5: case 6: case 7:
      std::cout << head << "Selected North Sea Crust Pinch Model (Tetra WCG).\n";
      build_crustpinch(*this, args);
      break;
    case 16:
      std::cout << head << "Selected Spherical Earth Model (Spherical).\n";
      build_sphere_earth(*this, args);
      break;
    case 40:
      std::cout << head << "Selected Halfspace Model (Layered).\n";
      build_halfspace(*this, args);
      break;
    default: {
      std::cout << head << "Model selection code not known.\n";
      TextStream err;
      err << "Unknown compiled grid selection ID " << Selection << ".";
      throw Runtime(err.str());
    }
  }
}
