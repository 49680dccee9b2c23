// model.cpp -- see model.hpp.  Each builder cites the reference code whose
// behaviour it reproduces.
#include "model.hpp"

#include <algorithm>
#include <cstring>
#include <functional>
#include <iomanip>
#include <iostream>
#include <thread>

namespace {

const Real kInf = std::numeric_limits<Real>::infinity();

void put3(double dst[3], const R3::XYZ& v) { dst[0] = v.x(), dst[1] = v.y(), dst[2] = v.z(); }
R3::XYZ get3(const double s[3]) { return {s[0], s[1], s[2]}; }

r3d_face blank_face() {
  r3d_face f;
  std::memset(&f, 0, sizeof f);
  f.neighbor = -1;
  return f;
}

// Plane through n1,n2,n3, outward = side from which n1->n2->n3 runs
// counter-clockwise (reference media_cellface.cpp:174-190).
r3d_face plane_face(const R3::XYZ& n1, const R3::XYZ& n2, const R3::XYZ& n3) {
  r3d_face f = blank_face();
  R3::XYZ nrm = n1.VectorTo(n2).Cross(n1.VectorTo(n3));
  nrm.Normalize();
  put3(f.normal, nrm);
  put3(f.point, n1);
  return f;
}
// Same plane, oriented AWAY from the excluded fourth node
// (reference media_cellface.cpp:196-214).
r3d_face plane_face(const R3::XYZ& n1, const R3::XYZ& n2, const R3::XYZ& n3,
                    const R3::XYZ& excluded) {
  r3d_face f = plane_face(n1, n2, n3);
  if (get3(f.normal).Dot(n1.VectorTo(excluded)) > 0) put3(f.normal, get3(f.normal).Negative());
  return f;
}

r3d_cell blank_cell() {
  r3d_cell c;
  std::memset(&c, 0, sizeof c);
  c.scatterer = -1;
  for (auto& f : c.faces) f = blank_face();
  return c;
}

enum FaceShape { PLANE, CYLWALL, SPHERE };
FaceShape shape_of(int cell_kind, int face) {
  if (cell_kind == R3D_CELL_SPHERESHELL) return SPHERE;
  if (cell_kind == R3D_CELL_CYLINDER && face == 2) return CYLWALL;
  return PLANE;
}

// Signed distance "above" a face, positive outside
// (reference media_cellface.cpp:231-234, :473-476, :638-641).
Real distance_above(const r3d_face& f, FaceShape s, const R3::XYZ& p) {
  switch (s) {
    case PLANE: return get3(f.normal).Dot(get3(f.point).VectorTo(p));
    case CYLWALL: return std::sqrt(p.x() * p.x() + p.y() * p.y()) - f.radius;
    default: return f.radius > 0 ? p.Mag() - f.radius : -(f.radius + p.Mag());
  }
}

// Straight-ray distance to the exit through a face, with the reference's
// +/-inf sentinels (media_cellface.cpp:262-324 plane, :664-684 sphere).  Only
// the host-side surface finder needs it; the hot-path copies live in the
// engine and in the oracle.
Real linear_exit_distance(const r3d_face& f, FaceShape s, const R3::XYZ& loc, const R3::XYZ& dir) {
  if (s == PLANE) {
    Real d_sh = get3(f.normal).Dot(loc.VectorTo(get3(f.point)));
    Real d_fact = get3(f.normal).Dot(dir);
    if (d_fact < 0) return kInf;
    if (d_fact == 0) return d_sh < 0 ? -kInf : kInf;
    return d_sh / d_fact;
  }
  if (s == SPHERE) {
    bool outward = f.radius > 0;
    Real midpt = -loc.Dot(dir);
    Real urad = f.radius * f.radius + midpt * midpt - loc.MagSquared();
    if (urad <= 0) return outward ? -kInf : kInf;
    Real sq = std::sqrt(urad);
    if (outward) return midpt + sq;
    if (midpt <= 0) return kInf;
    return midpt - sq;
  }
  throw Invalid("linear_exit_distance: cylinder walls are never surface faces");
}

// Interface bookkeeping of CellFace::LinkTo (reference
// media_cellface.cpp:46-70): both faces learn their neighbour and agree on
// the discontinuity flag.
void link_faces(std::vector<r3d_cell>& cells, int ca, int fa, int cb, int fb, bool disc) {
  r3d_face& A = cells[ca].faces[fa];
  r3d_face& B = cells[cb].faces[fb];
  A.neighbor = cb, B.neighbor = ca;
  A.flags |= R3D_FACE_ADJOIN, B.flags |= R3D_FACE_ADJOIN;
  if (disc) A.flags |= R3D_FACE_DISCON, B.flags |= R3D_FACE_DISCON;
  else A.flags &= ~R3D_FACE_DISCON, B.flags &= ~R3D_FACE_DISCON;
}
void link_faces_keep_flag(std::vector<r3d_cell>& cells, int ca, int fa, int cb, int fb) {
  bool disc = ((cells[ca].faces[fa].flags | cells[cb].faces[fb].flags) & R3D_FACE_DISCON) != 0;
  link_faces(cells, ca, fa, cb, fb, disc);
}

// Solve the 4x4 system [x y z 1] c = v for the linear field through four
// nodes (reference media.cpp:366-389 via geom_r4.cpp:15-66).
void fit_linear(const R3::XYZ n[4], const Real v[4], Real grad[3], Real& at_origin) {
  Real a[4][5];
  for (int r = 0; r < 4; r++) {
    a[r][0] = n[r].x(), a[r][1] = n[r].y(), a[r][2] = n[r].z(), a[r][3] = 1, a[r][4] = v[r];
  }
  for (int c = 0; c < 4; c++) {
    int piv = c;
    for (int r = c + 1; r < 4; r++)
      if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    if (piv != c) std::swap(a[piv], a[c]);
    if (a[c][c] == 0) continue;
    Real inv = 1 / a[c][c];
    for (int k = c; k < 5; k++) a[c][k] *= inv;
    for (int r = 0; r < 4; r++) {
      if (r == c) continue;
      Real f = a[r][c];
      for (int k = c; k < 5; k++) a[r][k] -= f * a[c][k];
    }
  }
  grad[0] = a[0][4], grad[1] = a[1][4], grad[2] = a[2][4], at_origin = a[3][4];
}

void parallel_for(size_t n, const std::function<void(size_t, size_t)>& body) {
  unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
  if (n < 65536 || nt == 1) return body(0, n);
  std::vector<std::thread> pool;
  size_t chunk = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; t++) {
    size_t lo = t * chunk, hi = std::min(n, lo + chunk);
    if (lo < hi) pool.emplace_back(body, lo, hi);
  }
  for (auto& th : pool) th.join();
}

void integrate(std::vector<double>& v) {  // reference probability.cpp:21-35
  for (size_t i = 1; i < v.size(); i++) v[i] += v[i - 1];
}

}  // namespace

// ------------------------------------------------------------ ModelParams --
// reference model.cpp:30-86
static ModelParams::SeisRequest make_request(EarthCoords::Generic loc,
                                             ModelParams::axes_scheme_e ax, Real r_in, Real r_out,
                                             bool wavelengths) {
  ModelParams::SeisRequest r;
  r.Location = loc;
  r.Orientation = ax;
  r.GatherRadiusInner[0] = r.GatherRadiusInner[1] = r_in;
  r.GatherRadiusOuter[0] = r.GatherRadiusOuter[1] = r_out;
  r.RadiiUnitsAreWavelengths = wavelengths;
  return r;
}
void ModelParams::AddSeismometerByWavelength(EarthCoords::Generic loc, axes_scheme_e ax, Real wl) {
  mSReqList.push_back(make_request(loc, ax, 0, wl, true));
}
void ModelParams::AddSeismometerFixedRadius(EarthCoords::Generic loc, axes_scheme_e ax, Real r) {
  mSReqList.push_back(make_request(loc, ax, 0, r, false));
}
void ModelParams::AddSeismometerRing(EarthCoords::Generic loc, axes_scheme_e ax, Real r_in,
                                     Real r_out) {
  mSReqList.push_back(make_request(loc, ax, r_in, r_out, false));
}

// ------------------------------------------------------------------ Model --
struct Model::ScatStore {
  std::vector<double> cdf[4];
  std::vector<double> spol;
};

Model::~Model() = default;

Model::Model(const ModelParams& par, std::ostream* logp) {
  std::ostream& log = logp ? *logp : std::cout;
  log << "@@ __BEGIN_MODEL_INITIALIZATION__" << std::endl;
  mNumPhonons = par.NumPhonons;
  mOverrideMFP = par.OverrideMFP;
  mNoDeflect = par.NoDeflect;
  mDeviceTables = par.DeviceTables;
  mMFPOverride[0] = par.MFPOverride[0], mMFPOverride[1] = par.MFPOverride[1];

  // The coordinate system is process-global in the reference; start clean.
  ECS.Reset();
  ECS.SetEarthRadius(par.EarthRadius);
  ECS.SetEarthFlattening(par.Flatten);
  if (par.OcsRaw) ECS.SetOCSMapping(EarthCoords::OUT_NOTRANSFORM);

  log << "@@ __DISCRETIZING_TOA__" << std::endl;
  if (par.TOA_Degree < 0 || par.TOA_Degree > 12) throw Runtime("TOA degree out of range [0,12].");
  mNumTOA = (size_t)20 << (2 * par.TOA_Degree);
  if (!mDeviceTables) {   // (with --device-tables the engine generates the set itself, in HBM)
    mTOA = S2::TesselSphereIco(par.TOA_Degree);
    mTOAFlat.resize(mTOA.size() * 2);
    for (size_t i = 0; i < mTOA.size(); i++)
      mTOAFlat[2 * i] = mTOA[i].theta, mTOAFlat[2 * i + 1] = mTOA[i].phi;
  }
  log << "|\n|" << std::setw(8) << mNumTOA
      << "  Take-off angles initialized for event and scattering sources.\n"
      << "|          (TesselSphere of degree " << par.TOA_Degree << ".)\n|\n";

  ScatterParams::SetFrequencyHertz(par.Frequency);

  log << "@@ __CONSTRUCTING_GRID__" << std::endl << "|\n";
  switch (par.GridSource) {
    case ModelParams::GRID_UNSPEC:
      log << "|  Grid source unspecified. Assuming user-compiled with index 0.\n";
      mGrid.ConstructGridManual(0, par.CompiledArgs);
      break;
    case ModelParams::GRID_COMPILED:
      log << "|  Grid source: User-coded grid; Selection ID = " << par.CompiledSelector
          << " with " << par.CompiledArgs.size() << " args.\n";
      mGrid.ConstructGridManual(par.CompiledSelector, par.CompiledArgs);
      break;
    default: {
      TextStream msg;
      msg << "No handler exists for Grid Source '" << par.GridSource << "'.";
      throw Runtime(msg.str());
    }
  }
  log << "|\n|  " << std::setw(6) << mGrid.N() << "  Grid nodes initialized.  Arrangement: "
      << mGrid.Ni() << " x " << mGrid.Nj() << " x " << mGrid.Nk() << "\n|\n";
  if (!ECS.IsEarthFlattening() && !ECS.CurvedCoords())
    log << "|  Coordinate mapping is rectilinear.\n";
  if (ECS.IsEarthFlattening())
    log << "|  An Earth-flattening transformation was used with Earth Radius = "
        << ECS.GetEarthRadius() << "\n";
  if (ECS.CurvedCoords())
    log << "|  Level planes were curved with Earth Radius = " << ECS.GetEarthRadius() << "\n";
  log << "|\n";

  log << "@@ __STAGING_EARTH_MODEL_ONTO_GRID__" << std::endl << "|\n";
  const char* wait = par.TOA_Degree > 7 ? "|  (This may take a while.)\n" : "";
  switch (mGrid.GetModelType()) {
    case Grid::MOD_CYLINDER:
      log << "|  Building tilted-interface LAYERED model from plumbline grid... \n" << wait
          << std::flush;
      mDesc.cell_kind = R3D_CELL_CYLINDER;
      BuildCellArray_Cylinder(par.CylinderRange);
      break;
    case Grid::MOD_TETRAWCG:
      log << "|  Building TETRA model from Warped Cartesian Grid... \n" << wait << std::flush;
      mDesc.cell_kind = R3D_CELL_TETRA;
      BuildCellArray_WCGTetra();
      break;
    case Grid::MOD_SPHERESHELL:
      log << "|  Building SPHERICAL SHELL model... \n" << wait << std::flush;
      mDesc.cell_kind = R3D_CELL_SPHERESHELL;
      BuildCellArray_SphericalShells();
      break;
    default:
      throw Runtime("No handler exists for this Model type.");
  }
  log << "|\n|" << std::setw(8) << mCells.size() << "  Model cells constructed.\n|"
      << std::setw(8) << mScatStore.size() << "  Scatterer objects allocated among cells.\n|\n";

  log << "@@ __INITIALIZING_EVENT_SOURCE__" << std::endl;
  BuildSource(par);
  log << "@@ __INITIALIZING_SEISMOMETERS__" << std::endl;
  BuildSeismometers(par);

  // ---- flat description ---------------------------------------------------
  mScatDesc.resize(mScatStore.size());
  for (size_t s = 0; s < mScatStore.size(); s++) {
    r3d_scatterer& d = mScatDesc[s];
    d = r3d_scatterer{};
    for (int t = 0; t < 2; t++) d.mfp[t] = mScatInfo[s].mfp[t];
    const ScatterParams& sp = mScatParams[s];
    const double het[6] = {sp.GetNu(), sp.GetEps(), sp.GetA(), sp.GetKappa(), sp.GetL(), sp.GetGam0()};
    for (int k = 0; k < 6; k++) d.het[k] = het[k];
    d.psdf_numer = sp.GetPsdfNumer();
    d.mfp_fixed = mOverrideMFP ? 1u : 0u;
    if (mDeviceTables) continue;   // build-on-device form: no host tables
    for (int k = 0; k < 4; k++) d.cdf[k] = mScatStore[s]->cdf[k].data();
    d.spol = mScatStore[s]->spol.data();
    // reference scatterers.cpp:172-184: an incoming P can only go to GPP/GPS,
    // an incoming S only to GSP/GSS; stored cumulatively.
    const double tot[4] = {mScatStore[s]->cdf[0].back(), mScatStore[s]->cdf[1].back(),
                           mScatStore[s]->cdf[2].back(), mScatStore[s]->cdf[3].back()};
    const double wp[2][4] = {{tot[0], tot[1], 0, 0}, {0, 0, tot[2], tot[3]}};
    for (int in = 0; in < 2; in++) {
      double acc = 0;
      for (int k = 0; k < 4; k++) d.whole_cdf[in][k] = (acc += wp[in][k]);
    }
  }
  mDesc.n_cells = (int32_t)mCells.size();
  mDesc.cells = mCells.data();
  mDesc.n_scatterers = (int32_t)mScatDesc.size();
  mDesc.scatterers = mScatDesc.data();
  mDesc.n_seismometers = (int32_t)mSeis.size();
  mDesc.seismometers = mSeis.data();
  mDesc.n_toa = mNumTOA;
  mDesc.toa = mDeviceTables ? nullptr : mTOAFlat.data();
  mDesc.toa_degree = par.TOA_Degree;

  r3d_params& p = mDesc.params;
  p.ttl = par.PhononTTL;
  p.frequency = par.Frequency;
  p.time_per_bin = par.GetBinSize();
  p.n_bins = (uint32_t)std::floor(par.PhononTTL / par.GetBinSize());  // dataout.hpp:178-181
  p.no_deflect = mNoDeflect ? 1u : 0u;
  p.min_theta = 0.0000001;  // phonons.cpp:34-35
  p.max_theta = Geometry::Pi - 0.0000001;
  p.slow_concern = 0.001;   // phonons.cpp:33
  p.loop_concern = 1048576; // phonons.cpp:32
  R3::XYZ ec = ECS.CurvedCoords() ? ECS.GetEarthCenter() : R3::XYZ(0, 0, 0);
  put3(p.earth_center, ec);
  mMapCode = (int)ECS.Mapping(), mRadE = ECS.GetEarthRadius();

  log << "@@ __MODEL_INITIALIZATION_COMPLETE__" << std::endl << std::flush;
}

// reference scatterers.cpp:45-91 (sharing), :97-220 (tables, MFPs, dipoles)
int Model::ScattererFor(const ScatterParams& requested) {
  ScatterParams par = requested;
  if (mOverrideMFP && mNoDeflect)  // one dummy scatterer serves every cell
    par = ScatterParams(Elastic::HSneak(1.0, 0.0, 1.0, 1.0), 1.0, 1.0);
  for (size_t i = 0; i < mScatParams.size(); i++)
    if (par.CompareRoughly(mScatParams[i]) <= 0) return (int)i;

  const size_t n = mNumTOA;
  auto store = std::make_unique<ScatStore>();
  if (mDeviceTables) {   // the engine evaluates the tables in HBM; only the parameters travel
    const double nan = std::nan("");
    ScattererInfo info{par.GetNu(), par.GetEps(), par.GetA(),  par.GetKappa(),
                       par.GetL(),  par.GetGam0(), {nan, nan},  {nan, nan}};
    if (mOverrideMFP) info.mfp[0] = mMFPOverride[0], info.mfp[1] = mMFPOverride[1];
    mScatStore.push_back(std::move(store));
    mScatParams.push_back(par);
    mScatInfo.push_back(info);
    return (int)mScatStore.size() - 1;
  }
  for (auto& c : store->cdf) c.resize(n);
  store->spol.resize(n);
  parallel_for(n, [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; k++)
      par.GSATO(mTOA[k], store->cdf[0][k], store->cdf[1][k], store->cdf[2][k],
                store->cdf[3][k], store->spol[k]);
  });

  ScattererInfo info{par.GetNu(), par.GetEps(), par.GetA(),  par.GetKappa(),
                     par.GetL(),  par.GetGam0(), {0, 0},     {1.0, 1.0}};
  // Dipole moments: forward/backward character, sum(cos(theta) * p)
  // (scatterers.cpp:244-259); computed on the raw weights before integration.
  double raw_sum[4] = {0, 0, 0, 0}, raw_cos[4] = {0, 0, 0, 0};
  for (int c = 0; c < 4; c++)
    for (size_t k = 0; k < n; k++) {
      raw_sum[c] += store->cdf[c][k];
      raw_cos[c] += mTOA[k].z() * store->cdf[c][k];
    }
  for (auto& c : store->cdf) integrate(c);
  const double tot[4] = {store->cdf[0].back(), store->cdf[1].back(), store->cdf[2].back(),
                         store->cdf[3].back()};
  if (!mOverrideMFP) {
    // Equal-area approximation: mean of g over the TOA set is the inverse
    // mean free path (scatterers.cpp:195-220).
    info.mfp[0] = 1.0 / ((tot[0] + tot[1]) / (double)n);
    info.mfp[1] = 1.0 / ((tot[2] + tot[3]) / (double)n);
  } else {
    info.mfp[0] = mMFPOverride[0], info.mfp[1] = mMFPOverride[1];
  }
  if (!mNoDeflect) {
    auto frac = [](double part, double whole) { return whole == 0 ? 0.0 : part / whole; };
    double mom[4];
    for (int c = 0; c < 4; c++) mom[c] = frac(raw_cos[c], raw_sum[c]);
    info.dipole[0] = mom[0] * frac(tot[0], tot[0] + tot[1]) + mom[1] * frac(tot[1], tot[0] + tot[1]);
    info.dipole[1] = mom[2] * frac(tot[2], tot[2] + tot[3]) + mom[3] * frac(tot[3], tot[2] + tot[3]);
  }
  mScatStore.push_back(std::move(store));
  mScatParams.push_back(par);
  mScatInfo.push_back(info);
  return (int)mScatStore.size() - 1;
}

// reference model.cpp:647-724 + media.cpp:137-160
void Model::BuildCellArray_Cylinder(Real range) {
  const unsigned nc = mGrid.Nk() - 1;
  mCells.reserve(nc);
  for (unsigned k = 0; k < nc; k++) {
    const GridData top = mGrid.Node(0, 0, k).Data(GridNode::GN_BELOW);
    r3d_cell c = blank_cell();
    c.n_faces = 3;
    // Stair-step layer: only the TOP node's properties are used
    // (media.cpp:150-156, :185-196).
    c.vel_c[0] = top.Vp(), c.vel_c[1] = top.Vs();
    c.rho_c = top.Rho();
    c.q[0] = top.getQ().Qp(top.getV()), c.q[1] = top.getQ().Qs(top.getV());
    c.faces[0] = plane_face(mGrid.Node(0, 0, k).Loc(), mGrid.Node(1, 0, k).Loc(),
                            mGrid.Node(2, 0, k).Loc());
    c.faces[1] = plane_face(mGrid.Node(0, 0, k + 1).Loc(), mGrid.Node(2, 0, k + 1).Loc(),
                            mGrid.Node(1, 0, k + 1).Loc());
    c.faces[2].radius = range;  // shared loss wall, never linked
    c.scatterer = ScattererFor(ScatterParams(top.getV(), top.getHS()));
    mCells.push_back(c);
  }
  for (unsigned k = 1; k < nc; k++)
    link_faces(mCells, k - 1, 1, k, 0, mGrid.Node(0, 0, k).IsDiscontinuous());
  mCells[0].faces[0].flags |= R3D_FACE_COLLECT | R3D_FACE_REFLECT;
  mSurfaceFaces.push_back(0 * 4 + 0);
}

// reference model.cpp:732-795 + media.cpp:578-626
void Model::BuildCellArray_SphericalShells() {
  const unsigned nc = mGrid.Nk() - 1;
  mCells.reserve(nc);
  for (unsigned k = 0; k < nc; k++) {
    const GridData top = mGrid.Node(0, 0, k).Data(GridNode::GN_BELOW);
    const GridData bot = mGrid.Node(0, 0, k + 1).Data(GridNode::GN_ABOVE);
    const Real rt = mGrid.Node(0, 0, k).GetRawLoc().Radius(ECS);
    const Real rb = mGrid.Node(0, 0, k + 1).GetRawLoc().Radius(ECS);
    if (rt <= rb)
      throw Runtime("SphereShell: Top surface must have greater radius than bottom surface.");
    if (rb < 0) throw Runtime("SphereShell: Bottom radius is less than zero. Check grid.");
    if (bot.Vp() < top.Vp() || bot.Vs() < top.Vs())
      throw Runtime("SphereShell: Reverse velocity gradients within model cells not currently "
                    "supported. Ensure velocity at bottom of cell is greater than or equal to "
                    "the top of the cell.");
    if (top.Vp() <= 0 || top.Vs() <= 0 || bot.Vp() <= 0 || bot.Vs() <= 0)
      throw Runtime("SphereShell: Elastic velocities must be greater than zero. If goal is to "
                    "model liquid, try very small but non-zero velocites.");
    r3d_cell c = blank_cell();
    c.n_faces = 2;
    const Real dr2 = rt * rt - rb * rb;
    const Real vt[2] = {top.Vp(), top.Vs()}, vb[2] = {bot.Vp(), bot.Vs()};
    for (int t = 0; t < 2; t++) {  // v(r) = A r^2 + C through both faces
      c.vel_a[t] = (vt[t] - vb[t]) / dr2;
      c.vel_c[t] = vt[t] - c.vel_a[t] * rt * rt;
      c.zero_rad2[t] = -c.vel_c[t] / c.vel_a[t];
      Real zero_rad = c.zero_rad2[t] > 0 ? std::sqrt(c.zero_rad2[t]) : kInf;
      if (!(zero_rad > rt))  // the reference asserts this (media.cpp:621-624)
        throw Runtime("SphereShell: zero-velocity surface lies inside the cell.");
    }
    c.rho_a = (top.Rho() - bot.Rho()) / dr2;
    c.rho_c = top.Rho() - c.rho_a * rt * rt;
    c.q[0] = top.Qp(), c.q[1] = top.Qs();  // top value for the whole cell
    c.faces[0].radius = rt;                // outward normal
    c.faces[1].radius = -rb;               // inward normal
    c.scatterer = ScattererFor(ScatterParams(top.getV(), top.getHS()));
    mCells.push_back(c);
  }
  for (unsigned k = 1; k < nc; k++)
    link_faces(mCells, k - 1, 1, k, 0, mGrid.Node(0, 0, k).IsDiscontinuous());
  mCells[0].faces[0].flags |= R3D_FACE_COLLECT | R3D_FACE_REFLECT;
  mSurfaceFaces.push_back(0 * 4 + 0);
}

// reference model.cpp:880-934 (block walk), :1017-1148 (five-tetra pattern),
// :1181-1228 (links across blocks), media.cpp:353-395 (Tetra constructor).
void Model::BuildCellArray_WCGTetra() {
  const int nI = mGrid.Ni() - 1, nJ = mGrid.Nj() - 1, nK = mGrid.Nk() - 1;
  mCells.clear();
  mCells.reserve((size_t)nI * nJ * nK * 5);
  auto base_of = [&](int i, int j, int k) { return ((i * (nK * nJ) + j * nK + k) * 5); };
  enum { FA = 0, FB = 1, FC = 2, FD = 3 };

  // Corner numbering of a block: bit2 = +i, bit1 = +j, bit0 = +k; even
  // corners are on the block's top sheet, odd ones on its bottom sheet.
  // Natural pattern and its mirror image (corner c -> c^1 in k... the mirror
  // swaps top and bottom sheets AND the roles of the diagonals):
  static const int nat[5][4] = {{6, 5, 0, 3}, {1, 3, 5, 0}, {2, 0, 6, 3}, {7, 5, 3, 6}, {4, 6, 0, 5}};
  static const int mir[5][4] = {{7, 4, 1, 2}, {0, 2, 4, 1}, {3, 1, 7, 2}, {6, 4, 2, 7}, {5, 7, 1, 4}};
  // corner whose velocity feeds each tetra's scatterer (HetSpec: corner 0)
  static const int nat_vsrc[5] = {0, 0, 2, 6, 4};
  static const int mir_vsrc[5] = {4, 0, 2, 6, 4};
  // {tetra, three corners} whose discontinuity flags decide FACE_D
  static const int nat_dis[4][4] = {{4, 0, 4, 6}, {2, 0, 2, 6}, {1, 1, 5, 3}, {3, 5, 7, 3}};
  static const int mir_dis[4][4] = {{1, 0, 4, 2}, {3, 4, 2, 6}, {4, 1, 5, 7}, {2, 1, 7, 3}};

  for (int i = 0; i < nI; i++)
    for (int j = 0; j < nJ; j++)
      for (int k = 0; k < nK; k++) {
        const GridNode* node[8];
        GridData data[8];
        R3::XYZ loc[8];
        bool dis[8];
        for (int c = 0; c < 8; c++) {
          node[c] = &mGrid.RelNode(i, j, k, (c >> 2) & 1, (c >> 1) & 1, c & 1);
          data[c] = node[c]->Data((c & 1) ? GridNode::GN_ABOVE : GridNode::GN_BELOW);
          loc[c] = node[c]->Loc();
          dis[c] = node[c]->IsDiscontinuous();
        }
        const bool mirror = ((i + j + k) % 2) == 1;
        const int(*pat)[4] = mirror ? mir : nat;
        const int* vsrc = mirror ? mir_vsrc : nat_vsrc;
        const int(*dtab)[4] = mirror ? mir_dis : nat_dis;
        const int base = (int)mCells.size();

        for (int t = 0; t < 5; t++) {
          R3::XYZ n[4];
          Real vp[4], vs[4], rho[4], qp = 0, qs = 0;
          for (int m = 0; m < 4; m++) {
            const GridData& d = data[pat[t][m]];
            n[m] = loc[pat[t][m]];
            vp[m] = d.Vp(), vs[m] = d.Vs(), rho[m] = d.Rho();
            qp += d.Qp(), qs += d.Qs();
          }
          r3d_cell c = blank_cell();
          c.n_faces = 4;
          // face X lies opposite node X
          c.faces[FA] = plane_face(n[1], n[2], n[3], n[0]);
          c.faces[FB] = plane_face(n[2], n[3], n[0], n[1]);
          c.faces[FC] = plane_face(n[3], n[0], n[1], n[2]);
          c.faces[FD] = plane_face(n[0], n[1], n[2], n[3]);
          fit_linear(n, vp, c.vel_grad[0], c.vel_c[0]);
          fit_linear(n, vs, c.vel_grad[1], c.vel_c[1]);
          fit_linear(n, rho, c.rho_grad, c.rho_c);
          c.q[0] = qp / 4, c.q[1] = qs / 4;
          c.scatterer = ScattererFor(ScatterParams(data[vsrc[t]].getV(), data[0].getHS()));
          mCells.push_back(c);
        }
        for (int r = 0; r < 4; r++)
          if (dis[dtab[r][1]] || dis[dtab[r][2]] || dis[dtab[r][3]])
            mCells[base + dtab[r][0]].faces[FD].flags |= R3D_FACE_DISCON;
        // the core tetra touches each corner tetra through that one's face A
        for (int t = 1; t <= 4; t++) link_faces(mCells, base, t - 1, base + t, FA, false);
        if (k == 0) {  // top of the stack is the free surface
          const int s1 = mirror ? 1 : 2, s2 = mirror ? 3 : 4;
          mCells[base + s1].faces[FD].flags |= R3D_FACE_COLLECT | R3D_FACE_REFLECT;
          mCells[base + s2].faces[FD].flags |= R3D_FACE_COLLECT | R3D_FACE_REFLECT;
        }
        // join to the blocks already built behind (-i), left (-j), above (-k)
        if (i > 0) {
          int prev = base_of(i - 1, j, k);
          link_faces_keep_flag(mCells, prev + 4, FC, base + 1, FC);
          link_faces_keep_flag(mCells, prev + 3, FC, base + 2, FC);
        }
        if (j > 0) {
          int prev = base_of(i, j - 1, k);
          link_faces_keep_flag(mCells, prev + 2, FB, base + 1, FB);
          link_faces_keep_flag(mCells, prev + 3, FB, base + 4, FB);
        }
        if (k > 0) {
          int prev = base_of(i, j, k - 1);
          // the block above has the opposite parity: its bottom faces belong
          // to tetra 1,3 if it is natural, 2,4 if mirrored; ours likewise.
          int shift = (!mirror) ? 1 : 0;  // previous block mirrored <=> we are natural
          link_faces_keep_flag(mCells, prev + 1 + shift, FD, base + 1 + shift, FD);
          link_faces_keep_flag(mCells, prev + 3 + shift, FD, base + 3 + shift, FD);
        }
      }
}

// reference model.cpp:521-551 with MediumCell::IsPointInside (media.cpp:60-76)
int Model::FindCellContainingPoint(const R3::XYZ& loc) const {
  Real best = 0.1;  // tolerate 100 m of mismatch
  int best_cell = -1;
  for (size_t ci = 0; ci < mCells.size(); ci++) {
    const r3d_cell& c = mCells[ci];
    Real mismatch = distance_above(c.faces[0], shape_of(mDesc.cell_kind, 0), loc);
    for (int f = 1; f < c.n_faces; f++)
      mismatch = std::max(mismatch, distance_above(c.faces[f], shape_of(mDesc.cell_kind, f), loc));
    if (mismatch <= 0.) return (int)ci;
    if (mismatch < best) best = mismatch, best_cell = (int)ci;
  }
  return best_cell;
}

// reference model.cpp:562-594
R3::XYZ Model::FindSurface(R3::XYZ loc) const {
  if (mSurfaceFaces.size() > 1)
    throw std::invalid_argument(
        "FindSurface() doesn't know how to handle multiple surface faces yet.");
  if (mSurfaceFaces.empty()) return loc;  // tetra models register none
  int cell = mSurfaceFaces[0] / 4, face = mSurfaceFaces[0] % 4;
  R3::XYZ up = ECS.GetUp(loc);
  Real dist = linear_exit_distance(mCells[cell].faces[face], shape_of(mDesc.cell_kind, face), loc, up);
  return loc + up.ScaledBy(dist);
}

// reference model.cpp:431-447 + events.cpp:42-107 (Aki & Richards box 9.10)
void Model::BuildSource(const ModelParams& par) {
  Tensor::Tensor mt = par.EventSourceMT;
  mEventMTUser = par.EventSourceMT;
  mEventLoc = ECS.Convert(par.EventSourceLoc);
  mt.Transform(ECS.GetXYZToLocalNEDRotation(mEventLoc));
  const Real mxx = mt.xx(), myy = mt.yy(), mzz = mt.zz();
  const Real mxy = mt.xy(), mxz = mt.xz(), myz = mt.yz();
  r3d_source& s = mDesc.source;
  const double moment[6] = {mxx, myy, mzz, mxy, mxz, myz};
  for (int k = 0; k < 6; k++) s.moment[k] = moment[k];
  put3(s.loc, mEventLoc);
  s.cell = FindCellContainingPoint(mEventLoc);
  if (s.cell < 0) throw Runtime("Event source location is not inside any model cell.");
  if (mDeviceTables) {   // the engine evaluates the patterns in HBM; only the moment tensor travels
    for (int t = 0; t < 3; t++) s.whole_cdf[t] = std::nan(""), s.cdf[t] = nullptr;
    return;
  }
  const size_t n = mTOA.size();
  for (auto& v : mSrcCdf) v.resize(n);
  parallel_for(n, [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; k++) {
      const Real th = mTOA[k].Theta(), az = mTOA[k].Phi();
      const Real st = std::sin(th), ct = std::cos(th), sa = std::sin(az), ca = std::cos(az);
      const Real horiz = mxx * ca * ca + mxy * std::sin(2 * az) + myy * sa * sa - mzz;
      const Real vert = mxz * ca + myz * sa;
      Real p = st * st * horiz + 2 * st * ct * vert + mzz;
      Real sh = st * (0.5 * std::sin(2 * az) * (myy - mxx) + std::cos(2 * az) * mxy) +
                ct * (ca * myz - sa * mxz);
      Real sv = st * ct * horiz + (1.0 - 2 * st * st) * vert;
      mSrcCdf[0][k] = p * p, mSrcCdf[1][k] = sh * sh, mSrcCdf[2][k] = sv * sv;
    }
  });
  for (auto& v : mSrcCdf) integrate(v);
  double acc = 0;
  for (int t = 0; t < 3; t++) {
    s.whole_cdf[t] = (acc += mSrcCdf[t].back());
    s.cdf[t] = mSrcCdf[t].data();
  }
}

// reference model.cpp:456-496 + dataout.cpp:42-71
void Model::BuildSeismometers(const ModelParams& par) {
  for (const auto& sr : par.SeisRequests()) {
    const R3::XYZ where = FindSurface(ECS.Convert(sr.Location));
    Real scale[2] = {1, 1};
    if (sr.RadiiUnitsAreWavelengths) {
      int cc = FindCellContainingPoint(where);
      if (cc < 0) throw Runtime("Seismometer location is not inside any model cell.");
      const r3d_cell& c = mCells[cc];
      for (int t = 0; t < 2; t++) {
        Real v;
        switch (mDesc.cell_kind) {
          case R3D_CELL_CYLINDER: v = c.vel_c[t]; break;
          case R3D_CELL_TETRA: v = where.Dot(get3(c.vel_grad[t])) + c.vel_c[t]; break;
          default: v = c.vel_c[t] + c.vel_a[t] * where.MagSquared();
        }
        scale[t] = v / par.Frequency;  // wavelength
      }
    }
    r3d_seismometer s;
    std::memset(&s, 0, sizeof s);
    put3(s.loc, where);
    for (int t = 0; t < 2; t++) {
      s.r_in[t] = sr.GatherRadiusInner[t] * scale[t];
      s.r_out[t] = sr.GatherRadiusOuter[t] * scale[t];
      s.area[t] = (s.r_out[t] * s.r_out[t] - s.r_in[t] * s.r_in[t]) * Geometry::Pi;
    }
    R3::XYZ x1 = (sr.Orientation == ModelParams::AX_RTZ) ? ECS.GetRadial(mEventLoc, where)
                                                         : ECS.GetEast(where);
    R3::XYZ x3 = ECS.GetUp(where);
    R3::XYZ x2 = x3.Cross(x1);
    x2 = x2.IsSquaredZero() ? ECS.GetNorth(where) : x2.Unit();
    x1 = x2.Cross(x3);
    put3(s.axes[0], x1), put3(s.axes[1], x2), put3(s.axes[2], x3);
    mSeis.push_back(s);
    mSeisAxes.push_back(sr.Orientation == ModelParams::AX_RTZ ? "RTZ" : "ENZ");
  }
}

// ------------------------------------------------------------- TOA set ----
namespace {
struct V3 {
  double x, y, z;
};
V3 unit_sum(const V3& a, const V3& b) {
  V3 s{a.x + b.x, a.y + b.y, a.z + b.z};
  double m = std::sqrt(s.x * s.x + s.y * s.y + s.z * s.z);
  return {s.x / m, s.y / m, s.z / m};
}
void split(const V3& a, const V3& b, const V3& c, int depth, std::vector<S2::ThetaPhi>& out) {
  if (depth == 0) {
    V3 s{a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z};
    double m = std::sqrt(s.x * s.x + s.y * s.y + s.z * s.z);
    out.emplace_back(std::acos(s.z / m), std::atan2(s.y / m, s.x / m));
    return;
  }
  V3 ab = unit_sum(a, b), bc = unit_sum(b, c), ca = unit_sum(c, a);
  split(a, ab, ca, depth - 1, out);
  split(ab, b, bc, depth - 1, out);
  split(ca, bc, c, depth - 1, out);
  split(bc, ca, ab, depth - 1, out);
}
}  // namespace

std::vector<S2::ThetaPhi> S2::TesselSphereIco(int degree) {
  if (degree < 0 || degree > 12) throw Runtime("TOA degree out of range [0,12].");
  const double g = (1. + std::sqrt(5.0)) / 2.0;
  auto U = [](double x, double y, double z) {
    double m = std::sqrt(x * x + y * y + z * z);
    return V3{x / m, y / m, z / m};
  };
  // twelve icosahedron vertices: cyclic permutations of (+-1, 0, +-g)
  const V3 v[12] = {U(1, 0, g),  U(-1, 0, g),  U(1, 0, -g), U(-1, 0, -g), U(g, -1, 0),  U(g, 1, 0),
                    U(-g, -1, 0), U(-g, 1, 0), U(0, g, 1),  U(0, g, -1),  U(0, -g, 1), U(0, -g, -1)};
  enum { NF, NB, SF, SB, FL, FR, BL, BR, RN, RS, LN, LS };
  static const int face[20][3] = {
      {NF, NB, LN}, {NF, NB, RN}, {SF, SB, LS}, {SF, SB, RS}, {FL, FR, NF}, {FL, FR, SF}, {BR, BL, NB},
      {BR, BL, SB}, {LN, LS, FL}, {LN, LS, BL}, {RN, RS, FR}, {RN, RS, BR}, {NF, LN, FL}, {NF, RN, FR},
      {NB, LN, BL}, {NB, RN, BR}, {SF, LS, FL}, {SF, RS, FR}, {SB, LS, BL}, {SB, RS, BR}};
  std::vector<S2::ThetaPhi> out;
  out.reserve((size_t)20 << (2 * degree));
  for (const auto& f : face) split(v[f[0]], v[f[1]], v[f[2]], degree, out);
  return out;
}
