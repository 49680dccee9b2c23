// r3d_pack.h -- host-side repacking of the C-ABI model (include/r3d.h) into
// the engine's own table layout (r3d_tables.h).  Pure C++ (no HIP): the
// engine uploads the packed vectors to HBM; the test-only CPU emulation of the
// kernel (tests/emul) points the kernel arguments at them directly.
#ifndef R3D_PACK_H_
#define R3D_PACK_H_

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/r3d.h"
#include "r3d_tables.h"

namespace r3d {

struct PackedModel {
  std::vector<CellCyl> cyl;
  std::vector<CellTet> tet;
  std::vector<CellSph> sph;
  std::vector<RhoLin> rho;
  std::vector<ScatHead> scat_head;
  std::vector<ScatPtrs> scat_ptrs;   // filled with the HOST pointers of the model
  std::vector<double> toa_xyz;
  std::vector<SeisScan> seis_scan;
  std::vector<SeisHit> seis_hit;
  std::vector<uint32_t> grid_start, grid_items;
  std::vector<std::vector<GuideCell>> src_guide;    // [3]   (host emulation only: the engine builds its own in HBM)
  std::vector<std::vector<GuideCell>> scat_guide;   // [n_scat * 4]
  std::vector<std::vector<double>> spol_cs;         // [n_scat] (cosine, sine) pairs (host emulation only, likewise)
  KArgs args;                        // pointers refer to the vectors above / the model
  size_t cell_bytes() const {
    return cyl.size() * sizeof(CellCyl) + tet.size() * sizeof(CellTet) + sph.size() * sizeof(CellSph);
  }
};

inline uint32_t pack_flags(const r3d_cell& c) {
  uint32_t f = 0;
  for (int i = 0; i < c.n_faces && i < 4; i++) f |= (c.faces[i].flags & 0xFFu) << (8 * i);
  return f;
}
inline double dot3(const double a[3], const double b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

// Uniform hash over the seismometers' gather spheres.  Cell edge ~ a quarter of the median
// outer radius -- the tables live in HBM / L2, not in LDS, so the grid can be fine: an arrival then
// finds little more than the receivers it really lies in (with an edge of twice the radius, as the
// LDS-resident hash had it, the crust-pinch arrays gave ~20 candidates per arrival for 0.7
// catches) -- grown until the grid has at most ~4M cells and ~8M entries.  Insertion is
// conservative (bounding box of the sphere plus a rounding margin), so the candidates of a cell
// are a superset of the seismometers that can catch a phonon arriving anywhere inside it.
inline void build_seis_grid(const r3d_model_desc& m, SeisGrid& g, std::vector<uint32_t>& start,
                            std::vector<uint32_t>& items) {
  const int n = m.n_seismometers;
  g.n_cells = 0;
  g.dim[0] = g.dim[1] = g.dim[2] = 0;
  g.dim_f[0] = g.dim_f[1] = g.dim_f[2] = 0.0;
  g.inv_h = 0;
  g.origin[0] = g.origin[1] = g.origin[2] = 0;
  start.assign(2, 0);
  items.assign(1, 0);
  if (n == 0) return;
  std::vector<double> rad(n);
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int s = 0; s < n; s++) {
    const r3d_seismometer& S = m.seismometers[s];
    rad[s] = std::max(S.r_out[0], S.r_out[1]);
    for (int k = 0; k < 3; k++) {
      lo[k] = std::min(lo[k], S.loc[k] - rad[s]);
      hi[k] = std::max(hi[k], S.loc[k] + rad[s]);
    }
  }
  std::vector<double> sorted = rad;
  std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
  double h = std::max(0.25 * sorted[n / 2], 1e-6);
  int d[3];
  auto dims_for = [&](double hh) {
    double cells = 1;
    for (int k = 0; k < 3; k++) {
      d[k] = std::max(1, (int)std::ceil((hi[k] - lo[k]) / hh) + 1);
      cells *= d[k];
    }
    return cells;
  };
  auto entries_for = [&](double hh) {   // cells each sphere's bounding box covers, summed
    double e = 0;
    for (int s = 0; s < n; s++) {
      const double w = 2.0 * rad[s] / hh + 2.0;
      e += w * w * w;
    }
    return e;
  };
  while (dims_for(h) > 4.0e6 || entries_for(h) > 8.0e6) h *= 1.25;
  for (int k = 0; k < 3; k++) g.origin[k] = lo[k], g.dim[k] = d[k], g.dim_f[k] = (double)d[k];
  g.inv_h = 1.0 / h;
  g.n_cells = d[0] * d[1] * d[2];
  std::vector<std::vector<uint32_t>> buckets(g.n_cells);
  for (int s = 0; s < n; s++) {
    const r3d_seismometer& S = m.seismometers[s];
    int a0[3], a1[3];
    for (int k = 0; k < 3; k++) {
      double r = rad[s] * (1 + 1e-9) + 1e-9 * h;
      a0[k] = std::max(0, (int)std::floor((S.loc[k] - r - lo[k]) * g.inv_h) - 0);
      a1[k] = std::min(d[k] - 1, (int)std::floor((S.loc[k] + r - lo[k]) * g.inv_h));
    }
    for (int z = a0[2]; z <= a1[2]; z++)
      for (int y = a0[1]; y <= a1[1]; y++)
        for (int x = a0[0]; x <= a1[0]; x++)
          buckets[((size_t)z * d[1] + y) * d[0] + x].push_back((uint32_t)s);
  }
  start.assign(g.n_cells + 1, 0);
  for (int c = 0; c < g.n_cells; c++) start[c + 1] = start[c] + (uint32_t)buckets[c].size();
  items.clear();
  items.reserve(start.back() + 1);
  for (auto& b : buckets) items.insert(items.end(), b.begin(), b.end());
  if (items.empty()) items.push_back(0);
}

// Search guide of a cumulative table: guide[j] = smallest k with
// total * (j / G) <= cdf[k], j = 0..G.  A draw u in [j/G, (j+1)/G) has its
// answer inside [guide[j], guide[j+1]], so the bisection starts from a bracket
// of ~n/G entries instead of n, and returns the identical index.
inline void build_guide(const double* cdf, uint64_t n, uint32_t bits, std::vector<uint32_t>& g) {
  const uint64_t G = 1ull << bits;
  g.resize(G + 1);
  const double total = cdf[n - 1];
  uint64_t k = 0;
  for (uint64_t j = 0; j <= G; j++) {
    const double r = total * ((double)j / (double)G);
    while (k < n - 1 && !(r <= cdf[k])) k++;
    g[j] = (uint32_t)k;
  }
}
// ... and its cells (r3d_tables.h GuideCell): the bracket's ends and its first entries.
inline void build_guide_cells(const double* cdf, uint64_t n, uint32_t bits, std::vector<GuideCell>& cells) {
  std::vector<uint32_t> g;
  build_guide(cdf, n, bits, g);
  const uint64_t G = 1ull << bits;
  cells.resize(G);
  for (uint64_t j = 0; j < G; j++) {
    GuideCell& c = cells[j];
    c.k1 = g[j], c.k2 = g[j + 1];
    const bool direct = c.k2 - c.k1 <= (uint32_t)kGuideVals;
    for (int i = 0; i < kGuideVals; i++)
      c.c[i] = direct ? cdf[(uint64_t)c.k1 + i < c.k2 ? (uint64_t)c.k1 + i : c.k2] : cdf[guide_pivot(c.k1, c.k2, i)];
  }
}
inline uint32_t guide_bits_for(uint64_t n_toa) {
  uint32_t bits = 4;
  const int per = 2, cap = 20;   // ~2^per entries per bracket, at most 2^cap guide entries
  while ((int)bits < cap && (1ull << (bits + per)) < n_toa) bits++;
  return bits;
}

// ---- velocity-step classification of interior faces -------------------------
// Phonon::Refract (phonons.cpp:243-252) evaluates max over P,S of
// |2 (v2 - v1) / (v2 + v1)| at the crossing point and compares it with 1e-5.
// On a face each signed step f_t = 2 (v2 - v1) / (v2 + v1) is a ratio of affine
// functions of position (constants for layered and spherical cells) with a
// positive denominator, hence monotone along every line: over the face it lies
// between its values at the face's corners.  So
//   * if |f_t| is safely below the threshold at every corner for both wave types,
//     max_t |f_t| is below it everywhere: the face is SMOOTH;
//   * if for ONE wave type f_t is safely beyond the threshold WITH THE SAME SIGN
//     at every corner, |f_t| -- and with it the maximum -- is beyond it everywhere:
//     the face is a STEP.  (A corner-only lower bound on |f_t| would not do: where
//     f_t changes sign across the face it passes through zero in between.)
// Faces that satisfy neither keep the run-time test.  The 10 % guard band dwarfs any
// rounding in where exactly on the face the phonon sits.
inline double signed_step(double v1, double v2) { return 2 * (v2 - v1) / (v2 + v1); }

// steps[corner][type]
inline uint32_t classify_from_corner_steps(const double (*steps)[2], int n) {
  bool smooth = true;
  bool up[2] = {true, true}, down[2] = {true, true};
  for (int i = 0; i < n; i++)
    for (int t = 0; t < 2; t++) {
      const double s = steps[i][t];
      if (!(s == s)) return 0;   // NaN: decide at run time
      smooth = smooth && std::fabs(s) < 0.9e-5;
      up[t] = up[t] && s > 1.1e-5;
      down[t] = down[t] && s < -1.1e-5;
    }
  if (smooth) return F_SMOOTH;
  if (up[0] || down[0] || up[1] || down[1]) return F_STEP;
  return 0;
}

// Corner of a tetrahedron opposite face `skip`: intersection of the other three face planes.
inline bool tet_corner(const r3d_cell& c, int skip, double out[3]) {
  double A[3][4];
  int r = 0;
  for (int f = 0; f < 4; f++) {
    if (f == skip) continue;
    for (int k = 0; k < 3; k++) A[r][k] = c.faces[f].normal[k];
    A[r][3] = dot3(c.faces[f].normal, c.faces[f].point);
    r++;
  }
  for (int col = 0; col < 3; col++) {
    int piv = col;
    for (int i = col + 1; i < 3; i++)
      if (std::fabs(A[i][col]) > std::fabs(A[piv][col])) piv = i;
    if (A[piv][col] == 0) return false;
    for (int k = 0; k < 4; k++) std::swap(A[piv][k], A[col][k]);
    for (int i = 0; i < 3; i++) {
      if (i == col) continue;
      double f = A[i][col] / A[col][col];
      for (int k = col; k < 4; k++) A[i][k] -= f * A[col][k];
    }
  }
  for (int k = 0; k < 3; k++) out[k] = A[k][3] / A[k][k];
  return true;
}

inline uint32_t classify_velocity_step(const r3d_model_desc& m, int ci, int f) {
  const r3d_cell& c = m.cells[ci];
  const r3d_face& F = c.faces[f];
  if (!(F.flags & R3D_FACE_ADJOIN) || (F.flags & (R3D_FACE_DISCON | R3D_FACE_REFLECT))) return 0;
  const r3d_cell& o = m.cells[F.neighbor];
  double steps[3][2];
  if (m.cell_kind == R3D_CELL_CYLINDER) {
    for (int t = 0; t < 2; t++) steps[0][t] = signed_step(c.vel_c[t], o.vel_c[t]);
    return classify_from_corner_steps(steps, 1);
  }
  if (m.cell_kind == R3D_CELL_SPHERESHELL) {
    const double r2 = F.radius * F.radius;
    for (int t = 0; t < 2; t++)
      steps[0][t] = signed_step(c.vel_c[t] + c.vel_a[t] * r2, o.vel_c[t] + o.vel_a[t] * r2);
    return classify_from_corner_steps(steps, 1);
  }
  int n = 0;
  for (int corner = 0; corner < 4; corner++) {
    if (corner == f) continue;                 // corners ON face f are those opposite the other faces
    double x[3];
    if (!tet_corner(c, corner, x)) return 0;
    for (int t = 0; t < 2; t++)
      steps[n][t] = signed_step(dot3(c.vel_grad[t], x) + c.vel_c[t], dot3(o.vel_grad[t], x) + o.vel_c[t]);
    n++;
  }
  return classify_from_corner_steps(steps, n);
}

inline uint32_t pack_flags_classified(const r3d_model_desc& m, int ci) {
  uint32_t f = pack_flags(m.cells[ci]);
  for (int i = 0; i < m.cells[ci].n_faces && i < 4; i++) f |= classify_velocity_step(m, ci, i) << (8 * i);
  return f;
}

// for_engine: the engine evaluates the unit vectors of the take-off set and the source's search
// guides in HBM itself (and may be given neither the set nor the source tables: their
// build-on-device forms), so those are left out here; the test-only host emulation packs everything.
inline void pack_model(const r3d_model_desc& m, PackedModel& pm, bool for_engine = false) {
  KArgs& a = pm.args;
  std::memset(&a, 0, sizeof a);
  const r3d_params& par = m.params;
  const double kPiF = -3.14159265358979323846 * par.frequency;

  // ---- cells ----
  if (m.cell_kind == R3D_CELL_CYLINDER) {
    pm.cyl.resize((size_t)2 * m.n_cells);   // one record per cell and ray type (r3d_tables.h)
    for (int i = 0; i < m.n_cells; i++) {
      const r3d_cell& c = m.cells[i];
      const uint32_t flags = pack_flags_classified(m, i);
      for (int t = 0; t < 2; t++) {
        CellCyl& d = pm.cyl[(size_t)2 * i + t];
        std::memset(&d, 0, sizeof d);
        d.v = c.vel_c[t], d.att = kPiF / c.q[t];
        d.rho = c.rho_c;
        for (int f = 0; f < 2; f++) {
          for (int k = 0; k < 3; k++) d.n[f][k] = c.faces[f].normal[k];
          d.d[f] = dot3(c.faces[f].normal, c.faces[f].point);
          d.nbr[f] = (c.faces[f].flags & R3D_FACE_ADJOIN) ? c.faces[f].neighbor : -1;
        }
        d.flags = flags;
        d.scat = c.scatterer;
      }
    }
    a.cyl_radius2 = m.cells[0].faces[2].radius * m.cells[0].faces[2].radius;
    a.cells = pm.cyl.data();
  } else if (m.cell_kind == R3D_CELL_TETRA) {
    pm.tet.resize((size_t)2 * m.n_cells);   // one record per cell and ray type (r3d_tables.h CellTet)
    pm.rho.resize(m.n_cells);
    for (int i = 0; i < m.n_cells; i++) {
      const r3d_cell& c = m.cells[i];
      const uint32_t flags = pack_flags_classified(m, i);
      for (int t = 0; t < 2; t++) {
        CellTet& d = pm.tet[(size_t)2 * i + t];
        std::memset(&d, 0, sizeof d);
        for (int k = 0; k < 3; k++) d.g[k] = c.vel_grad[t][k];
        d.v0 = c.vel_c[t];
        d.inv_gmag = 1.0 / std::sqrt(dot3(c.vel_grad[t], c.vel_grad[t]));
        d.att = kPiF / c.q[t];
        for (int f = 0; f < 4; f++) {
          for (int k = 0; k < 3; k++) d.n[f][k] = c.faces[f].normal[k];
          d.d[f] = dot3(c.faces[f].normal, c.faces[f].point);
          const uint32_t nbr = (c.faces[f].flags & R3D_FACE_ADJOIN) ? (uint32_t)(c.faces[f].neighbor + 1) : 0u;
          d.link[f] = nbr | (((flags >> (8 * f)) & 0x3Fu) << kTetNbrBits) |
                      ((((uint32_t)c.scatterer >> (4 * f)) & 0xFu) << 28);
        }
      }
      for (int k = 0; k < 3; k++) pm.rho[i].g[k] = c.rho_grad[k];
      pm.rho[i].c = c.rho_c;
    }
    a.cells = pm.tet.data();
    a.rho = pm.rho.data();
  } else {
    pm.sph.resize((size_t)2 * m.n_cells);
    for (int i = 0; i < m.n_cells; i++) {
      const r3d_cell& c = m.cells[i];
      const uint32_t flags = pack_flags_classified(m, i);
      for (int t = 0; t < 2; t++) {
        CellSph& d = pm.sph[(size_t)2 * i + t];
        std::memset(&d, 0, sizeof d);
        d.a = c.vel_a[t], d.c = c.vel_c[t], d.zero_rad2 = c.zero_rad2[t];
        d.att = kPiF / c.q[t];
        d.rho_a = c.rho_a, d.rho_c = c.rho_c;
        for (int f = 0; f < 2; f++) {
          d.radius[f] = c.faces[f].radius;
          d.nbr[f] = (c.faces[f].flags & R3D_FACE_ADJOIN) ? c.faces[f].neighbor : -1;
        }
        d.flags = flags;
        d.scat = c.scatterer;
      }
    }
    a.cells = pm.sph.data();
  }
  a.n_cells = m.n_cells;

  // ---- scatterers ----
  a.guide_bits = guide_bits_for(m.n_toa);
  pm.scat_guide.assign((size_t)m.n_scatterers * 4, {});
  pm.spol_cs.assign((size_t)m.n_scatterers, {});
  pm.src_guide.assign(3, {});
  pm.scat_head.resize(m.n_scatterers);
  pm.scat_ptrs.resize(m.n_scatterers);
  for (int s = 0; s < m.n_scatterers; s++) {
    const r3d_scatterer& S = m.scatterers[s];
    for (int t = 0; t < 2; t++) {
      pm.scat_head[s].mfp[t] = S.mfp[t];
      for (int k = 0; k < 4; k++) pm.scat_head[s].whole[t][k] = S.whole_cdf[t][k];
    }
    if (!S.cdf[0]) {   // build-on-device form: the engine fills head, tables and guides
      for (int k = 0; k < 4; k++) pm.scat_ptrs[s].cdf[k] = nullptr, pm.scat_ptrs[s].guide[k] = nullptr;
      pm.scat_ptrs[s].spol_cs = nullptr;
      continue;
    }
    for (int k = 0; k < 4; k++) {
      pm.scat_ptrs[s].cdf[k] = S.cdf[k];
      pm.scat_head[s].total[k] = S.cdf[k][m.n_toa - 1];
      pm.scat_ptrs[s].guide[k] = nullptr;
      if (!for_engine) {   // (the engine makes the guides in HBM, from the tables it has copied in)
        build_guide_cells(S.cdf[k], m.n_toa, a.guide_bits, pm.scat_guide[s * 4 + k]);
        pm.scat_ptrs[s].guide[k] = pm.scat_guide[s * 4 + k].data();
      }
    }
    pm.scat_ptrs[s].spol_cs = nullptr;
    if (!for_engine) {   // (the engine makes the pairs in HBM too)
      std::vector<double>& cs = pm.spol_cs[s];
      cs.resize(2 * m.n_toa);
      for (uint64_t k = 0; k < m.n_toa; k++) cs[2 * k] = std::cos(S.spol[k]), cs[2 * k + 1] = std::sin(S.spol[k]);
      pm.scat_ptrs[s].spol_cs = cs.data();
    }
  }
  a.scat_head = pm.scat_head.data();
  a.scat_ptrs = pm.scat_ptrs.data();
  a.n_scat = m.n_scatterers;

  // ---- take-off directions as unit vectors; theta nudged away from the poles
  //      as Phonon::nudge_if_singular does (phonons.hpp:335-344) ----
  if (!for_engine) pm.toa_xyz.resize(m.n_toa * 4);
  for (uint64_t k = 0; k < m.n_toa && !for_engine; k++) {
    double th = m.toa[2 * k], ph = m.toa[2 * k + 1];
    if (th < par.min_theta) th = par.min_theta;
    if (th > par.max_theta) th = par.max_theta;
    pm.toa_xyz[4 * k] = std::cos(th), pm.toa_xyz[4 * k + 1] = std::cos(ph);
    pm.toa_xyz[4 * k + 2] = std::sin(ph), pm.toa_xyz[4 * k + 3] = std::sin(th);
  }
  a.toa_dir = pm.toa_xyz.data();
  a.n_toa = m.n_toa;
  a.nodeflect_dir[0] = std::cos(par.min_theta);
  a.nodeflect_dir[1] = 1.0, a.nodeflect_dir[2] = 0.0;
  a.nodeflect_dir[3] = std::sin(par.min_theta);

  // ---- source ----
  for (int k = 0; k < 3; k++) {
    a.src_cdf[k] = m.source.cdf[k];
    a.src_total[k] = m.source.cdf[0] ? m.source.cdf[k][m.n_toa - 1] : 0.0;
    if (!for_engine) {
      build_guide_cells(m.source.cdf[k], m.n_toa, a.guide_bits, pm.src_guide[k]);
      a.src_guide[k] = pm.src_guide[k].data();
    }
    a.src_whole[k] = m.source.whole_cdf[k];
    a.src_loc[k] = m.source.loc[k];
    a.earth_center[k] = par.earth_center[k];
  }
  a.src_cell = m.source.cell;

  // ---- seismometers ----
  const int ns = std::max(1, m.n_seismometers);
  pm.seis_scan.assign(ns, SeisScan{});
  pm.seis_hit.assign(ns, SeisHit{});
  for (int s = 0; s < m.n_seismometers; s++) {
    const r3d_seismometer& S = m.seismometers[s];
    for (int k = 0; k < 3; k++) {
      pm.seis_scan[s].loc[k] = S.loc[k];
      for (int j = 0; j < 3; j++) pm.seis_hit[s].axes[k][j] = S.axes[k][j];
    }
    for (int t = 0; t < 2; t++) {
      pm.seis_scan[s].r_in[t] = S.r_in[t], pm.seis_scan[s].r_out[t] = S.r_out[t];
      pm.seis_hit[s].inv_norm[t] = 1.0 / (par.time_per_bin * S.area[t]);
    }
  }
  a.seis_scan = pm.seis_scan.data();
  a.seis_hit = pm.seis_hit.data();
  a.n_seis = m.n_seismometers;
  build_seis_grid(m, a.grid, pm.grid_start, pm.grid_items);
  a.grid.start = pm.grid_start.data();
  a.grid.items = pm.grid_items.data();

  // ---- scalars ----
  a.n_bins = par.n_bins;
  a.n_bins_f = (double)par.n_bins;
  a.no_deflect = par.no_deflect;
  a.ttl = par.ttl;
  a.time_per_bin = par.time_per_bin;
  a.inv_time_per_bin = 1.0 / par.time_per_bin;
  a.slow_concern = par.slow_concern;
  a.loop_concern = par.loop_concern;
  a.lds_cells_off = 0xFFFFFFFFu;
}

}  // namespace r3d
#endif
