// r3d_engine.hip -- the MI355X (gfx950) phonon-transport engine: the C-ABI of include/r3d.h
// around the traversal kernel of r3d_pool.h.
//
// Kernel design in one paragraph (details: r3d_pool.h): persistent grid, one
// 768-thread workgroup per CU; the histories in flight live in LDS (a pool of ~1000 slots per
// workgroup) and sit in one of five queues -- MOVE, COLLECT, RT, SCATTER, FREE; a wave takes 64
// slots of ONE queue and runs that phase for all of them at once, so every phase executes at
// (nearly) full width instead of under the partial masks of a one-phonon-per-lane loop; small
// read-only tables (cells of layered and spherical models, scatterer heads) are staged in LDS,
// tetra cells, receiver tables, CDFs (GBs at TOA degree 9) and bins stay in HBM / L2; bins are
// accumulated through per-workgroup LDS accumulators into native fp64 / u64 global atomics; RNG
// is counter-based Philox keyed by history id (r3d_rng.h): which lane, wave, launch or GPU serves a
// history changes its result by rounding at most (a few series pick their form by a vote of the
// wave, r3d_math.h all_lanes) and not at all in the reproducible build (make repro).  The kernels
// themselves are instantiated in r3d_kernels_kind.hip, one translation unit per cell kind.
//
// The product has no CPU path: without a HIP device every entry point fails with an error
// message.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/r3d.h"
#include "r3d_pack.h"
#include "r3d_step.h"
#include "r3d_tables_build.h"

#include "r3d_kernels.h"
#include "r3d_rccl.h"   // the shards' blocks are summed on the devices (r3d_node_run, r3d_comm_reduce)
namespace r3d {

// ------------------------------------------------------------------- engine --
thread_local std::string g_error;

#define R3D_HIP_OK(call)                                                              \
  do {                                                                                \
    hipError_t err__ = (call);                                                        \
    if (err__ != hipSuccess) {                                                        \
      g_error = std::string(#call) + ": " + hipGetErrorString(err__);                 \
      return fail_value;                                                              \
    }                                                                                 \
  } while (0)

// Entry points make the engine's device current for their own calls and restore the caller's
// on return: a host program (or torch) whose current device differs from the engine's keeps it.
struct DeviceGuard {
  int prev = -1;
  hipError_t status;
  explicit DeviceGuard(int device) {
    status = hipGetDevice(&prev);
    if (status != hipSuccess) prev = -1;
    if (prev == device) prev = -1;   // nothing to switch, nothing to restore
    else status = hipSetDevice(device);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define R3D_ON_DEVICE(dev)          \
  DeviceGuard guard__(dev);         \
  R3D_HIP_OK(guard__.status)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t upload(const void* src, size_t n) {
    bytes = n;
    hipError_t e = hipMalloc(&p, n ? n : 8);
    if (e != hipSuccess) return e;
    if (n) e = hipMemcpy(p, src, n, hipMemcpyHostToDevice);
    return e;
  }
  hipError_t alloc_zero(size_t n) {
    bytes = n;
    hipError_t e = hipMalloc(&p, n ? n : 8);
    if (e != hipSuccess) return e;
    return hipMemset(p, 0, n ? n : 8);
  }
};

}  // namespace r3d

using namespace r3d;

struct r3d_engine {
  int device = 0;
  int kind = 0;
  int res = RES_ALL;   // which tables live in LDS (RES_*)
  size_t carry_bytes = 0;
  int n_seis = 0;
  uint32_t n_bins = 0;
  KArgs args{};
  size_t lds_bytes = 0;
  int grid_blocks = 0;
  hipStream_t stream = nullptr;
  std::vector<std::unique_ptr<DevBuf>> bufs;
  DevBuf d_energy, d_counts, d_scalars, d_next;
  // launches on different streams may be in flight together (a caller overlapping one
  // batch's drain with the next batch): each takes its own work counter from a small ring
  static constexpr unsigned kCounters = 64;
  // ... and its own pair of timing events, so that r3d_kernel_ms(e, launch) reads the launch
  // it names and not whichever recorded last
  hipEvent_t ev0[kCounters] = {}, ev1[kCounters] = {};
  uint64_t launches = 0;   // launches enqueued so far; launch id k used slot (k - 1) % kCounters
  std::unique_ptr<DevBuf> d_carry;   // pool image per workgroup of the grid (r3d_run_device_carry)
  bool carry_pending = false;
  uint64_t carry_seed = 0;
  std::unique_ptr<DevBuf> d_volume;
  void* volume_ext = nullptr;   // caller-owned counters (r3d_engine_set_volume_buffer)
  size_t volume_len = 0;
  std::unique_ptr<DevBuf> d_evlog, d_evlog_count;
  std::unique_ptr<DevBuf> d_pfinals;   // final records out of the production kernels (r3d_engine_set_production_finals)
  uint64_t pfinals_base = 0, pfinals_cap = 0;   // ... of the histories base <= id < base + cap
  struct ScatStats {
    double mfp[2], dipole[2], total[4];
  };
  std::vector<ScatStats> scat_stats;
  std::vector<ScatPtrs> scat_ptrs;   // device addresses of every scatterer's tables
  std::vector<const double*> d_spol; // ... and of its polarisation ANGLES (the kernel reads their cosine / sine pairs)
  uint64_t n_toa = 0;
  const double* d_toa = nullptr;     // the take-off set, (theta, phi) pairs
  const double* d_src[3] = {nullptr, nullptr, nullptr};
  double src_whole[3] = {0, 0, 0};

  DevBuf* keep(std::unique_ptr<DevBuf> b) {
    bufs.push_back(std::move(b));
    return bufs.back().get();
  }
  ~r3d_engine() {
    for (unsigned i = 0; i < kCounters; i++) {
      if (ev0[i]) (void)hipEventDestroy(ev0[i]);
      if (ev1[i]) (void)hipEventDestroy(ev1[i]);
    }
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

template <class T>
const T* upload_vec(r3d_engine* e, const std::vector<T>& v, hipError_t* err) {
  auto b = std::make_unique<DevBuf>();
  hipError_t r = b->upload(v.data(), v.size() * sizeof(T));
  if (r != hipSuccess) *err = r;
  return reinterpret_cast<const T*>(e->keep(std::move(b))->p);
}
const double* upload_doubles(r3d_engine* e, const double* src, size_t n, hipError_t* err) {
  auto b = std::make_unique<DevBuf>();
  hipError_t r = b->upload(src, n * sizeof(double));
  if (r != hipSuccess) *err = r;
  return reinterpret_cast<const double*>(e->keep(std::move(b))->p);
}

// The search guide of a cumulative table that is already in HBM (r3d_tables.h GuideCell), made there.
const GuideCell* make_guide(r3d_engine* e, const double* d_cdf, uint64_t n, uint32_t bits, hipError_t* err) {
  auto g = std::make_unique<DevBuf>();
  if (hipError_t r = g->alloc_zero((size_t(1) << bits) * sizeof(GuideCell)); r != hipSuccess) *err = r;
  GuideCell* d_guide = reinterpret_cast<GuideCell*>(e->keep(std::move(g))->p);
  if (*err == hipSuccess)
    if (hipError_t r = build_guide_on_device(d_cdf, n, bits, d_guide, nullptr); r != hipSuccess) *err = r;
  return d_guide;
}

// The (cosine, sine) pairs of a scatterer's polarisation angles (r3d_tables.h ScatPtrs::spol_cs), made in HBM.
const double* make_spol_cs(r3d_engine* e, const double* d_spol, uint64_t n, hipError_t* err) {
  auto g = std::make_unique<DevBuf>();
  if (hipError_t r = g->alloc_zero(2 * n * sizeof(double)); r != hipSuccess) *err = r;
  double* d_cs = reinterpret_cast<double*>(e->keep(std::move(g))->p);
  if (*err == hipSuccess)
    if (hipError_t r = build_spol_cs_on_device(d_spol, n, d_cs, nullptr); r != hipSuccess) *err = r;
  return d_cs;
}

// The traversal kernels live in one translation unit per cell kind (r3d_kernels_kind.hip, compiled
// three times): each kind gets the instruction scheduler that suits it (Makefile) and the three
// compile side by side.
hipError_t launch_any(r3d_engine* e, const KArgs& a, bool trace, hipStream_t s, bool drain_only = false) {
  const unsigned grid = (unsigned)e->grid_blocks;
  switch (e->kind) {
    case R3D_CELL_CYLINDER: return launch_pool_cyl(e->res, trace, drain_only, grid, e->lds_bytes, s, a);
    case R3D_CELL_TETRA: return launch_pool_tet(e->res, trace, drain_only, grid, e->lds_bytes, s, a);
    default: return launch_pool_sph(e->res, trace, drain_only, grid, e->lds_bytes, s, a);
  }
}

hipError_t set_lds_attr(const r3d_engine* e) {
  switch (e->kind) {
    case R3D_CELL_CYLINDER: return pool_lds_attr_cyl(e->res, (int)e->lds_bytes);
    case R3D_CELL_TETRA: return pool_lds_attr_tet(e->res, (int)e->lds_bytes);
    default: return pool_lds_attr_sph(e->res, (int)e->lds_bytes);
  }
}

bool check_model(const r3d_model_desc* m) {
  if (!m) return g_error = "null model", false;
  if (m->cell_kind < 0 || m->cell_kind > 2) return g_error = "unknown cell kind", false;
  if (m->n_cells <= 0 || !m->cells) return g_error = "model has no cells", false;
  if (m->n_scatterers <= 0 || !m->scatterers) return g_error = "model has no scatterers", false;
  if (m->n_toa == 0) return g_error = "model has no take-off angles", false;
  if (!m->toa && (m->toa_degree < 0 || m->toa_degree > 12 || m->n_toa != ((uint64_t)20 << (2 * m->toa_degree))))
    return g_error = "take-off set to be generated: n_toa must be 20 * 4^toa_degree, degree 0..12", false;
  if (m->params.n_bins == 0) return g_error = "zero time bins", false;
  // index widths of the device layout: guides and samplers hold take-off indices in 32 bits,
  // a catch's (seismometer, bin, type) travels as one 32-bit word
  if (m->n_toa >= (uint64_t(1) << 32)) return g_error = "more than 2^32 take-off angles", false;
  if ((uint64_t)std::max(0, m->n_seismometers) * m->params.n_bins >= (uint64_t(1) << 31))
    return g_error = "seismometers x time bins must stay below 2^31", false;
  if (m->source.cell < 0 || m->source.cell >= m->n_cells) return g_error = "source cell out of range", false;
  // a tetra record packs neighbour + 1 into 22 bits and the scatterer index into 16 (r3d_tables.h CellTet)
  if (m->cell_kind == R3D_CELL_TETRA && ((uint32_t)m->n_cells >= kTetNbrMask || m->n_scatterers > 65536))
    return g_error = "tetra models are limited to 4 194 302 cells and 65 536 scatterers", false;
  const int want_faces = m->cell_kind == R3D_CELL_CYLINDER ? 3 : m->cell_kind == R3D_CELL_TETRA ? 4 : 2;
  for (int i = 0; i < m->n_cells; i++) {
    const r3d_cell& c = m->cells[i];
    if (c.n_faces != want_faces) return g_error = "cell face count does not match cell kind", false;
    if (c.scatterer < 0 || c.scatterer >= m->n_scatterers) return g_error = "cell scatterer out of range", false;
    for (int f = 0; f < c.n_faces; f++) {
      const r3d_face& F = c.faces[f];
      if ((F.flags & R3D_FACE_ADJOIN) && (F.neighbor < 0 || F.neighbor >= m->n_cells))
        return g_error = "face neighbour out of range", false;
    }
  }
  for (int s = 0; s < m->n_scatterers; s++)
    for (int k = 0; k < 4; k++)
      if (m->scatterers[s].cdf[0] && (!m->scatterers[s].cdf[k] || !m->scatterers[s].spol))
        return g_error = "scatterer table missing", false;
  for (int k = 0; k < 3; k++)
    if (m->source.cdf[0] && !m->source.cdf[k]) return g_error = "source table missing", false;
  return true;
}

}  // namespace

extern "C" {

const char* r3d_last_error(void) { return g_error.c_str(); }
const char* r3d_version(void) { return "radiative3d_amd engine r1 (gfx950, fp64)"; }

r3d_engine* r3d_engine_create(const r3d_model_desc* m, int device) { return r3d_engine_create_ex(m, device, nullptr); }

r3d_engine* r3d_engine_create_ex(const r3d_model_desc* m, int device, const r3d_engine_opts* opts) {
  r3d_engine* const fail_value = nullptr;
  if (!check_model(m)) return nullptr;
  // The carve-up's knobs (include/r3d.h r3d_engine_opts): defaults unless the caller -- a test that must
  // reach a given kernel variant on a small model, a tuning run -- names them.  Nothing here reads the
  // environment.
  r3d_engine_opts o;
  o.size = sizeof o, o.residency = -1, o.pool_slots = 0, o.accumulator_bits = -1, o.lds_reserve = 0;
  if (opts) {
    if (opts->size != sizeof(r3d_engine_opts))
      return g_error = "r3d_engine_opts.size does not match this library's sizeof(r3d_engine_opts)", nullptr;
    o = *opts;
    if (o.residency < -1 || o.residency > RES_NONE)
      return g_error = "r3d_engine_opts.residency must be -1 (automatic), 0, 1 or 2", nullptr;
    if (o.accumulator_bits < -1 || o.accumulator_bits > 8 || (o.accumulator_bits > 0 && o.accumulator_bits < 5))
      return g_error = "r3d_engine_opts.accumulator_bits must be -1 (automatic), 0 (none) or 5..8", nullptr;
    if (o.pool_slots > kSlotStride)
      return g_error = "r3d_engine_opts.pool_slots exceeds the pool's 1024 slots", nullptr;
    if (o.lds_reserve > 160u * 1024u)
      return g_error = "r3d_engine_opts.lds_reserve exceeds the 160 KB of a CU", nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_error = "no HIP device available: the engine has no CPU path";
    return nullptr;
  }
  if (device < 0 || device >= ndev) {
    g_error = "device index out of range";
    return nullptr;
  }
  R3D_ON_DEVICE(device);
  auto e = std::make_unique<r3d_engine>();
  e->device = device;
  e->kind = m->cell_kind;
  e->n_seis = m->n_seismometers;
  e->n_bins = m->params.n_bins;
  KArgs& a = e->args;
  PackedModel pm;
  pack_model(*m, pm, /*for_engine*/ true);
  a = pm.args;
  const size_t cell_bytes = pm.cell_bytes();
  hipError_t err = hipSuccess;
  // ---- move every table into HBM and point the launch arguments at it ----
  switch (m->cell_kind) {
    case R3D_CELL_CYLINDER: a.cells = upload_vec(e.get(), pm.cyl, &err); break;
    case R3D_CELL_TETRA:
      a.cells = upload_vec(e.get(), pm.tet, &err);
      a.rho = upload_vec(e.get(), pm.rho, &err);
      break;
    default: a.cells = upload_vec(e.get(), pm.sph, &err);
  }
  e->n_toa = m->n_toa;
  e->scat_stats.resize(m->n_scatterers);
  e->d_spol.assign(m->n_scatterers, nullptr);
  // ---- the take-off set: uploaded, or generated here (r3d_tables_build.hip) ----
  const double* d_toa = nullptr;   // (theta, phi) pairs
  {
    auto toa_buf = std::make_unique<DevBuf>();
    if (m->toa) {
      if (hipError_t r = toa_buf->upload(m->toa, m->n_toa * 2 * sizeof(double)); r != hipSuccess) err = r;
    } else {
      if (hipError_t r = toa_buf->alloc_zero(m->n_toa * 2 * sizeof(double)); r != hipSuccess) err = r;
      if (err == hipSuccess)
        err = build_toa_on_device(m->toa_degree, m->n_toa, reinterpret_cast<double*>(toa_buf->p), nullptr);
    }
    d_toa = reinterpret_cast<const double*>(toa_buf->p);
    e->d_toa = d_toa;
    e->keep(std::move(toa_buf));
  }
  for (int s = 0; s < m->n_scatterers; s++) {
    const r3d_scatterer& S = m->scatterers[s];
    r3d_engine::ScatStats& st = e->scat_stats[s];
    const double nan = std::nan("");
    if (S.cdf[0]) {   // host-built tables: copy them in
      for (int k = 0; k < 4; k++) {
        pm.scat_ptrs[s].cdf[k] = upload_doubles(e.get(), S.cdf[k], m->n_toa, &err);
        pm.scat_ptrs[s].guide[k] = make_guide(e.get(), pm.scat_ptrs[s].cdf[k], m->n_toa, a.guide_bits, &err);
        st.total[k] = S.cdf[k][m->n_toa - 1];
      }
      e->d_spol[s] = upload_doubles(e.get(), S.spol, m->n_toa, &err);
      pm.scat_ptrs[s].spol_cs = make_spol_cs(e.get(), e->d_spol[s], m->n_toa, &err);
      st.mfp[0] = S.mfp[0], st.mfp[1] = S.mfp[1], st.dipole[0] = st.dipole[1] = nan;
      continue;
    }
    // build-on-device form (r3d_tables_build.hip)
    double* d_cdf[4];
    for (int k = 0; k < 4; k++) {
      auto b = std::make_unique<DevBuf>();
      if (hipError_t r = b->alloc_zero(m->n_toa * sizeof(double)); r != hipSuccess) err = r;
      d_cdf[k] = reinterpret_cast<double*>(e->keep(std::move(b))->p);
    }
    auto sp = std::make_unique<DevBuf>();
    if (hipError_t r = sp->alloc_zero(m->n_toa * sizeof(double)); r != hipSuccess) err = r;
    double* d_spol = reinterpret_cast<double*>(e->keep(std::move(sp))->p);
    if (err != hipSuccess) break;
    double cos_sums[4];
    if (hipError_t r = build_scatterer_tables(S.het, S.psdf_numer, d_toa, m->n_toa, d_cdf, d_spol, st.total,
                                              cos_sums, nullptr);
        r != hipSuccess) {
      err = r;
      break;
    }
    for (int k = 0; k < 4; k++) {
      pm.scat_ptrs[s].cdf[k] = d_cdf[k];
      pm.scat_ptrs[s].guide[k] = make_guide(e.get(), d_cdf[k], m->n_toa, a.guide_bits, &err);
      pm.scat_head[s].total[k] = st.total[k];
    }
    e->d_spol[s] = d_spol;
    pm.scat_ptrs[s].spol_cs = make_spol_cs(e.get(), d_spol, m->n_toa, &err);
    // Mean free paths, conversion table and dipole moments from the totals, as the host
    // builder derives them (scatterers.cpp:172-220, :244-259)
    const double* tot = st.total;
    const double n = (double)m->n_toa;
    st.mfp[0] = S.mfp_fixed ? S.mfp[0] : 1.0 / ((tot[0] + tot[1]) / n);
    st.mfp[1] = S.mfp_fixed ? S.mfp[1] : 1.0 / ((tot[2] + tot[3]) / n);
    auto frac = [](double part, double whole) { return whole == 0 ? 0.0 : part / whole; };
    st.dipole[0] = frac(cos_sums[0], tot[0]) * frac(tot[0], tot[0] + tot[1]) +
                   frac(cos_sums[1], tot[1]) * frac(tot[1], tot[0] + tot[1]);
    st.dipole[1] = frac(cos_sums[2], tot[2]) * frac(tot[2], tot[2] + tot[3]) +
                   frac(cos_sums[3], tot[3]) * frac(tot[3], tot[2] + tot[3]);
    const double wp[2][4] = {{tot[0], tot[1], 0, 0}, {0, 0, tot[2], tot[3]}};
    for (int t = 0; t < 2; t++) {
      pm.scat_head[s].mfp[t] = st.mfp[t];
      double acc = 0;
      for (int k = 0; k < 4; k++) pm.scat_head[s].whole[t][k] = (acc += wp[t][k]);
    }
  }
  if (err == hipSuccess) err = hipDeviceSynchronize();
  e->scat_ptrs = pm.scat_ptrs;
  a.scat_head = upload_vec(e.get(), pm.scat_head, &err);
  a.scat_ptrs = upload_vec(e.get(), pm.scat_ptrs, &err);
  {   // unit vectors of the take-off directions: evaluated in HBM from the (theta, phi) pairs
    auto b = std::make_unique<DevBuf>();
    if (hipError_t r = b->alloc_zero(m->n_toa * 4 * sizeof(double)); r != hipSuccess) err = r;
    if (err == hipSuccess)
      err = build_toa_xyz_on_device(d_toa, m->n_toa, m->params.min_theta, m->params.max_theta,
                                    reinterpret_cast<double*>(b->p), nullptr);
    a.toa_dir = reinterpret_cast<const double*>(e->keep(std::move(b))->p);
  }
  {   // the source's cumulative radiation patterns: copied in, or evaluated here; guides made here
    double* d_src[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < 3; k++) {
      auto b = std::make_unique<DevBuf>();
      if (m->source.cdf[0]) {
        if (hipError_t r = b->upload(m->source.cdf[k], m->n_toa * sizeof(double)); r != hipSuccess) err = r;
      } else if (hipError_t r = b->alloc_zero(m->n_toa * sizeof(double)); r != hipSuccess) {
        err = r;
      }
      d_src[k] = reinterpret_cast<double*>(e->keep(std::move(b))->p);
      a.src_cdf[k] = d_src[k], e->d_src[k] = d_src[k];
    }
    if (!m->source.cdf[0] && err == hipSuccess) {
      double tot[3];
      err = build_source_tables(m->source.moment, d_toa, m->n_toa, d_src, tot, nullptr);
      double acc = 0;
      for (int k = 0; k < 3; k++) a.src_total[k] = tot[k], a.src_whole[k] = (acc += tot[k]);
    }
    for (int k = 0; k < 3; k++) e->src_whole[k] = a.src_whole[k];
    for (int k = 0; k < 3 && err == hipSuccess; k++)
      a.src_guide[k] = make_guide(e.get(), d_src[k], m->n_toa, a.guide_bits, &err);
    if (err == hipSuccess) err = hipDeviceSynchronize();
  }
  a.seis_scan = upload_vec(e.get(), pm.seis_scan, &err);
  a.seis_hit = upload_vec(e.get(), pm.seis_hit, &err);
  a.grid.start = upload_vec(e.get(), pm.grid_start, &err);
  a.grid.items = upload_vec(e.get(), pm.grid_items, &err);
  if (err != hipSuccess) {
    g_error = std::string("uploading model tables: ") + hipGetErrorString(err);
    return nullptr;
  }

  // ---- LDS carve-up and launch geometry ----
  a.grid_n_items = (uint32_t)pm.grid_items.size();
  {
    // Pool kernel (r3d_pool.h): the phonon pool (1024 slots of 128 B in eight record arrays) and its
    // rings, the bin accumulators, and beside them what fits of the small tables: the scatterer
    // heads, then the cell records (two per cell: one per ray type).
    auto align16 = [](size_t x) { return (x + 15) & ~size_t(15); };
    // static LDS of the kernel: queue control words and tallies, 208 bytes; the diagnostic builds'
    // per-phase timers (R3D_PHASE_TIMING, make variant) add 320
#ifdef R3D_PHASE_TIMING
    size_t kStatic = 1024;
#else
    size_t kStatic = 512;
#endif
    // opts.lds_reserve: LDS the carve-up must leave alone, as if the model's tables were that much
    // larger -- how the tests reach the paths for models with more cells
    kStatic += (size_t)o.lds_reserve / 16 * 16;
    const size_t kLds = 160 * 1024;
    const size_t head_bytes = (size_t)m->n_scatterers * sizeof(ScatHead);
    const size_t scat_bytes = head_bytes + (size_t)m->n_scatterers * sizeof(ScatPtrs);   // heads + table addresses
    // opts.residency = 1 / 2: run the kernel variant that keeps the cell records (1) or also the
    // scatterer heads (2) in HBM although they would fit in LDS, so that every compiled variant can
    // be held against the oracle on any model
    const int force_res = o.residency < 0 ? 0 : o.residency;
    uint32_t acc_bits = m->n_seismometers > 0 ? 8u : 0u;   // 256 accumulators = 13 KB ...
    if (o.accumulator_bits >= 0) acc_bits = (uint32_t)o.accumulator_bits;
    const size_t pool_bytes = (size_t)kSlotStride * kSlotBytes, ring_bytes = (size_t)Q_NUM * kSlotStride * sizeof(uint16_t);
    // what the pool, its rings and `acc` bytes of accumulators leave for tables (64: alignment of up to four blocks)
    auto room_beside = [&](size_t acc) {
      const size_t fixed = kStatic + pool_bytes + ring_bytes + 64 + acc;
      return fixed < kLds ? kLds - fixed : size_t(0);
    };
    // (a layered or spherical model whose cell records do not fit beside 256 accumulators but do beside
    //  128 gets 128: the hot first-arrival bins are a few dozen, and records from LDS are worth more)
    if (acc_bits == 8 && o.accumulator_bits < 0 && m->cell_kind != R3D_CELL_TETRA) {
      if (cell_bytes + scat_bytes > room_beside(kAccEntryBytes << 8) && cell_bytes + scat_bytes <= room_beside(kAccEntryBytes << 7))
        acc_bits = 7;
    }
    const size_t acc_bytes = acc_bits ? (kAccEntryBytes << acc_bits) : 0;
    // The pool's field arrays have kSlotStride entries each (r3d_pool.h: a constant distance between a
    // slot's fields), 128 KB in all, and the rings one entry per slot: what is staged beside them is
    // what the remaining ~20 KB hold -- the bin accumulators first, then the scatterer heads, then
    // the cell records.
    const size_t room = room_beside(acc_bytes);
    const bool scat_fit = scat_bytes <= room && force_res < RES_NONE;
    const bool cells_fit = scat_fit && cell_bytes + scat_bytes <= room && force_res < RES_TABLES &&
                           m->cell_kind != R3D_CELL_TETRA;
    e->res = cells_fit ? RES_ALL : scat_fit ? RES_TABLES : RES_NONE;
    size_t off = 0;
    a.lds_cells_off = a.lds_scat_off = a.lds_seis_off = a.lds_hit_off = a.lds_grid_off = 0xFFFFFFFFu;
    if (cells_fit) a.lds_cells_off = (uint32_t)off, off = align16(off + cell_bytes);
    a.lds_scatptr_off = 0xFFFFFFFFu;
    if (scat_fit) {
      a.lds_scat_off = (uint32_t)off, off = align16(off + head_bytes);
      a.lds_scatptr_off = (uint32_t)off, off = align16(off + scat_bytes - head_bytes);
    }
    a.lds_acc_off = (uint32_t)off, a.acc_bits = acc_bits;
    if (acc_bits) off = align16(off + acc_bytes);
    uint32_t slots = kSlotStride, cap = kSlotStride;
    if (o.pool_slots) {   // fewer slots in circulation (tests of a crowded pool, tuning)
      uint32_t want = o.pool_slots / 64 * 64;   // at least a slot per lane of the workgroup
      if (want < (uint32_t)kPoolBlock) want = (uint32_t)kPoolBlock;
      if (want < slots) {
        slots = want, cap = 64;
        while (cap < slots) cap <<= 1;
      }
    }
    a.pool_slots = slots, a.pool_ring_mask = cap - 1;
    a.lds_pool_off = (uint32_t)off, off = align16(off + pool_bytes);
    a.lds_ring_off = (uint32_t)off, off = align16(off + (size_t)Q_NUM * cap * sizeof(uint16_t));
    e->lds_bytes = off;
    if (e->lds_bytes + kStatic > kLds) {
      g_error = "internal error: LDS carve-up exceeds the 160 KB of a CU";
      return nullptr;
    }
  }
  hipDeviceProp_t prop;
  R3D_HIP_OK(hipGetDeviceProperties(&prop, device));
  if (e->lds_bytes > 64 * 1024) R3D_HIP_OK(set_lds_attr(e.get()));
  // persistent grid: one workgroup per CU (the pool takes the CU's whole LDS), all resident at
  // once, so every workgroup is running while there is work and none queues behind
  e->grid_blocks = prop.multiProcessorCount;
  e->carry_bytes = (size_t)e->grid_blocks * kSlotStride * kSlotBytes;

  // ---- result scratch, work counter, stream, events ----
  R3D_HIP_OK(e->d_energy.alloc_zero((size_t)std::max(1, e->n_seis) * e->n_bins * R3D_N_ENERGY * sizeof(double)));
  R3D_HIP_OK(e->d_counts.alloc_zero((size_t)std::max(1, e->n_seis) * e->n_bins * R3D_N_COUNT * sizeof(uint64_t)));
  R3D_HIP_OK(e->d_scalars.alloc_zero(R3D_N_SCALARS * sizeof(uint64_t)));
  R3D_HIP_OK(e->d_next.alloc_zero(r3d_engine::kCounters * sizeof(unsigned long long)));
  R3D_HIP_OK(hipStreamCreate(&e->stream));
  for (unsigned i = 0; i < r3d_engine::kCounters; i++) {
    R3D_HIP_OK(hipEventCreate(&e->ev0[i]));
    R3D_HIP_OK(hipEventCreate(&e->ev1[i]));
  }
  return e.release();
}

int r3d_engine_carry_pending(const r3d_engine* e) { return (e && e->carry_pending) ? 1 : 0; }

int r3d_engine_close(r3d_engine* e) {
  const int fail_value = 1;
  if (!e) return 0;
  if (e->carry_pending)
    return g_error = "histories carried over from the last r3d_run_device_carry launch are still in the engine: "
                     "flush them (final != 0) before closing it, or their tallies and bins are lost", 1;
  {
    R3D_ON_DEVICE(e->device);
    R3D_HIP_OK(hipDeviceSynchronize());
    delete e;
  }
  return 0;
}

void r3d_engine_destroy(r3d_engine* e) {
  if (!e) return;
  if (e->carry_pending)   // (a void call cannot refuse: leave the fact where the caller can find it)
    g_error = "r3d_engine_destroy: carried histories were dropped unflushed (see r3d_engine_close)";
  DeviceGuard guard(e->device);
  delete e;
}

size_t r3d_energy_len(const r3d_engine* e) { return e ? (size_t)e->n_seis * e->n_bins * R3D_N_ENERGY : 0; }
size_t r3d_counts_len(const r3d_engine* e) { return e ? (size_t)e->n_seis * e->n_bins * R3D_N_COUNT : 0; }

// carry: 0 none, 1 resume carried histories and park the unfinished ones, 2 resume and finish all
static int enqueue(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, double* d_energy,
                   uint64_t* d_counts, uint64_t* d_scalars, r3d_final* d_finals, hipStream_t s,
                   int carry = 0) {
  const int fail_value = 1;
  if (!e) return g_error = "null engine", 1;
  if (!d_energy || !d_counts || !d_scalars) return g_error = "null result buffer", 1;
  R3D_ON_DEVICE(e->device);
  KArgs a = e->args;
  a.n = n, a.first_id = first_id, a.seed = seed;
  // the launch's work counter and event pair: slot (id - 1) % kCounters of the id it will get
  // once it is enqueued -- what r3d_kernel_ms(e, id) reads; a rejected call takes nothing
  const unsigned slot = (unsigned)(e->launches % r3d_engine::kCounters);
  a.next = reinterpret_cast<unsigned long long*>(e->d_next.p) + slot;
  a.energy = d_energy;
  a.counts = reinterpret_cast<unsigned long long*>(d_counts);
  a.scalars = reinterpret_cast<unsigned long long*>(d_scalars);
  a.finals = d_finals;
  a.carry_in = a.carry_out = nullptr;
  if (a.pfinals && n && (first_id < e->pfinals_base || first_id - e->pfinals_base > e->pfinals_cap || n > e->pfinals_cap - (first_id - e->pfinals_base)))
    return g_error = "the production finals buffer does not cover this launch's ids (r3d_engine_set_production_finals)", 1;
  bool must_launch = n > 0;
  bool pending_after = e->carry_pending;
  if (carry) {
    if (d_finals) return g_error = "final records and carry-over cannot be combined", 1;
    if (e->carry_pending && seed != e->carry_seed)
      return g_error = "carried histories were started under another seed", 1;
    if (!e->d_carry) {
      auto buf = std::make_unique<DevBuf>();
      R3D_HIP_OK(buf->alloc_zero(e->carry_bytes));
      e->d_carry = std::move(buf);
    }
    if (e->carry_pending) a.carry_in = e->d_carry->p, must_launch = true;
    if (carry == 1) a.carry_out = e->d_carry->p;
    pending_after = (carry == 1) && (n > 0 || e->carry_pending);
  }
  R3D_HIP_OK(hipMemsetAsync(a.next, 0, sizeof(unsigned long long), s));
  R3D_HIP_OK(hipEventRecord(e->ev0[slot], s));
  if (must_launch)
    R3D_HIP_OK(launch_any(e, a, d_finals != nullptr || a.evlog != nullptr, s, /*drain_only*/ carry == 2 && n == 0));
  R3D_HIP_OK(hipEventRecord(e->ev1[slot], s));
  // enqueued: only now does the engine's state move on
  if (carry) e->carry_pending = pending_after, e->carry_seed = seed;
  e->launches++;
  return 0;
}

int r3d_run_device(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, double* d_energy,
                   uint64_t* d_counts, uint64_t* d_scalars, r3d_final* d_finals, void* stream) {
  // NULL means HIP's default (null) stream, which is also what torch's
  // default stream is, so later work queued there is ordered after the kernel.
  return enqueue(e, n, first_id, seed, d_energy, d_counts, d_scalars, d_finals,
                 reinterpret_cast<hipStream_t>(stream));
}

int r3d_run_device_carry(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, double* d_energy,
                         uint64_t* d_counts, uint64_t* d_scalars, void* stream, int final) {
  return enqueue(e, n, first_id, seed, d_energy, d_counts, d_scalars, nullptr,
                 reinterpret_cast<hipStream_t>(stream), final ? 2 : 1);
}

static int run_host(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, r3d_result* out,
                    r3d_final* finals) {
  const int fail_value = 1;
  if (!e) return g_error = "null engine", 1;
  if (!out || !out->energy || !out->counts) return g_error = "null result", 1;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipMemsetAsync(e->d_energy.p, 0, e->d_energy.bytes, e->stream));
  R3D_HIP_OK(hipMemsetAsync(e->d_counts.p, 0, e->d_counts.bytes, e->stream));
  R3D_HIP_OK(hipMemsetAsync(e->d_scalars.p, 0, e->d_scalars.bytes, e->stream));
  DevBuf d_finals;
  if (finals) R3D_HIP_OK(d_finals.alloc_zero(n * sizeof(r3d_final)));
  if (enqueue(e, n, first_id, seed, reinterpret_cast<double*>(e->d_energy.p),
              reinterpret_cast<uint64_t*>(e->d_counts.p), reinterpret_cast<uint64_t*>(e->d_scalars.p),
              finals ? reinterpret_cast<r3d_final*>(d_finals.p) : nullptr, e->stream))
    return 1;
  R3D_HIP_OK(hipStreamSynchronize(e->stream));
  const size_t ne = r3d_energy_len(e), nc = r3d_counts_len(e);
  std::vector<double> he(ne);
  std::vector<uint64_t> hc(nc);
  uint64_t hs[R3D_N_SCALARS];
  if (ne) R3D_HIP_OK(hipMemcpy(he.data(), e->d_energy.p, ne * sizeof(double), hipMemcpyDeviceToHost));
  if (nc) R3D_HIP_OK(hipMemcpy(hc.data(), e->d_counts.p, nc * sizeof(uint64_t), hipMemcpyDeviceToHost));
  R3D_HIP_OK(hipMemcpy(hs, e->d_scalars.p, sizeof hs, hipMemcpyDeviceToHost));
  if (finals) R3D_HIP_OK(hipMemcpy(finals, d_finals.p, n * sizeof(r3d_final), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < ne; i++) out->energy[i] += he[i];
  for (size_t i = 0; i < nc; i++) out->counts[i] += hc[i];
  out->n_lost += hs[0], out->n_timeout += hs[1], out->n_invalid += hs[2];
  for (int r = 0; r < R3D_INV_NUM; r++) out->invalid_reasons[r] += hs[3 + r];
  for (int k = 0; k < R3D_EV_NUM; k++) out->events[k] += hs[3 + R3D_INV_NUM + k];
  return 0;
}

int r3d_run(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, r3d_result* out) {
  return run_host(e, n, first_id, seed, out, nullptr);
}

// ---- a node: one engine per shard, kept alive across runs, the shards' blocks summed ON THE DEVICES ------------
// (include/r3d.h r3d_node_*).  Each run: every shard's kernel is enqueued on its engine's stream into the engine's
// own block in HBM; then ONE grouped reduce (sum) per buffer -- f64 energies, u64 counts, u64 counters: reduce_block,
// r3d_rccl.h -- over the shards' streams brings the job's totals to shard 0's device, and the host reads that one
// block.  This is the reference's "replicas + combine" (scripts/do-parallel.sh:23-29, vis/seisplot/combine.m:26-33)
// with the combine done by RCCL over xGMI.  The blocks are added on the HOST instead (r3d_node_reduction says
// "host") when there is nothing for RCCL to do or it cannot do it: a node of one shard (its block is the job's),
// shards that share a GPU (RCCL refuses a communicator that names a device twice -- tests, or a user who wants
// it), no usable librccl, a communicator that could not be formed, or one that failed in an earlier run.
struct r3d_node {
  std::vector<int> devices;
  std::vector<r3d_engine*> engines;
  std::vector<ncclComm_t> comms;   // empty: host sum
  std::string host_why;            // why the host sums (when it does)
  DevBuf sum_energy, sum_counts, sum_scalars;   // on devices[0]: where the reduce puts the job's totals
  size_t ne = 0, nc = 0;
  uint64_t runs = 0;
};

// A communicator that failed in the middle of a group cannot be waited for (its kernels may be waiting for peers
// that were never enqueued): every rank's is aborted, which also ends what they have in flight, and the node goes
// on with the host sum.
static void node_abort_comms(r3d_node* nd, const std::string& why) {
  if (const Rccl* R = rccl())
    for (size_t g = 0; g < nd->comms.size(); g++) {
      DeviceGuard on(nd->devices[g]);
      if (nd->comms[g]) (void)R->CommAbort(nd->comms[g]);
    }
  nd->comms.clear();
  nd->host_why = why;
}

void r3d_node_destroy(r3d_node* nd) {
  if (!nd) return;
  if (const Rccl* R = rccl())
    for (size_t g = 0; g < nd->comms.size(); g++) {
      DeviceGuard on(nd->devices[g]);
      if (nd->comms[g]) (void)R->CommDestroy(nd->comms[g]);
    }
  for (r3d_engine* e : nd->engines)
    if (e) r3d_engine_destroy(e);
  DeviceGuard on(nd->devices.empty() ? 0 : nd->devices[0]);   // (the reduced block is freed on its device)
  delete nd;
}

r3d_node* r3d_node_create(const r3d_model_desc* model, const int* devices, int n_devices) {
  if (!model) return g_error = "null model", nullptr;
  if (n_devices < 1 || !devices) return g_error = "at least one device is needed (the engine has no CPU path)", nullptr;
  auto nd = std::make_unique<r3d_node>();
  nd->devices.assign(devices, devices + n_devices);
  nd->ne = (size_t)model->n_seismometers * model->params.n_bins * R3D_N_ENERGY;
  nd->nc = (size_t)model->n_seismometers * model->params.n_bins * R3D_N_COUNT;
  nd->engines.assign(n_devices, nullptr);
  // the engines are built side by side (tables made in HBM: tens of milliseconds; uploaded: seconds each)
  std::vector<std::string> errors(n_devices);
  {
    std::vector<std::thread> pool;
    for (int g = 0; g < n_devices; g++)
      pool.emplace_back([&, g] {
        nd->engines[g] = r3d_engine_create(model, devices[g]);
        if (!nd->engines[g]) errors[g] = "shard " + std::to_string(g) + " (device " + std::to_string(devices[g]) + "): " + g_error;
      });
    for (auto& t : pool) t.join();
  }
  for (int g = 0; g < n_devices; g++)
    if (!errors[g].empty()) {
      g_error = errors[g];
      r3d_node_destroy(nd.release());
      return nullptr;
    }
  bool distinct = true;
  for (int g = 0; g < n_devices; g++)
    for (int h = 0; h < g; h++) distinct = distinct && devices[g] != devices[h];
  if (n_devices == 1) nd->host_why = "one shard: its block is the job's";
  else if (!distinct) nd->host_why = "shards share a device";
  else {
    std::string why;
    const Rccl* R = rccl(&why);
    if (!R) nd->host_why = why;
    else {
      nd->comms.assign(n_devices, nullptr);
      const ncclResult_t r = R->CommInitAll(nd->comms.data(), n_devices, devices);
      if (r != ncclSuccess) {   // (the job can still be run: the blocks are then added on the host)
        nd->comms.clear();
        nd->host_why = std::string("ncclCommInitAll: ") + R->GetErrorString(r);
      }
    }
  }
  if (!nd->comms.empty()) {
    DeviceGuard on(devices[0]);
    if (on.status != hipSuccess || nd->sum_energy.alloc_zero(std::max<size_t>(nd->ne, 1) * sizeof(double)) != hipSuccess ||
        nd->sum_counts.alloc_zero(std::max<size_t>(nd->nc, 1) * sizeof(uint64_t)) != hipSuccess ||
        nd->sum_scalars.alloc_zero(R3D_N_SCALARS * sizeof(uint64_t)) != hipSuccess) {
      g_error = "r3d_node_create: no memory for the reduced block";
      r3d_node_destroy(nd.release());
      return nullptr;
    }
  }
  return nd.release();
}

int r3d_node_size(const r3d_node* nd) { return nd ? (int)nd->engines.size() : 0; }
r3d_engine* r3d_node_engine(r3d_node* nd, int shard) {
  if (!nd || shard < 0 || shard >= (int)nd->engines.size()) return g_error = "r3d_node_engine: no such shard", nullptr;
  return nd->engines[shard];
}
const char* r3d_node_reduction(const r3d_node* nd) { return !nd ? "" : nd->comms.empty() ? "host" : "rccl"; }
const char* r3d_node_reduction_note(const r3d_node* nd) { return !nd ? "" : nd->host_why.c_str(); }

int r3d_node_run(r3d_node* nd, uint64_t n, uint64_t first_id, uint64_t seed, r3d_result* out) {
  const int fail_value = 1;
  if (!nd) return g_error = "null node", 1;
  if (!out || !out->energy || !out->counts) return g_error = "null result", 1;
  const int N = (int)nd->engines.size();
  const size_t ne = nd->ne, nc = nd->nc;
  // every shard's launch, asynchronously, each on its engine's stream into its engine's block
  for (int g = 0; g < N; g++) {
    r3d_engine* e = nd->engines[g];
    const uint64_t lo = n / N * g + std::min<uint64_t>(g, n % N);
    const uint64_t cnt = n / N + ((uint64_t)g < n % N ? 1 : 0);
    auto shard_error = [&](const std::string& what) {
      g_error = "shard " + std::to_string(g) + " (device " + std::to_string(nd->devices[g]) + "): " + what;
      for (int h = 0; h <= g; h++) {   // what was enqueued runs out before anyone reads or reuses the blocks
        DeviceGuard on(nd->devices[h]);
        (void)hipStreamSynchronize(nd->engines[h]->stream);
      }
      return 1;
    };
    DeviceGuard on(e->device);
    if (on.status != hipSuccess || hipMemsetAsync(e->d_energy.p, 0, e->d_energy.bytes, e->stream) != hipSuccess ||
        hipMemsetAsync(e->d_counts.p, 0, e->d_counts.bytes, e->stream) != hipSuccess ||
        hipMemsetAsync(e->d_scalars.p, 0, e->d_scalars.bytes, e->stream) != hipSuccess)
      return shard_error("cannot clear the result block");
    if (enqueue(e, cnt, first_id + lo, seed, reinterpret_cast<double*>(e->d_energy.p), reinterpret_cast<uint64_t*>(e->d_counts.p),
                reinterpret_cast<uint64_t*>(e->d_scalars.p), nullptr, e->stream))
      return shard_error(g_error);
  }
  std::vector<double> he(std::max<size_t>(ne, 1), 0.0);
  std::vector<uint64_t> hc(std::max<size_t>(nc, 1), 0);
  uint64_t hs[R3D_N_SCALARS] = {0};
  auto wait_all = [&]() -> int {
    for (int g = 0; g < N; g++) {
      R3D_ON_DEVICE(nd->devices[g]);
      R3D_HIP_OK(hipStreamSynchronize(nd->engines[g]->stream));
    }
    return 0;
  };
  bool summed = false;
  if (!nd->comms.empty()) {
    // one grouped reduce per buffer, in stream order behind each shard's kernel: the sums land on devices[0]
    const Rccl& R = *rccl();
    ncclResult_t bad = R.GroupStart();
    if (bad == ncclSuccess) {
      for (int g = 0; g < N && bad == ncclSuccess; g++) {
        r3d_engine* e = nd->engines[g];
        DeviceGuard on(e->device);   // (each rank's calls with its own device current)
        // (the receive buffers only mean something on the root; the other ranks name their own block, in place)
        bad = reduce_block(R, nd->comms[g], e->stream, 0, e->d_energy.p, g == 0 ? nd->sum_energy.p : e->d_energy.p, ne,
                           e->d_counts.p, g == 0 ? nd->sum_counts.p : e->d_counts.p, nc,
                           e->d_scalars.p, g == 0 ? nd->sum_scalars.p : e->d_scalars.p, R3D_N_SCALARS);
      }
      const ncclResult_t end = R.GroupEnd();   // (a call that fails inside the group must not leave the group open)
      if (bad == ncclSuccess) bad = end;
    }
    if (bad == ncclSuccess) {
      if (wait_all()) return 1;
      R3D_ON_DEVICE(nd->devices[0]);
      if (ne) R3D_HIP_OK(hipMemcpy(he.data(), nd->sum_energy.p, ne * sizeof(double), hipMemcpyDeviceToHost));
      if (nc) R3D_HIP_OK(hipMemcpy(hc.data(), nd->sum_counts.p, nc * sizeof(uint64_t), hipMemcpyDeviceToHost));
      R3D_HIP_OK(hipMemcpy(hs, nd->sum_scalars.p, sizeof hs, hipMemcpyDeviceToHost));
      summed = true;
    } else {
      // Some ranks' reduces may be enqueued and waiting for peers that never will be: nobody waits for them.  The
      // communicators are aborted (that also ends their kernels), the node sums on the host from now on -- and this
      // run too: the shards' kernels wrote their own blocks, which a reduce to shard 0's separate buffer left alone.
      node_abort_comms(nd, std::string("the RCCL reduce failed (") + R.GetErrorString(bad) + "): communicators aborted");
    }
  }
  if (!summed) {
    if (wait_all()) return 1;
    std::vector<double> te(std::max<size_t>(ne, 1));
    std::vector<uint64_t> tc(std::max<size_t>(nc, 1));
    uint64_t ts[R3D_N_SCALARS];
    for (int g = 0; g < N; g++) {
      r3d_engine* e = nd->engines[g];
      R3D_ON_DEVICE(e->device);
      if (ne) R3D_HIP_OK(hipMemcpy(te.data(), e->d_energy.p, ne * sizeof(double), hipMemcpyDeviceToHost));
      if (nc) R3D_HIP_OK(hipMemcpy(tc.data(), e->d_counts.p, nc * sizeof(uint64_t), hipMemcpyDeviceToHost));
      R3D_HIP_OK(hipMemcpy(ts, e->d_scalars.p, sizeof ts, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < ne; i++) he[i] += te[i];
      for (size_t i = 0; i < nc; i++) hc[i] += tc[i];
      for (int i = 0; i < R3D_N_SCALARS; i++) hs[i] += ts[i];
    }
  }
  // (nothing was added to *out until every shard had run and the sums were read)
  for (size_t i = 0; i < ne; i++) out->energy[i] += he[i];
  for (size_t i = 0; i < nc; i++) out->counts[i] += hc[i];
  out->n_lost += hs[0], out->n_timeout += hs[1], out->n_invalid += hs[2];
  for (int r = 0; r < R3D_INV_NUM; r++) out->invalid_reasons[r] += hs[3 + r];
  for (int k = 0; k < R3D_EV_NUM; k++) out->events[k] += hs[3 + R3D_INV_NUM + k];
  nd->runs++;
  return 0;
}

// ---- one process per GPU: this rank's handle of an RCCL communicator (include/r3d.h r3d_comm_*) ------------------
struct r3d_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 0, device = 0;
  bool broken = false;
};

int r3d_comm_unique_id(unsigned char id[R3D_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == R3D_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id) return g_error = "null id", 1;
  std::string why;
  const Rccl* R = rccl(&why);
  if (!R) return g_error = why, 1;
  ncclUniqueId u;
  const ncclResult_t r = R->GetUniqueId(&u);
  if (r != ncclSuccess) return g_error = std::string("ncclGetUniqueId: ") + R->GetErrorString(r), 1;
  std::memcpy(id, &u, sizeof u);
  return 0;
}

r3d_comm* r3d_comm_create(const unsigned char id[R3D_COMM_ID_BYTES], int rank, int n_ranks, int device) {
  if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return g_error = "r3d_comm_create: bad rank or id", nullptr;
  std::string why;
  const Rccl* R = rccl(&why);
  if (!R) return g_error = why, nullptr;
  DeviceGuard on(device);
  if (on.status != hipSuccess) return g_error = std::string("r3d_comm_create: device ") + std::to_string(device) + ": " + hipGetErrorString(on.status), nullptr;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  auto c = std::make_unique<r3d_comm>();
  c->rank = rank, c->device = device;
  ncclResult_t r = R->CommInitRank(&c->comm, n_ranks, u, rank);
  if (r != ncclSuccess) return g_error = std::string("ncclCommInitRank: ") + R->GetErrorString(r), nullptr;
  r = R->CommCount(c->comm, &c->n_ranks);   // (what RCCL itself says the communicator's size is)
  if (r != ncclSuccess || c->n_ranks != n_ranks) {
    g_error = "r3d_comm_create: the communicator reports " + std::to_string(c->n_ranks) + " ranks, " + std::to_string(n_ranks) + " were asked for";
    (void)R->CommAbort(c->comm);
    return nullptr;
  }
  return c.release();
}

int r3d_comm_reduce(r3d_comm* c, double* d_energy, uint64_t n_energy, uint64_t* d_counts, uint64_t n_counts,
                    uint64_t* d_scalars, uint64_t n_scalars, int root, void* stream) {
  if (!c || !c->comm) return g_error = "null communicator", 1;
  if (c->broken) return g_error = "r3d_comm_reduce: the communicator failed earlier and was aborted", 1;
  if (root >= c->n_ranks) return g_error = "r3d_comm_reduce: no such root", 1;
  const Rccl& R = *rccl();
  DeviceGuard on(c->device);
  if (on.status != hipSuccess) return g_error = std::string("r3d_comm_reduce: ") + hipGetErrorString(on.status), 1;
  ncclResult_t bad = R.GroupStart();
  if (bad == ncclSuccess) {
    bad = reduce_block(R, c->comm, static_cast<hipStream_t>(stream), root, d_energy, d_energy, n_energy, d_counts, d_counts, n_counts,
                       d_scalars, d_scalars, n_scalars);
    const ncclResult_t end = R.GroupEnd();
    if (bad == ncclSuccess) bad = end;
  }
  if (bad != ncclSuccess) {
    g_error = std::string("r3d_comm_reduce: ") + R.GetErrorString(bad);
    (void)R.CommAbort(c->comm);   // (what is enqueued may wait for peers for ever: ended here, not waited for)
    c->comm = nullptr, c->broken = true;
    return 1;
  }
  return 0;
}

int r3d_comm_describe(const r3d_comm* c, r3d_comm_info* info) {
  if (!c || !info) return g_error = "null argument", 1;
  std::memset(info, 0, sizeof *info);
  const Rccl* R = rccl();
  info->n_ranks = c->n_ranks, info->rank = c->rank, info->device = c->device;
  info->rccl_version = R ? R->version : 0;
  if (R) std::snprintf(info->library, sizeof info->library, "%s", R->where.c_str());
  hipUUID uuid;
  if (hipDeviceGetUuid(&uuid, c->device) == hipSuccess)
    for (int i = 0; i < 16; i++) std::snprintf(info->device_uuid + 2 * i, 3, "%02x", (unsigned)(unsigned char)uuid.bytes[i]);
  return 0;
}

void r3d_comm_destroy(r3d_comm* c) {
  if (!c) return;
  if (c->comm)
    if (const Rccl* R = rccl()) {
      DeviceGuard on(c->device);
      (void)R->CommDestroy(c->comm);
    }
  delete c;
}

int r3d_run_model_on(const r3d_model_desc* model, uint64_t n, uint64_t first_id, uint64_t seed,
                     const int* devices, int n_devices, r3d_result* out) {
  if (!model || !out || !out->energy || !out->counts) return g_error = "null argument", 1;
  r3d_node* nd = r3d_node_create(model, devices, n_devices);
  if (!nd) return 1;
  const int rc = r3d_node_run(nd, n, first_id, seed, out);
  const std::string keep = rc ? g_error : std::string();
  r3d_node_destroy(nd);
  if (rc) g_error = keep;
  return rc;
}

int r3d_run_model(const r3d_model_desc* model, uint64_t n, uint64_t first_id, uint64_t seed, int n_gpus,
                  r3d_result* out) {
  if (n_gpus < 1) return g_error = "n_gpus must be at least 1 (the engine has no CPU path)", 1;
  std::vector<int> devices(n_gpus);
  for (int g = 0; g < n_gpus; g++) devices[g] = g;
  return r3d_run_model_on(model, n, first_id, seed, devices.data(), n_gpus, out);
}

int r3d_run_traced(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, r3d_result* out,
                   r3d_final* finals) {
  if (!finals) return g_error = "null finals", 1;
  return run_host(e, n, first_id, seed, out, finals);
}

// ---- self-test of the lean elementary functions (include/r3d.h r3d_selftest_math) ----
__global__ void selftest_math_kernel(int which, const double* x, const double* y, double* out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;   // (n is padded to whole waves by the caller's choice of inputs or not: the votes see active lanes only)
  const double a = x[i], b = y ? y[i] : 0.0;
  double s = 0.0, c = 0.0, r = 0.0;
  switch (which) {   // (wave-uniform)
    case 0: r = exp_lean(a); break;
    case 1: r = log_lean(a); break;
    case 2: r = atanh_lean(a); break;
    case 3: r = asin_small(a); break;
    case 4: r = angle_from_sincos(a, b); break;
    case 5: rotation(a, &s, &c), r = s; break;
    case 6: rotation(a, &s, &c), r = c; break;
    case 7: r = frcp(a); break;
    case 8: r = frsqrt(a); break;
    case 9: r = fsqrt(a); break;
    default: r = a - a; break;
  }
  out[i] = r;
}
int r3d_selftest_math(int device, int which, const double* x, const double* y, double* out, uint64_t n) {
  const int fail_value = 1;
  if (!x || !out || which < 0 || which > 9) return g_error = "r3d_selftest_math: bad arguments", 1;
  if (n == 0) return 0;
  R3D_ON_DEVICE(device);
  DevBuf dx, dy, dout;
  R3D_HIP_OK(dx.upload(x, n * sizeof(double)));
  if (y) R3D_HIP_OK(dy.upload(y, n * sizeof(double)));
  R3D_HIP_OK(dout.upload(x, n * sizeof(double)));
  selftest_math_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(which, (const double*)dx.p, y ? (const double*)dy.p : nullptr,
                                                                      (double*)dout.p, n);
  R3D_HIP_OK(hipGetLastError());
  R3D_HIP_OK(hipMemcpy(out, dout.p, n * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

#ifdef R3D_PHASE_TIMING
// diagnostic builds only: per queue of the pool kernel, batches served / lanes filled / wave cycles
// since the last call (slot 6 of the first row: idle polls), summed over the three kernel units
int r3d_debug_pool_stats(unsigned long long out[40]) {
  for (int i = 0; i < 40; i++) out[i] = 0;
  unsigned long long part[40];
  int (*const read[3])(unsigned long long*) = {pool_stats_cyl, pool_stats_tet, pool_stats_sph};
  for (int k = 0; k < 3; k++) {
    if (read[k](part)) return 1;
    for (int i = 0; i < 40; i++) out[i] += part[i];
  }
  return 0;
}
#endif

int r3d_engine_scatterer_stats(const r3d_engine* e, int s, double out[8]) {
  if (!e || !out || s < 0 || s >= (int)e->scat_stats.size()) return g_error = "scatterer index out of range", 1;
  const r3d_engine::ScatStats& st = e->scat_stats[s];
  out[0] = st.mfp[0], out[1] = st.mfp[1], out[2] = st.dipole[0], out[3] = st.dipole[1];
  for (int k = 0; k < 4; k++) out[4 + k] = st.total[k];
  return 0;
}

int r3d_engine_download_scatterer(r3d_engine* e, int s, double* cdf[4], double* spol) {
  const int fail_value = 1;
  if (!e || s < 0 || s >= (int)e->scat_ptrs.size()) return g_error = "scatterer index out of range", 1;
  R3D_ON_DEVICE(e->device);
  const size_t bytes = e->n_toa * sizeof(double);
  for (int k = 0; k < 4; k++)
    if (cdf && cdf[k]) R3D_HIP_OK(hipMemcpy(cdf[k], e->scat_ptrs[s].cdf[k], bytes, hipMemcpyDeviceToHost));
  if (spol) R3D_HIP_OK(hipMemcpy(spol, e->d_spol[s], bytes, hipMemcpyDeviceToHost));
  return 0;
}

int r3d_engine_download_source(r3d_engine* e, double* cdf[3], double whole[3]) {
  const int fail_value = 1;
  if (!e) return g_error = "null engine", 1;
  R3D_ON_DEVICE(e->device);
  for (int k = 0; k < 3; k++) {
    if (cdf && cdf[k]) R3D_HIP_OK(hipMemcpy(cdf[k], e->d_src[k], e->n_toa * sizeof(double), hipMemcpyDeviceToHost));
    if (whole) whole[k] = e->src_whole[k];
  }
  return 0;
}

int r3d_engine_download_toa(r3d_engine* e, double* toa) {
  const int fail_value = 1;
  if (!e || !toa) return g_error = "null argument", 1;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipMemcpy(toa, e->d_toa, e->n_toa * 2 * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int r3d_engine_set_event_log(r3d_engine* e, uint32_t mask, uint64_t capacity) {
  const int fail_value = 1;
  if (!e) return g_error = "null engine", 1;
  R3D_ON_DEVICE(e->device);
  KArgs& a = e->args;
  a.evlog = nullptr, a.evlog_count = nullptr, a.evlog_cap = 0, a.evlog_mask = 0;
  e->d_evlog.reset(), e->d_evlog_count.reset();
  if (mask == 0 || capacity == 0) return 0;
  auto buf = std::make_unique<DevBuf>(), cnt = std::make_unique<DevBuf>();
  R3D_HIP_OK(buf->alloc_zero(capacity * sizeof(r3d_event)));
  R3D_HIP_OK(cnt->alloc_zero(sizeof(unsigned long long)));
  a.evlog = buf->p, a.evlog_count = reinterpret_cast<unsigned long long*>(cnt->p);
  a.evlog_cap = capacity, a.evlog_mask = mask & R3D_RPT_ALL;
  e->d_evlog = std::move(buf), e->d_evlog_count = std::move(cnt);
  return 0;
}

int r3d_engine_set_production_finals(r3d_engine* e, uint64_t base_id, uint64_t capacity) {
  const int fail_value = 1;
  if (!e) return g_error = "null engine", 1;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipDeviceSynchronize());   // (launches that write into the old buffer)
  if (e->carry_pending) return g_error = "r3d_engine_set_production_finals: histories are carried over from a launch made under the old setting (close the chain first)", 1;
  KArgs& a = e->args;
  a.pfinals = nullptr, e->pfinals_base = 0, e->pfinals_cap = 0;
  e->d_pfinals.reset();
  if (capacity == 0) return 0;
  auto buf = std::make_unique<DevBuf>();
  R3D_HIP_OK(buf->alloc_zero(capacity * sizeof(r3d_final)));
  R3D_HIP_OK(hipMemset(buf->p, 0xFF, capacity * sizeof(r3d_final)));   // (fate 255: no record written)
  // (the kernel indexes by the history id itself: the address less base_id records)
  a.pfinals = reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(buf->p) - (uintptr_t)base_id * sizeof(r3d_final));
  e->pfinals_base = base_id, e->pfinals_cap = capacity;
  e->d_pfinals = std::move(buf);
  return 0;
}

int r3d_production_finals_read(r3d_engine* e, r3d_final* out, uint64_t first, uint64_t count) {
  const int fail_value = 1;
  if (!e || !e->d_pfinals) return g_error = "no production finals attached", 1;
  if (first > e->pfinals_cap || count > e->pfinals_cap - first) return g_error = "r3d_production_finals_read: range beyond the buffer", 1;
  if (count == 0) return 0;
  if (!out) return g_error = "null output", 1;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipDeviceSynchronize());
  R3D_HIP_OK(hipMemcpy(out, reinterpret_cast<const r3d_final*>(e->d_pfinals->p) + first, count * sizeof(r3d_final), hipMemcpyDeviceToHost));
  for (uint64_t i = 0; i < count; i++)
    if (out[i].fate != 255) out[i].amp = std::exp(out[i].amp);   // (the kernel left ln(amplitude): r3d_pool.h finish)
  return 0;
}

uint64_t r3d_event_log_count(r3d_engine* e) {
  const uint64_t fail_value = ~uint64_t(0);
  if (!e || !e->d_evlog_count) return 0;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipDeviceSynchronize());
  unsigned long long n = 0;
  R3D_HIP_OK(hipMemcpy(&n, e->d_evlog_count->p, sizeof n, hipMemcpyDeviceToHost));
  return n;
}

uint64_t r3d_event_log_read(r3d_engine* e, r3d_event* out, uint64_t max, int reset) {
  const uint64_t fail_value = ~uint64_t(0);
  if (!e || !e->d_evlog) return g_error = "no event log attached", fail_value;
  const uint64_t total = r3d_event_log_count(e);
  if (total == fail_value) return fail_value;
  const uint64_t n = std::min<uint64_t>(std::min<uint64_t>(total, e->args.evlog_cap), max);
  if (n && !out) return g_error = "null output", fail_value;
  if (n) R3D_HIP_OK(hipMemcpy(out, e->d_evlog->p, n * sizeof(r3d_event), hipMemcpyDeviceToHost));
  if (reset) R3D_HIP_OK(hipMemset(e->d_evlog_count->p, 0, sizeof(unsigned long long)));
  return n;
}

static int attach_volume(r3d_engine* e, const r3d_volume_desc* v, void* d_counters) {
  const int fail_value = 1;
  if (!e) return g_error = "null engine", 1;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipStreamSynchronize(e->stream));
  e->d_volume.reset();
  e->volume_len = 0;
  e->volume_ext = nullptr;
  KArgs& a = e->args;
  a.vol = nullptr;
  if (!v) return 0;
  if (!(v->frame_dt > 0) || v->n_frames == 0 || v->dims[0] == 0 || v->dims[1] == 0 || v->dims[2] == 0 ||
      !(v->cell_size[0] > 0) || !(v->cell_size[1] > 0) || !(v->cell_size[2] > 0))
    return g_error = "volume grid: dimensions, cell sizes and frame length must be positive", 1;
  const size_t len = (size_t)2 * v->n_frames * v->dims[2] * v->dims[1] * v->dims[0];
  if (d_counters) {
    e->volume_ext = d_counters;
  } else {
    auto buf = std::make_unique<DevBuf>();
    R3D_HIP_OK(buf->alloc_zero(len * sizeof(unsigned int)));
    e->d_volume = std::move(buf);
  }
  for (int k = 0; k < 3; k++) {
    a.vol_origin[k] = v->origin[k], a.vol_inv_cell[k] = 1.0 / v->cell_size[k], a.vol_dim[k] = v->dims[k];
    a.vol_dim_f[k] = (double)v->dims[k];
  }
  a.vol_frames = v->n_frames, a.vol_frames_f = (double)v->n_frames;
  a.vol_inv_dt = 1.0 / v->frame_dt;
  a.vol = reinterpret_cast<unsigned int*>(d_counters ? d_counters : e->d_volume->p);
  e->volume_len = len;
  return 0;
}

int r3d_engine_set_volume(r3d_engine* e, const r3d_volume_desc* v) { return attach_volume(e, v, nullptr); }

int r3d_engine_set_volume_buffer(r3d_engine* e, const r3d_volume_desc* v, uint32_t* d_counters) {
  if (v && !d_counters) return g_error = "null volume buffer", 1;
  return attach_volume(e, v, d_counters);
}

size_t r3d_volume_len(const r3d_engine* e) { return e ? e->volume_len : 0; }

void* r3d_volume_device_ptr(r3d_engine* e) {
  if (!e) return nullptr;
  return e->volume_ext ? e->volume_ext : (e->d_volume ? e->d_volume->p : nullptr);
}

int r3d_volume_read(r3d_engine* e, uint32_t* out, int reset) {
  const int fail_value = 1;
  void* const vol = r3d_volume_device_ptr(e);
  if (!vol) return g_error = "no volume grid attached", 1;
  if (!out) return g_error = "null output", 1;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipDeviceSynchronize());   // (runs may have been enqueued on caller streams)
  R3D_HIP_OK(hipMemcpy(out, vol, e->volume_len * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (reset) R3D_HIP_OK(hipMemset(vol, 0, e->volume_len * sizeof(uint32_t)));
  return 0;
}

int r3d_volume_read_range(r3d_engine* e, uint64_t begin, uint64_t count, uint32_t* out) {
  const int fail_value = 1;
  void* const vol = r3d_volume_device_ptr(e);
  if (!vol) return g_error = "no volume grid attached", 1;
  if (begin > e->volume_len || count > e->volume_len - begin) return g_error = "r3d_volume_read_range: range beyond the grid", 1;
  if (count == 0) return 0;
  if (!out) return g_error = "null output", 1;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipDeviceSynchronize());   // (runs may have been enqueued on caller streams)
  R3D_HIP_OK(hipMemcpy(out, reinterpret_cast<const uint32_t*>(vol) + begin, count * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return 0;
}

int r3d_volume_reduce_by_frame(r3d_engine* const* engines, int n, uint32_t* frames, uint64_t* saturated) {
  const int fail_value = 1;
  if (!engines || n < 1) return g_error = "r3d_volume_reduce_by_frame: at least one engine is needed", 1;
  for (int g = 0; g < n; g++) {
    if (!engines[g] || !r3d_volume_device_ptr(engines[g])) return g_error = "r3d_volume_reduce_by_frame: an engine without a grid", 1;
    const KArgs &a = engines[g]->args, &a0 = engines[0]->args;
    if (engines[g]->volume_len != engines[0]->volume_len || a.vol_frames != a0.vol_frames || a.vol_dim[0] != a0.vol_dim[0] ||
        a.vol_dim[1] != a0.vol_dim[1] || a.vol_dim[2] != a0.vol_dim[2])
      return g_error = "r3d_volume_reduce_by_frame: the engines' grids differ in shape", 1;
    for (int h = 0; h < g; h++)
      if (r3d_volume_device_ptr(engines[h]) == r3d_volume_device_ptr(engines[g]))
        return g_error = "r3d_volume_reduce_by_frame: two engines share one grid", 1;
  }
  const uint64_t len = engines[0]->volume_len;
  if (len >= (uint64_t(1) << 32)) return g_error = "r3d_volume_reduce_by_frame: grids beyond 2^32 cells do not fit a pair's index", 1;
  const uint64_t n_frames = engines[0]->args.vol_frames, frame_cells = len / (2 * n_frames);
  // frames [lo, hi) of owner g: contiguous and balanced, as radiative3d_amd/parallel.py shard_range cuts them
  auto frame_lo = [&](int g) { return n_frames / n * g + std::min<uint64_t>(g, n_frames % n); };
  if (saturated) *saturated = 0;
  const uint64_t cap = std::max<uint64_t>(1024, len / 16);
  for (int src = 0; src < n; src++) {
    r3d_engine* S = engines[src];
    // every launch into this grid must have finished (they may sit on caller streams)
    {
      R3D_ON_DEVICE(S->device);
      R3D_HIP_OK(hipDeviceSynchronize());
    }
  }
  // Phase 1, nothing modified: every source COUNTS the non-zero cells of the other owners' frame ranges of its grid (the
  // compaction kernel with no room to write: it counts what it would have written).  A source whose pairs would not fit
  // the limit is found HERE, before any owner's grid has been added to: the call then fails with every grid as it was.
  auto compact_others = [&](int src, uint32_t* pairs, uint64_t room, uint64_t* d_count, std::vector<uint64_t>* ends) -> int {
    r3d_engine* S = engines[src];
    const uint32_t* grid = reinterpret_cast<const uint32_t*>(r3d_volume_device_ptr(S));
    for (int owner = 0; owner < n; owner++) {
      if (owner != src)
        for (uint64_t t = 0; t < 2; t++) {
          const uint64_t b = (t * n_frames + frame_lo(owner)) * frame_cells, e = (t * n_frames + frame_lo(owner + 1)) * frame_cells;
          if (e > b && r3d_volume_compact(S->device, grid, b, e, pairs, room, d_count, S->stream)) return 1;
        }
      if (ends) {   // (pairs written -- or counted -- once this owner's frames are done)
        R3D_HIP_OK(hipStreamSynchronize(S->stream));
        R3D_HIP_OK(hipMemcpy(&(*ends)[owner], d_count, sizeof(uint64_t), hipMemcpyDeviceToHost));
      }
    }
    return 0;
  };
  std::vector<std::vector<uint64_t>> ends(n > 1 ? n : 0);
  for (int src = 0; src < n && n > 1; src++) {
    r3d_engine* S = engines[src];
    R3D_ON_DEVICE(S->device);
    DevBuf count, nowhere;
    R3D_HIP_OK(count.alloc_zero(sizeof(uint64_t)));
    R3D_HIP_OK(nowhere.alloc_zero(2 * sizeof(uint32_t)));
    ends[src].assign(n, 0);
    if (compact_others(src, reinterpret_cast<uint32_t*>(nowhere.p), 0, reinterpret_cast<uint64_t*>(count.p), &ends[src])) return 1;
    if (ends[src][n - 1] > cap)
      return g_error = "r3d_volume_reduce_by_frame: the grid of engine " + std::to_string(src) + " is too full for the pair buffer (" +
                       std::to_string(ends[src][n - 1]) + " non-zero cells in the other owners' frames, room for " + std::to_string(cap) +
                       ": a sixteenth of the grid); no grid has been modified", 1;
  }
  // Phase 2, source by source: its pairs written into a buffer of exactly their number (the other owners' frames of a
  // source's grid are touched by nobody: the count of phase 1 stands), sent to their owners, added there, and the buffer
  // freed before the next source's is made -- one pair buffer alive at a time, however many shards share a device.
  // (A failure from here on is a failed HIP call -- a lost device, no memory for the pairs --, and leaves the owners'
  // frames partly summed: the grids are then undefined.)
  for (int src = 0; src < n && n > 1; src++) {
    r3d_engine* S = engines[src];
    const std::vector<uint64_t>& E = ends[src];
    if (E[n - 1] == 0) continue;
    DevBuf pairs, count;   // (on the source's device; freed where this iteration ends)
    {
      R3D_ON_DEVICE(S->device);
      R3D_HIP_OK(pairs.alloc_zero(E[n - 1] * 2 * sizeof(uint32_t)));
      R3D_HIP_OK(count.alloc_zero(sizeof(uint64_t)));
      if (compact_others(src, reinterpret_cast<uint32_t*>(pairs.p), E[n - 1], reinterpret_cast<uint64_t*>(count.p), nullptr)) return 1;
      R3D_HIP_OK(hipStreamSynchronize(S->stream));
      uint64_t written = 0;
      R3D_HIP_OK(hipMemcpy(&written, count.p, sizeof written, hipMemcpyDeviceToHost));
      if (written != E[n - 1]) return g_error = "r3d_volume_reduce_by_frame: internal error: a grid changed between the count and the compaction", 1;
    }
    for (int owner = 0; owner < n; owner++) {
      const uint64_t lo = owner ? E[owner - 1] : 0, cnt = E[owner] - lo;
      if (owner == src || cnt == 0) continue;
      r3d_engine* D = engines[owner];
      DeviceGuard on_dst(D->device);
      R3D_HIP_OK(on_dst.status);
      DevBuf got, flags;
      R3D_HIP_OK(got.alloc_zero(cnt * 2 * sizeof(uint32_t)));
      R3D_HIP_OK(flags.alloc_zero(2 * sizeof(uint64_t)));
      // (the pairs to the owner's device: a copy between peers, or within the device when the engines share one)
      R3D_HIP_OK(hipMemcpyPeer(got.p, D->device, reinterpret_cast<const uint32_t*>(pairs.p) + 2 * lo, S->device, cnt * 2 * sizeof(uint32_t)));
      if (r3d_volume_scatter_add(D->device, reinterpret_cast<uint32_t*>(r3d_volume_device_ptr(D)), len,
                                 reinterpret_cast<const uint32_t*>(got.p), cnt, reinterpret_cast<uint64_t*>(flags.p), D->stream))
        return 1;
      R3D_HIP_OK(hipStreamSynchronize(D->stream));
      uint64_t f[2];
      R3D_HIP_OK(hipMemcpy(f, flags.p, sizeof f, hipMemcpyDeviceToHost));
      if (f[1]) return g_error = "r3d_volume_reduce_by_frame: internal error: pairs outside the grid", 1;
      if (saturated) *saturated += f[0];
    }
    DeviceGuard on_src(S->device);   // (the pair buffer is freed on its device)
  }
  if (frames)
    for (int g = 0; g <= n; g++) frames[g] = (uint32_t)frame_lo(g);
  return 0;
}

uint64_t r3d_launch_count(const r3d_engine* e) { return e ? e->launches : 0; }

int r3d_engine_variant(const r3d_engine* e) { return e ? e->kind * 4 + e->res : -1; }
uint32_t r3d_engine_pool_slots(const r3d_engine* e) { return e ? e->args.pool_slots : 0; }
uint32_t r3d_engine_accumulators(const r3d_engine* e) { return e && e->args.acc_bits ? 1u << e->args.acc_bits : 0u; }

double r3d_kernel_ms(r3d_engine* e, uint64_t launch) {
  if (!e || launch == 0 || launch > e->launches || e->launches - launch >= r3d_engine::kCounters) return -1.0;
  DeviceGuard guard(e->device);
  const unsigned slot = (unsigned)((launch - 1) % r3d_engine::kCounters);
  if (hipEventSynchronize(e->ev1[slot]) != hipSuccess) return -1.0;
  float ms = 0;
  if (hipEventElapsedTime(&ms, e->ev0[slot], e->ev1[slot]) != hipSuccess) return -1.0;
  return (double)ms;
}

double r3d_last_kernel_ms(r3d_engine* e) { return e ? r3d_kernel_ms(e, e->launches) : -1.0; }

}  // extern "C"
