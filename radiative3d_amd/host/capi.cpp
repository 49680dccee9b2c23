// capi.cpp -- C-ABI of the host model builder (include/r3d_host.h).
// Exceptions never cross the boundary: they become a NULL / non-zero return
// plus a thread-local message.
#include <cstring>
#include <sstream>

#include "../../include/r3d_host.h"
#include <fstream>

#include "cmdline.hpp"
#include "dataout.hpp"

namespace {
thread_local std::string g_error;
}

struct r3dh_model {
  ModelParams params;
  MissionParams mission;
  std::ostringstream log;
  std::unique_ptr<Model> model;
  std::string grid_dump, text;
};

extern "C" {

r3dh_model* r3dh_model_from_args(int argc, const char* const* argv) {
  try {
    auto h = std::make_unique<r3dh_model>();
    std::vector<std::string> tokens(argv, argv + argc);
    ParseCommandLine(tokens, h->params, h->mission);
    h->model = std::make_unique<Model>(h->params, &h->log);
    return h.release();
  } catch (const std::exception& e) {
    g_error = e.what();
  } catch (...) {
    g_error = "unknown error";
  }
  return nullptr;
}

void r3dh_model_free(r3dh_model* m) { delete m; }

const r3d_model_desc* r3dh_model_desc(const r3dh_model* m) { return m ? &m->model->Desc() : nullptr; }

uint64_t r3dh_num_phonons(const r3dh_model* m) { return m ? (uint64_t)m->params.NumPhonons : 0; }
uint64_t r3dh_seed(const r3dh_model* m) { return m ? (uint64_t)m->mission.Seed : 0; }

const char* r3dh_model_log(const r3dh_model* m) {
  static thread_local std::string s;
  s = m ? m->log.str() : "";
  return s.c_str();
}

const char* r3dh_grid_dump(r3dh_model* m) {
  if (!m) return "";
  try {
    // The dump goes through the global coordinate system, which still holds
    // this model's mapping only if no other model was built since.
    std::ostringstream os;
    m->model->GetGridRef().DumpGridToAscii(os);
    m->grid_dump = os.str();
  } catch (const std::exception& e) {
    g_error = e.what();
    m->grid_dump.clear();
  }
  return m->grid_dump.c_str();
}

int r3dh_scatterer_info(const r3dh_model* m, int i, double out[10]) {
  if (!m || i < 0 || i >= (int)m->model->Scatterers().size()) return 1;
  const ScattererInfo& s = m->model->Scatterers()[i];
  const double v[10] = {s.nu, s.eps, s.a, s.kappa, s.el, s.gam0, s.mfp[0], s.mfp[1], s.dipole[0], s.dipole[1]};
  for (int k = 0; k < 10; k++) out[k] = v[k];
  return 0;
}

const char* r3dh_scatterer_dump(r3dh_model* m) {
  if (!m) return "";
  std::ostringstream os;
  PrintAllScatteringStats(*m->model, os);
  m->text = os.str();
  return m->text.c_str();
}

const char* r3dh_params_echo(r3dh_model* m) {
  if (!m) return "";
  std::ostringstream os;
  OutputModelParams(m->params, os);
  m->text = os.str();
  return m->text.c_str();
}

const char* r3dh_write_outputs(r3dh_model* m, const r3d_result* result, const char* outdir,
                               const char* trace_path, const char* mparams_path) {
  if (!m || !result || !trace_path) return nullptr;
  try {
    std::ostringstream console;
    std::ofstream trace(trace_path);
    if (!trace) throw Runtime(std::string("cannot open ") + trace_path);
    OutputPostSimSummary(*m->model, *result, outdir ? outdir : "", console, trace);
    if (mparams_path) {
      std::ofstream f(mparams_path);
      OutputModelParamsOctave(m->params, f);
    }
    m->text = console.str();
    return m->text.c_str();
  } catch (const std::exception& e) {
    g_error = e.what();
  }
  return nullptr;
}

int r3dh_model_set_scatterer_stats(r3dh_model* m, int s, const double mfp[2], const double dipole[2]) {
  if (!m || !mfp || !dipole || s < 0 || s >= (int)m->model->Scatterers().size()) return 1;
  m->model->SetScattererStats(s, mfp, dipole);
  return 0;
}
int r3dh_model_device_tables(const r3dh_model* m) { return m && m->model->DeviceTables() ? 1 : 0; }

uint32_t r3dh_report_mask(const char* keywords) {
  try {
    return ReportMaskFromKeywords(keywords ? keywords : "");
  } catch (const std::exception& e) {
    g_error = e.what();
  }
  return ~uint32_t(0);
}

uint32_t r3dh_model_report_mask(const r3dh_model* m) {
  return m ? r3dh_report_mask(m->mission.Reports.c_str()) : 0;
}

const char* r3dh_write_reports(r3dh_model* m, const r3d_event* events, uint64_t n, const char* path) {
  if (!m || (n && !events)) return nullptr;
  try {
    // (the lines go through the global coordinate system, which still holds this model's
    //  mapping only if no other model was built since)
    if (path) {
      std::ofstream f(path);
      if (!f) throw Runtime(std::string("cannot open ") + path);
      OutputReports(events, (size_t)n, f);
      m->text.clear();
    } else {
      std::ostringstream os;
      OutputReports(events, (size_t)n, os);
      m->text = os.str();
    }
    return m->text.c_str();
  } catch (const std::exception& e) {
    g_error = e.what();
  }
  return nullptr;
}

int r3dh_model_coordinates(const r3dh_model* m, int* map_code, double* earth_radius, int* flattened) {
  if (!m) return 1;
  if (map_code) *map_code = m->model->MapCode();
  if (earth_radius) *earth_radius = m->model->EarthRadius();
  if (flattened) *flattened = m->params.Flatten ? 1 : 0;
  return 0;
}

int r3dh_grid_size(const r3dh_model* m, int dims[3]) {
  if (!m || !dims) return 1;
  const Grid& g = m->model->GetGridRef();
  dims[0] = (int)g.Ni(), dims[1] = (int)g.Nj(), dims[2] = (int)g.Nk();
  return 0;
}

int r3dh_grid_nodes(const r3dh_model* m, r3dh_grid_node* out, size_t capacity) {
  if (!m || !out) return 1;
  try {
    // (through the global coordinate system, which still holds this model's mapping only if no
    //  other model was built since -- as for r3dh_grid_dump)
    const Grid& g = m->model->GetGridRef();
    if ((size_t)g.N() > capacity) throw Runtime("r3dh_grid_nodes: output too small");
    size_t at = 0;
    for (Index k = 0; k < g.Nk(); k++)
      for (Index j = 0; j < g.Nj(); j++)
        for (Index i = 0; i < g.Ni(); i++) {
          const GridNode& n = g.Node(i, j, k);
          r3dh_grid_node& o = out[at++];
          std::memset(&o, 0, sizeof o);
          const R3::XYZ loc = n.Loc();
          o.loc[0] = loc.x(), o.loc[1] = loc.y(), o.loc[2] = loc.z();
          o.radius = ECS.CurvedCoords() ? n.GetRawLoc().Radius(ECS) : 0.0;
          o.n_sets = n.NumAttributeSets();
          if (o.n_sets == 0) continue;
          for (int s = 0; s < 2; s++) {
            const GridData d = n.Data(s == 0 ? GridNode::GN_ABOVE : GridNode::GN_BELOW);
            const double v[9] = {d.Vp(), d.Vs(), d.Rho(), d.Qp(), d.Qs(), d.getHS().nu(), d.getHS().eps(),
                                 d.getHS().a(), d.getHS().kappa()};
            for (int q = 0; q < 9; q++) o.side[s][q] = v[q];
          }
        }
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
  }
  return 1;
}

int r3dh_grid_nodes_raw(const r3dh_model* m, r3dh_grid_node_raw* out, size_t capacity) {
  if (!m || !out) return 1;
  try {
    const Grid& g = m->model->GetGridRef();
    if ((size_t)g.N() > capacity) throw Runtime("r3dh_grid_nodes_raw: output too small");
    size_t at = 0;
    for (Index k = 0; k < g.Nk(); k++)
      for (Index j = 0; j < g.Nj(); j++)
        for (Index i = 0; i < g.Ni(); i++) {
          const GridNode& n = g.Node(i, j, k);
          r3dh_grid_node_raw& o = out[at++];
          std::memset(&o, 0, sizeof o);
          const EarthCoords::Generic raw = n.GetRawLoc();
          o.x[0] = raw.x1(), o.x[1] = raw.x2(), o.x[2] = raw.x3();
          o.n_sets = n.NumAttributeSets();
          for (int s = 0; s < o.n_sets && s < 2; s++) {
            const GridData d = n.RawData(s);
            int missing;
            Real qp, qs, qk;
            d.getQ().Stored(missing, qp, qs, qk);
            const double v[11] = {d.Vp(), d.Vs(), d.Rho(), (double)missing, qp, qs, qk, d.getHS().nu(),
                                  d.getHS().eps(), d.getHS().a(), d.getHS().kappa()};
            for (int q = 0; q < 11; q++) o.set[s][q] = v[q];
          }
        }
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
  }
  return 1;
}

int r3dh_seismometer_axes(const r3dh_model* m, int i) {
  if (!m || i < 0 || i >= (int)m->model->SeisAxesDesc().size()) return -1;
  return m->model->SeisAxesDesc()[i] == "RTZ" ? 1 : 0;
}

const char* r3dh_last_error(void) { return g_error.c_str(); }

}  // extern "C"
