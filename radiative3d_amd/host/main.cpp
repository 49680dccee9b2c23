// main.cpp -- command-line shell around the HIP engine, flag-compatible with
// the reference's `main` (reference main.cpp:27-136) so that the do-*.sh
// drivers run unchanged: same option tokens, same stdout markers
// ("@@ __PHASE__", "#  R3D_GRID:", "#  BEGIN SCATTERER DUMP:"), same output
// files.  The N-phonon loop itself (Model::RunSimulation, model.cpp:602-633)
// is one call into the engine's C-ABI.
//
// Differences a user can see: `--seed=S`, `--gpus=N` / `--devices=a,b,...` and `--scatter-grid=...` are accepted (the
// reference seeds from the clock, is single-process and has no event histogram); the `--reports` stream is written
// after the run, grouped by history, where the reference writes its lines as they happen (the engine appends binary
// records in HBM: include/r3d.h r3d_event); tables are built in HBM unless `--host-tables` is given.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <thread>
#include <unistd.h>

#include "../../include/r3d.h"
#include "../csrc/r3d_physics.h"   // rt_weights(): the --rtcoef-test mission
#include "cmdline.hpp"
#include "dataout.hpp"

namespace {

void print_banner() {
  std::cout << "**\n"
            << "**  Radiative3D / MI355X engine - radiative transport in 3D Earth models\n"
            << "**\n"
            << "**  Propagation on AMD Instinct MI355X (HIP, fp64); model definition,\n"
            << "**  options and output formats follow Radiative3D\n"
            << "**  (https://github.com/christophersanborn/Radiative3D).\n"
            << "**\n"
            << "**  BUILD STATS:  " << r3d_version() << "\n"
            << "**                Floating-point representation: " << (8 * sizeof(Real)) << "-bit\n"
            << "**\n**\n";
}

// RTCoef::RunRTCoefTest(100, 10,8,4, 8,4,2) + PrintChosenRaytype (rtcoef.cpp:604-742),
// without the random "Result" column.
void run_rtcoef_test(unsigned n_sini, double rho1, double a1, double b1, double rho2, double a2,
                     double b2) {
  using namespace r3d;
  const int width = 14;
  const char* names[3] = {"RAY_P", "RAY_SH", "RAY_SV"};   // raytype order P, SH, SV (raytype.hpp)
  for (int irt = 0; irt < 3; irt++) {
    std::cout << "\n##RTCoef Probability Test:\n##\n##       Input Raytype:  " << names[irt] << "\n##\n"
              << "##     Reflection Side:  (rho,alpha,beta) = ( " << rho1 << " " << a1 << " " << b1
              << " )\n"
              << "##   Transmission Side:  (rho,alpha,beta) = ( " << rho2 << " " << a2 << " " << b2
              << " )\n##\n";
    for (unsigned i = 0; i < n_sini; i++) {
      const double theta = kPi90 * ((double)i / (n_sini - 1));
      const V3 fnorm = v3(0, 0, 1), dir = v3(sin(theta), 0.0, cos(theta));
      const double sini = dot(in_plane_unit_perp(fnorm, dir), dir);
      Iface f;
      f.normal = fnorm, f.has_neighbor = true;
      f.rhoR = rho1, f.vR[0] = a1, f.vR[1] = b1, f.rhoT = rho2, f.vT[0] = a2, f.vT[1] = b2;
      double w[RT_NUM], det2;
      rt_weights(f, sini, irt, w, det2);
      if (i == 0)
        std::cout << "##" << std::setw(width) << "Sine_in" << std::setw(width) << "Prob_R_P"
                  << std::setw(width) << "Prob_T_P" << std::setw(width) << "Prob_R_SV"
                  << std::setw(width) << "Prob_T_SV" << std::setw(width) << "Prob_R_SH"
                  << std::setw(width) << "Prob_T_SH" << "\n\n";
      std::cout << "  " << std::setw(width) << sini;
      for (int k : {R_P, T_P, R_SV, T_SV, R_SH, T_SH}) std::cout << std::setw(width) << w[k] / det2;
      std::cout << "\n";
    }
  }
}

// --event-test (main.cpp:86-104, sources.cpp:71-87): radiation patterns on a
// degree-3 take-off set, as GMT symbol rows.
void run_event_test(const ModelParams& par) {
  std::cout << "@@ __EVENT_SOURCE_TEST__" << std::endl;
  ModelParams p = par;
  p.TOA_Degree = 3;
  p.GridSource = ModelParams::GRID_COMPILED;
  if (p.CompiledSelector == 0) p.CompiledSelector = 40, p.CompiledArgs.clear();
  std::ostringstream sink;
  p.DeviceTables = false;   // (this mission prints the host builder's source tables)
  Model m(p, &sink);
  const r3d_source& s = m.Desc().source;
  const size_t n = m.Desc().n_toa;
  const char* sym[3] = {"c", "-", "y"};
  const double total = s.whole_cdf[2];
  for (int t = 0; t < 3; t++) {
    const double share = (s.whole_cdf[t] - (t ? s.whole_cdf[t - 1] : 0.0)) / total;
    const double mag = s.cdf[t][n - 1];
    for (size_t k = 0; k < n; k++) {
      const double diff = mag == 0 ? 0 : (s.cdf[t][k] - (k ? s.cdf[t][k - 1] : 0.0)) / mag;
      const S2::ThetaPhi& a = m.TOA()[k];
      std::cout << std::setw(16) << Geometry::RtoD * a.Phi() << std::setw(16)
                << 90.0 - Geometry::RtoD * a.Theta() << std::setw(16)
                << std::sqrt(share * diff * n) * 0.2 << "  " << sym[t] << std::endl;
    }
  }
}

// What --scatter-grid asks for, and where it goes.
struct GridJob {
  bool on = false;
  r3d_volume_desc desc{};
  std::string header_path, raw_path, raw_name;
};

// A scatter grid is checked BEFORE the run (a 1e8-history job must not find out at its end that its grid
// cannot be reduced or written): the pair exchange of r3d_volume_reduce_by_frame carries 32-bit cell indices,
// and the output directory must take the files.
void check_grid_job(const GridJob& grid, size_t n_shards) {
  if (!grid.on) return;
  const unsigned long long cells = 2ull * grid.desc.dims[0] * grid.desc.dims[1] * grid.desc.dims[2] * grid.desc.n_frames;
  if (n_shards > 1 && cells >= (1ull << 32))
    throw Runtime("--scatter-grid: 2 x NX x NY x NZ x FRAMES = " + std::to_string(cells) + " cells do not fit the 32-bit cell "
                  "indices the shards' grids are added with (r3d_volume_reduce_by_frame); use one device or a coarser grid.");
  const std::string probe = grid.raw_path + ".part";
  std::ofstream f(probe.c_str(), std::ios::binary);
  if (!f) throw Runtime("--scatter-grid: cannot write " + probe);
  f.close();
  std::remove(probe.c_str());
}

// File descriptor 1 pointed at stderr for the lifetime of the object (what C libraries underneath write to stdout).
struct StdoutToStderr {
  int saved = -1;
  StdoutToStderr() {
    std::cout.flush();
    std::fflush(stdout);
    saved = dup(1);
    if (saved >= 0) dup2(2, 1);
  }
  ~StdoutToStderr() {
    std::fflush(stdout);
    if (saved >= 0) dup2(saved, 1), close(saved);
  }
};

// The replacement for Model::RunSimulation()'s loop: a node (include/r3d.h r3d_node_*) shards the id range over
// the requested devices, one engine per entry, and sums the shards' blocks on the devices (RCCL; on the host when
// two shards share a device).  With a scatter grid every shard's engine fills its own grid in HBM; the grids are
// then added by frame (r3d_volume_reduce_by_frame: every engine ends with the job's counts for its share of the
// frames) and each engine's frames written to the raw file.
void run_simulation(const Model& model, uint64_t n, uint64_t seed, r3d_node* node, r3d_result& total,
                    std::vector<double>& energy, std::vector<uint64_t>& counts, uint32_t report_mask,
                    std::vector<r3d_event>& events, uint64_t& events_dropped, const GridJob& grid) {
  const r3d_model_desc& d = model.Desc();
  const int gpus = r3d_node_size(node);
  const size_t ne = (size_t)d.n_seismometers * d.params.n_bins * R3D_N_ENERGY;
  const size_t nc = (size_t)d.n_seismometers * d.params.n_bins * R3D_N_COUNT;
  energy.assign(ne, 0.0), counts.assign(nc, 0);
  total = r3d_result{};
  total.energy = energy.data(), total.counts = counts.data();
  std::vector<r3d_engine*> engines;
  std::vector<uint64_t> caps(gpus, 0);
  for (int g = 0; g < gpus; g++) {
    r3d_engine* e = r3d_node_engine(node, g);
    engines.push_back(e);
    // report stream: room for 256 events per history, at most 2^26 records (6.4 GB) per GPU
    const uint64_t cnt = n / gpus + ((uint64_t)g < n % gpus ? 1 : 0);
    caps[g] = std::min<uint64_t>(std::max<uint64_t>(cnt, 1) * 256, uint64_t(1) << 26);
    if (report_mask && r3d_engine_set_event_log(e, report_mask, caps[g])) throw Runtime(r3d_last_error());
    if (grid.on && r3d_engine_set_volume(e, &grid.desc)) throw Runtime(r3d_last_error());
  }
  if (r3d_node_run(node, n, 0, seed, &total)) throw Runtime(r3d_last_error());
  std::cout << "|  Shards: " << gpus << " (summed by " << r3d_node_reduction(node)
            << (*r3d_node_reduction_note(node) ? std::string(": ") + r3d_node_reduction_note(node) : std::string()) << ")\n";
  if (report_mask)
    for (int g = 0; g < gpus; g++) {   // in shard order: ids ascend across shards
      const uint64_t reported = r3d_event_log_count(engines[g]);
      const size_t at = events.size(), got = (size_t)std::min<uint64_t>(reported, caps[g]);
      events.resize(at + got);
      if (got && r3d_event_log_read(engines[g], events.data() + at, got, 0) == ~uint64_t(0)) throw Runtime(r3d_last_error());
      events_dropped += reported - got;
      if (r3d_engine_set_event_log(engines[g], 0, 0)) throw Runtime(r3d_last_error());   // (its HBM back before the grids are added)
    }
  if (grid.on) {
    std::vector<uint32_t> frames(gpus + 1);
    uint64_t saturated = 0;
    if (r3d_volume_reduce_by_frame(engines.data(), gpus, frames.data(), &saturated)) throw Runtime(r3d_last_error());
    const uint64_t fc = (uint64_t)grid.desc.dims[0] * grid.desc.dims[1] * grid.desc.dims[2], nf = grid.desc.n_frames;
    unsigned long long binned = 0;
    // (written under a temporary name and renamed when complete: a failed run leaves no half-written grid behind)
    const std::string part = grid.raw_path + ".part";
    std::string err;
    {
      std::ofstream raw(part.c_str(), std::ios::binary);
      std::vector<uint32_t> buf;
      // the file is count[type][frame][z][y][x]: for each wave type the owners' frame ranges in turn
      for (uint64_t t = 0; t < 2 && err.empty(); t++)
        for (int g = 0; g < gpus && err.empty(); g++) {
          const uint64_t cnt = (uint64_t)(frames[g + 1] - frames[g]) * fc;
          buf.resize(cnt);
          if (cnt && r3d_volume_read_range(engines[g], (t * nf + frames[g]) * fc, cnt, buf.data())) err = r3d_last_error();
          for (uint32_t v : buf) binned += v;
          raw.write(reinterpret_cast<const char*>(buf.data()), (std::streamsize)(cnt * sizeof(uint32_t)));
        }
      if (err.empty() && !raw) err = "cannot write " + part;
    }
    if (err.empty() && std::rename(part.c_str(), grid.raw_path.c_str())) err = "cannot rename " + part + " to " + grid.raw_path;
    if (!err.empty()) {
      std::remove(part.c_str());
      throw Runtime(err);
    }
    std::ofstream hdr(grid.header_path.c_str());
    const double lo[3] = {grid.desc.origin[0], grid.desc.origin[1], grid.desc.origin[2]};
    const double hi[3] = {lo[0] + grid.desc.cell_size[0] * grid.desc.dims[0], lo[1] + grid.desc.cell_size[1] * grid.desc.dims[1],
                          lo[2] + grid.desc.cell_size[2] * grid.desc.dims[2]};
    OutputScatterGridHeader(grid.desc.dims, grid.desc.n_frames, lo, hi, grid.desc.frame_dt, grid.raw_name, binned, saturated, hdr);
    std::cout << "|  Scatter-event grid: " << binned << " events binned into " << grid.desc.dims[0] << " x " << grid.desc.dims[1]
              << " x " << grid.desc.dims[2] << " cells x " << nf << " frames x 2 wave types -> " << grid.raw_path << "\n";
  }
}

struct NodeHolder {   // (the node goes with the scope, whichever way it is left)
  r3d_node* node = nullptr;
  ~NodeHolder() {
    if (node) r3d_node_destroy(node);
  }
};

}  // namespace

int main(int argc, char* argv[]) {
  print_banner();
  MissionParams mission;
  ModelParams par;
  try {
    ParseCommandLine(std::vector<std::string>(argv + 1, argv + argc), par, mission);
  } catch (std::exception& e) {
    std::cout << "** Error processing command-line options\n** Message: " << e.what()
              << "\n** Exiting...\n";
    return 1;
  }
  if (mission.bHelpMsg) {
    std::cout << "\nOptions follow the Radiative3D manual (doc/MANUAL.md of the reference);\n"
              << "additional: --seed=<n>  --gpus=<n>  --devices=<a,b,...>  --host-tables (a simulation run builds the\n"
              << "take-off set, source and scattering tables in HBM unless told otherwise)  --device-tables\n"
              << "--scatter-grid=NX,NY,NZ,FRAMES,X0,Y0,Z0,X1,Y1,Z1 [--scatter-grid-file=<name>]: SCT / REF events per wave\n"
              << "type, frame and model-space cell, written as <name>.octv + <name>.u32 under --output-dir\n\n";
    return 0;
  }
  // A simulation run makes its tables where it uses them (seconds of host work and GBs of upload
  // at TOA degree 9 become milliseconds); the diagnostic missions, which print host tables and
  // must work without a GPU, keep the host builder.
  if (mission.bRunSim && !par.HostTables) par.DeviceTables = true;
  OutputModelParams(par, std::cout);
  if (mission.bOutputModParamsOctv) {
    std::string fn = mission.OutputDir.empty() ? mission.FNModParamsOctv
                                               : mission.OutputDir + "/" + mission.FNModParamsOctv;
    std::ofstream f(fn.c_str());
    OutputModelParamsOctave(par, f);
  }
  uint32_t report_mask = 0;
  try {
    report_mask = ReportMaskFromKeywords(mission.Reports);
  } catch (std::exception& e) {
    std::cout << "** Error processing command-line options\n** Message: " << e.what()
              << "\n** Exiting...\n";
    return 1;
  }
  if (mission.bRTCoefTest) run_rtcoef_test(100, 10, 8, 4, 8, 4, 2);
  const char* phase = "while constructing Earth model:";
  try {
    if (mission.bSourcePatternTest) run_event_test(par);
    if (mission.bRunSim || mission.bDumpGrid) {
      Model model(par);
      phase = "during model retrospective output:";
      if (mission.bDumpGrid) model.GetGridRef().DumpGridToAscii();
      // the devices of the run, and -- for a simulation run or tables made in HBM -- the node: one engine per
      // shard on the devices NAMED (nothing is built on device 0 unless it is one of them)
      std::vector<int> devices = mission.Devices;
      if (devices.empty())
        for (int g = 0; g < std::max(1, mission.Gpus); g++) devices.push_back(g);
      GridJob grid;
      if (mission.bRunSim && mission.bScatterGrid) {
        grid.on = true;
        for (int k = 0; k < 3; k++) {
          grid.desc.origin[k] = mission.GridLo[k], grid.desc.dims[k] = mission.GridDims[k];
          grid.desc.cell_size[k] = (mission.GridHi[k] - mission.GridLo[k]) / mission.GridDims[k];
        }
        grid.desc.n_frames = mission.GridFrames;
        grid.desc.frame_dt = par.PhononTTL / mission.GridFrames;
        const std::string dir = mission.OutputDir.empty() ? "" : mission.OutputDir + "/";
        grid.raw_name = mission.ScatterGridFile + ".u32";
        grid.raw_path = dir + grid.raw_name, grid.header_path = dir + mission.ScatterGridFile + ".octv";
        check_grid_job(grid, devices.size());
      }
      NodeHolder held;
      if (mission.bRunSim || model.DeviceTables()) {
        const std::vector<int> on = mission.bRunSim ? devices : std::vector<int>(1, devices[0]);
        {
          // (RCCL prints its version banner on stdout when a communicator is formed: this program's stdout is the
          //  reference's output format, so whatever libraries say while the node is built goes to stderr)
          StdoutToStderr quiet;
          held.node = r3d_node_create(&model.Desc(), on.data(), (int)on.size());
        }
        if (!held.node) throw Runtime(r3d_last_error());
      }
      if (model.DeviceTables()) {   // the tables (and so the MFPs the dump prints) are made in HBM
        r3d_engine* e0 = r3d_node_engine(held.node, 0);
        for (int s = 0; s < model.Desc().n_scatterers; s++) {
          double st[8];
          if (r3d_engine_scatterer_stats(e0, s, st)) throw Runtime(r3d_last_error());
          model.SetScattererStats(s, st, st + 2);
        }
      }
      PrintAllScatteringStats(model, std::cout);
      phase = "during simulation execution:";
      if (mission.bRunSim) {
        std::cout << "@@ __BEGINNING_SIMULATION__" << std::endl;
        r3d_result res;
        std::vector<double> energy;
        std::vector<uint64_t> counts;
        std::vector<r3d_event> events;
        uint64_t dropped = 0;
        run_simulation(model, (uint64_t)std::max(0L, par.NumPhonons), mission.Seed, held.node, res, energy, counts,
                       report_mask, events, dropped, grid);
        if (report_mask) {   // the reference writes them as they happen: stdout, or --report-file
          if (mission.ReportFile.empty()) {
            OutputReports(events.data(), events.size(), std::cout);
          } else {
            const std::string fn = mission.OutputDir.empty() ? mission.ReportFile
                                                             : mission.OutputDir + "/" + mission.ReportFile;
            std::ofstream f(fn.c_str());
            OutputReports(events.data(), events.size(), f);
          }
          if (dropped) std::cerr << "Note: " << dropped << " report lines did not fit the event buffer.\n";
        }
        std::cerr << "100% of " << par.NumPhonons << " have been cast.\n";
        std::cout << "@@ __SIMULATION_COMPLETE__" << std::endl;
        // seis_traces_asc.dat is opened in the CWD whatever --output-dir says (dataout.hpp:332)
        std::ofstream trace("seis_traces_asc.dat");
        OutputPostSimSummary(model, res, mission.OutputDir, std::cout, trace);
      }
    }
  } catch (std::exception& e) {
    std::cout << "**\n** Error " << phase << "\n** What: " << e.what() << "\n** Exiting...\n";
    return 1;
  }
  return 0;
}
