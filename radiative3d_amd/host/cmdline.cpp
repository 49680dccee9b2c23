// cmdline.cpp -- see cmdline.hpp.
#include "cmdline.hpp"

#include <cctype>
#include <cstdlib>
#include <deque>
#include <iostream>
#include <map>

namespace {

enum OptID {
  OPT_FREQ, OPT_NUMBER, OPT_TTLIVE, OPT_TOA, OPT_COMPSELECT, OPT_MODARGS, OPT_CYLRANGE,
  OPT_FLATTEN, OPT_EARTHRAD, OPT_EVENT_MT, OPT_EVENT_LOC, OPT_OVR_MFP, OPT_NODEFLECT,
  OPT_REPORTS, OPT_REPORT_FILE, OPT_OUTDIR, OPT_OCSRAW, OPT_SEISBINS, OPT_SEISBINSIZE,
  OPT_SEISARRAY, OPT_SEIS_P2P, OPT_SEIS_P2PW, OPTM_HELP, OPTM_DUMPGRID, OPTM_PARAMOUTFN,
  OPTM_RTTEST, OPTM_EVENTTEST, OPTM_RUNSIM, OPTX_SEED, OPTX_GPUS, OPTX_DEVTABLES, OPTX_HOSTTABLES,
  OPTX_DEVICES, OPTX_SCATGRID, OPTX_SCATGRID_FILE
};

const std::map<std::string, OptID>& option_table() {
  static const std::map<std::string, OptID> t = {
      {"-F", OPT_FREQ}, {"--frequency", OPT_FREQ},
      {"-N", OPT_NUMBER}, {"--num-phonons", OPT_NUMBER},
      {"-T", OPT_TTLIVE}, {"--timetolive", OPT_TTLIVE},
      {"-A", OPT_TOA}, {"--toa-degree", OPT_TOA},
      {"--grid-compiled", OPT_COMPSELECT},
      {"--model-args", OPT_MODARGS}, {"--model-compiled-args", OPT_MODARGS},
      {"--range", OPT_CYLRANGE}, {"--cylinder-range", OPT_CYLRANGE},
      {"--flatten", OPT_FLATTEN},
      {"--earthrad", OPT_EARTHRAD}, {"--earthradius", OPT_EARTHRAD},
      {"-E", OPT_EVENT_MT}, {"--source", OPT_EVENT_MT},
      {"-L", OPT_EVENT_LOC}, {"--source-loc", OPT_EVENT_LOC},
      {"--mfpoverride", OPT_OVR_MFP}, {"--overridemfp", OPT_OVR_MFP},
      {"--nodeflect", OPT_NODEFLECT}, {"--no-deflect", OPT_NODEFLECT},
      {"--reports", OPT_REPORTS}, {"--report-file", OPT_REPORT_FILE},
      {"--output-dir", OPT_OUTDIR},
      {"--ocsnotransform", OPT_OCSRAW}, {"--ocsraw", OPT_OCSRAW},
      {"--binspercycle", OPT_SEISBINS}, {"--bins", OPT_SEISBINS},
      {"--binsize", OPT_SEISBINSIZE},
      {"--seis-array", OPT_SEISARRAY},
      {"--seis-p2p", OPT_SEIS_P2P}, {"--seis-p2pw", OPT_SEIS_P2PW},
      {"--help", OPTM_HELP}, {"--dump-grid", OPTM_DUMPGRID},
      {"--mparams-outfile", OPTM_PARAMOUTFN},
      {"--rtcoef-test", OPTM_RTTEST}, {"--event-test", OPTM_EVENTTEST},
      {"--run-simulation", OPTM_RUNSIM}, {"--run-sim", OPTM_RUNSIM},
      {"--seed", OPTX_SEED}, {"--gpus", OPTX_GPUS}, {"--device-tables", OPTX_DEVTABLES},
      {"--host-tables", OPTX_HOSTTABLES}, {"--devices", OPTX_DEVICES},
      {"--scatter-grid", OPTX_SCATGRID}, {"--scatter-grid-file", OPTX_SCATGRID_FILE}};
  return t;
}

bool looks_like_option(const std::string& s) {
  return s.size() >= 2 && s[0] == '-' && (s[1] == '-' || std::isalpha((unsigned char)s[1]));
}

// One option with its comma-separated value list.
struct Opt {
  std::string token;
  std::deque<std::string> values;

  bool has() const { return !values.empty(); }
  std::string text() {
    if (values.empty()) throw Runtime("Required value not provided for " + token + ".");
    std::string v = values.front();
    values.pop_front();
    return v;
  }
  Real real() {
    std::string v = text();
    char* end = nullptr;
    Real r = std::strtod(v.c_str(), &end);
    if (end == v.c_str() || *end != '\0')
      throw Runtime("Invalid value: cannot interpret '" + v + "' as type Real.");
    return r;
  }
  long integer() {
    std::string v = text();
    if (!v.empty()) {
      const char* zeros = nullptr;
      switch (v.back()) {
        case 'K': zeros = "000"; break;
        case 'M': zeros = "000000"; break;
        case 'B': zeros = "000000000"; break;
      }
      if (zeros) v = v.substr(0, v.size() - 1) + zeros;
    }
    char* end = nullptr;
    long r = std::strtol(v.c_str(), &end, 10);
    if (end == v.c_str() || *end != '\0')
      throw Runtime("Invalid value: cannot interpret '" + v + "' as type Integer.");
    return r;
  }
  R3::XYZ xyz() {
    Real x = real(), y = real(), z = real();
    return {x, y, z};
  }
};

// Linear receiver array between two points given as raw coordinate triples
// (reference main.cpp:340-392): positions AND gather radii are interpolated in
// the user's coordinate tuple space, not along geodesics.
void add_p2p_array(Opt& o, ModelParams& par, bool force_wavelengths) {
  const bool two_gathers = (o.values.size() == 10);
  R3::XYZ origin = o.xyz(), dest = o.xyz();
  Real offset = o.real();
  Real g1 = o.real(), g2 = g1;
  if (two_gathers) g2 = o.real();
  long n = o.integer();
  const R3::XYZ span = origin.VectorTo(dest);
  const R3::XYZ dir = span.Unit();
  const R3::XYZ begin = origin + dir.ScaledBy(offset);
  const Real gap = n > 1 ? begin.VectorTo(dest).Mag() / (n - 1) : 0.0;
  static bool warned = false;
  for (long i = 0; i < n; i++) {
    R3::XYZ p = begin + dir.ScaledBy(i * gap);
    Real frac = origin.VectorTo(p).Mag() / span.Mag();
    Real gather = g1 + frac * (g2 - g1);
    if (!warned || force_wavelengths) {
      std::cout << "Warning: Seis array interpolation ignores coordinate "
                << "system and could produce distorted results.\n";
      warned = true;
    }
    EarthCoords::Generic at(p.x(), p.y(), p.z());
    if (two_gathers && !force_wavelengths)
      par.AddSeismometerFixedRadius(at, ModelParams::AX_RTZ, gather);
    else
      par.AddSeismometerByWavelength(at, ModelParams::AX_RTZ, gather);
  }
}

void set_source(Opt& o, ModelParams& par) {
  std::string kind = o.text();
  if (kind == "EQ") {
    par.EventSourceMT = Tensor::USGS(0, -1, 1, 0, 0, 0);
  } else if (kind == "EXPL") {
    par.EventSourceMT = Tensor::USGS(1, 1, 1, 0, 0, 0);
  } else if (kind == "USGS") {
    if (o.values.size() < 6)
      throw Runtime("USGS keyword expects six numeric moment tensor elements.");
    Real m[6];
    for (Real& v : m) v = o.real();
    par.EventSourceMT = Tensor::USGS(m[0], m[1], m[2], m[3], m[4], m[5]);
  } else if (kind == "SDR") {
    std::vector<Real> a;
    while (o.has()) a.push_back(o.real());
    auto iso_of = [](Real iso) {
      return (iso > 1.0 || iso < -1.0) ? Tensor::SDR::IsoFracFromIsoAngle(iso) : iso;
    };
    if (a.size() == 3) {
      par.EventSourceMT = Tensor::SDR(a[0], a[1], a[2]);
    } else if (a.size() == 4) {
      par.EventSourceMT = Tensor::SDR(a[0], a[1], a[2], iso_of(a[3]));
    } else if (a.size() == 5) {
      Real iso = a[3], moment = a[4];
      if (moment == 0.0) {  // old argument order (moment, iso)
        std::swap(iso, moment);
        std::cout << "Warning: SDR Event Specification: Swapped iso and moment arguments.\n"
                  << "Warning: (Note new argument order: "
                  << "--source=SDR,strike,dip,rake,iso,moment)\n";
      }
      par.EventSourceMT = Tensor::SDR(a[0], a[1], a[2], iso_of(iso), moment);
    } else {
      throw Runtime("SDR keyword expects SDR,strike,dip,rake[,iso[,moment]].");
    }
  } else {
    throw Runtime(kind + " is not a valid source mechanism keyword.");
  }
}

}  // namespace

void ParseCommandLine(const std::vector<std::string>& tokens, ModelParams& par,
                      MissionParams& mission) {
  for (size_t i = 0; i < tokens.size(); i++) {
    const std::string& tk = tokens[i];
    if (!looks_like_option(tk)) continue;  // stray value: a no-op, as in the reference
    Opt o;
    std::string vals;
    size_t eq = tk.find('=');
    if (tk[1] == '-' && eq != std::string::npos) {
      o.token = tk.substr(0, eq);
      vals = tk.substr(eq + 1);
    } else {
      o.token = tk;
      if (i + 1 < tokens.size() && !looks_like_option(tokens[i + 1])) vals = tokens[++i];
    }
    for (size_t p = 0; !vals.empty();) {
      size_t c = vals.find(',', p);
      o.values.push_back(vals.substr(p, c == std::string::npos ? c : c - p));
      if (c == std::string::npos) break;
      p = c + 1;
    }
    auto it = option_table().find(o.token);
    if (it == option_table().end()) throw Runtime("Unrecognized option: " + o.token);

    switch (it->second) {
      case OPT_FREQ: par.Frequency = o.real(); break;
      case OPT_NUMBER: par.NumPhonons = o.integer(); break;
      case OPT_TTLIVE: par.PhononTTL = o.real(); break;
      case OPT_TOA: par.TOA_Degree = (int)o.integer(); break;
      case OPT_COMPSELECT:
        par.GridSource = ModelParams::GRID_COMPILED;
        par.CompiledSelector = o.has() ? (int)o.integer() : 0;
        break;
      case OPT_MODARGS:
        while (o.has()) par.CompiledArgs.push_back(o.real());
        break;
      case OPT_CYLRANGE: par.CylinderRange = o.real(); break;
      case OPT_FLATTEN: par.Flatten = true; break;
      case OPT_EARTHRAD: par.EarthRadius = o.real(); break;
      case OPT_EVENT_MT: set_source(o, par); break;
      case OPT_EVENT_LOC: {
        Real x = o.real(), y = o.real(), z = o.real();
        par.EventSourceLoc = EarthCoords::Generic(x, y, z);
      } break;
      case OPT_OVR_MFP:
        par.OverrideMFP = true;
        par.MFPOverride[0] = o.real();
        par.MFPOverride[1] = o.real();
        break;
      case OPT_NODEFLECT: par.NoDeflect = true; break;
      case OPT_REPORTS:
        mission.Reports.clear();
        if (!o.has()) mission.Reports = "ALL_ON";
        while (o.has()) {
          std::string kw = o.text();
          static const char* ok[] = {"ALL_ON", "ALL_OFF", "GEN", "SCT", "REF", "COL",
                                     "CEL",    "LST",     "TMO", "INV", "SCATTERS"};
          bool known = false;
          for (const char* k : ok) known |= (kw == k);
          if (!known)
            throw Runtime("Valid report keywords are: ALL_ON, ALL_OFF, GEN, SCT, REF, COL, "
                          "CEL, LST, TMO, INV, or SCATTERS.");
          mission.Reports += (mission.Reports.empty() ? "" : ",") + kw;
        }
        break;
      case OPT_REPORT_FILE: mission.ReportFile = o.text(); break;
      case OPT_OUTDIR: mission.OutputDir = o.text(); break;
      case OPT_OCSRAW: par.OcsRaw = true; break;
      case OPT_SEISBINS:
        par.TimeBinsPerCycle = o.real();
        par.TimeBinSize = 0;
        break;
      case OPT_SEISBINSIZE:
        par.TimeBinsPerCycle = 0;
        par.TimeBinSize = o.real();
        break;
      case OPT_SEISARRAY:
        throw Runtime("Arg: --seisarray: Disabled; use --seis-p2p instead.");
      case OPT_SEIS_P2P: add_p2p_array(o, par, false); break;
      case OPT_SEIS_P2PW: add_p2p_array(o, par, true); break;
      case OPTM_HELP: mission.bHelpMsg = true; break;
      case OPTM_RUNSIM: mission.bRunSim = true; break;
      case OPTM_DUMPGRID: mission.bDumpGrid = true; break;
      case OPTM_PARAMOUTFN:
        mission.bOutputModParamsOctv = true;
        mission.FNModParamsOctv = o.text();
        break;
      case OPTM_RTTEST: mission.bRTCoefTest = true, mission.bRunSim = false; break;
      case OPTM_EVENTTEST: mission.bSourcePatternTest = true, mission.bRunSim = false; break;
      case OPTX_SEED: mission.Seed = (unsigned long)std::strtoull(o.text().c_str(), nullptr, 0); break;
      case OPTX_GPUS: mission.Gpus = (int)o.integer(); break;
      case OPTX_DEVICES:
        mission.Devices.clear();
        while (o.has()) mission.Devices.push_back((int)o.integer());
        if (mission.Devices.empty()) throw Runtime("--devices needs at least one device index.");
        break;
      case OPTX_SCATGRID: {
        long v[4];
        for (int k = 0; k < 4; k++) v[k] = o.integer();
        for (int k = 0; k < 3; k++) mission.GridLo[k] = o.real();
        for (int k = 0; k < 3; k++) mission.GridHi[k] = o.real();
        for (int k = 0; k < 3; k++) {
          if (v[k] < 1 || v[k] > 0xFFFFFFFFL || !(mission.GridHi[k] > mission.GridLo[k]))
            throw Runtime("--scatter-grid=NX,NY,NZ,FRAMES,X0,Y0,Z0,X1,Y1,Z1: cell counts must be positive (and below 2^32) and X1 > X0 etc.");
          mission.GridDims[k] = (unsigned)v[k];
        }
        if (v[3] < 1 || v[3] > 0xFFFFFFFFL) throw Runtime("--scatter-grid: FRAMES must be positive (and below 2^32).");
        mission.GridFrames = (unsigned)v[3];
        mission.bScatterGrid = true;
        break;
      }
      case OPTX_SCATGRID_FILE: mission.ScatterGridFile = o.text(); break;
      case OPTX_DEVTABLES: par.DeviceTables = true; break;
      case OPTX_HOSTTABLES: par.HostTables = true, par.DeviceTables = false; break;
    }
  }
}
