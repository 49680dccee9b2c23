// cmdline.hpp -- flag-compatible command-line front end.
//
// Accepts the option tokens of the reference's parser (cmdline.hpp:253-312)
// in the forms the do-*.sh drivers emit (scripts/do-fundamentals.sh:396-419):
// `--key=v1,v2,...`, `--key value`, bare `--key`, and the single-letter
// aliases -F -N -T -A -E -L.  Integer values take K/M/B suffixes
// (cmdline.cpp:343-390).  Engine-only additions: --seed, --gpus, --devices, --host-tables / --device-tables,
// --scatter-grid / --scatter-grid-file (the scatter-event histogram of a video run, written as a file).
#ifndef R3DH_CMDLINE_HPP_
#define R3DH_CMDLINE_HPP_

#include <string>
#include <vector>

#include "model.hpp"

struct MissionParams {
  bool bHelpMsg = false;
  bool bRunSim = true;
  bool bDumpGrid = false;
  bool bOutputModParamsOctv = false;
  bool bRTCoefTest = false;
  bool bSourcePatternTest = false;
  Text FNModParamsOctv;
  Text OutputDir;
  Text ReportFile;
  Text Reports;          // keyword list as given (INV, ALL_ON, ...)
  unsigned long Seed = 0x5EED;
  int Gpus = 1;
  std::vector<int> Devices;   // --devices=a,b,...: one shard per entry (a device may repeat); overrides --gpus
  // --scatter-grid=NX,NY,NZ,FRAMES,X0,Y0,Z0,X1,Y1,Z1: count SCT / REF events per wave type, frame
  // floor(t / (TTL / FRAMES)) and cell of the model-space box [X0,X1) x [Y0,Y1) x [Z0,Z1) -- the histogram the
  // reference's video scripts build from the report stream (vis/scattervid/scattervid_above.m:111)
  bool bScatterGrid = false;
  unsigned GridDims[3] = {0, 0, 0}, GridFrames = 0;
  double GridLo[3] = {0, 0, 0}, GridHi[3] = {0, 0, 0};
  Text ScatterGridFile = "scattergrid";   // <name>.octv (header) + <name>.u32 (counters), under --output-dir
};

// Fills `params` / `mission` from argv-style tokens (program name excluded).
// Throws Runtime on unknown options or malformed values, like the
// reference's process_option (main.cpp:202-634).
void ParseCommandLine(const std::vector<std::string>& tokens, ModelParams& params,
                      MissionParams& mission);

#endif
