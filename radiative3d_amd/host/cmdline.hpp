// cmdline.hpp -- flag-compatible command-line front end.
//
// Accepts the option tokens of the reference's parser (cmdline.hpp:253-312)
// in the forms the do-*.sh drivers emit (scripts/do-fundamentals.sh:396-419):
// `--key=v1,v2,...`, `--key value`, bare `--key`, and the single-letter
// aliases -F -N -T -A -E -L.  Integer values take K/M/B suffixes
// (cmdline.cpp:343-390).  Engine-only additions: --seed, --gpus.
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
};

// Fills `params` / `mission` from argv-style tokens (program name excluded).
// Throws Runtime on unknown options or malformed values, like the
// reference's process_option (main.cpp:202-634).
void ParseCommandLine(const std::vector<std::string>& tokens, ModelParams& params,
                      MissionParams& mission);

#endif
