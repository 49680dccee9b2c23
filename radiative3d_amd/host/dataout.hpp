// dataout.hpp -- output writers: the reference's file and stdout formats,
// produced from the flat model + the engine's result block.
//
// Restates, byte for byte in layout, what the reference's DataReporter /
// Seismometer / ModelParams / Scatterer print (reference dataout.cpp:222-406,
// :623-694; model.cpp:88-197; scatterers.cpp:420-478), so the do-*.sh drivers
// and the Octave scripts under vis/ consume this engine's runs unchanged:
//   seis_%03d.octv        one GNU-Octave text struct per seismometer
//   seis_traces_asc.dat   all traces, ASCII (always in the CWD, dataout.hpp:332)
//   out_mparams.octv      run parameters (--mparams-outfile)
//   stdout                parameter echo, scatterer dump, post-sim loss summary
#ifndef R3DH_DATAOUT_HPP_
#define R3DH_DATAOUT_HPP_

#include <ostream>
#include <string>

#include "model.hpp"

// ModelParams::Output (model.cpp:88-132) and ::OutputOctaveText (:134-197)
void OutputModelParams(const ModelParams& par, std::ostream& out);
void OutputModelParamsOctave(const ModelParams& par, std::ostream& out);

// Scatterer::PrintAllScatteringStats (scatterers.cpp:420-478).  The "address"
// column (a heap pointer in the reference) carries the scatterer's index.
void PrintAllScatteringStats(const Model& model, std::ostream& out);

// Seismometer::OutputOctaveText (dataout.cpp:284-406) for seismometer `s`.
void OutputSeismometerOctave(const Model& model, const r3d_result& res, int s, std::ostream& out);

// DataReporter::OutputPostSimSummary (dataout.cpp:623-694): loss report to
// `console`, description + trace of every seismometer to `trace`, and one
// seis_NNN.octv per seismometer into `outdir` ("" = current directory).
void OutputPostSimSummary(const Model& model, const r3d_result& res, const std::string& outdir,
                          std::ostream& console, std::ostream& trace);

// --reports keywords (reference main.cpp:223-258) -> R3D_RPT_* mask.  `csv` is the keyword
// list as given ("ALL_ON", "GEN,SCT,REF", "SCATTERS", ...); empty = none.
uint32_t ReportMaskFromKeywords(const std::string& csv);

// DataReporter::output_phonon_dataline (dataout.cpp:484-520) for every record, grouped by
// history id in the order the events happened (the reference runs histories one after the
// other; the engine's buffer interleaves them).  The "cell:" column, a heap address in the
// reference, carries the cell index.
void OutputReports(const r3d_event* ev, size_t n, std::ostream& out);

// Header of a scatter-event grid written by --scatter-grid (no counterpart in the reference, whose video
// pipeline bins its report stream in Octave, vis/scattervid/scattervid_above.m:111): GNU/Octave text with
// the grid's shape, box, frame length and the name of the raw file beside it, which holds
// count[type P,S][frame][iz][iy][ix] as little-endian uint32 (x fastest) -- in Octave:
//   c = reshape(fread(fopen(GridFile), Inf, "uint32"), GridDims(1), GridDims(2), GridDims(3), GridFrames, 2);
void OutputScatterGridHeader(const unsigned dims[3], unsigned frames, const double lo[3], const double hi[3],
                             double frame_dt, const std::string& raw_file, unsigned long long events_binned,
                             unsigned long long saturated_cells, std::ostream& out);

#endif
