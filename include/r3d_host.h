/* r3d_host.h -- C-ABI of the host-side model builder (libr3d_host.so).
 *
 * The builder is the C++ restatement of the reference's model construction
 * (Model::Model, model.cpp:220-501) that produces the flat r3d_model_desc the
 * engine consumes.  It takes the reference's own command-line tokens
 * (cmdline.hpp:253-312, as assembled by scripts/do-fundamentals.sh:396-419),
 * so a test or driver describes a run exactly as a do-*.sh script does.
 */
#ifndef R3D_HOST_H_
#define R3D_HOST_H_

#include "r3d.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct r3dh_model r3dh_model;

/* Build a model from argv-style option tokens (program name NOT included),
 * e.g. {"--grid-compiled=40", "--model-args=...", "--toa-degree=4", ...}.
 * Returns NULL on error (r3dh_last_error()).                                */
r3dh_model* r3dh_model_from_args(int argc, const char* const* argv);
void        r3dh_model_free(r3dh_model* m);

/* The flat tables; valid until r3dh_model_free.                             */
const r3d_model_desc* r3dh_model_desc(const r3dh_model* m);

/* Run parameters that are not part of the tables.                           */
uint64_t r3dh_num_phonons(const r3dh_model* m);   /* --num-phonons           */
uint64_t r3dh_seed(const r3dh_model* m);          /* --seed (default 0x5EED) */

/* Build log (the "@@ __PHASE__" / "|" lines the reference prints to stdout,
 * model.cpp:222-498).                                                       */
const char* r3dh_model_log(const r3dh_model* m);

/* ASCII grid dump, identical in layout to `--dump-grid`
 * (grid.cpp:376-405).  Valid until the model is freed.                      */
const char* r3dh_grid_dump(r3dh_model* m);

/* Scatterer summary row i: out[10] = nu, eps, a, kappa, el, gam0,
 * MFP_P, MFP_S, dipole_P, dipole_S (scatterers.cpp:420-478). Returns 0 ok.  */
int r3dh_scatterer_info(const r3dh_model* m, int i, double out[10]);

/* Scatterer dump block ("#  BEGIN SCATTERER DUMP:" ... "#  END SCATTERERS",
 * scatterers.cpp:420-478) and the run-parameter echo (model.cpp:88-132).    */
const char* r3dh_scatterer_dump(r3dh_model* m);
const char* r3dh_params_echo(r3dh_model* m);

/* Write the reference's output files for a finished run (dataout.cpp:623-694):
 * seis_NNN.octv into `outdir` ("" = cwd), the ASCII traces into `trace_path`,
 * out_mparams.octv-style parameters into `mparams_path` (NULL to skip).
 * Returns the post-sim console summary text, or NULL on error.              */
const char* r3dh_write_outputs(r3dh_model* m, const r3d_result* result, const char* outdir,
                               const char* trace_path, const char* mparams_path);

/* For a model built with --device-tables (scattering tables left to the engine):
 * record what r3d_engine_scatterer_stats() returned, so that the scatterer dump
 * and r3dh_scatterer_info show the engine's numbers.  Returns 0 ok.            */
int r3dh_model_set_scatterer_stats(r3dh_model* m, int s, const double mfp[2], const double dipole[2]);
int r3dh_model_device_tables(const r3dh_model* m);   /* 1 if built with --device-tables */

/* --reports keyword list ("ALL_ON", "GEN,SCT,REF", "SCATTERS", ... as on the
 * reference's command line, main.cpp:223-258) -> R3D_RPT_* mask; (uint32_t)-1 on
 * an unknown keyword.  The mask the model's own argv asked for:            */
uint32_t r3dh_report_mask(const char* keywords);
uint32_t r3dh_model_report_mask(const r3dh_model* m);

/* Print event records in the reference's report-line format
 * (dataout.cpp:484-520), grouped by history.  path != NULL: write that file and
 * return ""; path == NULL: return the text.  NULL on error.                 */
const char* r3dh_write_reports(r3dh_model* m, const r3d_event* events, uint64_t n, const char* path);

/* The coordinate system the model was built in (reference ecs.hpp:242-257): map_code 0 ENU_ORTHO,
 * 1 RAE_ORTHO, 2 RAE_CURVED, 3 RAE_SPHERICAL; the Earth radius; whether --flatten applied.
 * And the axes scheme of seismometer i (model.cpp:486-491): 0 ENZ, 1 RTZ, -1 out of range.
 * For tests that restate the builders (oracle/r3d_tables_oracle.cpp).  Returns 0 ok.         */
int r3dh_model_coordinates(const r3dh_model* m, int* map_code, double* earth_radius, int* flattened);
/* The grid as the cell builders see it (reference grid.hpp:341-403): dims = ni, nj, nk; nodes in the
 * grid's own order (k slowest, then j, then i), each with its model-space location (GridNode::Loc),
 * its radius from the Earth's centre (curved mappings; else 0), the number of attribute sets given
 * (2 = a first-order discontinuity) and Data(GN_ABOVE) / Data(GN_BELOW) after the coordinate system's
 * conversion: vp vs rho qp qs nu eps a kappa.  Valid while this model is the most recently built one. */
typedef struct r3dh_grid_node {
  double  loc[3];
  double  radius;
  double  side[2][9];
  int32_t n_sets;
  int32_t pad_;
} r3dh_grid_node;
/* The same nodes as the model definition wrote them, BEFORE the coordinate system's conversion: the
 * raw coordinate triple (GridNode::GetRawLoc) and the attribute sets in the order given, each as
 * vp vs rho | which Q is the unknown (0 Qp, 1 Qs, 2 Qk) and the stored Qp Qs Qk | nu eps a kappa.  */
typedef struct r3dh_grid_node_raw {
  double  x[3];
  double  set[2][11];
  int32_t n_sets;
  int32_t pad_;
} r3dh_grid_node_raw;
int r3dh_grid_nodes_raw(const r3dh_model* m, r3dh_grid_node_raw* out, size_t capacity);
int r3dh_grid_size(const r3dh_model* m, int dims[3]);
int r3dh_grid_nodes(const r3dh_model* m, r3dh_grid_node* out, size_t capacity);
int r3dh_seismometer_axes(const r3dh_model* m, int i);

const char* r3dh_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
