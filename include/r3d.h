/* r3d.h -- C-ABI of the MI355X phonon-transport engine (libr3d_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of Radiative3D: the body of
 * Model::RunSimulation() (reference model.cpp:602-633), i.e. N independent
 * calls of ShearDislocation::GenerateEventPhonon() (events.cpp:111-124)
 * followed by Phonon::Propagate() (phonons.cpp:540-682) and everything those
 * two call.  The reference has no plugin/FFI interface; the seam is "a fully
 * built immutable model in, filled seismometer bins + loss counters out".
 *
 * Everything crossing the boundary is plain C: POD structs, pointers, sizes.
 * No C++ types, no torch types, no exceptions, no exit().  A maintainer of the
 * reference would fill r3d_model_desc from the live objects of its Model
 * (see INTEGRATION.md for the field-by-field mapping and the stub to add to
 * model.cpp) and call r3d_run() instead of the for-loop at model.cpp:611-625.
 *
 * Units follow the reference: km, s, km/s, arbitrary density.  All reals are
 * fp64 (reference typedefs.hpp:90, Real = double).
 */
#ifndef R3D_H_
#define R3D_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ray types (reference raytype.hpp:12-20) --------------------------- */
enum { R3D_RAY_P = 0, R3D_RAY_S = 1 };

/* ---- cell kinds (reference media.hpp: RCUCylinder :312, Tetra :400,
 *      SphereShell :467).  A model is homogeneous in kind
 *      (model.cpp:378-411). ---------------------------------------------- */
enum { R3D_CELL_CYLINDER = 0, R3D_CELL_TETRA = 1, R3D_CELL_SPHERESHELL = 2 };

/* ---- face flags (reference media_cellface.hpp:120-142) ----------------- */
enum {
  R3D_FACE_COLLECT = 1u, /* mCollect   : report to seismometers on arrival  */
  R3D_FACE_REFLECT = 2u, /* mReflect   : free surface, full R/T, no transmit*/
  R3D_FACE_ADJOIN  = 4u, /* mAdjoin    : has a neighbour cell               */
  R3D_FACE_DISCON  = 8u  /* mGridDiscon: first-order discontinuity -> R/T   */
};

/* One bounding face of a cell.
 *  plane faces  (PlaneFace,   media_cellface.hpp:273-276): normal + point
 *  sphere faces (SphereFace,  media_cellface.hpp:375-378): signed radius
 *               (+R outward-normal top face, -R inward-normal bottom face)
 *  cylinder wall(CylinderFace,media_cellface.hpp:326-327): radius, no
 *               neighbour, no flags (the shared static loss face,
 *               media.cpp:124-125).                                         */
typedef struct r3d_face {
  double   normal[3];
  double   point[3];
  double   radius;
  int32_t  neighbor;   /* index of the cell across the face, -1 if none     */
  uint32_t flags;
} r3d_face;

/* One medium cell.  Which members are meaningful depends on the model's
 * cell kind:
 *  CYLINDER   : vel_c = mVelTop, rho_c = mDensity, q = mQ; faces 0=top plane,
 *               1=bottom plane, 2=cylinder wall      (media.hpp:312-331)
 *  TETRA      : vel_grad/vel_c = mVelGrad/mVel0, rho_grad/rho_c, q = mQ;
 *               faces 0..3 = FACE_A..FACE_D            (media.hpp:400-408)
 *  SPHERESHELL: v(r) = vel_a r^2 + vel_c, rho(r) = rho_a r^2 + rho_c,
 *               zero_rad2 = mZeroRadius2; faces 0=top, 1=bottom
 *                                                       (media.hpp:467-478) */
typedef struct r3d_cell {
  double   vel_c[2];
  double   vel_a[2];
  double   vel_grad[2][3];
  double   rho_c;
  double   rho_a;
  double   rho_grad[3];
  double   q[2];
  double   zero_rad2[2];
  int32_t  scatterer;   /* index into r3d_model_desc.scatterers             */
  int32_t  n_faces;     /* 3 / 4 / 2                                        */
  r3d_face faces[4];
} r3d_cell;

/* Scatterer (reference scatterers.hpp:157,181 + sources.hpp:130-135).
 * cdf[k] are the INTEGRATED (cumulative, un-normalised) distributions over
 * the take-off-angle set for GPP, GPS, GSP, GSS (probability.cpp:21-35);
 * whole_cdf[in] is the 4-entry cumulative conversion table for an incoming
 * P (in=0) or S (in=1) phonon (scatterers.cpp:172-184).                     */
typedef struct r3d_scatterer {
  double        mfp[2];
  double        whole_cdf[2][4];
  const double* cdf[4];      /* each n_toa long                             */
  const double* spol;        /* n_toa, S->S polarisation (scatparams.cpp:114)*/
  /* Build-on-device form.  With cdf[0] == NULL the engine evaluates the tables
   * itself, in HBM, from the medium's heterogeneity parameters (what
   * Scatterer::Scatterer does on the host, scatterers.cpp:97-220, with
   * ScatterParams::GSATO, scatparams.cpp:75-194); whole_cdf, and mfp unless
   * mfp_fixed, are then outputs -- read them with r3d_engine_scatterer_stats. */
  double        het[6];      /* nu, eps, a, kappa, el = omega/Vs, gam0 = Vp/Vs
                                (scatparams.hpp:60-77)                       */
  double        psdf_numer;  /* 8 pi^1.5 eps^2 a^3 Gamma(kappa+1.5)/Gamma(kappa),
                                the numerator of PSATO (scatparams.cpp:175-190)*/
  uint32_t      mfp_fixed;   /* keep mfp[] as given (--overridemfp)          */
  uint32_t      pad_;
} r3d_scatterer;

/* Event source (reference events.hpp:57-61, sources.hpp:130-135):
 * cumulative P / SH / SV radiation patterns over the TOA set.              */
typedef struct r3d_source {
  double        loc[3];
  int32_t       cell;
  int32_t       pad_;
  double        whole_cdf[3];
  const double* cdf[3];      /* each n_toa long                             */
  /* Build-on-device form.  With cdf[0] == NULL the engine evaluates the three
   * radiation patterns itself, in HBM, from the moment tensor rotated into
   * the local north-east-down frame at the source (what
   * ShearDislocation::ShearDislocation does on the host, events.cpp:66-105);
   * whole_cdf is then an output (r3d_engine_download_source).               */
  double        moment[6];   /* xx, yy, zz, xy, xz, yz                       */
} r3d_source;

/* Seismometer (reference dataout.hpp:102-129, ctor dataout.cpp:42-71).     */
typedef struct r3d_seismometer {
  double loc[3];
  double axes[3][3];   /* X1, X2, X3 unit vectors                           */
  double r_in[2];      /* inner gather radius by ray type (0 = disc)        */
  double r_out[2];     /* outer gather radius by ray type                   */
  double area[2];      /* pi (r_out^2 - r_in^2)                             */
} r3d_seismometer;

/* Scalar run parameters = the class statics the hot path reads
 * (phonons.cpp:31-36, media.cpp:32, dataout.cpp:33-34,
 *  scatterers.hpp:114, ecs.hpp:242-257).                                   */
typedef struct r3d_params {
  double   ttl;            /* Phonon::cm_ttl                                */
  double   frequency;      /* MediumCell::cmPhononFreq (Hz)                 */
  double   time_per_bin;   /* Seismometer::cmTimePerBin                     */
  uint32_t n_bins;         /* Seismometer::cmNumBins                        */
  uint32_t no_deflect;     /* Scatterer::cm_NoDeflect_b                     */
  double   min_theta;      /* Phonon::cm_min_theta                          */
  double   max_theta;      /* Phonon::cm_max_theta                          */
  double   slow_concern;   /* Phonon::cm_slow_concern                       */
  uint64_t loop_concern;   /* Phonon::cm_loop_concern                       */
  double   earth_center[3];/* ECS earth centre (sphere-shell models)        */
} r3d_params;

/* The whole immutable model, as read by the hot path.                      */
typedef struct r3d_model_desc {
  int32_t                cell_kind;
  int32_t                n_cells;
  const r3d_cell*        cells;
  int32_t                n_scatterers;
  int32_t                n_seismometers;
  const r3d_scatterer*   scatterers;
  const r3d_seismometer* seismometers;
  uint64_t               n_toa;
  const double*          toa;        /* n_toa x (theta, phi)                 */
  r3d_source             source;
  r3d_params             params;
  /* Build-on-device form of the take-off set.  With toa == NULL the engine
   * generates it itself: the icosahedron-based tessellation of the sphere of
   * degree toa_degree (reference S2::TesselSphere, geom_s2.cpp:60-292;
   * n_toa must be 20 * 4^toa_degree), in the order of the host builder's.   */
  int32_t                toa_degree;
  int32_t                pad_;
} r3d_model_desc;

/* Invalid-phonon reason slots (reference dataout.hpp:229-237, bit order).  */
enum {
  R3D_INV_PATH_NAN = 0, R3D_INV_TIME_NAN, R3D_INV_PATH_NEGATIVE,
  R3D_INV_TIME_NEGATIVE, R3D_INV_STUCK, R3D_INV_SLOW, R3D_INV_LOOP_EXCEED,
  R3D_INV_NUM
};

/* Event counters (diagnostic; one increment where the reference would emit
 * the corresponding report line, dataout.cpp:526-617).                     */
enum {
  R3D_EV_GENERATED = 0, /* GEN */
  R3D_EV_ITERATIONS,    /* loop iterations of Phonon::Propagate             */
  R3D_EV_SCATTER,       /* SCT */
  R3D_EV_COLLECT,       /* COL (collection-face arrivals)                   */
  R3D_EV_CATCH,         /* seismometer bin increments                       */
  R3D_EV_REFLECT,       /* REF (free surface + interface reflections)       */
  R3D_EV_TRANSFER,      /* CEL */
  R3D_EV_RTSOLVE,       /* Refraction_FullRT calls                          */
  R3D_EV_VOLUME_OUT,    /* SCT / REF events that fell outside an attached event grid (r3d_engine_set_volume): with
                           a grid attached, the grid's total = SCATTER + REFLECT - VOLUME_OUT, exactly            */
  R3D_EV_NUM
};

#define R3D_N_ENERGY 5  /* X, Y, Z, P, S  (BinRecord, dataout.hpp:77-93)    */
#define R3D_N_COUNT  2  /* n_P, n_S                                         */

/* Result block.  `energy` and `counts` are caller-allocated and are
 * ACCUMULATED into (so shards can be chained); the scalar counters are
 * likewise added to.
 *   energy[(s*n_bins + b)*5 + c],  counts[(s*n_bins + b)*2 + t]            */
typedef struct r3d_result {
  double*   energy;
  uint64_t* counts;
  uint64_t  n_lost;      /* DataReporter::mNumLost    (dataout.cpp:591-598) */
  uint64_t  n_timeout;   /* mNumTimeout               (dataout.cpp:600-607) */
  uint64_t  n_invalid;   /* mNumInvalid               (dataout.cpp:611-617) */
  uint64_t  invalid_reasons[R3D_INV_NUM];
  uint64_t  events[R3D_EV_NUM];
} r3d_result;

/* Optional per-history final record, for parity tests (the reference's LST /
 * TMO / INV report lines carry the same fields, dataout.cpp:484-520).      */
typedef struct r3d_final {
  double   time, path, amp;
  double   loc[3];
  double   dir[3];
  uint32_t moves;
  uint8_t  fate;        /* 1 lost, 2 timeout, 3 invalid                     */
  uint8_t  type;        /* ray type at the end                              */
  uint16_t n_catch;     /* seismometer catches along the way                */
} r3d_final;

typedef struct r3d_engine r3d_engine;   /* opaque: tables resident in HBM   */

/* Copy the model into HBM on `device` and build the engine's own device
 * layout.  Returns NULL on error (see r3d_last_error).  Thread-compatible:
 * one engine per thread.                                                   */
r3d_engine* r3d_engine_create(const r3d_model_desc* model, int device);
/* The same with the kernel's LDS carve-up in the caller's hands -- for tests that must reach a
 * given compiled kernel variant on a small model, and for tuning runs.  r3d_engine_create is this
 * call with opts == NULL: every field automatic.  The library reads no environment variable.   */
typedef struct r3d_engine_opts {
  uint32_t size;             /* sizeof(r3d_engine_opts): a mismatch is refused                  */
  int32_t  residency;        /* -1 automatic; else run at least this far down the list of table
                                residencies: 0 cell records + scatterer heads staged in LDS,
                                1 the heads only, 2 neither (r3d_engine_variant)                */
  uint32_t pool_slots;       /* history slots of a workgroup's pool: 0 = 1024 (all of them);
                                rounded down to a multiple of 64, at least the workgroup size   */
  int32_t  accumulator_bits; /* -1 automatic; 0 no bin accumulators in LDS; 5..8: 2^bits entries */
  uint32_t lds_reserve;      /* bytes of LDS the carve-up leaves alone, as if the model's tables
                                were that much larger                                            */
} r3d_engine_opts;
r3d_engine* r3d_engine_create_ex(const r3d_model_desc* model, int device, const r3d_engine_opts* opts);
void        r3d_engine_destroy(r3d_engine* e);
/* Checked form of destroy: waits for the engine's launches, and REFUSES (returns
 * nonzero, engine left intact) while histories carried over by r3d_run_device_carry
 * are still in it -- their tallies and bins would be lost; flush with final != 0
 * first.  r3d_engine_destroy drops them and leaves a note in r3d_last_error.
 * Every entry point makes the engine's device current for its own work and
 * restores the caller's current device before it returns.                    */
int         r3d_engine_close(r3d_engine* e);
int         r3d_engine_carry_pending(const r3d_engine* e);   /* 1 if a chain awaits its flush */

/* Scatterer s as the engine holds it: out[0..1] = MFP P, S; out[2..3] = dipole
 * moments P, S (scatterers.cpp:244-259; NaN for host-built tables, whose raw
 * weights the engine never sees); out[4..7] = totals of the four cumulative
 * tables.  Returns 0 on success.                                            */
int r3d_engine_scatterer_stats(const r3d_engine* e, int s, double out[8]);
/* Copy scatterer s's tables from HBM (n_toa doubles each; any pointer may be
 * NULL to skip): for parity tests of the build-on-device form.              */
int r3d_engine_download_scatterer(r3d_engine* e, int s, double* cdf[4], double* spol);

/* The source tables and the take-off set as the engine holds them (for parity
 * tests of their build-on-device forms): cdf[k] n_toa doubles each, whole[3]
 * the cumulative P / SH / SV totals, toa n_toa (theta, phi) pairs; any pointer
 * may be NULL to skip.                                                       */
int r3d_engine_download_source(r3d_engine* e, double* cdf[3], double whole[3]);
int r3d_engine_download_toa(r3d_engine* e, double* toa);

/* Sizes of the result block for this engine's model.                       */
size_t r3d_energy_len(const r3d_engine* e);   /* doubles                    */
size_t r3d_counts_len(const r3d_engine* e);   /* uint64s                    */

/* Run histories [first_id, first_id + n) with RNG key `seed` and ADD the
 * outcome into *out (host memory).  Replaces model.cpp:611-625.
 * Returns 0 on success.  Histories are keyed by id, so the union of any
 * partition of an id range gives the same result up to fp64 summation
 * order.                                                                   */
int r3d_run(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed,
            r3d_result* out);

/* Device-resident variant for multi-GPU jobs: accumulates into caller-owned
 * DEVICE buffers (e.g. the data_ptr of a torch tensor that is all-reduced
 * over RCCL afterwards) on `stream` (a hipStream_t, or NULL for the default
 * stream).  d_scalars holds 3 + R3D_INV_NUM + R3D_EV_NUM uint64 counters in
 * the order n_lost, n_timeout, n_invalid, invalid_reasons[], events[].
 * Asynchronous: returns after enqueueing.  d_finals may be NULL.           */
/* (Up to 64 r3d_run_device launches of one engine may be in flight at a time, on
 * different streams and into different buffers: a batch ends in a drain phase in
 * which ever fewer lanes still carry a history -- the longest histories are ~40
 * times the mean -- and the next batch's workgroups fill the CUs it frees.)      */
int r3d_run_device(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed,
                   double* d_energy, uint64_t* d_counts, uint64_t* d_scalars,
                   r3d_final* d_finals, void* stream);

/* The seam in one call: engines on devices 0 .. n_gpus-1, the id
 * range [first_id, first_id + n) cut into n_gpus contiguous shards run concurrently,
 * and the shards' results summed on the devices (RCCL, see r3d_node_run below) and
 * ADDED into *out.  Equals r3d_run on one engine for the same ids (integers exactly, energies
 * to summation order).  Returns 0 on success.                                 */
int r3d_run_model(const r3d_model_desc* model, uint64_t n, uint64_t first_id, uint64_t seed,
                  int n_gpus, r3d_result* out);
/* The same with the devices named: shard g of n_devices runs on devices[g] (a device may
 * appear more than once -- the shards then share it).  r3d_run_model is this call with
 * devices 0 .. n_gpus-1.  If any shard fails, nothing is added to *out and the message names
 * the shard and its device.  (Reference seam: model.cpp:602-633; replicas summed as in
 * vis/seisplot/combine.m:26-33.)                                               */
int r3d_run_model_on(const r3d_model_desc* model, uint64_t n, uint64_t first_id, uint64_t seed,
                     const int* devices, int n_devices, r3d_result* out);

/* A NODE: the same seam for a host that runs more than one job on a model (r3d_run_model_on builds and
 * drops a node per call: tables made or uploaded every time).  r3d_node_create: one engine per entry of
 * `devices`, kept until r3d_node_destroy.  r3d_node_run: the id range cut into contiguous shards as
 * above, every shard's kernel enqueued on its engine's stream into that engine's own block in HBM, and
 * the blocks -- f64 energies, u64 counts, u64 counters -- SUMMED ON THE DEVICES by one grouped RCCL
 * reduce per buffer (ncclReduce, sum, to devices[0], in stream order behind the kernels: xGMI between
 * the GPUs of a node); the host reads that one block and ADDS it into *out.  This is the reference's
 * "replicas + combine" (scripts/do-parallel.sh:23-29, vis/seisplot/combine.m:26-33: replicas' traces
 * add).  The blocks are added on the HOST instead where RCCL has nothing to do or cannot do it: a node
 * of one shard, shards that share a device (RCCL refuses a communicator that names a device twice), no
 * usable librccl in the process, a communicator that could not be formed, or one that failed in an
 * earlier run (it is aborted, never waited for, and that run is summed on the host too) --
 * r3d_node_reduction says which ("rccl" / "host"), r3d_node_reduction_note why the host.  librccl is
 * bound at first use: the copy already in the process if there is one (a Python host with torch), else
 * the loader's search path, else /opt/rocm/lib.  Counts and counters equal one engine's run of the same
 * ids exactly, energies to summation order.  If a shard fails nothing is added to *out and the message
 * names shard and device.
 * r3d_node_engine: shard g's engine (to attach an event log or a grid before a run, to read them
 * after); it stays the node's.                                                        */
typedef struct r3d_node r3d_node;
r3d_node* r3d_node_create(const r3d_model_desc* model, const int* devices, int n_devices);
int r3d_node_run(r3d_node* node, uint64_t n, uint64_t first_id, uint64_t seed, r3d_result* out);
int r3d_node_size(const r3d_node* node);
r3d_engine* r3d_node_engine(r3d_node* node, int shard);
const char* r3d_node_reduction(const r3d_node* node);
const char* r3d_node_reduction_note(const r3d_node* node);
void r3d_node_destroy(r3d_node* node);

/* ONE PROCESS PER GPU: the same reduction for a job whose shards are processes (a launcher starts one
 * rank per device; each rank owns an engine and a result block in HBM, r3d_run_device).  The ranks
 * form an RCCL communicator through this library -- rank 0 makes the 128-byte id
 * (r3d_comm_unique_id), the host carries it to the other ranks by whatever channel it has, every rank
 * calls r3d_comm_create(id, rank, n_ranks, its device) -- and r3d_comm_reduce sums the block's three
 * buffers over the ranks IN PLACE, in stream order on `stream`: into rank `root`'s buffers
 * (root >= 0; the other ranks' buffers are then scratch), or into every rank's (root < 0).  It is the
 * code r3d_node_run reduces with, so a job of N processes and a job of N shards in one process add
 * their replicas the same way (vis/seisplot/combine.m:26-33).  A reduce that fails aborts the
 * communicator (never waits on it) and every later call on it fails.  r3d_comm_describe: the size RCCL
 * itself reports (ncclCommCount), this rank, its device and that device's UUID, the RCCL version and
 * which library file was bound.                                                        */
#define R3D_COMM_ID_BYTES 128
typedef struct r3d_comm r3d_comm;
typedef struct r3d_comm_info {
  int32_t n_ranks, rank, device, rccl_version;
  char device_uuid[40];    /* 32 hex digits */
  char library[256];
} r3d_comm_info;
int r3d_comm_unique_id(unsigned char id[R3D_COMM_ID_BYTES]);
r3d_comm* r3d_comm_create(const unsigned char id[R3D_COMM_ID_BYTES], int rank, int n_ranks, int device);
int r3d_comm_reduce(r3d_comm* comm, double* d_energy, uint64_t n_energy, uint64_t* d_counts, uint64_t n_counts,
                    uint64_t* d_scalars, uint64_t n_scalars, int root, void* stream);
int r3d_comm_describe(const r3d_comm* comm, r3d_comm_info* info);
void r3d_comm_destroy(r3d_comm* comm);

/* r3d_run_device for a CHAIN of batches (same engine, same seed, launches in stream
 * order).  A batch ends in a drain phase in which ever fewer lanes still carry a
 * history, and the longest histories are ~40 times the mean: ~8 of the 25 ms of a
 * 1e7-history NSCP launch.  With final == 0 the histories still in flight when the
 * batch's ids run out stay in the engine (HBM) and are resumed by its next carry
 * launch, so every launch runs full; final != 0 also runs everything carried to its
 * end (n may be 0: flush only).  The SUM over the chain's launches equals r3d_run on
 * the union of their id ranges (integers exactly, energies to summation order); a
 * single launch's buffers hold what that launch executed.                     */
int r3d_run_device_carry(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed,
                         double* d_energy, uint64_t* d_counts, uint64_t* d_scalars,
                         void* stream, int final);

/* Like r3d_run, additionally returning the per-history final records
 * (finals[i] for id first_id + i; caller-allocated, n entries).            */
int r3d_run_traced(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed,
                   r3d_result* out, r3d_final* finals);

/* The same records out of the PRODUCTION kernels that end histories for good (r3d_run_traced runs the diagnostic
 * kernel, another compilation).  r3d_engine_set_production_finals attaches an engine-owned buffer of `capacity`
 * records; from then on a history with base_id <= id < base_id + capacity leaves its final record when it ends in
 *   - a SELF-CONTAINED launch (r3d_run, r3d_run_device, r3d_node_run, r3d_run_device_carry with final != 0 and
 *     ids of its own): every history of the launch, or
 *   - a chain's FLUSH (r3d_run_device_carry with final != 0): the histories carried into it.
 * A chain's STEP launches (r3d_run_device_carry with final == 0) write NO record: their kernel is compiled without
 * that code (it cost the launch 3.5 %), so a history that ends inside a step launch keeps fate 255 -- "never written"
 * -- and is accounted for through the bins and counters it leaves; launches that run the diagnostic kernel (an event
 * log or r3d_run_traced's records attached) do not write these records either.  (A test-only compilation with the
 * step kernel's stores in, `make variant DEFS=-DR3D_STEP_FINALS=1`, is held against the oracle per history.)
 * Capacity 0 detaches; while a buffer is attached a launch whose ids it does not cover is refused, and a buffer
 * cannot be attached or detached while histories are carried over.  r3d_production_finals_read copies records
 * [first, first + count) to the host (waits for the engine's launches).  n_catch is 0xFFFF in all of them (only
 * the diagnostic kernel counts catches per history); `amp` is exponentiated on the host by the read.  Costs a run
 * without a buffer one scalar test per batch in which a history ends.                                        */
int r3d_engine_set_production_finals(r3d_engine* e, uint64_t base_id, uint64_t capacity);
int r3d_production_finals_read(r3d_engine* e, r3d_final* out, uint64_t first, uint64_t count);

/* ---- optional volumetric scatter-event grid ---------------------------------
 * The reference's "scattervid" data are one text line per SCT / REF event
 * (dataout.cpp:484-520, 570-577), cut down to (t, x, y, z) per resulting wave
 * type by vis/scattervid/preprocess.sh:17-29 and binned per frame floor(t/dt)
 * by scattervid_above.m:111 -- 6.6 KB of text per history.  The engine keeps
 * the histogram those scripts build, in HBM:
 *     count[type(P,S)][frame][iz][iy][ix]   (uint32, model coordinates)
 * incremented with one atomic at every scatter (SCT) and reflection (REF)
 * event that falls inside the grid and the frame range.                      */
typedef struct r3d_volume_desc {
  double   origin[3];     /* model-space corner of cell (0,0,0)               */
  double   cell_size[3];
  uint32_t dims[3];       /* nx, ny, nz                                       */
  uint32_t n_frames;
  double   frame_dt;      /* seconds per frame                                */
} r3d_volume_desc;

/* Attach (or with v == NULL detach) a volume grid; allocates and zeroes
 * 2 * n_frames * nz * ny * nx uint32 counters in HBM.  Subsequent runs
 * accumulate into it.                                                        */
int    r3d_engine_set_volume(r3d_engine* e, const r3d_volume_desc* v);
size_t r3d_volume_len(const r3d_engine* e);            /* counters (0 if none) */
/* Copy the counters to `out` (r3d_volume_len entries); reset != 0 zeroes them
 * on the device afterwards.                                                  */
int    r3d_volume_read(r3d_engine* e, uint32_t* out, int reset);
/* The same for counters [begin, begin + count) only (e.g. the frames an engine holds job totals for after
 * r3d_volume_reduce_by_frame); no reset.                                      */
int    r3d_volume_read_range(r3d_engine* e, uint64_t begin, uint64_t count, uint32_t* out);
/* Device address of the counters, for an RCCL reduction across ranks.        */
void*  r3d_volume_device_ptr(r3d_engine* e);
/* The same grid in CALLER-OWNED device memory (r3d_volume_len counters, zeroed
 * by the caller; e.g. a torch tensor that is reduced over RCCL afterwards, as
 * r3d_run_device does for the bins).  v == NULL detaches.                    */
int    r3d_engine_set_volume_buffer(r3d_engine* e, const r3d_volume_desc* v, uint32_t* d_counters);

/* ---- the grid between ranks ---------------------------------------------------
 * Replicas add (vis/seisplot/combine.m:26-33), and a rank's grid is sparse: ~10 events per
 * history touch < 5 % of config 5's 2.5e9 cells.  These two calls are what a multi-GPU job needs
 * to add the grids of its ranks WITHOUT moving 10 GB per rank: every rank compacts, per owner of
 * a range of frames, the non-zero counters of that range into (index, count) pairs; the pairs
 * travel point to point (RCCL send / recv); the owner adds what it receives into its own range
 * (radiative3d_amd/parallel.py DeviceVolume.reduce_scatter_frames_).  Asynchronous on `stream`
 * (a hipStream_t; NULL = the default stream); d_* are device pointers on `device`.
 *
 * r3d_volume_compact: appends the non-zero counters of d_counters[begin, end) to d_pairs as pairs
 * of uint32 {index, count} -- index counted from d_counters, so end <= 2^32 -- starting at slot
 * *d_n, and adds their number to *d_n (a device counter the caller zeroes; it may pass `capacity`
 * pairs: what does not fit is counted, not written).
 * r3d_volume_scatter_add: d_counters[index] += count for n pairs, index < len, saturating at
 * 2^32 - 1.  d_flags[0] counts the cells that reached the ceiling, d_flags[1] pairs whose index was
 * out of range (not written); the caller zeroes both.                              */
int r3d_volume_compact(int device, const uint32_t* d_counters, uint64_t begin, uint64_t end, uint32_t* d_pairs,
                       uint64_t capacity, uint64_t* d_n, void* stream);
int r3d_volume_scatter_add(int device, uint32_t* d_counters, uint64_t len, const uint32_t* d_pairs, uint64_t n,
                           uint64_t* d_flags, void* stream);

/* The same reduction for a host that drives its GPUs from ONE process (r3d_run_model_on's way; no
 * communication library): engines[0 .. n-1] -- one per shard of the job, each with its own grid of one
 * shape (r3d_engine_set_volume), on any devices, the same device included -- end up holding the job's
 * counts for their share of the frames: engine g for frames [frames[g], frames[g + 1]) of both wave
 * types (`frames`: n + 1 entries, may be NULL; the cut is contiguous and balanced), the rest of its grid
 * keeps that engine's own counts.  Pairs travel by hipMemcpyPeer.  *saturated (may be NULL) receives the
 * number of cells that reached 2^32 - 1.  Waits for every launch of the engines.  Returns 0 on success.
 * In two phases: first every engine COUNTS what it would send (the compaction with no room to write) -- a grid
 * with more than a sixteenth of its cells non-zero in the other owners' frames fails the call here, with every
 * grid as it was --; only then, source by source, the pairs are written into a buffer of exactly their number,
 * travel, are added by their owners, and the buffer is freed: one pair buffer is alive at a time, so shards
 * that share a device need the room of one.  After a failure in the second phase (a failed HIP call) the grids
 * are undefined.                                                                                         */
int r3d_volume_reduce_by_frame(r3d_engine* const* engines, int n, uint32_t* frames, uint64_t* saturated);

/* ---- optional per-event report stream --------------------------------------
 * The reference's `--reports[=KEYWORDS]` (main.cpp:223-258) writes one text line
 * per event with the phonon's state at that moment (dataout.cpp:484-520): GEN
 * after generation (events.cpp:120), SCT after a scatter (phonons.cpp:616), COL
 * on arrival at a collection face, with the incident state (:630), REF / CEL
 * after a reflection / hand-over (:643, :659-661), LST / TMO / INV when the
 * history ends (:550-598, :675).  The engine appends the same records, in
 * binary, to a buffer in HBM; records of one history appear in the order they
 * happened (histories interleave).  host: r3dh_write_reports() prints them in
 * the reference's line format.                                               */
enum {
  R3D_RPT_GEN = 1u, R3D_RPT_SCT = 2u, R3D_RPT_REF = 4u, R3D_RPT_COL = 8u,
  R3D_RPT_CEL = 16u, R3D_RPT_LST = 32u, R3D_RPT_TMO = 64u, R3D_RPT_INV = 128u,
  R3D_RPT_ALL = 255u
};
typedef struct r3d_event {
  uint64_t id;          /* history id (mSID)                                */
  double   time, path, amp;
  double   loc[3];      /* model coordinates                                */
  double   dir[3];      /* unit vector                                      */
  uint32_t cell;        /* cell index (the reference prints the address)    */
  uint32_t moves;       /* mMoveCount                                       */
  uint8_t  tag;         /* 0 GEN 1 SCT 2 REF 3 COL 4 CEL 5 LST 6 TMO 7 INV  */
  uint8_t  type;        /* ray type                                         */
  uint8_t  pad_[6];
} r3d_event;            /* 96 bytes                                         */

/* Attach an event buffer of `capacity` records for the tags in `mask`
 * (R3D_RPT_*); mask == 0 or capacity == 0 detaches.  Subsequent runs append;
 * events beyond the capacity are counted but not stored.                     */
int      r3d_engine_set_event_log(r3d_engine* e, uint32_t mask, uint64_t capacity);
/* Events reported since the buffer was attached or last reset (may exceed the
 * capacity).                                                                 */
uint64_t r3d_event_log_count(r3d_engine* e);
/* Copy up to `max` stored records to `out`; returns the number copied, or
 * (uint64_t)-1 on error.  reset != 0 empties the buffer afterwards.          */
uint64_t r3d_event_log_read(r3d_engine* e, r3d_event* out, uint64_t max, int reset);

/* Duration in milliseconds of the traversal kernel launch enqueued by the
 * most recent r3d_run / r3d_run_device[_carry] call on this engine, measured
 * with HIP events on the stream it was launched on (blocks until it has
 * completed).  Launches are numbered from 1 in enqueue order: r3d_launch_count
 * is the number so far, r3d_kernel_ms reads any of the 64 most recent (each has
 * its own event pair, so overlapping launches on several streams are timed
 * separately); -1 for a launch that is not on record.                        */
double   r3d_last_kernel_ms(r3d_engine* e);
uint64_t r3d_launch_count(const r3d_engine* e);
double   r3d_kernel_ms(r3d_engine* e, uint64_t launch);

/* Which compiled kernel variant serves this engine: cell kind * 4 + table residency
 * (0 cell records and scatterer heads staged in LDS, 1 the heads only, 2 neither), and the
 * number of history slots of a workgroup's pool.  For tests that must name the code object
 * they compare with the oracle.                                               */
int      r3d_engine_variant(const r3d_engine* e);
uint32_t r3d_engine_pool_slots(const r3d_engine* e);
/* Entries of a workgroup's table of bin accumulators in LDS (0: none -- a model without receivers).  */
uint32_t r3d_engine_accumulators(const r3d_engine* e);

/* Number of scalar counters r3d_run_device expects.                         */
#define R3D_N_SCALARS (3 + R3D_INV_NUM + R3D_EV_NUM)

/* Self-test hook: evaluates one of the kernel's own elementary functions
 * (radiative3d_amd/csrc/r3d_math.h -- the traversal uses these instead of the
 * device library's exp / log / atanh / asin / atan2 / sincos) on the device,
 * 64 consecutive elements per wave, so a test can feed waves whose lanes fall
 * into different tiers of the wave-voted routines.
 * which: 0 exp_lean(x)  1 log_lean(x)  2 atanh_lean(x)  3 asin_small(x)
 *        4 angle_from_sincos(x, y)  5 / 6 sine / cosine of rotation(x)
 *        7 frcp(x)  8 frsqrt(x)  9 fsqrt(x)
 * x, y (may be NULL where unused), out: host arrays of n doubles.  0 on success. */
int r3d_selftest_math(int device, int which, const double* x, const double* y, double* out, uint64_t n);

/* Message for the last failing call on this thread.                        */
const char* r3d_last_error(void);

/* Library version / build string.                                          */
const char* r3d_version(void);

#ifdef __cplusplus
}
#endif
#endif /* R3D_H_ */
