#!/usr/bin/env python3
"""bench.py -- phonon-histories/s of the HIP engine, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload: the reference's headline configuration (BASELINE.json configs[1]: NSCP
crust-pinch, do-crustpinch.sh arguments, TOA degree 9, 1e7 histories per GPU per step).
--config selects the other BASELINE configurations (halfspace, lopnor, sphere = SphereEarth
with the deep source, crustpinch_volume = crust-pinch video run + the 10 GB scatter-event grid).

Called directly with --gpus N > 1 (no launcher environment) the program starts N rank
processes itself, before anything touches a GPU (radiative3d_amd/launch.py); under
torch.distributed.run it is one rank of the launcher's job.  Either way the world size must
equal --gpus, and `n_gpus` in the output is the process group's size.

A step = one pass of the hot path (GenerateEventPhonon + Propagate) over a fresh batch of
history ids per GPU; tables are resident in HBM before the timed region; the per-receiver bins
stay in HBM and are summed over ranks with one RCCL all-reduce per buffer at the end of the job,
inside the timed region (weak scaling: every rank runs the same count).  Rank 0 prints one JSON line.

Beside that figure the line carries `single_launch` (one self-contained launch of the step's size) and `job`:
the BASELINE configuration AS STATED -- config 1: 1e5 histories, config 2: 1e7, configs 3-5: 1e8 divided over the
ranks -- as one self-contained launch per rank + the reductions, timed from the first launch to the reduced result
("scaling": "strong").  Config 5's 10 GB event grid is reduced by frame (--volume-reduce).
"""
import argparse
import hashlib
import re
import json
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBS = 8000.0
# What the chip delivers in RANDOM 64-byte sectors from a table of GBs (tools/microbench/hbm_gather.hip, recorded in
# profiles/r05/hbm_gather.log: 51.2 G sectors/s at 12, 16 or 32 waves per CU, 1 to 8 fetches in flight per lane): the
# bound of the table look-ups (take-off spray, scattering), which read one sector per fetch.
HBM_GATHER_PEAK_GBS = 3280.0
N_SIMD = 256 * 4
MAX_CLOCK_GHZ = 2.4
PROFILE_ROUND = "r06"


# BASELINE.json `configs`, as stated: the histories of ONE job of each configuration (configs 3-5 are
# jobs of a whole node: their histories are divided over the ranks -- strong scaling)
JOB_HISTORIES = {"halfspace": 100_000, "crustpinch": 10_000_000, "lopnor": 100_000_000, "sphere": 100_000_000,
                 "crustpinch_volume": 100_000_000}


def workloads():
    from radiative3d_amd import configs as C
    return {
        "halfspace": dict(
            args=lambda deg: C.halfspace(deg, one_receiver=True), histories=10_000_000, volume=None,
            label="Halfspace (do-halfspace.sh arguments, one receiver as BASELINE config 1 names)"),
        "crustpinch": dict(
            args=C.crustpinch, histories=10_000_000, volume=None,
            label="NSCP crust-pinch (do-crustpinch.sh arguments)"),
        "lopnor": dict(
            args=C.lopnor, histories=10_000_000, volume=None,
            label="LopNorCyl (do-lopnor.sh arguments, explosion source)"),
        "sphere": dict(
            args=lambda deg: C.sphere(deg, source_depth=-600), histories=10_000_000, volume=None,
            label="SphereEarth (do-spherical.sh arguments, deep double-couple source at 600 km)"),
        "crustpinch_volume": dict(
            args=C.crustpinch_vids, histories=10_000_000, volume=C.CRUSTPINCH_VOLUME,
            label="NSCP crust-pinch video run (do-crustpinch-vids.sh arguments: pinned mean free paths, no "
                  "deflection) with a dense scatter-event grid 2 x 300 x 64 x 256 x 256 uint32 = 10 GB"),
    }


def algorithmic_bytes_per_history(ev, n_toa, n_seis, cell_kind):
    """must-touch bytes per history of the REFERENCE's algorithm from
    per-history event counts (every receiver tested on every surface arrival, every cell record
    from memory).  Reported as `contract_bytes`; the engine does not move most of them."""
    probes = 2 + math.ceil(math.log2(n_toa))
    b_cell = {0: 150, 1: 340, 2: 120}[cell_kind]
    return (probes * 8 + 16
            + ev["iterations"] * b_cell
            + ev["transfer"] * 64
            + ev["rtsolve"] * 112
            + ev["scatter"] * (probes * 8 + 8 + 16)
            + ev["collect"] * n_seis * 48
            + ev["catch"] * 56
            + 104)


def contract_terms(ev, n_toa, n_seis, cell_kind):
    """algorithmic_bytes_per_history term by term (the same formula): which of the reference's must-touch
    bytes are what -- the all-receiver scan of every surface arrival is most of them."""
    probes = 2 + math.ceil(math.log2(n_toa))
    b_cell = {0: 150, 1: 340, 2: 120}[cell_kind]
    return {"source_draw": probes * 8 + 16,
            "cell_records": ev["iterations"] * b_cell,
            "cell_hand_overs": ev["transfer"] * 64,
            "interface_solves": ev["rtsolve"] * 112,
            "scatter_draws": ev["scatter"] * (probes * 8 + 8 + 16),
            "receiver_scan_all_receivers_per_arrival": ev["collect"] * n_seis * 48,
            "catches": ev["catch"] * 56,
            "history_state": 104}


def phase_floor(ev, cell_kind):
    """Vector instructions per history if nothing but each phase's common path were executed: the static counts of
    tools/microbench/phase_floor.hip (profiles/<round>/phase_floor.json, made by tools/phase_floor.py on these kernel
    sources) times this run's own event counts.  No queue, slot or tally code, no rare branch, the short tier of every
    series; light face events (hand-overs, Snell bends) and the receiver candidates that do not catch are not counted:
    a FLOOR.  achieved / floor falls when instructions are removed from the kernel and the floor stands."""
    rel = os.path.join("profiles", PROFILE_ROUND, "phase_floor.json")
    try:
        rec = json.load(open(os.path.join(REPO, rel)))
    except (OSError, ValueError):
        return None, rel, "no floor file"
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None, rel, "counted on other kernel sources"
    v = rec["valu"]
    move = v[{0: "move_cyl", 1: "move_tet", 2: "move_sph"}[cell_kind]]
    terms = {"spray": ev["generated"] * v["spray"], "moves": ev["iterations"] * move, "interface_solves": ev["rtsolve"] * v["rt"],
             "scatterings": ev["scatter"] * v["scatter"], "arrivals": ev["collect"] * v["collect_arrival"],
             "catches": ev["catch"] * (v["collect_candidate"] + v["collect_catch"])}
    return {"per_history": sum(terms.values()), "by_term": terms, "per_event": {"move": move, **{k: v[k] for k in v if not k.startswith("move_")}}}, rel, None


def ta_busy(rec):
    try:
        return rec["counters"]["TA_TA_BUSY_sum"]["mean_per_step_launch"] / (256.0 * rec["GRBM_GUI_ACTIVE"] / 8.0)
    except (KeyError, TypeError, ZeroDivisionError):
        return None


def kernel_source_hash(csrc=None):
    """sha256 over the traversal kernel's code, its launch geometry and build flags: what the recorded
    counters are keyed by.  Comments and white space do not count (a reworded comment leaves the machine
    code as it was); neither do the translation units no traversal kernel is built from or launched by (the
    table builders and the grid's compaction / add kernels)."""
    h = hashlib.sha256()
    csrc = csrc or os.path.join(REPO, "radiative3d_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".h", ".hip")) and name not in ("r3d_volume.hip", "r3d_tables_build.hip", "r3d_tables_build.h"):
            h.update(name.encode())
            text = open(os.path.join(csrc, name), encoding="utf-8").read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)          # block comments
            text = re.sub(r"//[^\n]*", " ", text)                       # line comments (no string in these sources holds //)
            h.update(" ".join(text.split()).encode())
    for line in open(os.path.join(REPO, "Makefile")):   # (HIPFLAGS and the per-kind HIPFLAGS_CYL / _TET / _SPH)
        if line.startswith("HIPFLAGS") or line.lstrip().startswith("-mllvm"):
            h.update(line.encode())
    return h.hexdigest()[:16]


def recorded_counters(config, toa_degree, n):
    """Per-launch hardware counters of this workload's traversal kernel from the committed
    rocprofv3 --pmc passes (profiles/<round>/pmc_<config>.json, written by tools/pmc_summary.py
    from tools/collect_profiles.sh).  PMC collection needs the profiler around the process, so
    bench.py cannot take them in line: they are RECORDED figures, named as such in the output,
    and are dropped when they were taken on other kernel sources or another workload."""
    rel = os.path.join("profiles", PROFILE_ROUND, f"pmc_{config}.json")
    try:
        rec = json.load(open(os.path.join(REPO, rel)))
    except (OSError, ValueError):
        return None, rel, "no counter file"
    if rec.get("toa_degree") != toa_degree or rec.get("histories_per_launch") != n:
        return None, rel, "recorded on another workload size"
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None, rel, "recorded on other kernel sources"
    return rec, rel, None


def usable_cores(cap=16):
    """Host cores this process may really use: affinity, cgroup CPU quota, and the
    GPU box's per-GPU share (16) as an upper bound."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def cpu_baseline(model, target_histories=10_000_000, budget_s=25.0):
    """The oracle (CPU port of the reference's algorithm) on this box's host cores: one thread
    per core, each on its own id range.  The sample is the 1e7 histories north_star names when
    they fit ~budget_s of wall time on these cores, else what does (stated in `sample`).
    Returns the JSON object, the per-thread results (independent batches, for the envelope
    check) and the histories per thread."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_ffi
    cores = usable_cores()
    probe = 2000
    t = time.perf_counter()
    oracle_ffi.run(model, probe, first_id=1 << 50)
    per_core_rate = probe / (time.perf_counter() - t)
    per_thread = max(probe, min(int(per_core_rate * budget_s), -(-target_histories // cores)))

    def work(i):
        return oracle_ffi.run(model, per_thread, first_id=(1 << 50) + (i + 1) * per_thread)

    t = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:   # ctypes releases the GIL during the call
        parts = list(pool.map(work, range(cores)))
    dt = time.perf_counter() - t
    n_cpu = cores * per_thread
    line = {"value": n_cpu / dt, "unit": "histories/s", "cores": cores, "kind": "port",
            "sample": f"{n_cpu} histories of the same workload "
                      f"({per_thread} per thread x {cores} threads, {dt:.1f} s), oracle/r3d_oracle.cpp"}
    return line, parts, per_thread


def batch_moments(energies, counts):
    """(mean, variance of the mean, total counts) per (seis, bin, component) from equally
    sized independent batches of per-history-normalised energies."""
    import numpy as np
    e = np.stack(energies)
    b = e.shape[0]
    return e.mean(0), e.var(0, ddof=1) / b, np.sum(counts, axis=0)


def envelope_agreement(gpu_mom, cpu_mom, n_gpu, n_cpu, min_count=25):
    """BASELINE metric part 2: RMS over seismometer, component (X,Y,Z,P,S)
    and time bin of (e_gpu - e_cpu) / sigma, e = Trace / N, the two runs on disjoint history
    ids (independent samples); near 1 means agreement, the target is <= 2.

    sigma comes from batch means: each run is split into equal independent batches and the
    variance of a bin's mean is the sample variance of its batch means / batches.  (The
    shortcut sigma^2 = e^2 / n from the bin's count assumes equal energy per catch; catches
    differ by orders of magnitude, and two oracle runs against each other score 5.2 with it.
    That figure is kept as rms_sigma_poisson.)"""
    import numpy as np
    eg, vg, ng = gpu_mom
    ec, vc, nc = cpu_mom
    ng = ng.sum(-1).astype(float)[..., None]
    nc = nc.sum(-1).astype(float)[..., None]
    sel = np.broadcast_to((nc >= min_count) & (ng >= min_count), eg.shape) & (ec > 0) & (eg > 0)
    if not sel.any():
        return {"rms_sigma": None, "bins": 0, "gpu_histories": int(n_gpu), "cpu_histories": int(n_cpu),
                "min_count": min_count}
    z = (eg - ec)[sel] / np.sqrt((vg + vc)[sel])
    zp = (eg - ec)[sel] / np.sqrt((eg ** 2 / np.maximum(ng, 1) + ec ** 2 / np.maximum(nc, 1))[sel])
    return {"rms_sigma": float(np.sqrt(np.mean(z ** 2))), "bins": int(sel.sum()),
            "gpu_histories": int(n_gpu), "cpu_histories": int(n_cpu), "min_count": min_count,
            "rms_sigma_poisson": float(np.sqrt(np.mean(zp ** 2))),
            "definition": "RMS over (seis, component, bin) of (e_gpu-e_cpu)/sigma on independent id "
                          "ranges; sigma^2 = variance of batch means (GPU and CPU batches summed)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: a timed region of two seconds or more on every configuration's step launch of 7-60 ms, so that a
    #  sampler that looks at the GPU every few seconds sees the run; the whole default run stays within a few minutes)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="crustpinch",
                    choices=["halfspace", "crustpinch", "lopnor", "sphere", "crustpinch_volume"])
    ap.add_argument("--histories", type=int, default=None, help="histories per GPU per step (default: per config)")
    ap.add_argument("--toa-degree", type=int, default=9)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--self-contained", action="store_true",
                    help="every step a self-contained launch that drains its own stragglers "
                         "(default: steps chained, one flush launch at the end of the timed region)")
    ap.add_argument("--reduce-per-step", action="store_true",
                    help="all-reduce every step's bins (default: every launch adds into the rank's own block "
                         "and the blocks are summed over ranks once, after the flush)")
    ap.add_argument("--no-job", action="store_true",
                    help="skip the literal BASELINE job (one self-contained run of the configuration's stated "
                         "size, timed from its first launch to the reduced bins; printed as `job`)")
    ap.add_argument("--volume-reduce", default="auto", choices=["auto", "sparse", "dense", "allreduce"],
                    help="config 5's event grid over ranks: by frame as (index, count) pairs point to point "
                         "(sparse), by frame with one reduce per owner (dense), whichever fits (auto), or the "
                         "whole grid on every rank (allreduce: round 3's form)")
    ap.add_argument("--timed-only", action="store_true",
                    help="stop after the timed region: no single-launch, device-table, CPU-baseline or envelope "
                         "legs (profiling passes: every dispatch is then a step or flush launch of the chain)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, form the process group (gloo, no GPU), print its size and exit: "
                         "the CPU check of the launcher path")
    return ap.parse_args(argv)


def rendezvous_only(args, rank, world):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        emit({"n_gpus": dist.get_world_size(), "ranks_seen": int(t.item()), "rendezvous": "ok"})
    dist.barrier()
    dist.destroy_process_group()


_JSON_OUT = None


def claim_stdout():
    """Keep file descriptor 1 for the JSON line alone: the model builder (C++, radiative3d_amd/host) and the GPU
    runtime write their banners to stdout, so everything else that goes there is sent to stderr from here on."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    out = _JSON_OUT or sys.stdout
    out.write(json.dumps(line) + "\n")
    out.flush()


def main():
    args = parse_args()
    from radiative3d_amd.launch import spawn_ranks, under_launcher
    launched = under_launcher()
    if not launched and (args.gpus > 1 or args.rendezvous_only):
        # called directly: start the ranks here, before anything touches a GPU
        sys.exit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.rendezvous_only:
        return rendezvous_only(args, rank, world)

    import torch
    import torch.distributed as dist
    from radiative3d_amd import Engine, Model
    from radiative3d_amd.parallel import Comm, DeviceResult, DeviceVolume, shard_range

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if launched:   # also at world size 1, so the RCCL path is the one that runs
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        assert dist.get_world_size() == args.gpus
    n_gpus = dist.get_world_size() if launched else 1

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    wl = workloads()[args.config]
    n = args.histories or wl["histories"]
    seed = 0x5EED

    # ---- build the model (host) and put it in HBM: not timed --------------
    # Every rank lets its engine evaluate the tables in HBM (--device-tables), at every N: the
    # points of a scaling curve then run on tables of one provenance, and N processes do not
    # each spend the host's cores on the same GBs of tables.  (The CPU baseline builds its own
    # host tables further down; the two builders agree to 1e-12, tests/test_device_tables.py.)
    model_args = wl["args"](args.toa_degree) + ["--device-tables"]
    t0 = time.perf_counter()
    model = Model(model_args)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    engine = Engine(model, device=local_rank)
    t_upload = time.perf_counter() - t0
    note(f"{args.config}: model built in {t_build:.1f} s, tables in HBM after {t_upload:.1f} s")
    # The bins are reduced by the PRODUCT's communicator (r3d_comm_*: the code `./main --devices` reduces with); the
    # torch.distributed group carries its id, the barriers and the small gathers of this program.  Should the library's
    # communicator not form, the run goes on with torch.distributed's all-reduce and says so in `collective`.
    comm, comm_note = None, None
    if launched:
        try:
            comm = Comm.form(device)
        except RuntimeError as exc:
            comm_note = str(exc)
            note(f"r3d_comm: {exc}; the bins go through torch.distributed")
    result = DeviceResult(model, device, comm)     # this rank's running total; the job's after the one all-reduce
    step_res = DeviceResult(model, device, comm)   # (--reduce-per-step only)
    volume = None
    if wl["volume"]:
        volume = DeviceVolume(engine, device=device, **wl["volume"])
    stream = torch.cuda.current_stream(device)

    # A lone batch ends in a drain phase: the work counter is exhausted and ever fewer lanes
    # still carry a history (the longest NSCP histories are ~40 times the mean).  The steps
    # therefore form a carry chain (r3d_run_device_carry): the histories still in flight when a
    # step's ids run out stay in the engine and are resumed by the next step's launch, and one
    # flush launch after the last step runs the stragglers to their end.  Results do not
    # depend on it.  --self-contained times lone launches instead; the default run reports
    # them too, as `single_launch`.
    #
    # Reduction: every launch ADDS into this rank's result block in HBM, and the blocks are summed
    # over ranks ONCE, after the flush, inside the timed region -- the reference's replicas +
    # combine (scripts/do-parallel.sh:23-29, vis/seisplot/combine.m:26-33).  --reduce-per-step
    # all-reduces every step's block instead (a rank-synchronising point every step).
    def launch(count, first, carry, launches):
        target = step_res if args.reduce_per_step else result
        if args.reduce_per_step:
            step_res.zero_()
        engine.run_device(count, first, seed, *target.pointers(), stream=stream.cuda_stream, carry=carry)
        if launches is not None:
            launches.append(engine.launch_count())
        if args.reduce_per_step:
            step_res.allreduce_()     # no-op without a process group
            result.add_(step_res)

    def step(i, launches=None):
        # every step and every rank gets its own disjoint id range
        launch(n, (i * world + rank) * n, None if args.self_contained else "carry", launches)

    def flush(launches=None):
        if not args.self_contained:
            launch(0, 0, "final", launches)

    def sync():
        if launched:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_volume():
        if args.volume_reduce == "allreduce":
            volume.allreduce_()
        else:   # rank r ends with the job's counts for its frames ("keep sharded by frame")
            volume.reduce_scatter_frames_(mode=args.volume_reduce)

    for i in range(args.warmup):
        step(i)
    flush()
    if not args.reduce_per_step:
        result.allreduce_()
    if volume is not None and args.warmup:   # (connections and the pair buffer are set up outside the timed region)
        reduce_volume()
    sync()
    result.zero_()
    if volume is not None:
        volume.zero_()
    sync()
    # per-launch kernel durations: the engine records a HIP event pair around every launch on
    # the stream it is launched on (r3d_kernel_ms); read after the timed region
    step_launches, flush_launches = [], []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, step_launches)
    flush(flush_launches)
    if not args.reduce_per_step:
        result.allreduce_()               # the one reduction of the job's bins (RCCL over xGMI)
    t_vol0 = time.perf_counter()
    if volume is not None:   # the job's event grid: summed over ranks once, at the end
        torch.cuda.synchronize()          # (so that the reduction is timed on its own)
        t_vol0 = time.perf_counter()
        reduce_volume()
    sync()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    volume_reduce_s = (t1 - t_vol0) if volume is not None else None
    events_binned = volume.job_total() if volume is not None else None   # (a collective: every rank calls it)
    volume_line = None
    if volume is not None:   # (as the timed region left it: the job leg below reduces the grid again)
        volume_line = {"shape": list(volume.shape), "bytes": volume.counters.numel() * 4,
                          "events_binned": events_binned,
                          "reduce_over_ranks_s": volume_reduce_s, "saturated_cells": volume.saturated,
                          "reduction": args.volume_reduce,
                          "frames_held_by_rank_0": None if volume.owned is None else list(volume.owned),
                          "reduced_as": volume.timing.get("mode") if args.volume_reduce != "allreduce" else
                                        (None if volume.widened is None else
                                         ("int64, saturating" if volume.widened else "int32 in place")),
                          "phases_rank_0": {k: v for k, v in volume.timing.items() if k != "mode"}}
    step_ms = [ms for ms in (engine.kernel_ms(k) for k in step_launches) if ms >= 0]   # (the 64 most recent)
    flush_ms = [ms for ms in (engine.kernel_ms(k) for k in flush_launches) if ms >= 0]
    per_rank, roster = None, None
    if launched:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own kernel times, so that an imbalance between the GPUs is visible
        mine = torch.tensor([min(step_ms, default=-1.0), max(step_ms, default=-1.0),
                             sum(step_ms) / max(1, len(step_ms)), sum(flush_ms)],
                            dtype=torch.float64, device=device)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        per_rank = [dict(zip(("step_ms_min", "step_ms_max", "step_ms_mean", "flush_ms"),
                             (round(float(v), 4) for v in row))) for row in everyone]
        # who took part, as the devices themselves report it
        me = comm.describe() if comm is not None else {"rank": rank, "device": local_rank, "device_uuid": str(getattr(torch.cuda.get_device_properties(device), "uuid", ""))}
        roster = [None] * world
        dist.all_gather_object(roster, {"rank": me["rank"], "device": me["device"], "device_uuid": me["device_uuid"]})
        if volume is not None:   # every rank's own time in the grid's reduction, and what it put on the wire
            mine = torch.tensor([volume_reduce_s, float(volume.timing.get("bytes_sent") or 0)],
                                dtype=torch.float64, device=device)
            everyone = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(everyone, mine)
            for row, v in zip(per_rank, everyone):
                row["volume_reduce_s"], row["volume_bytes_sent"] = round(float(v[0]), 5), int(v[1])
    note(f"timed {args.steps} steps in {elapsed:.3f} s")

    # ---- the literal BASELINE job (model.cpp:602-633 is the job): the configuration's stated number of
    # histories, divided over the ranks, as ONE self-contained launch per rank that drains its own
    # stragglers, then the reduction of the bins (and of config 5's grid) -- timed from the first launch
    # to the reduced result, tables resident.  Strong scaling: the job's size does not grow with N.
    res_timed = result.to_result() if rank == 0 else None   # (the timed region's totals: the job leg reuses the block)
    job = None
    if not args.timed_only and not args.no_job:
        n_job = JOB_HISTORIES[args.config]
        lo, hi = shard_range(n_job, rank, world)
        times = []
        for rep in range(4):   # (the first is a warm-up: it sizes the allocator's pools for this launch size)
            result.zero_()
            if volume is not None:
                volume.zero_()
            sync()
            tj = time.perf_counter()
            engine.run_device(hi - lo, (7 << 40) + rep * n_job + lo, seed, *result.pointers(), stream=stream.cuda_stream)
            result.allreduce_()
            if volume is not None:
                torch.cuda.synchronize()
                reduce_volume()
            sync()
            dt = time.perf_counter() - tj
            if launched:
                t = torch.tensor([dt], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            if rep:
                times.append((dt, engine.last_kernel_ms()))
        times.sort()
        dt, kernel_ms = times[len(times) // 2]
        job = {"histories": n_job, "histories_per_gpu": hi - lo, "ms": 1e3 * dt, "value": n_job / dt,
               "unit": "histories/s", "scaling": "strong", "kernel_ms_rank_0": kernel_ms,
               "note": "median of 3: one self-contained launch per rank (drains its own stragglers) + the "
                       "reduction of the bins" + (" and of the event grid" if volume is not None else "") +
                       ", first launch to reduced result, max over ranks"}
        note(f"job of {n_job} histories: {1e3 * dt:.2f} ms")

    line = None
    if rank == 0:
        res = res_timed
        total = res.events["generated"]            # histories summed over ranks and passes
        assert total == args.steps * n * world, (total, args.steps, n, world)
        assert res.n_lost + res.n_timeout + res.n_invalid == total
        ev = {k: v / total for k, v in res.events.items()}
        value = args.steps * n * world / elapsed
        avg_step_ms = sum(step_ms) / max(1, len(step_ms))
        # kernel time per step with the chain's flush launch shared out over its steps
        kernel_ms_per_step = (sum(step_ms) + sum(flush_ms)) / max(1, len(step_ms))
        kind_name = {0: "cylinder", 1: "tetra", 2: "sphere"}[model.desc.cell_kind]

        # ---- what bounds the kernel (recorded counters, measured time) ----
        rec, rec_path, why_not = recorded_counters(args.config, args.toa_degree, n)
        b_hist = algorithmic_bytes_per_history(ev, model.n_toa, model.n_seismometers, model.desc.cell_kind)
        floor, floor_path, floor_why_not = phase_floor(ev, model.desc.cell_kind)
        lane_peak = N_SIMD * MAX_CLOCK_GHZ * 16.0      # G lane-instructions/s: 1024 SIMDs x 16 lanes a clock x 2.4 GHz
        roofline = {
            "bound": "valu",
            "kernel": f"pool_kernel<{kind_name}>", "kernel_ms_step_avg": avg_step_ms,
            "kernel_ms_flush": flush_ms, "kernel_ms_per_step_incl_flush": kernel_ms_per_step,
            "note": "the traversal is fp64 vector code with divergent gathers; no MFMA, 4-16 % of HBM peak (the hbm object). "
                    "achieved = USEFUL lane-instructions per second of a step launch: the static vector-instruction count of "
                    "each phase's common path (floor, below) x this run's own event counts / this run's measured launch "
                    "time; peak = 1024 SIMDs x 16 lanes a clock x 2.4 GHz; frac = useful_frac = achieved / peak.  How busy "
                    "the vector unit was whatever it executed -- useful or queue / slot / tally code, full or masked "
                    "lanes -- is valu_busy (SQ_ACTIVE_INST_VALU x 4 cycles, recorded per launch by rocprofv3 --pmc, / "
                    "this run's measured launch time / (1024 SIMDs x 2.4 GHz))",
            "achieved": None, "peak": lane_peak, "unit": "G lane-instructions/s", "frac": None, "useful_frac": None,
            "events_per_history": {k: round(v, 4) for k, v in ev.items()},
        }
        if floor is not None:
            useful = floor["per_history"] * n / (avg_step_ms * 1e-3) / 1e9
            roofline.update({"achieved": useful, "frac": useful / lane_peak, "useful_frac": useful / lane_peak})
            roofline["floor_lane_insts_per_history"] = floor["per_history"]
            roofline["floor"] = {"source": floor_path, "by_term": floor["by_term"], "valu_insts_per_event": floor["per_event"],
                                 "note": "static v_* counts of each phase's common path (tools/microbench/phase_floor.hip) x this run's "
                                         "event counts: no queue / slot / tally code, no rare branch, no light face event"}
        else:
            roofline["floor_lane_insts_per_history"] = None
            roofline["floor"] = f"none ({floor_why_not}: {floor_path})"
        if rec is not None:
            busy_cycles = 4.0 * rec["SQ_ACTIVE_INST_VALU"]
            busy = busy_cycles / (avg_step_ms * 1e-3) / 1e9 / (N_SIMD * MAX_CLOCK_GHZ)
            traffic = rec.get("hbm_traffic_bytes_per_launch")
            roofline.update({
                "valu_busy": busy,
                "traffic": traffic, "counters": "recorded", "counters_source": rec_path,
                "valu_insts_per_launch": rec.get("SQ_INSTS_VALU"),
                "valu_busy_at_recorded_clock": rec.get("valu_busy"),
                "lanes_active_per_valu_inst": rec.get("lane_activity"),
                # busy x lanes: the share of the chip's lane-issue slots that carried a lane's instruction
                "lane_weighted_frac": busy * rec["lane_activity"] if rec.get("lane_activity") else None,
                # what one history costs in lane-instructions (falls when instructions are removed; valu_busy does not)
                "valu_lane_insts_per_history": rec["SQ_INSTS_VALU"] * 64.0 * rec["lane_activity"] / n
                if rec.get("lane_activity") and rec.get("SQ_INSTS_VALU") else None,
                # the texture addressers' busy share of a step launch (TA_TA_BUSY_sum over 256 units x the launch's cycles):
                # a 64-lane gather costs the unit ~48 cycles whatever its width, and the tetra kernel's cell records
                # (twelve 16-byte gathers a move) keep it busier than the vector unit
                "ta_busy": ta_busy(rec),
                # which of the two units is the busier one for this workload (at the recorded launch time), and its share -- occupancies,
                # not proofs of a bound: a sixth fewer gathers did not shorten the NSCP launch
                "binding": "texture addresser" if (ta_busy(rec) or 0.0) > (rec.get("valu_busy") or 0.0) else "vector issue",
                "binding_busy": max(ta_busy(rec) or 0.0, rec.get("valu_busy") or 0.0),
                "hbm": {"bytes_per_launch": traffic,
                        "GBps": traffic / (avg_step_ms * 1e-3) / 1e9 if traffic else None,
                        "frac_of_peak": traffic / (avg_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None,
                        "peak_GBps": HBM_PEAK_GBS,
                        "frac_of_gather_peak": traffic / (avg_step_ms * 1e-3) / 1e9 / HBM_GATHER_PEAK_GBS if traffic else None,
                        "gather_peak_GBps": HBM_GATHER_PEAK_GBS,
                        "gather_peak_source": "measured here: tools/microbench/hbm_gather.hip, profiles/r05/hbm_gather.log "
                                              "(random 64-byte sectors of a 4 GB table)",
                        "source": rec_path}})
            if floor is not None and roofline.get("valu_lane_insts_per_history"):
                roofline["achieved_over_floor"] = roofline["valu_lane_insts_per_history"] / floor["per_history"]
        else:
            roofline.update({"valu_busy": None, "traffic": None, "counters": f"none ({why_not}: {rec_path})"})
        contract = {
            "per_history": b_hist, "per_launch": b_hist * n,
            "by_term": contract_terms(ev, model.n_toa, model.n_seismometers, model.desc.cell_kind),
            "GBps": n * b_hist / (avg_step_ms * 1e-3) / 1e9,
            "over_hbm_peak": n * b_hist / (avg_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "SURVEY.md 8(d) must-touch bytes of the reference's algorithm (all receivers tested per "
                    "surface arrival, cell records from memory) / measured launch time; not HBM traffic -- the "
                    "receiver hash and the LDS-resident tables never move most of these bytes, hence > peak"}

        reduction = ("one all-reduce of every step's bins" if args.reduce_per_step else
                     "one all-reduce of the bins at the end of the job (inside the timed region)")
        line = {
            "metric": "phonon-histories/sec", "value": value, "unit": "histories/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{wl['label']}, TOA degree {args.toa_degree}, {n} histories per GPU per step, "
                                   f"{model.n_seismometers} seismometers x {model.n_bins} bins; `value` is the rate of "
                                   f"{args.steps} such steps CHAINED (unfinished histories carried into the next step's "
                                   "launch, one flush launch at the end, inside the timed region); the BASELINE "
                                   f"configuration as stated -- {JOB_HISTORIES[args.config]} histories over the ranks as one "
                                   "self-contained launch each + the reduction -- is `job` (`job.value`, strong scaling)"
                                   if not args.self_contained else
                                   f"{wl['label']}, TOA degree {args.toa_degree}, {n} histories per GPU per step, "
                                   f"{model.n_seismometers} seismometers x {model.n_bins} bins; every step a self-contained launch",
                       "name": args.config,
                       "histories_per_gpu_per_step": n, "toa_degree": args.toa_degree,
                       "cells": model.n_cells, "scatterers": model.n_scatterers,
                       "parallelism": f"history-id shards x{world}, {reduction}; "
                                      + ("every step a self-contained launch" if args.self_contained else
                                         "steps chained (unfinished histories carried into the next "
                                         "step's launch, one flush launch at the end, inside the timed region)")},
            "roofline": roofline, "contract_bytes": contract,
            "collective": {"bins": ("r3d_comm_reduce (libr3d_hip.so: one grouped ncclAllReduce of the block's three buffers)"
                                    if comm is not None else "torch.distributed all_reduce" if launched else None),
                           "rccl_ranks": comm.describe()["n_ranks"] if comm is not None else None,
                           "rccl_version": comm.describe()["rccl_version"] if comm is not None else None,
                           "rccl_library": comm.describe()["library"] if comm is not None else None,
                           "r3d_comm_error": comm_note,
                           "ranks": roster if launched else None,
                           "control_plane": dist.get_backend() if launched else None,
                           "process_group": bool(launched), "per_rank_kernel_ms": per_rank},
            "host": {"model_build_s": round(t_build, 2), "engine_create_s": round(t_upload, 2),
                     "tables": "device-built"},
        }
        if job is not None:
            line["job"] = job
        if volume is not None:
            line["volume"] = volume_line

        if args.timed_only:
            line["cpu_baseline"] = None
            emit(line)
        # ---- one self-contained launch of the same size (drains its own stragglers) ----
        if not args.self_contained and not args.timed_only:
            lone = []
            for i in range(3):
                step_res.zero_()
                engine.run_device(n, (1 << 40) + i * n, seed, *step_res.pointers(), stream=stream.cuda_stream)
                torch.cuda.synchronize()
                lone.append(engine.last_kernel_ms())
            lone.sort()
            line["single_launch"] = {"histories": n, "kernel_ms": lone[1], "value": n / (lone[1] * 1e-3),
                                     "unit": "histories/s",
                                     "note": "median of 3 self-contained launches (each drains its own "
                                             "stragglers), this rank only"}

    # ---- CPU baseline and envelope agreement (any world size) --------------------------------
    # Rank 0 times the oracle on its host cores, on host-built tables of the same model; the GPU
    # side of the envelope check is then run by ALL ranks together -- each batch's id range
    # sharded over the ranks, the bins all-reduced -- so the figure also covers the sharded path.
    do_cpu = not args.timed_only and not args.no_cpu_baseline
    if do_cpu:
        plan = torch.zeros(2, dtype=torch.int64, device=device)   # batches, histories per batch
        cpu_parts = None
        if rank == 0:
            note("building host tables and timing the CPU baseline (oracle) ...")
            try:   # (whatever happens here, the other ranks are waiting in the broadcast below)
                t0 = time.perf_counter()
                host_model = Model(wl["args"](args.toa_degree))
                line["host"]["host_tables_model_build_s"] = round(time.perf_counter() - t0, 2)
                line["cpu_baseline"], cpu_parts, per_thread = cpu_baseline(host_model)
                plan[0], plan[1] = len(cpu_parts), per_thread
            except Exception as exc:   # noqa: BLE001 -- reported in the line, the timed figures stand
                line["cpu_baseline"] = None
                line["cpu_baseline_error"] = f"{type(exc).__name__}: {exc}"
                cpu_parts = None
        if launched:
            dist.broadcast(plan, src=0)
        n_batches, per_batch = int(plan[0].item()), int(plan[1].item())

        def gpu_batches(base):
            g_e, g_c = [], []
            for b in range(n_batches):   # untimed
                lo, hi = shard_range(per_batch, rank, world)
                step_res.zero_()
                engine.run_device(hi - lo, base + b * per_batch + lo, seed, *step_res.pointers(),
                                  stream=stream.cuda_stream)
                step_res.allreduce_()
                torch.cuda.synchronize()
                if rank == 0:
                    r = step_res.to_result()
                    g_e.append(r.energy / per_batch), g_c.append(r.counts)
            return batch_moments(g_e, g_c) if rank == 0 else None

        if n_batches == 0:   # (the CPU leg failed on rank 0: nothing to compare with)
            if rank == 0:
                emit(line)
            engine.close()
            if comm is not None:
                comm.close()
            if launched:
                dist.barrier()
                dist.destroy_process_group()
            return
        note("envelope check: GPU batches ...")
        # the GPU sample mirrors the CPU sample's batch structure (same count, same size), on ids
        # beyond every range used above: two independent samples of 1e7 histories each when the
        # CPU got that far
        gpu_mom = gpu_batches(1 << 44)
        # calibration of the statistic: the same comparison with the CPU sample replaced by a
        # second independent GPU sample (both sides the same code); with few batches and
        # heavy-tailed bins it need not sit at exactly 1
        gpu_mom2 = gpu_batches(1 << 52)
        if rank == 0:
            n_side = per_batch * n_batches
            cpu_mom = batch_moments([p.energy / per_batch for p in cpu_parts], [p.counts for p in cpu_parts])
            line["envelope"] = envelope_agreement(gpu_mom, cpu_mom, n_side, n_side)
            line["envelope"]["gpu_side"] = (f"device-built tables, each batch sharded over {world} rank(s) and "
                                            "all-reduced" if launched else "device-built tables, one engine")
            line["envelope"]["cpu_side"] = "host-built tables, oracle/r3d_oracle.cpp"
            twin = envelope_agreement(gpu_mom, gpu_mom2, n_side, n_side)
            line["envelope"]["rms_sigma_gpu_vs_gpu_same_batches"] = twin["rms_sigma"]
    elif rank == 0 and not args.timed_only:
        line["cpu_baseline"] = None
    if rank == 0 and not args.timed_only:
        emit(line)
    engine.close()
    if comm is not None:
        comm.close()
    if launched:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
