#!/usr/bin/env python3
"""bench.py -- phonon-histories/s of the HIP engine on the reference's headline
configuration (BASELINE.json configs[1]: NSCP crust-pinch, do-crustpinch.sh
arguments, TOA degree 9, 1e7 histories per step), one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path (GenerateEventPhonon + Propagate) over a
fresh batch of 1e7 history ids per GPU; tables are resident in HBM before the
timed region; the per-receiver bins stay in HBM and are summed over ranks with
one RCCL all-reduce per buffer inside the timed region (weak scaling: every
rank runs the same count).  Rank 0 prints one JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_history(ev, n_toa, n_seis, cell_kind):
    """must-touch bytes per history from per-history event counts."""
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


def recorded_hbm_traffic(toa_degree, n):
    """HBM bytes per launch of the traversal kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE + WRITE_SIZE, KiB -> bytes; profiles/r01/pmc_counters_bench_nscp_deg9.json).
    PMC collection needs the profiler around the process, so bench.py cannot measure it in
    line; the figure is only reported when it was taken on this very workload."""
    path = os.path.join(REPO, "profiles", "r01", "pmc_counters_bench_nscp_deg9.json")
    if toa_degree != 9 or n != 10_000_000 or not os.path.exists(path):
        return None
    try:
        return float(json.load(open(path))["hbm_traffic_bytes_per_launch"])
    except (KeyError, ValueError, OSError):
        return None


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


def cpu_baseline(model, budget_s=12.0):
    """The oracle (CPU port of the reference's algorithm) on this box's host
    cores: one thread per core, each on its own id range, bounded to ~budget_s.
    Returns the JSON object, the per-thread results (independent batches, for the
    envelope check) and the histories per thread."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_ffi
    cores = usable_cores()
    t = time.perf_counter()
    oracle_ffi.run(model, 2000, first_id=1 << 50)
    per_core_rate = 2000 / (time.perf_counter() - t)
    per_thread = max(2000, int(per_core_rate * budget_s))

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
    z = (eg - ec)[sel] / np.sqrt((vg + vc)[sel])
    zp = (eg - ec)[sel] / np.sqrt((eg ** 2 / np.maximum(ng, 1) + ec ** 2 / np.maximum(nc, 1))[sel])
    return {"rms_sigma": float(np.sqrt(np.mean(z ** 2))), "bins": int(sel.sum()),
            "gpu_histories": int(n_gpu), "cpu_histories": int(n_cpu), "min_count": min_count,
            "rms_sigma_poisson": float(np.sqrt(np.mean(zp ** 2))),
            "definition": "RMS over (seis, component, bin) of (e_gpu-e_cpu)/sigma on independent id "
                          "ranges; sigma^2 = variance of batch means (GPU and CPU batches summed)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--histories", type=int, default=10_000_000, help="histories per GPU per step")
    ap.add_argument("--toa-degree", type=int, default=9)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--self-contained", action="store_true",
                    help="every step a self-contained launch that drains its own stragglers "
                         "(default: steps chained, one flush launch at the end of the timed region)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from radiative3d_amd import Engine, Model
    from radiative3d_amd.parallel import DeviceResult
    from radiative3d_amd.configs import crustpinch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    under_launcher = "RANK" in os.environ and "MASTER_PORT" in os.environ   # torch.distributed.run
    if under_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if under_launcher:   # also at world size 1, so the RCCL path is the one that runs
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    # ---- build the model (host) and put it in HBM: not timed --------------
    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    # One GPU: host-built tables, the ones the CPU baseline / envelope check also runs on.
    # Several ranks: every rank lets its engine evaluate the tables in HBM (--device-tables)
    # instead of N processes each spending the host's cores on the same 1.5 GB of tables.
    model_args = crustpinch(args.toa_degree) + (["--device-tables"] if world > 1 else [])
    t0 = time.perf_counter()
    model = Model(model_args)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    engine = Engine(model, device=local_rank)
    t_upload = time.perf_counter() - t0
    note(f"model built in {t_build:.1f} s, tables in HBM after {t_upload:.1f} s")
    result = DeviceResult(model, device)
    stream = torch.cuda.current_stream(device)
    n = args.histories
    seed = 0x5EED

    # A lone batch ends in a drain phase: the work counter is exhausted and ever fewer lanes
    # still carry a history (the longest NSCP histories are ~40 times the mean), about 8 of a
    # lone 1e7-history launch's 25 ms.  The steps therefore form a carry chain
    # (r3d_run_device_carry): the histories still in flight when a step's ids run out stay in
    # the engine and are resumed by the next step's launch, and one flush launch after the
    # last step runs the stragglers to their end.  Results do not depend on it.
    step_res = DeviceResult(model, device)

    def step(i, events=None):
        # every step and every rank gets its own disjoint id range; a step ends with the
        # whole-job bins of that launch (summed over ranks) added to the running total
        first = (i * world + rank) * n
        step_res.zero_()
        if events is not None:
            events[0].record(stream)
        engine.run_device(n, first, seed, *step_res.pointers(), stream=stream.cuda_stream,
                          carry=None if args.self_contained else "carry")
        if events is not None:
            events[1].record(stream)
        step_res.allreduce_()     # no-op at world == 1
        result.add_(step_res)

    def flush():
        if args.self_contained:
            return
        step_res.zero_()
        engine.run_device(0, 0, seed, *step_res.pointers(), stream=stream.cuda_stream, carry="final")
        step_res.allreduce_()
        result.add_(step_res)

    def sync():
        if under_launcher:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    flush()
    sync()
    result.zero_()
    sync()
    # per-launch kernel durations: HIP events on the launch stream, recorded around each
    # launch and read after the timed region
    kernel_events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                     for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, kernel_events[i])
    flush()
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in kernel_events]
    if under_launcher:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    note(f"timed {args.steps} steps in {elapsed:.3f} s")
    if rank == 0:
        res = result.to_result()
        total = res.events["generated"]            # histories summed over ranks and passes
        ev = {k: v / total for k, v in res.events.items()}
        b_hist = algorithmic_bytes_per_history(ev, model.n_toa, model.n_seismometers, model.desc.cell_kind)
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = n * b_hist / (avg_ms * 1e-3) / 1e9
        value = args.steps * n * world / elapsed
        line = {
            "metric": "phonon-histories/sec", "value": value, "unit": "histories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "NSCP crust-pinch (do-crustpinch.sh arguments), "
                                   f"TOA degree {args.toa_degree}, {n} histories per GPU per step, "
                                   "480 seismometers x 300 bins",
                       "histories_per_gpu_per_step": n, "toa_degree": args.toa_degree,
                       "cells": model.n_cells, "scatterers": model.n_scatterers,
                       "parallelism": f"history-id shards x{world}, one all-reduce of the bins per step; "
                                      + ("every step a self-contained launch" if args.self_contained else
                                         "steps chained (unfinished histories carried into the next "
                                         "step's launch, one flush launch at the end, inside the timed region)")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": recorded_hbm_traffic(args.toa_degree, n),
                         "kernel": "propagate_kernel<tetra>", "kernel_ms_avg": avg_ms,
                         "algorithmic_bytes_per_history": b_hist,
                         "events_per_history": {k: round(v, 4) for k, v in ev.items()}},
            "host": {"model_build_s": round(t_build, 2), "table_upload_s": round(t_upload, 2),
                     "tables": "device-built" if world > 1 else "host-built"},
        }
        # the same model with the scattering tables evaluated in HBM (--device-tables):
        # what a production run pays before its first history
        t0 = time.perf_counter()
        dev_model = Model(crustpinch(args.toa_degree) + ["--device-tables"])
        t_dev_host = time.perf_counter() - t0
        t0 = time.perf_counter()
        dev_engine = Engine(dev_model, device=local_rank)
        t_dev_engine = time.perf_counter() - t0
        dev_engine.close()
        line["host"]["device_tables"] = {"model_build_s": round(t_dev_host, 2),
                                         "engine_create_s": round(t_dev_engine, 2)}
        if not args.no_cpu_baseline and world == 1:
            note("timing the CPU baseline (oracle) ...")
            line["cpu_baseline"], cpu_parts, per_thread = cpu_baseline(model)
            note("envelope check: GPU batches ...")
            n_batch, nb = 32, max(1, n // 8)
            g_e, g_c = [], []
            for b in range(n_batch):   # untimed; ids beyond every range used above
                step_res.zero_()
                engine.run_device(nb, (1 << 44) + b * nb, seed, *step_res.pointers(), stream=stream.cuda_stream)
                torch.cuda.synchronize()
                r = step_res.to_result()
                g_e.append(r.energy / nb), g_c.append(r.counts)
            gpu_mom = batch_moments(g_e, g_c)
            cpu_mom = batch_moments([p.energy / per_thread for p in cpu_parts], [p.counts for p in cpu_parts])
            line["envelope"] = envelope_agreement(gpu_mom, cpu_mom, n_batch * nb, per_thread * len(cpu_parts))
            # calibration of the statistic: the same comparison with the CPU sample replaced by an
            # independent GPU sample of the CPU sample's batch structure (both sides the same code).
            # With few, unequal batches and heavy-tailed bins it sits above 1; the GPU-vs-CPU figure
            # is to be read against it.
            c_e, c_c = [], []
            for b in range(len(cpu_parts)):
                step_res.zero_()
                engine.run_device(per_thread, (1 << 52) + b * per_thread, seed, *step_res.pointers(),
                                  stream=stream.cuda_stream)
                torch.cuda.synchronize()
                r = step_res.to_result()
                c_e.append(r.energy / per_thread), c_c.append(r.counts)
            twin = envelope_agreement(gpu_mom, batch_moments(c_e, c_c), n_batch * nb, per_thread * len(cpu_parts))
            line["envelope"]["rms_sigma_gpu_vs_gpu_same_batches"] = twin["rms_sigma"]
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if under_launcher:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
