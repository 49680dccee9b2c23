"""Summarise the rocprofv3 --pmc passes of tools/collect_profiles.sh (DIR/pmc_<config>_*/) into
DIR/pmc_<config>.json: per-dispatch counter values of the traversal kernel, and the derived
figures bench.py's `roofline` object reads.  bench.py chains its steps (r3d_run_device_carry): the
step launches are pool_kernel dispatches, the chain's flush launches (stragglers only)
drain_kernel dispatches; they are averaged separately."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_hash, workloads)

out, config = sys.argv[1], sys.argv[2]
counters, kern = {}, None
for f in sorted(glob.glob(f"{out}/pmc_{config}_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    drain = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        if "drain_kernel" in row["Kernel_Name"]:
            drain[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
        if not ("pool_kernel" in row["Kernel_Name"]):
            continue
        acc[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
        if kern is None:
            kern = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                        "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in row}
    for name, per in acc.items():
        steps = [per[k] for k in sorted(per)]
        flush = list(drain[name].values())
        counters[name] = {"step_launches": len(steps), "mean_per_step_launch": sum(steps) / len(steps),
                          "min": min(steps), "max": max(steps),
                          "flush_launch_mean": (sum(flush) / len(flush)) if flush else None}


def mean(name):
    return counters[name]["mean_per_step_launch"] if name in counters else None


res = {"config": config, "toa_degree": 9, "histories_per_launch": bench.workloads()[config]["histories"],
       "kernel_source_hash": bench.kernel_source_hash(),
       "command": f"python3 bench.py --config {config} --steps 5 --warmup 1 --timed-only (one rocprofv3 --pmc pass "
                  "per counter group, no tracing)",
       "kernel": kern}
for name in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
             "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_SALU",
             "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
    res[name] = mean(name)
if mean("FETCH_SIZE") is not None and mean("WRITE_SIZE") is not None:
    # KiB per dispatch -> bytes.  The gfx950 x2 FETCH_SIZE correction of MI355X_MICROARCH.md holds for wide
    # coalesced streams (128-B requests tallied at 64 B); this kernel's reads are 8-16 B gathers, i.e. 64-B
    # requests, so the value is left uncorrected and the corrected one is given beside it as an upper bound.
    res["hbm_traffic_bytes_per_launch"] = 1024.0 * (mean("FETCH_SIZE") + mean("WRITE_SIZE"))
    res["hbm_traffic_bytes_per_launch_fetch_x2"] = 1024.0 * (2 * mean("FETCH_SIZE") + mean("WRITE_SIZE"))
if mean("SQ_ACTIVE_INST_VALU") and mean("GRBM_GUI_ACTIVE"):
    # SQ_ACTIVE_INST_* count quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
    res["valu_busy"] = 4.0 * mean("SQ_ACTIVE_INST_VALU") / (1024 * mean("GRBM_GUI_ACTIVE") / 8)
if mean("SQ_THREAD_CYCLES_VALU") and mean("SQ_ACTIVE_INST_VALU"):
    res["lane_activity"] = mean("SQ_THREAD_CYCLES_VALU") / (64 * mean("SQ_ACTIVE_INST_VALU"))
if mean("SQ_WAIT_ANY") and mean("SQ_WAVE_CYCLES"):
    res["wave_cycles_waiting"] = mean("SQ_WAIT_ANY") / mean("SQ_WAVE_CYCLES")
res["counters"] = counters
res["note"] = ("mean_per_step_launch / min / max are over the STEP launches (warm-up + 5 timed, each = "
               "histories_per_launch new histories per GPU); the flush launches of the run (stragglers only) are in "
               "flush_launch_mean.  FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES "
               "/ SQ_WAIT_* in quad-cycles; GRBM_GUI_ACTIVE in cycles summed over the 8 XCDs.")
json.dump(res, open(f"{out}/pmc_{config}.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ("counters", "note", "command")}, indent=1))
