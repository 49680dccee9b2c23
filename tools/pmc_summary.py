"""Summarise the rocprofv3 --pmc passes of tools/collect_profiles.sh (DIR/pmc*/): per-dispatch
counter values of the traversal kernel.  bench.py chains its steps (r3d_run_device_carry): the
step launches are propagate_kernel dispatches, the chain's flush launches (stragglers only)
drain_kernel dispatches; they are averaged separately."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
res, kern = {}, None
for f in sorted(glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    drain = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        if "drain_kernel" in row["Kernel_Name"]:
            drain[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
        if "propagate_kernel" not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
        if kern is None:
            kern = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                        "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in row}
    for name, per in acc.items():
        steps = [per[k] for k in sorted(per)]
        flush = list(drain[name].values())
        res[name] = {"dispatches": len(steps) + len(flush), "step_launches": len(steps),
                     "mean_per_dispatch": sum(steps) / len(steps), "min": min(steps), "max": max(steps),
                     "flush_launch_mean": (sum(flush) / len(flush)) if flush else None}
res["_kernel"] = kern
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    res["hbm_traffic_bytes_per_launch"] = 1024.0 * (res["FETCH_SIZE"]["mean_per_dispatch"] +
                                                    res["WRITE_SIZE"]["mean_per_dispatch"])
res["_note"] = ("rocprofv3 --pmc passes of `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline` (one pass per "
                "counter group, no tracing). mean_per_dispatch / min / max are over the STEP launches (one = 1e7 new "
                "histories); the two flush launches of the run (stragglers only) are in flush_launch_mean. "
                "FETCH_SIZE / WRITE_SIZE are in KiB per dispatch. The gfx950 x2 FETCH_SIZE correction of "
                "MI355X_MICROARCH.md applies to wide coalesced streams only; this kernel's reads are 8-16 B gathers, "
                "so the value is left uncorrected.")
json.dump(res, open(out + "/pmc_counters_bench_nscp_deg9.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k.startswith("hbm") or k == "_kernel"}, indent=1))
