"""Per-launch durations of the traversal kernel from a rocprofv3 kernel trace, in launch order
(tools/collect_profiles.sh): bench.py chains its steps (r3d_run_device_carry) -- step launches are
pool_kernel, the chain's flush launches (stragglers only) drain_kernel."""
import csv
import glob
import json
import sys

out, config = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(f"{out}/trace_{config}/**/*kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if "pool_kernel" in r["Kernel_Name"] or "drain_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
kinds = ["flush" if "drain_kernel" in r["Kernel_Name"] else "step" for r in rows]
steps = [m for m, k in zip(ms, kinds) if k == "step"]
flushes = [m for m, k in zip(ms, kinds) if k == "flush"]
json.dump({"config": config, "launches_ms": [{"kind": k, "ms": round(m, 4)} for m, k in zip(ms, kinds)],
           "step_launch_avg_ms": sum(steps) / max(1, len(steps)),
           "flush_launch_avg_ms": sum(flushes) / max(1, len(flushes))},
          open(f"{out}/kernel_launches_{config}.json", "w"), indent=1)
print("%s: step launches avg %.3f ms (%d), flush launches avg %.3f ms (%d)" %
      (config, sum(steps) / max(1, len(steps)), len(steps), sum(flushes) / max(1, len(flushes)), len(flushes)))
