#!/bin/bash
# tools/collect_profiles.sh TAG -- rocprofv3 evidence for the bench workload (run on the GPU box
# from the repo root; outputs under gpurun_out/profiles_TAG/, to be copied into profiles/).
#   1. kernel trace + stats of `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline`
#   2. PMC passes (each in its own run, no tracing): HBM traffic, SQ occupancy / issue counters
set -e
tag=${1:-run}
out=gpurun_out/profiles_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench --output-format csv -- $B > $out/bench_line_under_rocprofv3.json 2> $out/trace.log
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats_bench_nscp_deg9.csv
cp $(find $out/trace -name "*domain_stats.csv" | head -1) $out/domain_stats_bench_nscp_deg9.csv 2>/dev/null || true
# per-launch durations of the traversal kernel, in launch order (bench.py chains the steps,
# r3d_run_device_carry: step launches are propagate_kernel, the flush launches that run only the
# stragglers are drain_kernel)
python3 - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
f = sorted(glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if "propagate_kernel" in r["Kernel_Name"] or "drain_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
kinds = ["flush" if "drain_kernel" in r["Kernel_Name"] else "step" for r in rows]
steps = [m for m, k in zip(ms, kinds) if k == "step"]
flushes = [m for m, k in zip(ms, kinds) if k == "flush"]
json.dump({"launches_ms": [{"kind": k, "ms": round(m, 4)} for m, k in zip(ms, kinds)],
           "step_launch_avg_ms": sum(steps) / max(1, len(steps)),
           "flush_launch_avg_ms": sum(flushes) / max(1, len(flushes))},
          open(out + "/kernel_launches_bench_nscp_deg9.json", "w"), indent=1)
print("step launches avg %.3f ms (%d), flush launches avg %.3f ms (%d)" %
      (sum(steps) / max(1, len(steps)), len(steps), sum(flushes) / max(1, len(flushes)), len(flushes)))
PY
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set -d $out/pmc$i -o pmc --output-format csv -- $B > $out/pmc$i.log 2>&1
done
python3 tools/pmc_summary.py $out
head -3 $out/kernel_stats_bench_nscp_deg9.csv
tail -1 $out/bench_line_under_rocprofv3.json | cut -c1-300
