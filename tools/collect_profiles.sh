#!/bin/bash
# tools/collect_profiles.sh TAG [CONFIG ...] -- rocprofv3 evidence for the bench workloads (run on the
# GPU box from the repo root; outputs under gpurun_out/profiles_TAG/, to be copied into profiles/).
# Per config (default: all five of bench.py --config):
#   1. kernel trace + stats of `python3 bench.py --config C --steps 5 --warmup 1 --timed-only`
#   2. PMC passes (each in its own run, no tracing): HBM traffic, SQ issue / occupancy counters
# --timed-only keeps the run to the warm-up and the timed chain, so every pool_kernel dispatch
# is a chained step launch of the workload and every pool_drain_kernel dispatch a chain's flush.
set -e
tag=${1:-run}
shift || true
configs=${@:-crustpinch halfspace lopnor sphere crustpinch_volume}
out=gpurun_out/profiles_$tag
mkdir -p $out
export TMPDIR=/tmp
for c in $configs; do
  B="python3 bench.py --config $c --steps 5 --warmup 1 --timed-only"
  echo "== $c: kernel trace" >&2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_$c -o bench --output-format csv -- $B \
      > $out/bench_line_under_rocprofv3_$c.json 2> $out/trace_$c.log
  cp $(find $out/trace_$c -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$c.csv
  python3 tools/launch_list.py $out $c
  i=0
  for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    echo "== $c: pmc pass $i ($set)" >&2
    timeout -k 10 300 rocprofv3 --pmc $set -d $out/pmc_${c}_$i -o pmc --output-format csv -- $B > $out/pmc_${c}_$i.log 2>&1
  done
  python3 tools/pmc_summary.py $out $c
  rm -rf $out/trace_$c $out/pmc_${c}_[0-9]   # raw per-dispatch CSVs: tens of MB; the summaries stay
done
