#!/bin/bash
# tools/main_rccl_trace.sh OUTDIR -- kernel trace of `./main --devices=0` on the crust-pinch run (TOA degree 6, 1e6
# histories): the traversal kernel and the RCCL kernels of r3d_node_run's reduce in one list (profiles/r05/).
set -e
out=${1:-gpurun_out/main_rccl}; mkdir -p $out
export TMPDIR=/tmp
args=$(python3 -c 'from radiative3d_amd.configs import CONFIGS; print(" ".join(CONFIGS["crustpinch"](6)))')
rocprofv3 --kernel-trace --stats -d $out/trace -o main --output-format csv -- ./main $args --num-phonons=1M --seed=3 \
    --devices=0 --output-dir=$out > $out/main_stdout.log 2> $out/main_stderr.log
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats_main_devices0.csv
grep "Shards" $out/main_stdout.log
rm -rf $out/trace $out/seis_*.octv
