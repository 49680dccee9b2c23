#!/bin/bash
# tools/comm_rccl_trace.sh OUTDIR -- kernel trace of bench.py as a LAUNCHED one-rank job (the environment a launcher gives
# rank 0 of 1): the traversal kernels and whatever RCCL runs for r3d_comm_reduce's grouped all-reduce of the bins in one
# list.  (RCCL's ring kernels only exist between two or more ranks; at one rank a reduce is a device copy.)
set -e
out=${1:-gpurun_out/comm_rccl}; mkdir -p $out
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29517} HSA_ENABLE_IPC_MODE_LEGACY=0
rocprofv3 --kernel-trace --stats -d $out/trace -o bench --output-format csv -- python3 bench.py --gpus 1 --steps 5 --warmup 1 \
    --timed-only > $out/bench_line_launched_world1.json 2> $out/bench_launched_world1.log
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats_bench_launched_world1.csv
python3 -c "
import json; d = json.load(open('$out/bench_line_launched_world1.json')); print(json.dumps(d['collective'], indent=1))"
rm -rf $out/trace
